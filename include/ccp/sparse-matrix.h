// ccp/sparse-matrix.h — drop-in C++17 facade with the reference's `SparseMatrix<T, IndexType>`
// surface (reference: project/src/PhotoMontage/sparse-matrix.h, labs/lab3/src/OpenCVHW1/
// sparse-matrix.h), whose solvers run on an MI355X through the C ABI in ccp_gs.h.
//
// Same class name, member names, argument meaning and defaults as the reference, so a call site
// such as PhotoMontage.cpp:595-613 or main6.cc:238-249 compiles unchanged against this header
// (link with -lccp_gs).  What differs, deliberately:
//   * gaussSeidel / applyToVector execute on the GPU; there is no CPU fallback — without a
//     usable device they throw std::runtime_error (the reference never throws in release).
//   * gaussSeidel takes two optional trailing arguments the reference lacks: an initial guess
//     (mirrors conjugateGradient's 4th argument, sparse-matrix.h:396) and the sweep ordering.
//     The default ordering is ccp::Ordering::Lexicographic — bit-identical to the reference —
//     and ccp::Ordering::MultiColour is the fast red-black / multi-colour sweep, identical to
//     the reference applied to the colour-major permuted matrix.
//   * insert() on a matrix that is already on the device is applied there incrementally (no re-upload).
//   * insert() implements the semantics the reference's own test asserts (CheckEqual against a
//     dense mirror, main6.cc:19-33).  The reference's memmove counts lose or expose entries on
//     general input (sparse-matrix.h:196-198,219-221); this facade keeps rows sorted and exact.
//   * the host-side storage is written from scratch: same five arrays, same slack semantics
//     (explicit zeros and removed entries become free slots), different code.
//
// conjugateGradient (the solver the blend call sites use today) is provided too; its iterates
// match the reference to rounding (tree-ordered reductions); so are conjugateGradientEigen and conjugateGradientPaper.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <initializer_list>
#include <numeric>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "ccp_gs.h"

#ifdef USE_NAME_SPACE
namespace USE_NAME_SPACE {
#endif

namespace ccp {
enum class Ordering : int { Lexicographic = CCP_ORDER_LEXICOGRAPHIC, MultiColour = CCP_ORDER_MULTICOLOUR };

inline void throw_on(int status, const char *what)
{
    if (status != CCP_OK)
        throw std::runtime_error(std::string(what) + ": " + ccp_status_string(status));
}

// The C ABI reads rows()/cols() doubles through raw pointers: a short vector must fail here, not
// overrun the heap (the reference at least asserts b.size() == n_cols in debug builds, :351).
inline void need(bool ok, const char *what)
{
    if (!ok) throw std::invalid_argument(what);
}
}  // namespace ccp

// ---- free vector helpers (reference: sparse-matrix.h:45-105), serial left-to-right ------------
inline double manhattonDist(const std::vector<double> &a, const std::vector<double> &b)
{
    double acc = 0.0;
    for (std::size_t i = 0; i < a.size(); ++i) acc += std::abs(a[i] - b[i]);
    return acc;
}

template <typename T>
inline double veclen2(const std::vector<T> &a)
{
    double acc = 0.0;
    for (const auto &v : a) acc += v * v;
    return acc;
}

template <typename T>
inline double dotProd(const std::vector<T> &a, const std::vector<T> &b)
{
    double acc = 0.0;
    for (std::size_t i = 0; i < a.size(); ++i) acc += a[i] * b[i];
    return acc;
}

template <typename T>
inline void vecsub(const std::vector<T> &a, const std::vector<T> &b, std::vector<T> &out)
{
    for (std::size_t i = 0; i < a.size(); ++i) out[i] = a[i] - b[i];
}

template <typename T>
inline void vecadd(const std::vector<T> &a, const std::vector<T> &b, T scale_b, std::vector<T> &out)
{
    for (std::size_t i = 0; i < a.size(); ++i) out[i] = a[i] + scale_b * b[i];
}

template <typename T>
inline void vecadd(const std::vector<T> &src, const double inc, std::vector<T> &out)
{
    for (std::size_t i = 0; i < src.size(); ++i) out[i] = src[i] + inc;
}

template <typename T>
inline void vecmul(const std::vector<T> &src, const double scale, std::vector<T> &out)
{
    for (std::size_t i = 0; i < src.size(); ++i) out[i] = src[i] * scale;
}

template <typename T>
inline void vecmul(const std::vector<T> &a, const std::vector<T> &b, std::vector<T> &out)
{
    for (std::size_t i = 0; i < a.size(); ++i) out[i] = a[i] * b[i];
}

template <typename T, typename IndexType = int>
class SparseMatrix {
    static_assert(std::is_same<IndexType, int>::value || std::is_same<IndexType, std::int32_t>::value,
                  "the device path uses int32 indices, as SparseMatrix<double,int> in the reference");
    using Vector = std::vector<T>;

public:
    using Index = IndexType;
    using IndexVector = std::vector<Index>;

    struct Triplet {
        Index row;
        Index col;
        T val;
    };

    SparseMatrix() = default;
    SparseMatrix(SparseMatrix &&o) noexcept { take(std::move(o)); }
    SparseMatrix &operator=(SparseMatrix &&o) noexcept
    {
        if (this != &o) {
            release_device();
            take(std::move(o));
        }
        return *this;
    }
    // lab3 passes the matrix by value (main6.cc:20): copyable; the device handle is per object
    SparseMatrix(const SparseMatrix &o)
        : values_(o.values_), col_offset_(o.col_offset_), row_begin_(o.row_begin_),
          row_num_nze_(o.row_num_nze_), row_space_left_(o.row_space_left_), n_rows_(o.n_rows_), n_cols_(o.n_cols_),
          device_(o.device_), colour_(o.colour_), n_colours_(o.n_colours_)
    {
    }
    SparseMatrix &operator=(const SparseMatrix &o)
    {
        if (this != &o) {
            SparseMatrix tmp(o);
            release_device();
            take(std::move(tmp));
        }
        return *this;
    }
    ~SparseMatrix() { release_device(); }

    Index cols() const { return n_cols_; }
    Index rows() const { return n_rows_; }

    // Which device the solvers use (default 0).
    void setDevice(int device)
    {
        release_device();
        device_ = device;
    }

    T at(Index row, Index col) const
    {
        const Index k = find(row, col);
        return k >= 0 ? values_[k] : T(0);
    }
    T coeff(Index row, Index col) const { return at(row, col); }

    // ---- modification -------------------------------------------------------------------------
    void insert(const T &val, Index row, Index col) { insert(T(val), row, col); }

    // Reference: sparse-matrix.h:183-247.  Once the matrix is on the device an edit inside its shape is
    // forwarded (ccp_csr_insert) and applied there incrementally — the rows touched are re-laid in their
    // slices before the next solve; nothing is uploaded again.  An edit that grows the shape re-uploads.
    void insert(T &&val, Index row, Index col)
    {
        const double as_double = static_cast<double>(val);
        const bool inside = row >= 0 && row < n_rows_ && col >= 0 && col < n_cols_;
        if (val == T(0)) erase(row, col);
        else put(std::move(val), row, col);
        if (dev_ && !dirty_ && inside) {
            // the host arrays already hold the edit: should forwarding it fail, the device copy no longer matches them
            // and the next solve must upload again instead of running on a matrix that silently differs
            const int st = ccp_csr_insert(dev_, row, col, as_double);
            if (st != CCP_OK) dirty_ = true;
            ccp::throw_on(st, "ccp_csr_insert");
        } else {
            dirty_ = true;
        }
    }

    // How the device copy has been maintained so far: whole images built and uploaded, rows patched in
    // place, slices moved to the reserve, images dropped for a rebuild (ccp_csr_edit_stats).
    struct DeviceEditStats { long long edits, image_uploads, rows_patched, slices_relocated, image_rebuilds; };
    DeviceEditStats deviceEditStats() const
    {
        DeviceEditStats st{0, 0, 0, 0, 0};
        if (dev_) {
            int64_t v[5] = {0, 0, 0, 0, 0};
            ccp::throw_on(ccp_csr_edit_stats(dev_, &v[0], &v[1], &v[2], &v[3], &v[4]), "ccp_csr_edit_stats");
            st = DeviceEditStats{v[0], v[1], v[2], v[3], v[4]};
        }
        return st;
    }

    void initializeFromTriplets(Triplet *a, Index cnt)
    {
        for (Index i = 0; i < cnt; ++i) insert(a[i].val, a[i].row, a[i].col);
    }

    // Row-sorted, column-sorted COO; explicit zeros become free slots of their row; the shape is
    // estimated from the data (rows.back()+1, max col + 1), as the reference does (:270-275).
    void initializeFromVector(const IndexVector &rows, IndexVector &&cols, Vector &&vals)
    {
        dirty_ = true;
        values_ = std::move(vals);
        col_offset_ = std::move(cols);
        n_rows_ = rows.empty() ? 0 : rows.back() + 1;
        n_cols_ = 0;
        for (auto c : col_offset_) n_cols_ = std::max(n_cols_, c);
        n_cols_ += 1;
        row_begin_.assign(n_rows_, 0);
        row_num_nze_.assign(n_rows_, 0);
        row_space_left_.assign(n_rows_, 0);
        // each row keeps the span it arrived with; its non-zeros are packed to the front
        std::size_t k = 0;
        while (k < rows.size()) {
            const Index r = rows[k];
            std::size_t e = k;
            while (e < rows.size() && rows[e] == r) ++e;
            std::size_t w = k;
            for (std::size_t q = k; q < e; ++q) {
                if (values_[q] == T(0)) continue;
                values_[w] = values_[q];
                col_offset_[w] = col_offset_[q];
                ++w;
            }
            row_num_nze_[r] = static_cast<Index>(w - k);
            row_space_left_[r] = static_cast<Index>(e - w);
            k = e;
        }
        Index run = 0;
        for (Index r = 0; r < n_rows_; ++r) {
            row_begin_[r] = run;
            run += row_num_nze_[r] + row_space_left_[r];
        }
    }

    void initialize(int row, int col)
    {
        dirty_ = true;
        n_rows_ = row;
        n_cols_ = col;
        values_.clear();
        col_offset_.clear();
        row_begin_.assign(n_rows_, 0);
        row_num_nze_.assign(n_rows_, 0);       // (the reference leaves this unsized, :321-330)
        row_space_left_.assign(n_rows_, 0);
    }

    void initialize(int row, int col, std::initializer_list<T> x)
    {
        Vector v(x);
        IndexVector r(v.size()), c(v.size());
        std::size_t k = 0;
        for (int i = 0; i < row; ++i)
            for (int j = 0; j < col; ++j, ++k) {
                r[k] = i;
                c[k] = j;
            }
        initializeFromVector(r, std::move(c), std::move(v));
    }

    // Raw CSR hand-off from Eigen (ConvertFromEigen, project/src/PhotoMontage/utils.cc:5-15):
    // valuePtr/allocatedSize, outerIndexPtr/outerSize, innerIndexPtr/innerSize,
    // innerNonZeroPtr/outerSize (nullptr for a compressed matrix).
    void initializeFromEigenRowMajor(const T *values, Index n_values, const Index *row_offset, Index n_row_offset,
                                     const Index *col_offset, Index n_col_offset, const Index *non_zeros,
                                     Index n_non_zeros)
    {
        dirty_ = true;
        n_rows_ = n_row_offset;
        n_cols_ = n_col_offset;
        values_.assign(values, values + n_values);
        col_offset_.assign(col_offset, col_offset + n_values);
        row_begin_.assign(row_offset, row_offset + n_row_offset);
        row_num_nze_.assign(n_rows_, 0);
        row_space_left_.assign(n_rows_, 0);
        // rows whose recorded start is the end of the buffer are the trailing all-zero rows
        Index first_empty = n_rows_;
        for (Index i = 0; i < n_rows_; ++i)
            if (row_begin_[i] == n_values) {
                first_empty = i;
                break;
            }
        if (non_zeros != nullptr) {
            for (Index i = 0; i < n_rows_ && i < n_non_zeros; ++i) row_num_nze_[i] = non_zeros[i];
            const Index tail = first_empty > 0 ? row_begin_[first_empty - 1] + row_num_nze_[first_empty - 1] : 0;
            for (Index i = first_empty; i < n_rows_; ++i) {
                row_begin_[i] = tail;
                row_num_nze_[i] = 0;
            }
            for (Index i = 0; i < n_rows_; ++i) {
                const Index next = (i + 1 < n_rows_) ? row_begin_[i + 1] : n_values;
                row_space_left_[i] = std::max<Index>(0, next - row_begin_[i] - row_num_nze_[i]);
            }
        } else {
            for (Index i = 0; i < first_empty; ++i) {
                const Index next = (i + 1 < n_rows_) ? row_begin_[i + 1] : n_values;
                row_num_nze_[i] = next - row_begin_[i];
            }
            // (the reference parks these rows at n_values-1, inside the previous row; not observable)
            for (Index i = first_empty; i < n_rows_; ++i) row_begin_[i] = n_values;
        }
    }

    // ---- solvers (device) ---------------------------------------------------------------------
    // Reference: sparse-matrix.h:350-380.  x0 = all ones unless `initialize` is given.
    std::vector<double> gaussSeidel(const std::vector<double> &b, double epsilon = 1e-6, int max_iteration = 1000,
                                    const std::vector<double> &initialize = std::vector<double>(),
                                    ccp::Ordering ordering = ccp::Ordering::Lexicographic)
    {
        ccp::need(static_cast<Index>(b.size()) == n_rows_, "gaussSeidel: b.size() != rows()");
        ccp::need(initialize.empty() || static_cast<Index>(initialize.size()) == n_rows_, "gaussSeidel: initialize.size() != rows()");
        sync_device();
        std::vector<double> x(b.size(), 1.0);
        ccp_gs_report rep{};
        ccp::throw_on(ccp_csr_gauss_seidel(dev_, b.data(), initialize.empty() ? nullptr : initialize.data(), x.data(),
                                           epsilon, max_iteration, 1, static_cast<int>(ordering), &rep),
                      "ccp_csr_gauss_seidel");
        last_report_ = rep;
        return x;
    }

    // Reference: sparse-matrix.h:396-434 (x0 = 0 unless `initialize` is given).
    std::vector<double> conjugateGradient(const std::vector<double> &b, double epsilon = 1e-16, int max_iteration = 1000,
                                          const std::vector<double> &initialize = std::vector<double>())
    {
        ccp::need(static_cast<Index>(b.size()) == n_rows_, "conjugateGradient: b.size() != rows()");
        ccp::need(initialize.empty() || static_cast<Index>(initialize.size()) == n_rows_, "conjugateGradient: initialize.size() != rows()");
        sync_device();
        std::vector<double> x(b.size(), 0.0);
        ccp_gs_report rep{};
        ccp::throw_on(ccp_csr_conjugate_gradient(dev_, b.data(), initialize.empty() ? nullptr : initialize.data(), x.data(),
                                                 epsilon, max_iteration, &rep),
                      "ccp_csr_conjugate_gradient");
        last_report_ = rep;
        return x;
    }

    // Reference: sparse-matrix.h:436-470 — the same recurrence as conjugateGradient from x0 = 0.
    std::vector<double> conjugateGradientPaper(const std::vector<double> &b, double epsilon = 1e-16, int max_iteration = 1000)
    {
        return conjugateGradient(b, epsilon, max_iteration);
    }

    // Reference: sparse-matrix.h:494-535 (Jacobi-preconditioned, x0 = 0; RunTest, utils.cc:99).
    std::vector<double> conjugateGradientEigen(const std::vector<double> &b, double epsilon = 1e-16, int max_iteration = 180)
    {
        ccp::need(static_cast<Index>(b.size()) == n_rows_, "conjugateGradientEigen: b.size() != rows()");
        sync_device();
        std::vector<double> x(b.size(), 0.0);
        ccp_gs_report rep{};
        ccp::throw_on(ccp_csr_conjugate_gradient_jacobi(dev_, b.data(), x.data(), epsilon, max_iteration, &rep),
                      "ccp_csr_conjugate_gradient_jacobi");
        last_report_ = rep;
        return x;
    }

    // Reference: sparse-matrix.h:382-393.  `out` must be pre-sized to rows().
    void applyToVector(const std::vector<double> &in, std::vector<double> &out)
    {
        ccp::need(static_cast<Index>(in.size()) >= n_cols_, "applyToVector: in.size() < cols()");
        ccp::need(static_cast<Index>(out.size()) >= n_rows_, "applyToVector: out.size() < rows()");
        sync_device();
        ccp::throw_on(ccp_csr_apply_to_vector(dev_, in.data(), out.data()), "ccp_csr_apply_to_vector");
    }

    // sqrt(sum (b - A x)^2 / sum b^2): the metric of SURVEY.md §8d (not in the reference).
    double relativeResidual(const std::vector<double> &b, const std::vector<double> &x)
    {
        ccp::need(static_cast<Index>(b.size()) >= n_rows_ && static_cast<Index>(x.size()) >= n_cols_, "relativeResidual: vector too short");
        sync_device();
        double rr = 0, bb = 0;
        ccp::throw_on(ccp_csr_residual_norm2(dev_, b.data(), x.data(), &rr, &bb), "ccp_csr_residual_norm2");
        return std::sqrt(rr) / std::sqrt(bb);
    }

    // Optional colouring for Ordering::MultiColour (e.g. (x+y)&1 of a grid); empty = greedy.
    void setColouring(const IndexVector &colour, int n_colours)
    {
        colour_ = colour;
        n_colours_ = n_colours;
        dirty_ = true;
    }

    const ccp_gs_report &lastReport() const { return last_report_; }

    // extract diagonal, inverted (reference: sparse-matrix.h:472-491) — host side, kept for CG users
    std::vector<T> extractDiagnolColInv()
    {
        std::vector<T> res(cols(), T(1));
        for (Index i = 0; i < n_rows_; ++i) {
            const Index k = find(i, i);
            if (k >= 0 && values_[k] != T(0)) res[i] = T(1) / values_[k];
        }
        return res;
    }

private:
    void take(SparseMatrix &&o)
    {
        values_ = std::move(o.values_);
        col_offset_ = std::move(o.col_offset_);
        row_begin_ = std::move(o.row_begin_);
        row_num_nze_ = std::move(o.row_num_nze_);
        row_space_left_ = std::move(o.row_space_left_);
        n_rows_ = o.n_rows_;
        n_cols_ = o.n_cols_;
        dev_ = o.dev_;
        o.dev_ = nullptr;
        device_ = o.device_;
        dirty_ = o.dirty_;
        colour_ = std::move(o.colour_);
        n_colours_ = o.n_colours_;
        last_report_ = o.last_report_;
        o.n_rows_ = o.n_cols_ = 0;
        o.dirty_ = true;
    }

    // position of (row, col) among the live entries of the row, or -1
    Index find(Index row, Index col) const
    {
        if (row < 0 || row >= n_rows_ || row_num_nze_[row] == 0) return -1;
        const auto first = col_offset_.begin() + row_begin_[row];
        const auto last = first + row_num_nze_[row];
        const auto it = std::lower_bound(first, last, col);
        return (it != last && *it == col) ? static_cast<Index>(it - col_offset_.begin()) : Index(-1);
    }

    void erase(Index row, Index col)
    {
        const Index k = find(row, col);
        if (k < 0) return;                                 // already zero
        const Index end = row_begin_[row] + row_num_nze_[row];
        std::move(values_.begin() + k + 1, values_.begin() + end, values_.begin() + k);
        std::move(col_offset_.begin() + k + 1, col_offset_.begin() + end, col_offset_.begin() + k);
        --row_num_nze_[row];
        ++row_space_left_[row];
    }

    void put(T &&val, Index row, Index col)
    {
        const Index k = find(row, col);
        if (k >= 0) {                                      // overwrite in place
            values_[k] = std::move(val);
            return;
        }
        if (col >= n_cols_) n_cols_ = col + 1;
        const Index begin = row_begin_[row];
        const Index end = begin + row_num_nze_[row];
        const Index pos = static_cast<Index>(std::lower_bound(col_offset_.begin() + begin, col_offset_.begin() + end, col) -
                                             col_offset_.begin());
        if (row_space_left_[row] > 0) {                    // use a free slot of the row
            std::move_backward(values_.begin() + pos, values_.begin() + end, values_.begin() + end + 1);
            std::move_backward(col_offset_.begin() + pos, col_offset_.begin() + end, col_offset_.begin() + end + 1);
            values_[pos] = std::move(val);
            col_offset_[pos] = col;
            --row_space_left_[row];
        } else {                                           // grow the buffers, later rows shift by one
            values_.insert(values_.begin() + pos, std::move(val));
            col_offset_.insert(col_offset_.begin() + pos, col);
            for (Index r = row + 1; r < n_rows_; ++r) ++row_begin_[r];
        }
        ++row_num_nze_[row];
    }

    void release_device()
    {
        if (dev_) ccp_csr_destroy(dev_);
        dev_ = nullptr;
        dirty_ = true;
    }

    // (re)upload the five arrays when the matrix changed since the last solve
    void sync_device()
    {
        if (!dev_) ccp::throw_on(ccp_csr_create(device_, &dev_), "ccp_csr_create");
        if (!dirty_) return;
        std::vector<double> vals(values_.begin(), values_.end());   // T=int matrices multiply as doubles
        ccp::throw_on(ccp_csr_upload(dev_, n_rows_, n_cols_, static_cast<int64_t>(vals.size()), vals.data(),
                                     col_offset_.data(), row_begin_.data(), row_num_nze_.data()),
                      "ccp_csr_upload");
        if (!colour_.empty())
            ccp::throw_on(ccp_csr_set_colouring(dev_, colour_.data(), n_colours_), "ccp_csr_set_colouring");
        dirty_ = false;
    }

    Vector values_;
    IndexVector col_offset_;
    IndexVector row_begin_;
    IndexVector row_num_nze_;      // live entries per row
    IndexVector row_space_left_;   // free slots per row
    Index n_rows_ = 0;
    Index n_cols_ = 0;

    ccp_csr *dev_ = nullptr;
    int device_ = 0;
    bool dirty_ = true;
    IndexVector colour_;
    int n_colours_ = 0;
    ccp_gs_report last_report_{};
};

#ifdef USE_NAME_SPACE
}
#endif
