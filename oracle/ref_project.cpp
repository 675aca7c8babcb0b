// ref_project.cpp — thin extern "C" driver around the UNMODIFIED reference header
//   /root/reference/project/src/PhotoMontage/sparse-matrix.h  (SparseMatrix<double,int>)
// compiled where it lies (see oracle/Makefile; output goes to oracle/_ref/ only).
// Test infrastructure: used to pin oracle/ccp_oracle.c and to generate tests/golden/.
// Nothing of the reference is copied here; only its public member functions are called.
//
// The header uses memmove / std::abs / sqrt without including their headers (MSVC gets
// them transitively), so the standard headers are included first.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "sparse-matrix.h"   // -I/root/reference/project/src/PhotoMontage

namespace {
using Mat = SparseMatrix<double, int>;

Mat make_eigen(const double *values, int n_values, const int *row_offset, int n_rows,
               const int *col_offset, int n_cols, const int *non_zeros)
{
    Mat m;
    m.initializeFromEigenRowMajor(values, n_values, row_offset, n_rows, col_offset, n_cols,
                                  non_zeros, non_zeros ? n_rows : 0);
    return m;
}
}  // namespace

extern "C" {

// gaussSeidel on a matrix ingested through initializeFromEigenRowMajor (the ConvertFromEigen
// hand-off, project/src/PhotoMontage/utils.cc:5-15).
int ref_gs_eigen(const double *values, int n_values, const int *row_offset, int n_rows,
                 const int *col_offset, int n_cols, const int *non_zeros,
                 const double *b, double epsilon, int max_iteration, double *x_out)
{
    Mat m = make_eigen(values, n_values, row_offset, n_rows, col_offset, n_cols, non_zeros);
    std::vector<double> bv(b, b + n_cols);
    std::vector<double> x = m.gaussSeidel(bv, epsilon, max_iteration);
    std::memcpy(x_out, x.data(), sizeof(double) * x.size());
    return (int)x.size();
}

int ref_spmv_eigen(const double *values, int n_values, const int *row_offset, int n_rows,
                   const int *col_offset, int n_cols, const int *non_zeros,
                   const double *in, double *out)
{
    Mat m = make_eigen(values, n_values, row_offset, n_rows, col_offset, n_cols, non_zeros);
    std::vector<double> iv(in, in + n_cols), ov(n_rows, 0.0);
    m.applyToVector(iv, ov);
    std::memcpy(out, ov.data(), sizeof(double) * ov.size());
    return 0;
}

// Dense scan of at(r,c) plus rows()/cols(): the only public view of the ingested storage.
int ref_dense_eigen(const double *values, int n_values, const int *row_offset, int n_rows,
                    const int *col_offset, int n_cols, const int *non_zeros,
                    double *dense /* n_rows*n_cols */, int *rows_cols /* 2 */)
{
    Mat m = make_eigen(values, n_values, row_offset, n_rows, col_offset, n_cols, non_zeros);
    rows_cols[0] = m.rows();
    rows_cols[1] = m.cols();
    for (int r = 0; r < n_rows; ++r)
        for (int c = 0; c < n_cols; ++c) dense[(size_t)r * n_cols + c] = m.at(r, c);
    return 0;
}

// initializeFromVector path + a sequence of insert(val,row,col); dense scan after every step.
// dense_steps holds (n_ops+1) snapshots of n_rows*n_cols.
int ref_vector_insert_scenario(const int *rows, const int *cols, const double *vals, int count,
                               const int *op_row, const int *op_col, const double *op_val,
                               int n_ops, int n_rows, int n_cols, double *dense_steps)
{
    Mat m;
    std::vector<int> r(rows, rows + count), c(cols, cols + count);
    std::vector<double> v(vals, vals + count);
    m.initializeFromVector(r, std::move(c), std::move(v));
    auto snap = [&](int step) {
        for (int i = 0; i < n_rows; ++i)
            for (int j = 0; j < n_cols; ++j)
                dense_steps[((size_t)step * n_rows + i) * n_cols + j] =
                    (i < m.rows() && j < m.cols()) ? m.at(i, j) : 0.0;
    };
    snap(0);
    for (int k = 0; k < n_ops; ++k) {
        m.insert(op_val[k], op_row[k], op_col[k]);
        snap(k + 1);
    }
    return 0;
}

int ref_gs_vector(const int *rows, const int *cols, const double *vals, int count,
                  const double *b, int n, double epsilon, int max_iteration, double *x_out)
{
    Mat m;
    std::vector<int> r(rows, rows + count), c(cols, cols + count);
    std::vector<double> v(vals, vals + count);
    m.initializeFromVector(r, std::move(c), std::move(v));
    std::vector<double> bv(b, b + n);
    std::vector<double> x = m.gaussSeidel(bv, epsilon, max_iteration);
    std::memcpy(x_out, x.data(), sizeof(double) * x.size());
    return (int)x.size();
}

// conjugateGradient with optional initial guess (sparse-matrix.h:396-434) — the solver the
// blend call site uses today; kept for the "next" row of SURVEY §8f.
int ref_cg_eigen(const double *values, int n_values, const int *row_offset, int n_rows,
                 const int *col_offset, int n_cols, const int *non_zeros,
                 const double *b, double epsilon, int max_iteration, const double *init,
                 double *x_out)
{
    Mat m = make_eigen(values, n_values, row_offset, n_rows, col_offset, n_cols, non_zeros);
    std::vector<double> bv(b, b + n_cols);
    std::vector<double> iv;
    if (init) iv.assign(init, init + n_cols);
    std::vector<double> x = m.conjugateGradient(bv, epsilon, max_iteration, iv);
    std::memcpy(x_out, x.data(), sizeof(double) * x.size());
    return (int)x.size();
}

// conjugateGradientEigen (sparse-matrix.h:494-535), the Jacobi-preconditioned variant RunTest uses
int ref_cg_jacobi(const double *values, int n_values, const int *row_offset, int n_rows,
                  const int *col_offset, int n_cols, const int *non_zeros,
                  const double *b, double epsilon, int max_iteration, double *x_out)
{
    Mat m = make_eigen(values, n_values, row_offset, n_rows, col_offset, n_cols, non_zeros);
    std::vector<double> bv(b, b + n_cols);
    std::vector<double> x = m.conjugateGradientEigen(bv, epsilon, max_iteration);
    std::memcpy(x_out, x.data(), sizeof(double) * x.size());
    return (int)x.size();
}

double ref_manhatton_dist(const double *a, const double *b, int n)
{
    std::vector<double> av(a, a + n), bv(b, b + n);
    return manhattonDist(av, bv);
}

double ref_veclen2(const double *a, int n)
{
    std::vector<double> av(a, a + n);
    return veclen2(av);
}

double ref_dot_prod(const double *a, const double *b, int n)
{
    std::vector<double> av(a, a + n), bv(b, b + n);
    return dotProd(av, bv);
}

// Timed solve for bench.py's cpu_baseline leg: steady_clock around gaussSeidel only, like
// the reference Timer (labs/lab4/src/OpenCVHW1/utils.h:275-304).  Returns seconds.
double ref_gs_eigen_timed(const double *values, int n_values, const int *row_offset, int n_rows,
                          const int *col_offset, int n_cols, const double *b, int max_iteration,
                          double *x_out);
// The same with the time of the ingest (initializeFromEigenRowMajor + the copy of b) reported beside it.
double ref_gs_eigen_timed_phases(const double *values, int n_values, const int *row_offset, int n_rows,
                                 const int *col_offset, int n_cols, const double *b, int max_iteration,
                                 double *ingest_seconds);
}

#include <chrono>
extern "C" double ref_gs_eigen_timed(const double *values, int n_values, const int *row_offset,
                                     int n_rows, const int *col_offset, int n_cols,
                                     const double *b, int max_iteration, double *x_out)
{
    Mat m = make_eigen(values, n_values, row_offset, n_rows, col_offset, n_cols, nullptr);
    std::vector<double> bv(b, b + n_cols);
    auto t0 = std::chrono::steady_clock::now();
    std::vector<double> x = m.gaussSeidel(bv, 0.0, max_iteration);
    auto t1 = std::chrono::steady_clock::now();
    if (x_out) std::memcpy(x_out, x.data(), sizeof(double) * x.size());
    return std::chrono::duration<double>(t1 - t0).count();
}

extern "C" double ref_gs_eigen_timed_phases(const double *values, int n_values, const int *row_offset,
                                            int n_rows, const int *col_offset, int n_cols,
                                            const double *b, int max_iteration, double *ingest_seconds)
{
    auto i0 = std::chrono::steady_clock::now();
    Mat m = make_eigen(values, n_values, row_offset, n_rows, col_offset, n_cols, nullptr);
    std::vector<double> bv(b, b + n_cols);
    auto t0 = std::chrono::steady_clock::now();
    std::vector<double> x = m.gaussSeidel(bv, 0.0, max_iteration);
    auto t1 = std::chrono::steady_clock::now();
    if (ingest_seconds) *ingest_seconds = std::chrono::duration<double>(t0 - i0).count();
    return std::chrono::duration<double>(t1 - t0).count();
}

// The header's IndexType template parameter at 64 bits: the only way the reference can hold SolveChannel's 16384 x 16384
// system (1,342,046,211 stored entries): with the default int, getNearestIndex's `(end + idx) / 2` (sparse-matrix.h:636)
// overflows for every row whose entries lie beyond position 2^30, the search wanders over gigabytes and at(i, i) comes
// back wrong.  Same unmodified header, same member functions.
extern "C" double ref_gs_eigen_timed_phases_i64(const double *values, long long n_values, const long long *row_offset,
                                                long long n_rows, const long long *col_offset, long long n_cols,
                                                const double *b, int max_iteration, double *ingest_seconds, double *x_out)
{
    using Mat64 = SparseMatrix<double, long long>;
    auto i0 = std::chrono::steady_clock::now();
    Mat64 m;
    m.initializeFromEigenRowMajor(values, n_values, row_offset, n_rows, col_offset, n_cols, nullptr, 0);
    std::vector<double> bv(b, b + n_cols);
    auto t0 = std::chrono::steady_clock::now();
    std::vector<double> x = m.gaussSeidel(bv, 0.0, max_iteration);
    auto t1 = std::chrono::steady_clock::now();
    if (ingest_seconds) *ingest_seconds = std::chrono::duration<double>(t0 - i0).count();
    if (x_out) std::memcpy(x_out, x.data(), sizeof(double) * x.size());
    return std::chrono::duration<double>(t1 - t0).count();
}
