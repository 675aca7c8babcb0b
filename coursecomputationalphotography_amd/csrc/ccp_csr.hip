// ccp_csr.hip — C ABI of the general slack-CSR path (Gauss-Seidel, SpMV, residual).
// See include/ccp_gs.h.  Host side: schedule construction (colouring / level scheduling) and the
// sliced-ELL re-tiling; device side: ccp_csr_kernels.hpp.
#include "ccp_csr_kernels.hpp"
#include "ccp_csr_region.hpp"
#include "ccp_cg.hpp"
#include "ccp_comm.hpp"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstdint>
#include <numeric>
#include <vector>
#include <memory>
#include <thread>
#include <atomic>
#include <mutex>
#include <chrono>
#include <unordered_map>

using namespace ccp;

namespace {

// Host array that is NOT value-initialised when it is sized (std::vector<T>::resize zero-fills — one thread touching
// 2.5 GB of fresh pages cost ccp_csr_upload 0.3 s at 41.75 M unknowns); every element is written by the parallel
// copy that follows a resize().
template <typename T>
struct HostArr {
    std::unique_ptr<T[]> p;
    size_t n = 0;
    void resize(size_t count)
    {
        p.reset(count ? new T[count] : nullptr);
        n = count;
    }
    size_t size() const { return n; }
    T *data() { return p.get(); }
    const T *data() const { return p.get(); }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
    void swap(HostArr &o)
    {
        p.swap(o.p);
        std::swap(n, o.n);
    }
    void assign(const T *first, const T *last)
    {
        resize((size_t)(last - first));
        if (n) std::memcpy(p.get(), first, n * sizeof(T));
    }
};

// Sliced-ELL image of the matrix for one row schedule, resident on the device.
struct Schedule {
    bool built = false;
    int n_groups = 0;                    // colours or levels, swept in order
    std::vector<int> group_slice_ptr;    // [n_groups+1] slices of each group
    std::vector<long> group_block_off;   // [n_groups+1] partial-sum block offsets
    int n_slices = 0;
    DevBuf<long> slice_off;
    DevBuf<int> slice_width, slice_row0, slice_rows, cols, perm;
    DevBuf<int> group_ptr_dev;           // group_slice_ptr on the device (one-workgroup and pipelined solvers)
    DevBuf<long> group_block_off_dev;
    int max_group_slices = 0;
    int level_span = -1;                 // lexicographic: max level difference of two coupled rows (-1: not a level schedule)
    DevBuf<double> vals;
    // host mirrors for incremental edits (ccp_csr_insert): where a row lives in the image and how much
    // room its slice has.  A slice is laid out with kSliceSlack spare entry columns and the arrays end in
    // a reserve, so that a growing row is patched in place, or its slice moved to the reserve, without
    // touching the rest of the image.
    std::vector<int> inv;                // original row -> permuted row
    std::vector<long> gstart;            // [n_groups+1] first permuted row of each group
    std::vector<int> group_of;           // original row -> group
    std::vector<long> h_soff;
    std::vector<int> h_swidth, h_scap;
    long entries_used = 0, entries_cap = 0;
    bool sort_by_permuted = false;
    void reset()
    {
        built = false;
        inv.clear(); gstart.clear(); group_of.clear(); h_soff.clear(); h_swidth.clear(); h_scap.clear();
        entries_used = entries_cap = 0;
        n_groups = n_slices = 0;
        group_slice_ptr.clear();
        group_block_off.clear();
        slice_off.release(); slice_width.release(); slice_row0.release(); slice_rows.release();
        cols.release(); perm.release(); vals.release(); group_ptr_dev.release(); group_block_off_dev.release();
        max_group_slices = 0;
        level_span = -1;
    }
    SellView view() const
    {
        return SellView{slice_off.p, slice_width.p, slice_row0.p, slice_rows.p, cols.p, vals.p};
    }
};

}  // namespace

struct ccp_csr {
    int device = 0;
    hipStream_t stream = nullptr;
    bool uploaded = false;
    int n_rows = 0, n_cols = 0;
    // host copy of the live entries in compressed form (slack removed), storage order kept
    std::vector<long> row_ptr;
    HostArr<int> col;
    HostArr<double> val;
    // the same three arrays resident on the device (uploaded once per compact host copy: the region recognition and
    // the construction of the sliced-ELL images run there)
    DevBuf<long> d_row_ptr;
    DevBuf<int> d_col;
    DevBuf<double> d_val;
    bool dev_csr_valid = false;            // row offsets and columns are resident
    bool dev_val_valid = false;            // ... and the values (only the images need them)
    // ccp_csr_upload starts the device copy of the compact host arrays on a thread of its own (pageable source: the
    // copy occupies its caller) and returns; whoever needs the device arrays, or is about to replace the host arrays,
    // joins it first (wait_device_upload)
    std::thread uploader;
    int uploader_status = CCP_OK;
    std::vector<int> user_colour;
    int user_n_colours = 0;
    std::vector<int> used_colour;          // the colouring the multi-colour schedule was built from
    int used_n_colours = 0;
    // Incremental edits (ccp_csr_insert; SparseMatrix::insert, sparse-matrix.h:183-247).  An edited row's
    // whole current content lives in `overlay` (the compact arrays above stay as uploaded until a schedule
    // has to be built again: materialise()); `touched` rows are re-laid in the device images before the
    // next solve (flush_edits).
    struct RowContent { std::vector<int> col; std::vector<double> val; };
    std::unordered_map<int, RowContent> overlay;
    std::vector<int> touched;
    long stat_uploads = 0, stat_rows_patched = 0, stat_slices_relocated = 0, stat_schedule_rebuilds = 0, stat_edits = 0;
    long stat_eager_skipped = 0;           // background copies of the structure skipped because the device allocation failed
    DevBuf<long> patch_base;
    DevBuf<int> patch_cap, patch_cols;
    DevBuf<long> patch_off;
    DevBuf<double> patch_vals;
    Schedule natural;        // identity order, one group: SpMV / residual
    Schedule multicolour;    // colour-major, columns sorted by permuted index (reference on P A P^T)
    Schedule lexicographic;  // level-major, original storage order kept inside a row
    DevBuf<double> x, b, tmp, partial;
    DevBuf<double> pipe_partial, pipe_eps, pipe_snap;   // pipelined level schedule: step sums per sweep, snapshot
    bool allow_pipeline = true;            // CCP_GS_PIPELINE=0: one launch per level and sweep
    DevBuf<double> cg_p, cg_p2, cg_ap;     // conjugate-gradient work vectors (allocated on first use; cg_p2: the fused loop's second direction buffer)
    DevBuf<double> cg_inv, cg_partial2;    // Jacobi-preconditioned variant: 1/a_ii, second pair of partial-sum regions
    DevBuf<CgState> cg_state;
    DevBuf<CsrSolveState> state;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // structured-matrix detection: -1 unknown, 0 not the SolveChannel Poisson matrix, else W
    int poisson_w = -1, poisson_h = 0;
    int compact_poisson_w = -1, compact_poisson_h = 0;   // the same question for the compact host copy alone (cached)
    ccp_grid *grid = nullptr;              // matrix-free twin used when the matrix is that Poisson matrix
    bool allow_structured = true;
    // raster-region detection (region_state): the matrix is the 5-point Laplacian of a pixel region with zero
    // Dirichlet values around it (diagonal 4, -1 to every 4-neighbour inside) — BASELINE configs[4].  Then a
    // Dirichlet-mask grid on a canvas the region is embedded in sweeps it matrix-free; `where` maps unknown i
    // to its element of the canvas planes.
    int region_state = -1;                 // -1 unknown, 0 no, 1 yes, 2 yes for the reference's order only (the canvas' parity
                                           //    is a 2-colouring of the region, but not the colouring the colour-ordered sweep uses)
    bool region_wants_two_colouring = false;   // state 0 only because the resolved colouring has more than two colours
    ccp_grid *region_grid = nullptr;
    DevBuf<long> region_where;
    std::vector<int> region_colour;        // (x + y) & 1 of the embedding: the colouring the grid sweep realises
    std::vector<int> greedy_colour;        // the library's greedy colouring of the matrix as it is now (computed once: the recognition
    int greedy_n_colours = 0;              //    and the schedule both ask for it); cleared by every upload, colouring and edit
    std::vector<int> auto_colour;          // no colouring from the caller and the matrix is a raster region: the canvas parity IS the
                                           // library's colouring (kept for the stored-matrix path too, as long as it stays proper)
    int region_w = 0, region_h = 0;
    bool allow_region = true;              // CCP_GS_MASKED=0 keeps such matrices on the sliced-ELL path
    int region_values_ok = -1;             // every stored value is 4 (diagonal) or -1: -1 unknown, 0 no, 1 yes (compact host copy)
    int region_solves = 0;                 // solves on the region grid so far (its tiling is tuned at the second)
    bool region_tuned = false;
    int last_path = 0;                     // CCP_PATH_* of the last solve
    long last_launches = 0;                // sweep launches of the last solve on a grid twin
    bool edited = false;                   // ccp_csr_insert changed the matrix since the upload
    bool allow_one_block = true;           // CCP_GS_ONE_BLOCK=0: always one launch per group
    // Row block of a matrix distributed over the ranks of a communicator (ccp_csr_upload_rows; SURVEY §8e: "row-block by
    // unknown index with a halo index list").  The handle then holds an EXTENDED square system: the columns its rows
    // reference outside the block ("ghosts") become empty rows of their own, numbered with the owned rows in the order
    // of their global indices —
    //     [ghosts below row_begin | owned rows | ghosts above the block]
    // — so every schedule, kernel and buffer of the single-GPU path applies unchanged: an empty row has no diagonal and
    // is skipped by the sweep (sparse-matrix.h:361), colour-major order of the extended rows is the global colour-major
    // order restricted to them (a row sums its couplings in the order the one-GPU sweep does: same bits), and the ghosts
    // a peer owns are one contiguous range per colour that the exchange receives straight into x.
    struct RowBlock {
        bool on = false;
        ccp_comm *comm = nullptr;
        int row_begin = 0, n_local = 0, n_global = 0, n_lo = 0, n_ghost = 0, n_colours = 0, peers = 0;
        std::vector<int> ghost;                              // global indices of the ghosts, ascending
        // natural order (applyToVector, residual): one message per peer
        std::vector<int> nat_send_cnt, nat_send_off, nat_recv_cnt, nat_recv_pos;      // [world]
        DevBuf<int> nat_send_pos;                            // extended indices to gather, peer by peer
        // colour-major order (the sweep): one message per peer and colour
        std::vector<int> col_send_cnt, col_send_off, col_recv_cnt, col_recv_pos;      // [n_colours * world]
        std::vector<long> col_seg_off;                       // [n_colours + 1] segment of col_send_pos / sendbuf per colour
        DevBuf<int> col_send_pos;                            // positions in the colour-major x to gather
        DevBuf<double> sendbuf;
        long nat_total = 0;                                  // entries of nat_send_pos
        long exchanges = 0, values_sent = 0;
        // Overlap: per colour the slices that hold a row some peer references ("edge") are swept first; their values
        // travel on stream_comm while the other slices of the colour are swept.  Off when the edge slices are more
        // than a quarter of the image (nothing left to hide behind) or CCP_GS_ROWS_OVERLAP=0.
        bool overlap = false;
        DevBuf<int> slice_list;                              // per colour: edge slices, then the others
        std::vector<int> edge_off, edge_cnt, inner_off, inner_cnt;    // [n_colours] ranges of slice_list
        std::vector<long> part_off;                          // [n_colours + 1] partial-sum blocks of a colour (edge launch, then inner)
        hipStream_t stream_comm = nullptr;
        hipEvent_t ev_edge = nullptr, ev_comm = nullptr;
        void reset()
        {
            if (stream_comm) (void)hipStreamDestroy(stream_comm);
            if (ev_edge) (void)hipEventDestroy(ev_edge);
            if (ev_comm) (void)hipEventDestroy(ev_comm);
            stream_comm = nullptr;
            ev_edge = ev_comm = nullptr;
            overlap = false;
            slice_list.release();
            edge_off.clear(); edge_cnt.clear(); inner_off.clear(); inner_cnt.clear(); part_off.clear();
            on = false;
            comm = nullptr;
            row_begin = n_local = n_global = n_lo = n_ghost = n_colours = peers = 0;
            ghost.clear();
            nat_send_cnt.clear(); nat_send_off.clear(); nat_recv_cnt.clear(); nat_recv_pos.clear();
            col_send_cnt.clear(); col_send_off.clear(); col_recv_cnt.clear(); col_recv_pos.clear(); col_seg_off.clear();
            nat_send_pos.release(); col_send_pos.release(); sendbuf.release();
            exchanges = values_sent = nat_total = 0;
        }
    } rb;
};

namespace {

// Host-side schedule construction is embarrassingly parallel per slice / per row: plain std::thread
// workers over contiguous ranges (no OpenMP runtime to depend on).
template <typename F>
void parallel_ranges(long n, long min_per_thread, F &&body)
{
    unsigned hw = std::thread::hardware_concurrency();
    long workers = std::max<long>(1, std::min<long>(hw ? hw : 1, 32));
    workers = std::max<long>(1, std::min<long>(workers, n / std::max<long>(1, min_per_thread)));
    if (workers <= 1) {
        body(0L, n);
        return;
    }
    std::vector<std::thread> pool;
    const long chunk = (n + workers - 1) / workers;
    for (long w = 0; w < workers; ++w) {
        const long lo = w * chunk, hi = std::min(n, lo + chunk);
        if (lo >= hi) break;
        pool.emplace_back([&body, lo, hi] { body(lo, hi); });
    }
    for (auto &t : pool) t.join();
}

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int bind(ccp_csr *m)
{
    if (!m) return CCP_ERR_BAD_ARG;
    if (hipSetDevice(m->device) != hipSuccess) return CCP_ERR_NO_DEVICE;
    return CCP_OK;
}

// lower[i] = rows j < i coupled to i (a_ij != 0 or a_ji != 0, structurally).
void build_lower(const ccp_csr *m, std::vector<long> &lptr, std::vector<int> &lidx)
{
    const int n = m->n_rows;
    std::vector<long> cnt((size_t)n + 1, 0);
    for (int i = 0; i < n; ++i)
        for (long k = m->row_ptr[i]; k < m->row_ptr[i + 1]; ++k) {
            const int c = m->col[k];
            if (c == i || c < 0 || c >= n) continue;
            cnt[(size_t)std::max(i, c) + 1]++;
        }
    lptr.assign((size_t)n + 1, 0);
    for (int i = 0; i < n; ++i) lptr[i + 1] = lptr[i] + cnt[i + 1];
    lidx.assign((size_t)lptr[n], 0);
    std::vector<long> fill(lptr.begin(), lptr.end() - 1);
    for (int i = 0; i < n; ++i)
        for (long k = m->row_ptr[i]; k < m->row_ptr[i + 1]; ++k) {
            const int c = m->col[k];
            if (c == i || c < 0 || c >= n) continue;
            const int hi = std::max(i, c), lo = std::min(i, c);
            lidx[fill[hi]++] = lo;
        }
}

// Greedy colouring in row order (first colour not used by an already coloured neighbour).
int greedy_colouring(const std::vector<long> &lptr, const std::vector<int> &lidx, int n, std::vector<int> &colour)
{
    colour.assign(n, 0);
    int n_colours = n ? 1 : 0;
    std::vector<int> mark;
    for (int i = 0; i < n; ++i) {
        const long deg = lptr[i + 1] - lptr[i];
        mark.assign((size_t)deg + 2, 0);
        for (long k = lptr[i]; k < lptr[i + 1]; ++k) {
            const int cj = colour[lidx[k]];
            if (cj <= deg) mark[cj] = 1;
        }
        int c = 0;
        while (mark[c]) ++c;
        colour[i] = c;
        n_colours = std::max(n_colours, c + 1);
    }
    return n_colours;
}

// level[i] = 1 + max level of coupled rows j < i: rows of one level are mutually uncoupled and
// sweeping the levels in order reproduces the reference's index-order sweep exactly.
int level_schedule(const std::vector<long> &lptr, const std::vector<int> &lidx, int n, std::vector<int> &level)
{
    level.assign(n, 0);
    int n_levels = n ? 1 : 0;
    for (int i = 0; i < n; ++i) {
        int lv = 0;
        for (long k = lptr[i]; k < lptr[i + 1]; ++k) lv = std::max(lv, level[lidx[k]] + 1);
        level[i] = lv;
        n_levels = std::max(n_levels, lv + 1);
    }
    return n_levels;
}

// A potential with phi(i) - phi(j) == 1 for EVERY coupled pair j < i, if one exists (5-point-like
// stencils on any region: phi = x + y up to a constant per connected component).  Rows of equal phi
// are uncoupled and ascending phi is a topological order of the sweep's dependences, so it is as
// valid a schedule as the levels — and its span is 1, which is what lets the sweeps pipeline
// (k_sell_gs_pipe) however irregular the region.  Levels alone can differ by hundreds across one
// coupling on an irregular mask.  Returns the number of groups, 0 if no such potential exists.
int unit_potential(const std::vector<long> &lptr, const std::vector<int> &lidx, int n, std::vector<int> &phi)
{
    // upper adjacency (j -> i > j) from the lower lists
    std::vector<long> uptr((size_t)n + 1, 0);
    for (int i = 0; i < n; ++i)
        for (long k = lptr[i]; k < lptr[i + 1]; ++k) uptr[(size_t)lidx[k] + 1]++;
    for (int i = 0; i < n; ++i) uptr[i + 1] += uptr[i];
    std::vector<int> uidx((size_t)uptr[n]);
    {
        std::vector<long> fill(uptr.begin(), uptr.end() - 1);
        for (int i = 0; i < n; ++i)
            for (long k = lptr[i]; k < lptr[i + 1]; ++k) uidx[fill[lidx[k]]++] = i;
    }
    const int unset = INT32_MIN;
    phi.assign(n, unset);
    std::vector<int> queue;
    queue.reserve(1024);
    int max_phi = 0;
    for (int root = 0; root < n; ++root) {
        if (phi[root] != unset) continue;
        // breadth-first over the component; potentials relative to the root, shifted to min 0 afterwards
        queue.clear();
        queue.push_back(root);
        phi[root] = 0;
        int lo = 0;
        for (size_t q = 0; q < queue.size(); ++q) {
            const int i = queue[q];
            for (long k = lptr[i]; k < lptr[i + 1]; ++k) {
                const int j = lidx[k];
                if (phi[j] == unset) {
                    phi[j] = phi[i] - 1;
                    lo = std::min(lo, phi[j]);
                    queue.push_back(j);
                } else if (phi[j] != phi[i] - 1) return 0;
            }
            for (long k = uptr[i]; k < uptr[i + 1]; ++k) {
                const int j = uidx[k];
                if (phi[j] == unset) {
                    phi[j] = phi[i] + 1;
                    queue.push_back(j);
                } else if (phi[j] != phi[i] + 1) return 0;
            }
        }
        for (int i : queue) {
            phi[i] -= lo;
            max_phi = std::max(max_phi, phi[i]);
        }
    }
    return n ? max_phi + 1 : 0;
}

template <typename T>
int upload_vec(DevBuf<T> &d, const std::vector<T> &h, hipStream_t s)
{
    CCP_TRY(d.alloc(h.size()));
    if (!h.empty()) CCP_HIP(hipMemcpyAsync(d.p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
    return CCP_OK;
}

// Build the sliced-ELL image for rows ordered by `group` (stable: ascending row index inside a
// group).  sort_by_permuted: order a row's entries by their PERMUTED column (what the reference
// would see on P A P^T); otherwise keep the original storage order.
constexpr int kSliceSlack = 2;          // spare entry columns per slice (never read: kernels stop at the live width)

int materialise(ccp_csr *m);

int build_schedule_host(ccp_csr *m, Schedule &sc, const std::vector<int> &group, int n_groups, bool sort_by_permuted)
{
    const int n = m->n_rows;
    const double t_begin = now_s();
    std::vector<long> gcount((size_t)n_groups + 1, 0);
    for (int i = 0; i < n; ++i) gcount[(size_t)group[i] + 1]++;
    for (int g = 0; g < n_groups; ++g) gcount[g + 1] += gcount[g];
    std::vector<int> perm(n), inv(n);               // perm[new] = old
    {
        std::vector<long> fill(gcount.begin(), gcount.end() - 1);
        for (int i = 0; i < n; ++i) perm[fill[group[i]]++] = i;
        for (int r = 0; r < n; ++r) inv[perm[r]] = r;
    }
    // columns of a non-square matrix beyond n_rows keep their index (only square systems are
    // solved; SpMV uses the identity schedule where inv is the identity anyway)
    auto map_col = [&](int c) { return (c >= 0 && c < n) ? inv[c] : c; };

    const double t_perm = now_s();
    std::vector<int> srow0, srows, swidth, scap;
    std::vector<long> soff;
    sc.group_slice_ptr.assign((size_t)n_groups + 1, 0);
    sc.group_block_off.assign((size_t)n_groups + 1, 0);
    long total = 0;
    for (int g = 0; g < n_groups; ++g) {
        for (long r = gcount[g]; r < gcount[g + 1]; r += kWave) {
            const int rows = (int)std::min<long>(kWave, gcount[g + 1] - r);
            int width = 0;
            for (int t = 0; t < rows; ++t) {
                const int old = perm[r + t];
                width = std::max(width, (int)(m->row_ptr[old + 1] - m->row_ptr[old]));
            }
            srow0.push_back((int)r);
            srows.push_back(rows);
            swidth.push_back(width);
            scap.push_back(width + kSliceSlack);
            soff.push_back(total);
            total += (long)(width + kSliceSlack) * kWave;
        }
        sc.group_slice_ptr[g + 1] = (int)srow0.size();
        const int slices = sc.group_slice_ptr[g + 1] - sc.group_slice_ptr[g];
        sc.group_block_off[g + 1] = sc.group_block_off[g] + (slices + kBlock / kWave - 1) / (kBlock / kWave);
    }
    sc.n_slices = (int)srow0.size();
    sc.n_groups = n_groups;
    const double t_fill0 = now_s();
    // uninitialised on purpose: every element is written below, by the thread that first touches its page
    // the arrays end in a reserve that relocated (grown) slices move into
    const long reserve = std::max<long>(1L << 16, total / 64);
    const size_t n_entries = (size_t)std::max<long>(total, 1);
    std::unique_ptr<int[]> cols(new int[n_entries]);
    std::unique_ptr<double[]> vals(new double[n_entries]);
    parallel_ranges(sc.n_slices, 256, [&](long s_lo, long s_hi) {
        std::vector<std::pair<int, double>> tmp;
        for (long s = s_lo; s < s_hi; ++s) {
            // padding: column -1, value 0
            const size_t base = (size_t)soff[s], len = (size_t)scap[s] * kWave;
            std::fill(cols.get() + base, cols.get() + base + len, -1);
            std::fill(vals.get() + base, vals.get() + base + len, 0.0);
            for (int t = 0; t < srows[s]; ++t) {
                const int old = perm[srow0[s] + t];
                tmp.clear();
                for (long k = m->row_ptr[old]; k < m->row_ptr[old + 1]; ++k)
                    tmp.emplace_back(map_col(m->col[k]), m->val[k]);
                if (sort_by_permuted) {
                    // stable insertion sort: rows are a handful of entries
                    for (size_t a = 1; a < tmp.size(); ++a) {
                        const auto v = tmp[a];
                        size_t b = a;
                        while (b > 0 && tmp[b - 1].first > v.first) {
                            tmp[b] = tmp[b - 1];
                            --b;
                        }
                        tmp[b] = v;
                    }
                }
                for (size_t k = 0; k < tmp.size(); ++k) {
                    cols[base + k * kWave + t] = tmp[k].first;
                    vals[base + k * kWave + t] = tmp[k].second;
                }
            }
        }
    });
    const double t_fill1 = now_s();
    CCP_TRY(upload_vec(sc.slice_off, soff, m->stream));
    CCP_TRY(upload_vec(sc.slice_width, swidth, m->stream));
    CCP_TRY(upload_vec(sc.slice_row0, srow0, m->stream));
    CCP_TRY(upload_vec(sc.slice_rows, srows, m->stream));
    if (total == 0) {
        cols[0] = -1;
        vals[0] = 0.0;
    }
    CCP_TRY(sc.cols.alloc(n_entries + (size_t)reserve));
    CCP_TRY(sc.vals.alloc(n_entries + (size_t)reserve));
    CCP_HIP(hipMemcpyAsync(sc.cols.p, cols.get(), n_entries * sizeof(int), hipMemcpyHostToDevice, m->stream));
    CCP_HIP(hipMemcpyAsync(sc.vals.p, vals.get(), n_entries * sizeof(double), hipMemcpyHostToDevice, m->stream));
    CCP_TRY(upload_vec(sc.perm, perm, m->stream));
    CCP_TRY(upload_vec(sc.group_ptr_dev, sc.group_slice_ptr, m->stream));
    CCP_TRY(upload_vec(sc.group_block_off_dev, sc.group_block_off, m->stream));
    sc.max_group_slices = 0;
    for (int g = 0; g < n_groups; ++g)
        sc.max_group_slices = std::max(sc.max_group_slices, sc.group_slice_ptr[g + 1] - sc.group_slice_ptr[g]);
    CCP_HIP(hipStreamSynchronize(m->stream));      // host vectors die at scope exit
    if (getenv("CCP_GS_DEBUG"))
        fprintf(stderr, "[ccp_gs] schedule of %d slices: permutation %.3f s, slice table %.3f s, fill %.3f s, upload %.3f s\n",
                sc.n_slices, t_perm - t_begin, t_fill0 - t_perm, t_fill1 - t_fill0, now_s() - t_fill1);
    sc.inv.swap(inv);
    sc.gstart.swap(gcount);
    sc.group_of = group;
    sc.h_soff.swap(soff);
    sc.h_swidth.swap(swidth);
    sc.h_scap.swap(scap);
    sc.entries_used = total;
    sc.entries_cap = (long)n_entries + reserve;
    sc.sort_by_permuted = sort_by_permuted;
    m->stat_uploads++;
    sc.built = true;
    return CCP_OK;
}

// The compact host copy on the device (once per copy).
void wait_device_upload(ccp_csr *m)
{
    if (m->uploader.joinable()) {
        m->uploader.join();
        if (m->uploader_status == CCP_OK) m->dev_csr_valid = true;     // (structure only: the values travel with the first image)
    }
}

// Start copying the STRUCTURE of the compact host copy (row offsets, columns) to the device in the background: the
// raster-region recognition reads it at the first solve.  Best effort — ensure_device_csr copies whatever is missing
// when something needs it, so a failed allocation here never fails the upload — and only where it can pay:
//  * a matrix whose row 0 is SolveChannel's [3, -1@1, -1@W] is examined on the host (detect_poisson) and swept
//    matrix-free on its grid twin: nothing on the device ever reads the stored matrix (18 GB at 16384^2);
//  * the values travel only when an image of the matrix is built (ensure_device_csr(m, true)).
// release_device_csr gives the copy back once a grid twin has taken the matrix over.
int start_device_upload(ccp_csr *m)
{
    wait_device_upload(m);
    m->dev_csr_valid = m->dev_val_valid = false;
    m->d_row_ptr.release();
    m->d_col.release();
    m->d_val.release();
    const int n = m->n_rows;
    const long nnz = m->row_ptr[n];
    if (n <= 0 || nnz < (1L << 22)) return CCP_OK;        // small matrices: copied when first needed
    if (m->allow_structured && n == m->n_cols && m->row_ptr[1] - m->row_ptr[0] == 3 && m->val[m->row_ptr[0]] == 3.0) return CCP_OK;
    const bool refuse = getenv("CCP_GS_FAIL_EAGER_ALLOC") && atoi(getenv("CCP_GS_FAIL_EAGER_ALLOC")) != 0;   // test seam
    if (refuse || m->d_row_ptr.alloc((size_t)n + 1) != CCP_OK || m->d_col.alloc((size_t)nnz) != CCP_OK) {
        (void)hipGetLastError();                          // the failed hipMalloc must not poison a later error check
        m->d_row_ptr.release();
        m->d_col.release();
        m->stat_eager_skipped++;
        return CCP_OK;
    }
    m->uploader_status = CCP_OK;
    m->uploader = std::thread([m, n, nnz] {
        hipStream_t s = nullptr;
        bool ok = hipSetDevice(m->device) == hipSuccess && hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess;
        ok = ok && hipMemcpyAsync(m->d_row_ptr.p, m->row_ptr.data(), sizeof(long) * ((size_t)n + 1), hipMemcpyHostToDevice, s) == hipSuccess;
        ok = ok && hipMemcpyAsync(m->d_col.p, m->col.data(), sizeof(int) * (size_t)nnz, hipMemcpyHostToDevice, s) == hipSuccess;
        ok = ok && hipStreamSynchronize(s) == hipSuccess;
        if (s) (void)hipStreamDestroy(s);
        if (!ok) m->uploader_status = CCP_ERR_HIP;
    });
    return CCP_OK;
}

// A grid twin sweeps, applies and checks the matrix from now on: the device copy of the stored matrix is dead weight
// (2.8 GB at the 8192^2 mask).  An edit or a solve that needs an image copies it again (ensure_device_csr).
void release_device_csr(ccp_csr *m)
{
    wait_device_upload(m);
    m->dev_csr_valid = m->dev_val_valid = false;
    m->d_row_ptr.release();
    m->d_col.release();
    m->d_val.release();
}

int ensure_device_csr(ccp_csr *m, bool with_values)
{
    wait_device_upload(m);
    const int n = m->n_rows;
    const long nnz = m->row_ptr[n];
    if (!m->dev_csr_valid) {
        CCP_TRY(m->d_row_ptr.alloc((size_t)n + 1));
        CCP_TRY(m->d_col.alloc((size_t)std::max<long>(nnz, 1)));
        CCP_HIP(hipMemcpyAsync(m->d_row_ptr.p, m->row_ptr.data(), sizeof(long) * ((size_t)n + 1), hipMemcpyHostToDevice, m->stream));
        if (nnz) CCP_HIP(hipMemcpyAsync(m->d_col.p, m->col.data(), sizeof(int) * (size_t)nnz, hipMemcpyHostToDevice, m->stream));
        m->dev_val_valid = false;
    }
    if (with_values && !m->dev_val_valid) {
        CCP_TRY(m->d_val.alloc((size_t)std::max<long>(nnz, 1)));
        if (nnz) CCP_HIP(hipMemcpyAsync(m->d_val.p, m->val.data(), sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice, m->stream));
    }
    CCP_HIP(hipStreamSynchronize(m->stream));
    m->dev_csr_valid = true;
    if (with_values) m->dev_val_valid = true;
    return CCP_OK;
}

// build_schedule with everything that is per row or per entry on the device: the stable ordering of the rows by group
// (radix sort), the inverse permutation, the slice widths and offsets (scan) and the fill of the image.  The host keeps
// what is per group and per slice (a few hundred thousand entries) and the mirrors incremental edits need.
// 41.75 M unknowns, 208 M entries: 0.65 s on the host (permutation, slice table, fill, 4.4 GB upload) -> ~30 ms.
int build_schedule_device(ccp_csr *m, Schedule &sc, const std::vector<int> &group, int n_groups, bool sort_by_permuted)
{
    const int n = m->n_rows;
    const double t_begin = now_s();
    CCP_TRY(ensure_device_csr(m, true));
    const double t_csr = now_s();
    hipStream_t s = m->stream;
    // rows per group (host: the caller's vector), slices per group
    std::vector<long> gcount((size_t)n_groups + 1, 0);
    {
        const unsigned hw = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
        const long chunk = ((long)n + hw - 1) / hw;
        std::vector<std::vector<long>> local(hw, std::vector<long>((size_t)n_groups, 0));
        std::atomic<int> bad{0};
        parallel_ranges(hw, 1, [&](long lo, long hi) {
            for (long w = lo; w < hi; ++w)
                for (long i = w * chunk; i < std::min<long>(n, (w + 1) * chunk); ++i) {
                    const int g = group[i];
                    if (g < 0 || g >= n_groups) bad.store(1, std::memory_order_relaxed);
                    else local[w][g]++;
                }
        });
        if (bad.load()) return CCP_ERR_STATE;
        for (unsigned w = 0; w < hw; ++w)
            for (int g = 0; g < n_groups; ++g) gcount[(size_t)g + 1] += local[w][g];
        for (int g = 0; g < n_groups; ++g) gcount[g + 1] += gcount[g];
    }
    std::vector<int> srow0, srows;
    sc.group_slice_ptr.assign((size_t)n_groups + 1, 0);
    sc.group_block_off.assign((size_t)n_groups + 1, 0);
    for (int g = 0; g < n_groups; ++g) {
        for (long r = gcount[g]; r < gcount[g + 1]; r += kWave) {
            srow0.push_back((int)r);
            srows.push_back((int)std::min<long>(kWave, gcount[g + 1] - r));
        }
        sc.group_slice_ptr[g + 1] = (int)srow0.size();
        const int slices = sc.group_slice_ptr[g + 1] - sc.group_slice_ptr[g];
        sc.group_block_off[g + 1] = sc.group_block_off[g] + (slices + kBlock / kWave - 1) / (kBlock / kWave);
    }
    sc.n_slices = (int)srow0.size();
    sc.n_groups = n_groups;
    const int n_slices = sc.n_slices;
    // the stable order of the rows by group
    DevBuf<int> d_group, d_keys, d_iota, d_inv;
    DevBuf<unsigned char> d_tmp;
    CCP_TRY(sc.perm.alloc((size_t)n));
    CCP_TRY(d_inv.alloc((size_t)n));
    const unsigned nb = (unsigned)std::max<long>(1, std::min<long>(8192, ((long)n + kBlock - 1) / kBlock));
    if (n_groups <= 1) {
        hipLaunchKernelGGL(k_iota, dim3(nb), dim3(kBlock), 0, s, sc.perm.p, (long)n);
    } else {
        CCP_TRY(d_group.alloc((size_t)n));
        CCP_TRY(d_keys.alloc((size_t)n));
        CCP_TRY(d_iota.alloc((size_t)n));
        CCP_HIP(hipMemcpyAsync(d_group.p, group.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_iota, dim3(nb), dim3(kBlock), 0, s, d_iota.p, (long)n);
        int bits = 1;
        while ((1L << bits) < n_groups) ++bits;
        size_t bytes = 0;
        CCP_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, d_group.p, d_keys.p, d_iota.p, sc.perm.p, n, 0, bits, s));
        CCP_TRY(d_tmp.alloc(bytes));
        CCP_HIP(hipcub::DeviceRadixSort::SortPairs(d_tmp.p, bytes, d_group.p, d_keys.p, d_iota.p, sc.perm.p, n, 0, bits, s));
    }
    hipLaunchKernelGGL(k_invert_perm, dim3(nb), dim3(kBlock), 0, s, sc.perm.p, d_inv.p, (long)n);
    CCP_HIP(hipGetLastError());
    // slices: widths, offsets
    CCP_TRY(upload_vec(sc.slice_row0, srow0, s));
    CCP_TRY(upload_vec(sc.slice_rows, srows, s));
    CCP_TRY(sc.slice_width.alloc((size_t)std::max(n_slices, 1)));
    CCP_TRY(sc.slice_off.alloc((size_t)std::max(n_slices, 1)));
    DevBuf<long> d_cells;
    CCP_TRY(d_cells.alloc((size_t)std::max(n_slices, 1)));
    const unsigned sb = (unsigned)std::max(1, (n_slices + kBlock / kWave - 1) / (kBlock / kWave));
    std::vector<int> swidth((size_t)n_slices), scap((size_t)n_slices);
    std::vector<long> soff((size_t)n_slices);
    long total = 0;
    if (n_slices) {
        hipLaunchKernelGGL(k_slice_widths, dim3(sb), dim3(kBlock), 0, s, m->d_row_ptr.p, sc.perm.p, sc.slice_row0.p, sc.slice_rows.p, n_slices,
                           kSliceSlack, sc.slice_width.p, d_cells.p);
        size_t bytes = 0;
        CCP_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, d_cells.p, sc.slice_off.p, n_slices, s));
        if (d_tmp.n < bytes) CCP_TRY(d_tmp.alloc(bytes));
        CCP_HIP(hipcub::DeviceScan::ExclusiveSum(d_tmp.p, bytes, d_cells.p, sc.slice_off.p, n_slices, s));
        CCP_HIP(hipMemcpyAsync(swidth.data(), sc.slice_width.p, sizeof(int) * (size_t)n_slices, hipMemcpyDeviceToHost, s));
        CCP_HIP(hipMemcpyAsync(soff.data(), sc.slice_off.p, sizeof(long) * (size_t)n_slices, hipMemcpyDeviceToHost, s));
        CCP_HIP(hipStreamSynchronize(s));
        for (int k = 0; k < n_slices; ++k) scap[k] = swidth[k] + kSliceSlack;
        total = soff[n_slices - 1] + (long)scap[n_slices - 1] * kWave;
    }
    const double t_table = now_s();
    // the image; the arrays end in a reserve that relocated (grown) slices move into
    const long reserve = std::max<long>(1L << 16, total / 64);
    const size_t n_entries = (size_t)std::max<long>(total, 1);
    CCP_TRY(sc.cols.alloc(n_entries + (size_t)reserve));
    CCP_TRY(sc.vals.alloc(n_entries + (size_t)reserve));
    if (total == 0) {
        CCP_HIP(hipMemsetAsync(sc.cols.p, 0xff, sizeof(int), s));
        CCP_HIP(hipMemsetAsync(sc.vals.p, 0, sizeof(double), s));
    } else if (sort_by_permuted) {
        hipLaunchKernelGGL((k_slice_fill<true>), dim3(sb), dim3(kBlock), 0, s, m->d_row_ptr.p, m->d_col.p, m->d_val.p, sc.perm.p, d_inv.p, n,
                           sc.slice_row0.p, sc.slice_rows.p, sc.slice_width.p, sc.slice_off.p, n_slices, kSliceSlack, sc.cols.p, sc.vals.p);
    } else {
        hipLaunchKernelGGL((k_slice_fill<false>), dim3(sb), dim3(kBlock), 0, s, m->d_row_ptr.p, m->d_col.p, m->d_val.p, sc.perm.p, d_inv.p, n,
                           sc.slice_row0.p, sc.slice_rows.p, sc.slice_width.p, sc.slice_off.p, n_slices, kSliceSlack, sc.cols.p, sc.vals.p);
    }
    CCP_HIP(hipGetLastError());
    CCP_TRY(upload_vec(sc.group_ptr_dev, sc.group_slice_ptr, s));
    CCP_TRY(upload_vec(sc.group_block_off_dev, sc.group_block_off, s));
    sc.max_group_slices = 0;
    for (int g = 0; g < n_groups; ++g)
        sc.max_group_slices = std::max(sc.max_group_slices, sc.group_slice_ptr[g + 1] - sc.group_slice_ptr[g]);
    // host mirrors for incremental edits
    sc.inv.resize((size_t)n);
    if (n) CCP_HIP(hipMemcpyAsync(sc.inv.data(), d_inv.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, s));
    CCP_HIP(hipStreamSynchronize(s));
    if (getenv("CCP_GS_DEBUG"))
        fprintf(stderr, "[ccp_gs] schedule of %d slices on the device: matrix upload %.3f s, ordering + slice table %.3f s, fill + mirrors %.3f s\n",
                n_slices, t_csr - t_begin, t_table - t_csr, now_s() - t_table);
    sc.gstart.swap(gcount);
    sc.group_of = group;
    sc.h_soff.swap(soff);
    sc.h_swidth.swap(swidth);
    sc.h_scap.swap(scap);
    sc.entries_used = total;
    sc.entries_cap = (long)n_entries + reserve;
    sc.sort_by_permuted = sort_by_permuted;
    m->stat_uploads++;
    sc.built = true;
    return CCP_OK;
}

int flush_edits(ccp_csr *m);

// patch_edits: rows edited since the upload (the overlay) are NOT merged into the compact host copy first (one pass
// over the whole matrix and a second upload of it: 0.7 s at 41.75 M unknowns for a thousand edited rows) — the image
// is built from the compact copy as it stands and the edited rows are re-laid in it as patches, the way later edits
// are.  Only for schedules whose row order does not depend on the edited structure (identity, a given colouring).
int build_schedule(ccp_csr *m, Schedule &sc, const std::vector<int> &group, int n_groups, bool sort_by_permuted, bool patch_edits = false)
{
    static const bool on_host = getenv("CCP_GS_SCHEDULE_HOST") && atoi(getenv("CCP_GS_SCHEDULE_HOST")) != 0;
    const bool patch = patch_edits && !m->overlay.empty();
    if (!patch) CCP_TRY(materialise(m));             // pending edits go into the compact host copy first
    if (!on_host && m->n_rows > 0) CCP_TRY(build_schedule_device(m, sc, group, n_groups, sort_by_permuted));
    else CCP_TRY(build_schedule_host(m, sc, group, n_groups, sort_by_permuted));
    if (patch) {
        for (const auto &kv : m->overlay) m->touched.push_back(kv.first);
        CCP_TRY(flush_edits(m));
        if (!sc.built) {
            // an edited row does not fit the order after all: merge and build again
            CCP_TRY(materialise(m));
            if (!on_host && m->n_rows > 0) CCP_TRY(build_schedule_device(m, sc, group, n_groups, sort_by_permuted));
            else CCP_TRY(build_schedule_host(m, sc, group, n_groups, sort_by_permuted));
        }
    }
    return CCP_OK;
}

// Current content of row i: the overlay if the row was edited since the upload, else the compact copy.
struct RowView { const int *col; const double *val; long len; };
RowView row_view(const ccp_csr *m, int i)
{
    auto it = m->overlay.find(i);
    if (it != m->overlay.end()) return RowView{it->second.col.data(), it->second.val.data(), (long)it->second.col.size()};
    const long a = m->row_ptr[i];
    return RowView{m->col.data() + a, m->val.data() + a, m->row_ptr[i + 1] - a};
}

// Merge the overlay into the compact host copy (one pass over the matrix; only when a schedule has to be
// built from scratch again — never on the edit-and-solve path).
int materialise(ccp_csr *m)
{
    if (m->overlay.empty()) return CCP_OK;
    wait_device_upload(m);                 // (it reads the arrays replaced below)
    const int n = m->n_rows;
    std::vector<long> ptr((size_t)n + 1, 0);
    for (int i = 0; i < n; ++i) ptr[i + 1] = ptr[i] + row_view(m, i).len;
    HostArr<int> col;
    HostArr<double> val;
    col.resize((size_t)ptr[n]);
    val.resize((size_t)ptr[n]);
    parallel_ranges(n, 1 << 15, [&](long lo, long hi) {
        for (long i = lo; i < hi; ++i) {
            const RowView r = row_view(m, (int)i);
            if (!r.len) continue;
            std::memcpy(&col[ptr[i]], r.col, sizeof(int) * (size_t)r.len);
            std::memcpy(&val[ptr[i]], r.val, sizeof(double) * (size_t)r.len);
        }
    });
    m->row_ptr.swap(ptr);
    m->col.swap(col);
    m->val.swap(val);
    m->overlay.clear();
    m->compact_poisson_w = -1;
    m->poisson_w = -1;
    m->region_values_ok = -1;
    m->dev_csr_valid = false;
    return CCP_OK;
}

// Is this exactly the matrix SolveChannel assembles (closed form) for some W x H?
// Row 0 of that matrix is [3 @0, -1 @1, -1 @W], which fixes W; then every row is compared.
bool row_is_poisson(int W, int H, int i, const RowView &rv)
{
    const int x = i % W, y = i / W;
    auto cell = [&](int cx, int cy) { return cx >= 0 && cy >= 0 && cx < W - 1 && cy < H - 1; };
    const bool up = cell(x, y - 1), left = cell(x - 1, y), here = cell(x, y);
    const int diag = (int)up + (int)left + 2 * (int)here + (i == 0 ? 1 : 0);
    long k = 0;
    const long end = rv.len;
    auto next_is = [&](int c, double v) {
        if (k >= end || rv.col[k] != c || rv.val[k] != v) return false;
        ++k;
        return true;
    };
    if (up && !next_is(i - W, -1.0)) return false;
    if (left && !next_is(i - 1, -1.0)) return false;
    if (diag && !next_is(i, (double)diag)) return false;
    if (here && (!next_is(i + 1, -1.0) || !next_is(i + W, -1.0))) return false;
    return k == end;
}

void detect_poisson(ccp_csr *m)
{
    if (m->poisson_w >= 0) return;
    m->poisson_w = 0;
    const int n = m->n_rows;
    if (n < 4 || m->n_cols != n) return;
    // the compact host copy is examined once (cached); rows edited since are examined on top of it
    if (m->compact_poisson_w < 0) {
        m->compact_poisson_w = 0;
        const long len0 = m->row_ptr[1] - m->row_ptr[0];
        const int W = len0 == 3 ? m->col[m->row_ptr[0] + 2] : 0;
        if (W >= 2 && n % W == 0 && n / W >= 2) {
            const int H = n / W;
            std::atomic<int> bad{0};
            parallel_ranges(n, 1 << 16, [&](long lo, long hi) {
                for (long i = lo; i < hi && !bad.load(std::memory_order_relaxed); ++i) {
                    const long a = m->row_ptr[i];
                    if (!row_is_poisson(W, H, (int)i, RowView{m->col.data() + a, m->val.data() + a, m->row_ptr[i + 1] - a}))
                        bad.store(1, std::memory_order_relaxed);
                }
            });
            if (!bad.load()) {
                m->compact_poisson_w = W;
                m->compact_poisson_h = H;
            }
        }
    }
    if (m->compact_poisson_w <= 0) return;
    for (const auto &kv : m->overlay)
        if (!row_is_poisson(m->compact_poisson_w, m->compact_poisson_h, kv.first,
                            RowView{kv.second.col.data(), kv.second.val.data(), (long)kv.second.col.size()}))
            return;
    m->poisson_w = m->compact_poisson_w;
    m->poisson_h = m->compact_poisson_h;
}

// ---- raster-region recognition ------------------------------------------------------------------------
// Is the matrix the 5-point Laplacian of a pixel region with zero Dirichlet values around it — diagonal 4,
// -1 to every 4-neighbour inside the region, unknowns numbered in raster order (what a gradient-domain
// blend restricted to a brush / label region assembles; BASELINE configs[4])?  The CSR carries no
// coordinates, so they are reconstructed: consecutive unknowns coupled to each other form horizontal runs,
// a coupling to an earlier, non-adjacent unknown is the pixel above; a weighted union-find over the runs
// turns the vertical couplings into relative positions, one rigid piece per connected component.  The
// pieces are laid out on a canvas (one empty pixel between them, shifted by one where needed so that
// (x + y) & 1 is the row's colour) and the result is VERIFIED row by row against the matrix — every
// coupling a 4-neighbour pair, every 4-neighbour pair a coupling — so a wrong guess (a run that continues
// straight down instead of to the right is indistinguishable locally) only ever costs the fast path.
struct RegionEmbedding {
    int W = 0, H = 0;
    std::vector<int> x, y;               // per unknown
};

// ---- what is per RUN (shared by the host and the device statement of the recognition) ------------------------------
// A link says: the unknown `du` places into run A lies directly above the unknown `di` places into run B.
struct RunLink { int at, A, B, du, di; };

// Relative positions of the runs from the links (weighted union-find: position of a run's first pixel relative to its
// root's), one rigid piece per connected component; the pieces are shelf-packed on a canvas with one empty pixel around
// each and shifted by one where the colour of a piece's first unknown asks for the other parity.  x0/y0: canvas position
// of every run's first pixel.  false: the links contradict each other, or the canvas would be unreasonably large.
bool layout_runs(int n, const std::vector<int> &run_start, const std::vector<int> &run_colour, std::vector<RunLink> &links,
                 std::vector<int> &x0, std::vector<int> &y0, int &canvas_w_out, int &canvas_h_out)
{
    const int n_runs = (int)run_start.size();
    std::sort(links.begin(), links.end(), [](const RunLink &a, const RunLink &b) { return a.at < b.at; });
    std::vector<int> parent((size_t)n_runs), ox((size_t)n_runs, 0), oy((size_t)n_runs, 0);
    for (int r = 0; r < n_runs; ++r) parent[r] = r;
    auto find = [&](int r, int &px, int &py) {
        int root = r, sx = 0, sy = 0;                       // pass 1: the root and r's position relative to it
        while (parent[root] != root) {
            sx += ox[root];
            sy += oy[root];
            root = parent[root];
        }
        int cur = r, cx = sx, cy = sy;                      // pass 2: hang the whole path under the root
        while (cur != root) {
            const int next = parent[cur], nx = cx - ox[cur], ny = cy - oy[cur];
            parent[cur] = root;
            ox[cur] = cx;
            oy[cur] = cy;
            cur = next;
            cx = nx;
            cy = ny;
        }
        px = sx;
        py = sy;
        return root;
    };
    for (const RunLink &l : links) {
        int ax, ay, bx, by;
        const int ra = find(l.A, ax, ay), rb = find(l.B, bx, by);
        const int want_x = ax + l.du - l.di, want_y = ay + 1;            // where B's first pixel must sit
        if (ra == rb) {
            if (bx != want_x || by != want_y) return false;
        } else {
            parent[rb] = ra;
            ox[rb] = want_x - bx;
            oy[rb] = want_y - by;
        }
    }
    // components: bounding boxes in root-relative coordinates, first run for the colour parity
    std::vector<int> comp_of_root((size_t)n_runs, -1), rx((size_t)n_runs), ry((size_t)n_runs), rroot((size_t)n_runs);
    struct Box { int minx, maxx, miny, maxy, first_run; int px, py; };
    std::vector<Box> box;
    for (int r = 0; r < n_runs; ++r) {
        int px, py;
        const int root = find(r, px, py);
        rx[r] = px;
        ry[r] = py;
        rroot[r] = root;
        const int len = (r + 1 < n_runs ? run_start[r + 1] : n) - run_start[r];
        if (comp_of_root[root] < 0) {
            comp_of_root[root] = (int)box.size();
            box.push_back(Box{px, px + len - 1, py, py, r, 0, 0});
        } else {
            Box &b = box[comp_of_root[root]];
            b.minx = std::min(b.minx, px);
            b.maxx = std::max(b.maxx, px + len - 1);
            b.miny = std::min(b.miny, py);
            b.maxy = std::max(b.maxy, py);
        }
    }
    // shelf layout with one empty pixel around every piece
    long area = 0;
    int widest = 0;
    for (const Box &b : box) {
        area += (long)(b.maxx - b.minx + 3) * (b.maxy - b.miny + 3);
        widest = std::max(widest, b.maxx - b.minx + 3);
    }
    if (area > std::max<long>(8L * n, 1L << 22) || area > 0x7fffffffL) return false;
    const int target_w = std::max(widest + 1, (int)std::ceil(std::sqrt((double)area)) + 2);
    int cur_x = 0, shelf_y = 0, shelf_h = 0, canvas_w = 0;
    for (Box &b : box) {
        const int w = b.maxx - b.minx + 3, h = b.maxy - b.miny + 3;
        if (cur_x > 0 && cur_x + w + 1 > target_w) {
            shelf_y += shelf_h;
            cur_x = 0;
            shelf_h = 0;
        }
        // first pixel of the piece (root-relative coordinates) and the parity its colour asks for
        const int fx = rx[b.first_run], fy = ry[b.first_run];
        int px = cur_x + 1 - b.minx, py = shelf_y + 1 - b.miny;            // translation of the piece
        if ((((fx + px) + (fy + py)) & 1) != (run_colour[b.first_run] & 1)) ++px;   // one pixel to the right fixes the parity
        b.px = px;
        b.py = py;
        cur_x += w + 1;
        shelf_h = std::max(shelf_h, h);
        canvas_w = std::max(canvas_w, cur_x + 1);
    }
    const int canvas_h = shelf_y + shelf_h + 1;
    if ((long)canvas_w * canvas_h > 0x7fffffffL) return false;
    x0.assign((size_t)n_runs, 0);
    y0.assign((size_t)n_runs, 0);
    for (int r = 0; r < n_runs; ++r) {
        const Box &b = box[comp_of_root[rroot[r]]];
        x0[r] = rx[r] + b.px;
        y0[r] = ry[r] + b.py;
    }
    canvas_w_out = canvas_w;
    canvas_h_out = canvas_h;
    return true;
}

bool embed_region(const ccp_csr *m, const std::vector<int> &colour, RegionEmbedding &E)
{
    const int n = m->n_rows;
    if (n < 1 || m->n_cols != n || !m->overlay.empty()) return false;
    const bool dbg = getenv("CCP_GS_DEBUG") != nullptr;
    double t_mark = now_s();
    auto lap = [&](const char *what) {
        if (dbg) {
            const double t = now_s();
            fprintf(stderr, "[ccp_gs]   embed_region: %s %.3f s\n", what, t - t_mark);
            t_mark = t;
        }
    };
    std::vector<int> up((size_t)n, -1);
    std::vector<unsigned char> has_left((size_t)n, 0);
    std::atomic<int> bad{0};
    parallel_ranges(n, 1 << 15, [&](long lo, long hi) {
        for (long i = lo; i < hi && !bad.load(std::memory_order_relaxed); ++i) {
            int lower[2], n_lower = 0, n_upper = 0;
            bool diag = false, ok = true;
            for (long k = m->row_ptr[i]; k < m->row_ptr[i + 1]; ++k) {
                const int c = m->col[k];
                const double v = m->val[k];
                if (c == (int)i) {
                    ok &= !diag && v == 4.0;
                    diag = true;
                } else if (c < (int)i) {
                    ok &= v == -1.0 && n_lower < 2;
                    if (n_lower < 2) lower[n_lower] = c;
                    ++n_lower;
                } else {
                    ok &= v == -1.0;
                    ++n_upper;
                }
            }
            ok &= diag && n_upper <= 2;
            if (ok && n_lower == 2) {
                ok = lower[1] == (int)i - 1 && lower[0] < (int)i - 1;
                up[i] = lower[0];
                has_left[i] = 1;
            } else if (ok && n_lower == 1) {
                if (lower[0] == (int)i - 1) has_left[i] = 1;     // taken as the pixel to the left (see above)
                else up[i] = lower[0];
            }
            if (!ok) bad.store(1, std::memory_order_relaxed);
        }
    });
    if (bad.load()) return false;
    lap("row classification");
    // runs
    std::vector<int> run_of((size_t)n), run_start;
    for (int i = 0; i < n; ++i) {
        if (!has_left[i]) run_start.push_back(i);
        run_of[i] = (int)run_start.size() - 1;
    }
    const int n_runs = (int)run_start.size();
    lap("runs");
    // one constraint per pair of runs is enough: a link parallel to its left neighbour's is skipped
    std::vector<RunLink> links;
    for (int i = 0; i < n; ++i) {
        const int u = up[i];
        if (u < 0) continue;
        if (has_left[i] && up[i - 1] == u - 1 && u >= 1 && has_left[u]) continue;
        const int A = run_of[u], B = run_of[i];
        links.push_back(RunLink{i, A, B, u - run_start[A], i - run_start[B]});
    }
    std::vector<int> run_colour((size_t)n_runs), x0, y0;
    for (int r = 0; r < n_runs; ++r) run_colour[r] = colour[run_start[r]];
    if (!layout_runs(n, run_start, run_colour, links, x0, y0, E.W, E.H)) return false;
    lap("union-find over the vertical couplings, components, layout");
    const int canvas_w = E.W, canvas_h = E.H;
    E.x.assign((size_t)n, 0);
    E.y.assign((size_t)n, 0);
    parallel_ranges(n, 1 << 16, [&](long lo, long hi) {
        for (long i = lo; i < hi; ++i) {
            const int r = run_of[i];
            E.x[i] = x0[r] + (int)(i - run_start[r]);
            E.y[i] = y0[r];
        }
    });
    lap("components, layout, coordinates");
    // verification against the matrix
    std::vector<int> ident((size_t)canvas_w * canvas_h, -1);
    for (int i = 0; i < n; ++i) {
        if (E.x[i] < 1 || E.y[i] < 1 || E.x[i] >= canvas_w - 1 || E.y[i] >= canvas_h - 1) return false;
        ident[(size_t)E.y[i] * canvas_w + E.x[i]] = i;
    }
    parallel_ranges(n, 1 << 15, [&](long lo, long hi) {
        for (long i = lo; i < hi && !bad.load(std::memory_order_relaxed); ++i) {
            const size_t at = (size_t)E.y[i] * canvas_w + E.x[i];
            bool ok = ident[at] == (int)i && ((E.x[i] + E.y[i]) & 1) == (colour[i] & 1);
            const int nb[4] = {ident[at - canvas_w], ident[at - 1], ident[at + 1], ident[at + canvas_w]};
            int present = 0, matched = 0;
            for (int q = 0; q < 4; ++q) present += nb[q] >= 0;
            for (long k = m->row_ptr[i]; k < m->row_ptr[i + 1]; ++k) {
                const int c = m->col[k];
                if (c == (int)i) continue;
                matched += (c == nb[0]) + (c == nb[1]) + (c == nb[2]) + (c == nb[3]);
            }
            const long off_diag = m->row_ptr[i + 1] - m->row_ptr[i] - 1;
            ok &= matched == present && off_diag == present;
            if (!ok) bad.store(1, std::memory_order_relaxed);
        }
    });
    lap("verification");
    return !bad.load();
}

// Do the stored values fit a region matrix (4 on the diagonal, -1 elsewhere)?  One parallel pass over the compact host
// copy, cached until the copy changes.
bool region_values_fit(ccp_csr *m)
{
    if (m->region_values_ok < 0) {
        std::atomic<int> bad{0};
        parallel_ranges(m->n_rows, 1 << 15, [&](long lo, long hi) {
            for (long i = lo; i < hi && !bad.load(std::memory_order_relaxed); ++i)
                for (long k = m->row_ptr[i]; k < m->row_ptr[i + 1]; ++k)
                    if (m->val[k] != (m->col[k] == (int)i ? 4.0 : -1.0)) {
                        bad.store(1, std::memory_order_relaxed);
                        break;
                    }
        });
        m->region_values_ok = bad.load() ? 0 : 1;
    }
    return m->region_values_ok == 1;
}

// The recognition with everything that is per unknown on the device (ccp_csr_region.hpp).  On success the region grid
// exists with its mask set, m->region_where holds every unknown's element of the canvas planes and W, H the canvas.
// false: not a region matrix (or a HIP failure: the caller then simply does not take the fast path).
bool embed_region_device(ccp_csr *m, std::vector<int> &colour, int &W, int &H, bool free_parity = false)
{
    const int n = m->n_rows;
    if (n < 1 || m->n_cols != n || !m->overlay.empty()) return false;
    const bool dbg = getenv("CCP_GS_DEBUG") != nullptr;
    double t_mark = now_s();
    auto lap = [&](const char *what) {
        if (dbg) {
            (void)hipStreamSynchronize(m->stream);
            const double t = now_s();
            fprintf(stderr, "[ccp_gs]   embed_region_device: %s %.3f s\n", what, t - t_mark);
            t_mark = t;
        }
    };
    if (!region_values_fit(m)) return false;
    lap("values");
    hipStream_t s = m->stream;
    if (ensure_device_csr(m, false) != CCP_OK) return false;
    struct Alias { const long *p; } d_ptr{m->d_row_ptr.p};
    struct AliasI { const int *p; } d_col{m->d_col.p};
    DevBuf<int> d_colour, d_up, d_flag, d_run_id, d_bad;
    DevBuf<unsigned char> d_left;
    if (d_colour.alloc((size_t)n) != CCP_OK || d_up.alloc((size_t)n) != CCP_OK || d_flag.alloc((size_t)n) != CCP_OK ||
        d_run_id.alloc((size_t)n) != CCP_OK || d_left.alloc((size_t)n) != CCP_OK || d_bad.alloc(4) != CCP_OK)
        return false;
    auto hip_ok = [](hipError_t e) { return e == hipSuccess; };
    // free_parity: no colouring to honour — every piece is laid out with its first pixel on even parity and the canvas
    // parity becomes the colouring (k_region_place writes it, it comes back below)
    if (!hip_ok(free_parity ? hipMemsetAsync(d_colour.p, 0, sizeof(int) * (size_t)n, s)
                            : hipMemcpyAsync(d_colour.p, colour.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, s)) ||
        !hip_ok(hipMemsetAsync(d_bad.p, 0, sizeof(int) * 4, s)))
        return false;
    lap("matrix (if not resident yet) and colours to the device");
    const unsigned blocks = (unsigned)(((long)n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_region_classify, dim3(blocks), dim3(kBlock), 0, s, d_ptr.p, d_col.p, n, d_up.p, d_left.p, d_flag.p, d_bad.p);
    // runs: inclusive scan of the run-start flags
    size_t tmp_bytes = 0;
    if (!hip_ok(hipcub::DeviceScan::InclusiveSum(nullptr, tmp_bytes, d_flag.p, d_run_id.p, n, s))) return false;
    DevBuf<unsigned char> d_tmp;
    if (d_tmp.alloc(tmp_bytes) != CCP_OK) return false;
    if (!hip_ok(hipcub::DeviceScan::InclusiveSum(d_tmp.p, tmp_bytes, d_flag.p, d_run_id.p, n, s))) return false;
    int head[2] = {0, 0};                                    // bad flag, number of runs
    if (!hip_ok(hipMemcpyAsync(&head[0], d_bad.p, sizeof(int), hipMemcpyDeviceToHost, s)) ||
        !hip_ok(hipMemcpyAsync(&head[1], d_run_id.p + (n - 1), sizeof(int), hipMemcpyDeviceToHost, s)) ||
        !hip_ok(hipStreamSynchronize(s)))
        return false;
    lap("classification, runs");
    if (head[0]) return false;
    const int n_runs = head[1];
    if (n_runs < 1) return false;
    DevBuf<int> d_run_start, d_run_colour, d_link, d_link_at;
    const int link_cap = n;                                  // (at most one link per unknown)
    if (d_run_start.alloc((size_t)n_runs) != CCP_OK || d_run_colour.alloc((size_t)n_runs) != CCP_OK ||
        d_link.alloc(4 * (size_t)link_cap) != CCP_OK || d_link_at.alloc((size_t)link_cap) != CCP_OK)
        return false;
    hipLaunchKernelGGL(k_region_run_starts, dim3(blocks), dim3(kBlock), 0, s, d_run_id.p, d_left.p, d_colour.p, n, d_run_start.p, d_run_colour.p);
    hipLaunchKernelGGL(k_region_links, dim3(blocks), dim3(kBlock), 0, s, d_up.p, d_left.p, d_run_id.p, d_run_start.p, n, d_bad.p + 1, link_cap,
                       d_link.p, d_link_at.p);
    int n_links = 0;
    if (!hip_ok(hipMemcpyAsync(&n_links, d_bad.p + 1, sizeof(int), hipMemcpyDeviceToHost, s)) || !hip_ok(hipStreamSynchronize(s))) return false;
    if (n_links < 0 || n_links > link_cap) return false;
    std::vector<int> run_start((size_t)n_runs), run_colour((size_t)n_runs), link4(4 * (size_t)n_links), link_at((size_t)n_links);
    if (!hip_ok(hipMemcpyAsync(run_start.data(), d_run_start.p, sizeof(int) * (size_t)n_runs, hipMemcpyDeviceToHost, s)) ||
        !hip_ok(hipMemcpyAsync(run_colour.data(), d_run_colour.p, sizeof(int) * (size_t)n_runs, hipMemcpyDeviceToHost, s)) ||
        (n_links && !hip_ok(hipMemcpyAsync(link4.data(), d_link.p, sizeof(int) * 4 * (size_t)n_links, hipMemcpyDeviceToHost, s))) ||
        (n_links && !hip_ok(hipMemcpyAsync(link_at.data(), d_link_at.p, sizeof(int) * (size_t)n_links, hipMemcpyDeviceToHost, s))) ||
        !hip_ok(hipStreamSynchronize(s)))
        return false;
    d_link.release();
    d_link_at.release();
    d_up.release();
    d_flag.release();
    lap("run starts, links, their download");
    std::vector<RunLink> links((size_t)n_links);
    for (int p = 0; p < n_links; ++p) links[p] = RunLink{link_at[p], link4[4 * (size_t)p], link4[4 * (size_t)p + 1], link4[4 * (size_t)p + 2], link4[4 * (size_t)p + 3]};
    std::vector<int> x0, y0;
    if (!layout_runs(n, run_start, run_colour, links, x0, y0, W, H)) return false;
    if (dbg) fprintf(stderr, "[ccp_gs]   embed_region_device: %d runs, %d links\n", n_runs, n_links);
    lap("union-find, components, layout (host, per run)");
    // the region grid (its layout decides where an unknown's canvas element is) and the per-unknown placement
    ccp_grid_desc d{W, H, 1, 0, H, 0, m->device, CCP_GRID_DIRICHLET_MASK};
    if (m->region_grid) ccp_grid_destroy(m->region_grid);
    m->region_grid = nullptr;
    if (ccp_grid_create(&d, &m->region_grid) != CCP_OK) return false;
    (void)grid_set_allow_swap(m->region_grid, true);       // (the layout is asked for at every solve)
    ccp_grid_layout lay{};
    if (ccp_grid_get_layout(m->region_grid, &lay) != CCP_OK) return false;
    const size_t plane = (size_t)lay.local_rows * 2 * (size_t)lay.pitch;
    DevBuf<int> d_x0, d_y0, d_ident;
    DevBuf<unsigned char> d_mask;
    if (upload_vec(d_x0, x0, s) != CCP_OK || upload_vec(d_y0, y0, s) != CCP_OK || d_ident.alloc((size_t)W * H) != CCP_OK ||
        d_mask.alloc(plane) != CCP_OK || m->region_where.alloc((size_t)n) != CCP_OK)
        return false;
    hipLaunchKernelGGL(k_fill_int, dim3(4096), dim3(kBlock), 0, s, d_ident.p, (long)W * H, -1);
    if (!hip_ok(hipMemsetAsync(d_mask.p, 0, plane, s))) return false;
    hipLaunchKernelGGL(k_region_place, dim3(blocks), dim3(kBlock), 0, s, d_run_id.p, d_run_start.p, d_x0.p, d_y0.p, d_colour.p, n, W, H, (long)lay.pitch,
                       d_ident.p, m->region_where.p, d_mask.p, d_bad.p + 2, free_parity ? 1 : 0);
    int bad = 0;
    if (!hip_ok(hipMemcpyAsync(&bad, d_bad.p + 2, sizeof(int), hipMemcpyDeviceToHost, s)) || !hip_ok(hipStreamSynchronize(s))) return false;   // (x0, y0 die below)
    if (bad) return false;
    hipLaunchKernelGGL(k_region_verify, dim3(blocks), dim3(kBlock), 0, s, d_ptr.p, d_col.p, d_run_id.p, d_run_start.p, d_x0.p, d_y0.p, n, W, d_ident.p,
                       d_bad.p + 3);
    if (!hip_ok(hipMemcpyAsync(&bad, d_bad.p + 3, sizeof(int), hipMemcpyDeviceToHost, s)) || !hip_ok(hipStreamSynchronize(s))) return false;
    lap("placement, verification");
    if (bad || hipGetLastError() != hipSuccess) return false;
    if (free_parity) {
        colour.resize((size_t)n);
        if (!hip_ok(hipMemcpyAsync(colour.data(), d_colour.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, s)) || !hip_ok(hipStreamSynchronize(s)))
            return false;
    }
    if (ccp_grid_set_stream(m->region_grid, s) != CCP_OK) return false;
    if (grid_set_mask_split_device(m->region_grid, d_mask.p, (long)n) != CCP_OK) return false;
    lap("mask into the grid");
    return true;
}

int resolve_colouring(ccp_csr *m, std::vector<int> &colour, int &nc);

// Recognise (once per upload) and set up the Dirichlet-mask twin.
bool bipartite_colouring(ccp_csr *m, std::vector<int> &colour);

int detect_region_impl(ccp_csr *m, bool for_reference_order);

int detect_region(ccp_csr *m, bool for_reference_order = false)
{
    const int before = m->region_state;
    const int st = detect_region_impl(m, for_reference_order);
    // recognised in colour order AND in index order: the grid twin sweeps, applies and checks the matrix from here on
    if (st == CCP_OK && before < 1 && m->region_state == 1 && !m->rb.on) release_device_csr(m);
    return st;
}

int detect_region_impl(ccp_csr *m, bool for_reference_order)
{
    if (m->region_state >= 1) return CCP_OK;
    if (m->region_state == 0 && !(for_reference_order && m->region_wants_two_colouring)) return CCP_OK;
    m->region_state = 0;
    m->region_wants_two_colouring = false;
    if (!m->allow_region || m->n_rows < 4 || m->n_rows != m->n_cols || !m->overlay.empty()) return CCP_OK;
    const double t0 = now_s();
    // cheap screen before anything O(nnz): row 0 must look like a region row
    {
        bool ok = false;
        for (long k = m->row_ptr[0]; k < m->row_ptr[1]; ++k) ok |= m->col[k] == 0 && m->val[k] == 4.0;
        if (!ok) return CCP_OK;
    }
    std::vector<int> colour;
    int nc = 0;
    static const bool host_recognition = getenv("CCP_GS_REGION_HOST") && atoi(getenv("CCP_GS_REGION_HOST")) != 0;
    if (m->user_colour.empty() && m->auto_colour.empty() && !host_recognition && m->used_colour.empty() && !m->multicolour.built) {
        // No colouring came with the matrix (the facade's case).  A greedy colouring in row order needs a third colour
        // on most masks whose pieces merge further down (1.1 s at 41.75 M unknowns, and the sweep then stays on the stored
        // matrix): lay the region out first and let the canvas parity be the colouring.
        int cw0 = 0, ch0 = 0;
        if (embed_region_device(m, colour, cw0, ch0, true)) {
            m->region_tuned = false;
            m->region_solves = 0;
            m->auto_colour = colour;
            m->region_colour.swap(colour);
            m->region_w = cw0;
            m->region_h = ch0;
            m->region_state = 1;
            m->multicolour.reset();                      // (an image built with another colouring is not this sweep's)
            if (getenv("CCP_GS_DEBUG"))
                fprintf(stderr, "[ccp_gs] raster-region Laplacian without a colouring: %d unknowns on a %d x %d canvas, canvas parity taken as the colouring, %.3f s\n",
                        m->n_rows, cw0, ch0, now_s() - t0);
            return CCP_OK;
        }
        if (m->region_grid) ccp_grid_destroy(m->region_grid);
        m->region_grid = nullptr;
        colour.clear();
    }
    CCP_TRY(resolve_colouring(m, colour, nc));
    bool order_only = false;
    if (nc != 2) {
        // No colouring was given and the greedy one needs a third colour somewhere.  The colour-ordered sweep then
        // runs with that colouring, on the stored matrix.  The index-order sweep does not care about colours: any
        // 2-colouring of the (bipartite) pixel graph gives the canvas its parity.
        m->region_wants_two_colouring = true;
        if (!for_reference_order || !bipartite_colouring(m, colour)) return CCP_OK;
        m->region_wants_two_colouring = false;
        order_only = true;
    }
    const double t_col = now_s();
    const long n = m->n_rows;
    int cw = 0, chh = 0;
    static const bool host_path = getenv("CCP_GS_REGION_HOST") && atoi(getenv("CCP_GS_REGION_HOST")) != 0;
    if (!host_path) {
        // everything per unknown on the device (ccp_csr_region.hpp); the host keeps what is per run
        if (!embed_region_device(m, colour, cw, chh)) {
            if (m->region_grid) ccp_grid_destroy(m->region_grid);
            m->region_grid = nullptr;
            if (getenv("CCP_GS_DEBUG")) fprintf(stderr, "[ccp_gs] not a raster-region Laplacian (%.3f s)\n", now_s() - t0);
            return CCP_OK;
        }
        m->region_tuned = false;                         // default tiling now; tuned once the matrix is solved repeatedly
        if (getenv("CCP_GS_DEBUG")) fprintf(stderr, "[ccp_gs]   detect_region (device): colouring %.3f s, recognition + grid %.3f s\n", t_col - t0, now_s() - t_col);
    } else {
        RegionEmbedding E;
        if (!embed_region(m, colour, E)) {
            if (getenv("CCP_GS_DEBUG")) fprintf(stderr, "[ccp_gs] not a raster-region Laplacian (%.3f s)\n", now_s() - t0);
            return CCP_OK;
        }
        const double t_emb = now_s();
        ccp_grid_desc d{E.W, E.H, 1, 0, E.H, 0, m->device, CCP_GRID_DIRICHLET_MASK};
        if (m->region_grid) ccp_grid_destroy(m->region_grid);
        m->region_grid = nullptr;
        CCP_TRY(ccp_grid_create(&d, &m->region_grid));
        (void)grid_set_allow_swap(m->region_grid, true);
        ccp_grid_layout lay{};
        CCP_TRY(ccp_grid_get_layout(m->region_grid, &lay));
        std::vector<unsigned char> mask((size_t)E.W * E.H, 0);
        std::vector<long> where((size_t)n);
        parallel_ranges(n, 1 << 16, [&](long lo, long hi) {
            for (long i = lo; i < hi; ++i) {
                const int x = E.x[i], y = E.y[i];
                mask[(size_t)y * E.W + x] = 1;
                where[i] = ((long)y * 2 + ((x + y) & 1)) * lay.pitch + (x >> 1);
            }
        });
        const double t_mask = now_s();
        CCP_TRY(ccp_grid_set_mask_host(m->region_grid, mask.data(), E.W));
        const double t_set = now_s();
        CCP_TRY(ccp_grid_tune(m->region_grid, 8, nullptr, nullptr, nullptr));      // depth (clamped to what a mask grid has) and chunk rows for this canvas: speed only
        m->region_tuned = true;
        const double t_tune = now_s();
        CCP_TRY(upload_vec(m->region_where, where, m->stream));
        CCP_HIP(hipStreamSynchronize(m->stream));
        if (getenv("CCP_GS_DEBUG"))
            fprintf(stderr, "[ccp_gs]   detect_region (host): colouring %.3f s, embedding %.3f s, grid + mask/where arrays %.3f s, mask upload %.3f s, tune %.3f s, where upload %.3f s\n",
                    t_col - t0, t_emb - t_col, t_mask - t_emb, t_set - t_mask, t_tune - t_set, now_s() - t_tune);
        cw = E.W;
        chh = E.H;
    }
    m->region_solves = 0;
    m->region_colour.swap(colour);
    m->region_w = cw;
    m->region_h = chh;
    m->region_state = order_only ? 2 : 1;
    if (getenv("CCP_GS_DEBUG"))
        fprintf(stderr, "[ccp_gs] raster-region Laplacian: %ld unknowns on a %d x %d canvas (%.0f %% filled), recognised in %.3f s\n", n,
                cw, chh, 100.0 * n / ((double)cw * chh), now_s() - t0);
    return CCP_OK;
}

// The user colouring (if any) must be the grid's red-black colouring with pixel 0 red, because
// that is the sweep order the matrix-free kernels implement.
bool colouring_is_checkerboard(const ccp_csr *m)
{
    if (m->user_colour.empty()) return true;          // greedy colouring of the full grid IS the checkerboard
    if (m->user_n_colours != 2) return false;
    const int W = m->poisson_w;
    for (int i = 0; i < m->n_rows; ++i)
        if (m->user_colour[i] != (((i % W) + (i / W)) & 1)) return false;
    return true;
}

int ensure_natural(ccp_csr *m)
{
    if (m->natural.built) return CCP_OK;
    std::vector<int> group((size_t)m->n_rows, 0);
    return build_schedule(m, m->natural, group, m->n_rows ? 1 : 0, false, true);
}

// The colouring of the multi-colour sweep: the caller's (checked to be proper) or greedy in row order.
int resolve_colouring(ccp_csr *m, std::vector<int> &colour, int &nc)
{
    const double t0 = now_s();
    // (the canvas parity of a recognised region stands in for a colouring only while the matrix IS that region: once an
    // edit has taken the region form away the library colours the rows as it would after a fresh upload of the edited matrix)
    const bool automatic = m->user_colour.empty() && !m->auto_colour.empty() && (int)m->auto_colour.size() == m->n_rows && !m->edited;
    if (!m->user_colour.empty() || automatic) {
        colour = automatic ? m->auto_colour : m->user_colour;
        nc = automatic ? 2 : m->user_n_colours;
        // a proper colouring: no stored off-diagonal entry couples two rows of one colour.  Rows edited since the
        // upload are checked in their current form (the overlay), the others in the compact host copy.
        std::atomic<int> bad{0};
        const int n = m->n_rows;
        std::vector<unsigned char> edited;
        if (!m->overlay.empty()) {
            edited.assign((size_t)n, 0);
            for (const auto &kv : m->overlay) {
                edited[(size_t)kv.first] = 1;
                for (int c : kv.second.col)
                    if (c != kv.first && c >= 0 && c < n && colour[c] == colour[kv.first]) bad.store(1, std::memory_order_relaxed);
            }
        }
        const unsigned char *skip = edited.empty() ? nullptr : edited.data();
        parallel_ranges(n, 1 << 16, [&](long lo, long hi) {
            for (long i = lo; i < hi && !bad.load(std::memory_order_relaxed); ++i) {
                if (skip && skip[i]) continue;
                for (long k = m->row_ptr[i]; k < m->row_ptr[i + 1]; ++k) {
                    const int c = m->col[k];
                    if (c != (int)i && c >= 0 && c < n && colour[c] == colour[i]) {
                        bad.store(1, std::memory_order_relaxed);
                        break;
                    }
                }
            }
        });
        if (bad.load() && !automatic) return CCP_ERR_UNSUPPORTED;
        if (bad.load()) {
            // an edit coupled two rows of one parity: the library's own colouring starts again (greedy)
            m->auto_colour.clear();
            return resolve_colouring(m, colour, nc);
        }
    } else if (!m->greedy_colour.empty() && (int)m->greedy_colour.size() == m->n_rows) {
        colour = m->greedy_colour;
        nc = m->greedy_n_colours;
        return CCP_OK;
    } else {
        CCP_TRY(materialise(m));
        std::vector<long> lptr;
        std::vector<int> lidx;
        build_lower(m, lptr, lidx);
        nc = greedy_colouring(lptr, lidx, m->n_rows, colour);
        m->greedy_colour = colour;
        m->greedy_n_colours = nc;
    }
    if (getenv("CCP_GS_DEBUG")) fprintf(stderr, "[ccp_gs] colouring (%d colours) in %.3f s\n", nc, now_s() - t0);
    return CCP_OK;
}

// A proper 2-colouring of the rows by breadth-first search over the stored couplings (false: an odd cycle, or a
// coupling stored in one direction only reaches a row the other way round with the wrong colour).
bool bipartite_colouring(ccp_csr *m, std::vector<int> &colour)
{
    const int n = m->n_rows;
    colour.assign((size_t)n, -1);
    std::vector<int> queue;
    queue.reserve((size_t)n);
    for (int root = 0; root < n; ++root) {
        if (colour[root] >= 0) continue;
        colour[root] = 0;
        queue.clear();
        queue.push_back(root);
        for (size_t head = 0; head < queue.size(); ++head) {
            const int i = queue[head];
            for (long k = m->row_ptr[i]; k < m->row_ptr[i + 1]; ++k) {
                const int c = m->col[k];
                if (c == i || c < 0 || c >= n) continue;
                if (colour[c] < 0) {
                    colour[c] = colour[i] ^ 1;
                    queue.push_back(c);
                } else if (colour[c] == colour[i]) {
                    return false;
                }
            }
        }
    }
    return true;
}

int ensure_multicolour(ccp_csr *m)
{
    if (m->multicolour.built) return CCP_OK;
    std::vector<int> colour;
    int nc = 0;
    CCP_TRY(resolve_colouring(m, colour, nc));
    CCP_TRY(build_schedule(m, m->multicolour, colour, nc, true, !m->user_colour.empty()));
    m->used_colour.swap(colour);
    m->used_n_colours = nc;
    return CCP_OK;
}

int ensure_lexicographic(ccp_csr *m)
{
    if (m->lexicographic.built) return CCP_OK;
    CCP_TRY(materialise(m));
    std::vector<long> lptr;
    std::vector<int> lidx;
    build_lower(m, lptr, lidx);
    std::vector<int> level;
    int nl = m->allow_pipeline ? unit_potential(lptr, lidx, m->n_rows, level) : 0;
    int span = 1;
    if (nl == 0) {
        nl = level_schedule(lptr, lidx, m->n_rows, level);
        span = 0;
        for (int i = 0; i < m->n_rows; ++i)
            for (long k = lptr[i]; k < lptr[i + 1]; ++k) span = std::max(span, level[i] - level[lidx[k]]);
    }
    CCP_TRY(build_schedule(m, m->lexicographic, level, nl, false));
    m->lexicographic.level_span = span;
    if (getenv("CCP_GS_DEBUG")) fprintf(stderr, "[ccp_gs] level schedule: %d levels, span %d\n", nl, span);
    return CCP_OK;
}

// ---- incremental edits -------------------------------------------------------------------------------
// Where original row i lives in a schedule's image.
void locate(const Schedule &sc, int i, int &slice, int &lane)
{
    const int p = sc.inv[i], g = sc.group_of[i];
    const long in_group = p - sc.gstart[g];
    slice = sc.group_slice_ptr[g] + (int)(in_group / kWave);
    lane = (int)(in_group % kWave);
}

// May the edited row keep its place in this schedule?  Every coupled pair must still be ordered the way
// the schedule promises: different colours (multi-colour); 0 < level(hi) - level(lo) <= span for lo < hi
// (level schedule, pipelined over sweeps); anything goes for the identity order of the SpMV.
bool row_still_fits(const ccp_csr *m, const Schedule &sc, int kind, int i, const RowView &r)
{
    if (kind == 0) return true;
    for (long k = 0; k < r.len; ++k) {
        const int c = r.col[k];
        if (c == i || c < 0 || c >= m->n_rows) continue;
        if (kind == 1) {
            if (sc.group_of[c] == sc.group_of[i]) return false;
        } else {
            const int lo = std::min(i, c), hi = std::max(i, c);
            const int d = sc.group_of[hi] - sc.group_of[lo];
            if (d <= 0 || d > sc.level_span) return false;
        }
    }
    return true;
}

// Re-lay the touched rows in every built image: one batched patch kernel per image, slices that outgrew
// their spare columns move into the reserve at the end of the arrays; an image whose ordering a new
// coupling breaks (same colour, wrong level) or whose reserve is used up is dropped and rebuilt from the
// host copy at the next solve that needs it.
int flush_edits(ccp_csr *m)
{
    if (m->touched.empty()) return CCP_OK;
    std::sort(m->touched.begin(), m->touched.end());
    m->touched.erase(std::unique(m->touched.begin(), m->touched.end()), m->touched.end());
    Schedule *scs[3] = {&m->natural, &m->multicolour, &m->lexicographic};
    for (int kind = 0; kind < 3; ++kind) {
        Schedule &sc = *scs[kind];
        if (!sc.built) continue;
        std::vector<long> base, off;
        std::vector<int> cap, pc;
        std::vector<double> pv;
        std::vector<std::pair<int, double>> tmp;
        bool drop = false;
        // pass 1: does every touched row keep its place, and how many entry columns does each touched slice need?
        std::unordered_map<int, int> slice_need;
        for (int i : m->touched) {
            const RowView r = row_view(m, i);
            if (!row_still_fits(m, sc, kind, i, r)) {
                drop = true;
                break;
            }
            int s, lane;
            locate(sc, i, s, lane);
            int &need = slice_need[s];
            need = std::max(need, (int)r.len);
        }
        // a slice that needs more than its spare columns moves to the reserve ONCE, before any of its rows is patched
        // (a patch carries the slice's final offset)
        if (!drop)
            for (const auto &kv : slice_need) {
                const int s = kv.first, need = kv.second;
                if (need > sc.h_scap[s]) {
                    const int new_cap = need + kSliceSlack;
                    if (sc.entries_used + (long)new_cap * kWave > sc.entries_cap) {
                        drop = true;
                        break;
                    }
                    const long dst = sc.entries_used;
                    hipLaunchKernelGGL(k_move_slice, dim3(1), dim3(kWave), 0, m->stream, sc.cols.p, sc.vals.p, sc.h_soff[s], dst,
                                       sc.h_swidth[s], new_cap);
                    CCP_HIP(hipGetLastError());
                    CCP_HIP(hipMemcpyAsync(sc.slice_off.p + s, &dst, sizeof(long), hipMemcpyHostToDevice, m->stream));
                    CCP_HIP(hipStreamSynchronize(m->stream));     // `dst` lives on this stack frame
                    sc.h_soff[s] = dst;
                    sc.h_scap[s] = new_cap;
                    sc.entries_used += (long)new_cap * kWave;
                    m->stat_slices_relocated++;
                }
                if (need > sc.h_swidth[s]) {                      // the live width only ever grows (padding is skipped by the kernels)
                    sc.h_swidth[s] = need;
                    CCP_HIP(hipMemcpyAsync(sc.slice_width.p + s, &sc.h_swidth[s], sizeof(int), hipMemcpyHostToDevice, m->stream));
                }
            }
        // pass 2: the patches
        if (!drop)
            for (int i : m->touched) {
                const RowView r = row_view(m, i);
                int s, lane;
                locate(sc, i, s, lane);
                tmp.clear();
                for (long k = 0; k < r.len; ++k) {
                    const int c = r.col[k];
                    tmp.emplace_back((c >= 0 && c < m->n_rows) ? sc.inv[c] : c, r.val[k]);
                }
                if (sc.sort_by_permuted) std::stable_sort(tmp.begin(), tmp.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
                base.push_back(sc.h_soff[s] + lane);
                cap.push_back(sc.h_scap[s]);
                off.push_back((long)pc.size());
                for (int k = 0; k < sc.h_scap[s]; ++k) {
                    const bool live = k < (int)tmp.size();
                    pc.push_back(live ? tmp[k].first : -1);
                    pv.push_back(live ? tmp[k].second : 0.0);
                }
            }
        if (drop) {
            sc.reset();
            m->stat_schedule_rebuilds++;
            continue;
        }
        CCP_TRY(upload_vec(m->patch_base, base, m->stream));
        CCP_TRY(upload_vec(m->patch_cap, cap, m->stream));
        CCP_TRY(upload_vec(m->patch_off, off, m->stream));
        CCP_TRY(upload_vec(m->patch_cols, pc, m->stream));
        CCP_TRY(upload_vec(m->patch_vals, pv, m->stream));
        hipLaunchKernelGGL(k_patch_rows, dim3((unsigned)base.size()), dim3(kWave), 0, m->stream, sc.cols.p, sc.vals.p, m->patch_base.p,
                           m->patch_cap.p, m->patch_off.p, m->patch_cols.p, m->patch_vals.p);
        CCP_HIP(hipGetLastError());
        CCP_HIP(hipStreamSynchronize(m->stream));             // host vectors die at scope exit
        m->stat_rows_patched += (long)base.size();
    }
    m->touched.clear();
    return CCP_OK;
}

// partial-sum scratch large enough for `blocks` block results of two doubles each
int ensure_partial(ccp_csr *m, long blocks)
{
    const size_t need = 2 * (size_t)std::max<long>(blocks, 1) + 2;
    if (m->partial.n >= need) return CCP_OK;
    return m->partial.alloc(need);
}

unsigned blocks_for(long n) { return (unsigned)std::max<long>(1, std::min<long>(4096, (n + kBlock - 1) / kBlock)); }

// ---- row block of a distributed matrix (ccp_csr::RowBlock) ---------------------------------------------------------

// Every rank's status word, gathered: the own status if it is a failure, CCP_ERR_STATE if only a peer failed.  The
// set-up of a row block is collective; a rank that found its arguments wrong must not leave the others waiting in the
// next message.
int rb_agree(ccp_comm *c, hipStream_t s, int status)
{
    const RcclApi *api = rccl_api();
    if (!api) return CCP_ERR_RCCL;
    int *dev = reinterpret_cast<int *>(c->scratch.p);
    std::vector<int> all((size_t)c->world);
    CCP_HIP(hipMemcpyAsync(dev + c->rank, &status, sizeof(int), hipMemcpyHostToDevice, s));
    CCP_RCCL(api->AllGather(dev + c->rank, dev, 1, ncclInt32, c->comm, s));
    CCP_HIP(hipMemcpyAsync(all.data(), dev, sizeof(int) * all.size(), hipMemcpyDeviceToHost, s));
    CCP_HIP(hipStreamSynchronize(s));
    if (status != CCP_OK) return status;
    for (int v : all)
        if (v != CCP_OK) return CCP_ERR_STATE;
    return CCP_OK;
}

// One grouped exchange on the handle's stream: `send_cnt[r]` doubles from sendbuf + send_off[r] to rank r, `recv_cnt[r]`
// doubles from rank r into x + recv_pos[r].
int rb_messages(ccp_csr *m, const int *send_cnt, const int *send_off, const int *recv_cnt, const int *recv_pos, double *x, hipStream_t s)
{
    ccp_csr::RowBlock &rb = m->rb;
    const RcclApi *api = rccl_api();
    if (!api || !rb.comm) return CCP_ERR_STATE;
    const int W = rb.comm->world;
    bool any = false;
    for (int r = 0; r < W; ++r) any |= send_cnt[r] > 0 || recv_cnt[r] > 0;
    if (!any) return CCP_OK;
    CCP_RCCL(api->GroupStart());
    ncclResult_t res = ncclSuccess;
    for (int r = 0; r < W && res == ncclSuccess; ++r) {
        if (send_cnt[r] > 0) {
            res = api->Send(rb.sendbuf.p + send_off[r], (size_t)send_cnt[r], ncclDouble, r, rb.comm->comm, s);
            rb.values_sent += send_cnt[r];
        }
        if (recv_cnt[r] > 0 && res == ncclSuccess)
            res = api->Recv(x + recv_pos[r], (size_t)recv_cnt[r], ncclDouble, r, rb.comm->comm, s);
    }
    const ncclResult_t e = api->GroupEnd();
    if (res != ncclSuccess) return rccl_fail(res, "ncclSend/ncclRecv", __FILE__, __LINE__);
    CCP_RCCL(e);
    rb.exchanges++;
    return CCP_OK;
}

// The rows of colour g this block owns have new values: hand the ones other blocks reference to their ghosts and
// take ours (m->x in colour-major order).
int rb_exchange_colour(ccp_csr *m, int g, hipStream_t s)
{
    ccp_csr::RowBlock &rb = m->rb;
    const int W = rb.comm->world;
    const long seg0 = rb.col_seg_off[(size_t)g], seg = rb.col_seg_off[(size_t)g + 1] - seg0;
    if (seg > 0) {
        hipLaunchKernelGGL((k_permute<true>), dim3(blocks_for(seg)), dim3(kBlock), 0, s, rb.sendbuf.p + seg0, m->x.p,
                           rb.col_send_pos.p + seg0, seg);
        CCP_HIP(hipGetLastError());
    }
    const size_t at = (size_t)g * W;
    return rb_messages(m, &rb.col_send_cnt[at], &rb.col_send_off[at], &rb.col_recv_cnt[at], &rb.col_recv_pos[at], m->x.p, s);
}

// All ghosts of a vector in natural (extended) order.
int rb_exchange_natural(ccp_csr *m, double *x)
{
    ccp_csr::RowBlock &rb = m->rb;
    const long total = rb.nat_total;
    if (total > 0) {
        hipLaunchKernelGGL((k_permute<true>), dim3(blocks_for(total)), dim3(kBlock), 0, m->stream, rb.sendbuf.p, x, rb.nat_send_pos.p, total);
        CCP_HIP(hipGetLastError());
    }
    return rb_messages(m, rb.nat_send_cnt.data(), rb.nat_send_off.data(), rb.nat_recv_cnt.data(), rb.nat_recv_pos.data(), x, m->stream);
}

// A vector of the block (n_local host values) into the extended natural order on the device: ghosts 0.
int rb_stage(ccp_csr *m, double *dst_dev, const double *host_local)
{
    const ccp_csr::RowBlock &rb = m->rb;
    CCP_HIP(hipMemsetAsync(dst_dev, 0, sizeof(double) * (size_t)std::max(m->n_rows, 1), m->stream));
    if (rb.n_local) CCP_HIP(hipMemcpyAsync(dst_dev + rb.n_lo, host_local, sizeof(double) * (size_t)rb.n_local, hipMemcpyHostToDevice, m->stream));
    return CCP_OK;
}


}  // namespace

extern "C" {

int ccp_csr_create(int device, ccp_csr **out)
try {
    if (!out) return CCP_ERR_BAD_ARG;
    *out = nullptr;
    CCP_TRY(select_device(device));
    ccp_csr *m = new (std::nothrow) ccp_csr();
    if (!m) return CCP_ERR_ALLOC;
    m->device = device;
    if (hipEventCreate(&m->ev0) != hipSuccess || hipEventCreate(&m->ev1) != hipSuccess) {
        ccp_csr_destroy(m);
        return CCP_ERR_HIP;
    }
    *out = m;
    return CCP_OK;
} CCP_ABI_CATCH

int ccp_csr_destroy(ccp_csr *m)
try {
    if (!m) return CCP_OK;
    (void)hipSetDevice(m->device);
    wait_device_upload(m);
    if (m->ev0) (void)hipEventDestroy(m->ev0);
    if (m->ev1) (void)hipEventDestroy(m->ev1);
    if (m->grid) ccp_grid_destroy(m->grid);
    if (m->region_grid) ccp_grid_destroy(m->region_grid);
    m->rb.reset();
    delete m;
    return CCP_OK;
} CCP_ABI_CATCH

int ccp_csr_upload(ccp_csr *m, int32_t n_rows, int32_t n_cols, int64_t n_values, const double *values,
                   const int32_t *col_offset, const int32_t *row_begin, const int32_t *row_num_nze)
try {
    CCP_TRY(bind(m));
    if (n_rows < 0 || n_cols < 0 || n_values < 0) return CCP_ERR_BAD_ARG;
    if (n_rows > 0 && (!row_begin || !row_num_nze)) return CCP_ERR_BAD_ARG;
    if (n_values > 0 && (!values || !col_offset)) return CCP_ERR_BAD_ARG;
    wait_device_upload(m);                 // (a copy of the previous matrix may still be in flight)
    // a failed upload must not leave the previous matrix half overwritten but still "uploaded"
    m->uploaded = false;
    m->natural.reset();
    m->multicolour.reset();
    m->lexicographic.reset();
    try {
        // row offsets of the compact copy: a two-pass prefix sum over fixed chunks of rows (41.75 M rows: the serial
        // loop was 50 ms of the upload)
        m->row_ptr.assign((size_t)n_rows + 1, 0);
        {
            const long K = std::max<long>(1, std::min<long>(256, n_rows / (1 << 16)));
            const long per = ((long)n_rows + K - 1) / K;
            std::vector<long> chunk_sum((size_t)K + 1, 0);
            std::atomic<int> bad_rows{0};
            parallel_ranges(K, 1, [&](long lo, long hi) {
                for (long c = lo; c < hi; ++c) {
                    long sum = 0;
                    for (long i = c * per; i < std::min<long>(n_rows, (c + 1) * per); ++i) {
                        const long nz = row_num_nze[i];
                        if (nz < 0 || (nz > 0 && (row_begin[i] < 0 || (long)row_begin[i] + nz > n_values))) bad_rows.store(1, std::memory_order_relaxed);
                        sum += std::max<long>(nz, 0);
                    }
                    chunk_sum[(size_t)c + 1] = sum;
                }
            });
            if (bad_rows.load()) return CCP_ERR_BAD_ARG;
            for (long c = 0; c < K; ++c) chunk_sum[c + 1] += chunk_sum[c];
            parallel_ranges(K, 1, [&](long lo, long hi) {
                for (long c = lo; c < hi; ++c) {
                    long run = chunk_sum[c];
                    for (long i = c * per; i < std::min<long>(n_rows, (c + 1) * per); ++i) {
                        run += row_num_nze[i];
                        m->row_ptr[i + 1] = run;
                    }
                }
            });
        }
        const long nnz = m->row_ptr[n_rows];
        m->col.resize((size_t)nnz);
        m->val.resize((size_t)nnz);
        std::atomic<int> bad{0};
        parallel_ranges(n_rows, 1 << 15, [&](long lo, long hi) {       // compaction of the live entries, by row ranges
            for (long i = lo; i < hi; ++i) {
                const long nz = row_num_nze[i];
                if (!nz) continue;
                std::memcpy(&m->col[m->row_ptr[i]], col_offset + row_begin[i], sizeof(int32_t) * (size_t)nz);
                std::memcpy(&m->val[m->row_ptr[i]], values + row_begin[i], sizeof(double) * (size_t)nz);
                // columns in range and STRICTLY increasing within a row, as the reference keeps them (at() is a binary
                // search, sparse-matrix.h:627-645): insert() relies on it (lower_bound) and so does the recognition of
                // region matrices (a duplicated coupling would be counted twice)
                for (long k = m->row_ptr[i]; k < m->row_ptr[i + 1]; ++k)
                    if (m->col[k] < 0 || m->col[k] >= n_cols || (k > m->row_ptr[i] && m->col[k] <= m->col[k - 1]))
                        bad.store(1, std::memory_order_relaxed);
            }
        });
        if (bad.load()) return CCP_ERR_BAD_ARG;
    } catch (const std::bad_alloc &) {
        return CCP_ERR_ALLOC;
    }
    m->n_rows = n_rows;
    m->n_cols = n_cols;
    m->natural.reset();
    m->multicolour.reset();
    m->lexicographic.reset();
    m->user_colour.clear();
    m->user_n_colours = 0;
    m->auto_colour.clear();
    m->greedy_colour.clear();
    m->used_colour.clear();
    m->used_n_colours = 0;
    m->overlay.clear();
    m->touched.clear();
    m->poisson_w = -1;
    m->poisson_h = 0;
    m->compact_poisson_w = -1;
    m->region_state = -1;
    m->region_values_ok = -1;
    m->dev_csr_valid = false;
    m->edited = false;
    m->rb.reset();                         // (ccp_csr_upload_rows sets the row block up again after this call)
    m->allow_region = true;
    if (const char *e = getenv("CCP_GS_MASKED")) m->allow_region = atoi(e) != 0;
    if (m->grid) ccp_grid_destroy(m->grid);
    m->grid = nullptr;
    if (m->region_grid) ccp_grid_destroy(m->region_grid);      // canvas-sized buffers of the previous matrix (GBs at 8192^2)
    m->region_grid = nullptr;
    m->region_wants_two_colouring = false;
    // a handle that held a row block (ccp_csr_upload_rows switches the one-GPU forms off) returns to the defaults
    m->allow_structured = m->allow_one_block = m->allow_pipeline = true;
    if (const char *e = getenv("CCP_GS_STRUCTURED")) m->allow_structured = atoi(e) != 0;
    if (const char *e = getenv("CCP_GS_ONE_BLOCK")) m->allow_one_block = atoi(e) != 0;
    if (const char *e = getenv("CCP_GS_PIPELINE")) m->allow_pipeline = atoi(e) != 0;
    const size_t vec = (size_t)std::max(std::max(n_rows, n_cols), 2);
    CCP_TRY(m->x.alloc(vec));
    CCP_TRY(m->b.alloc(vec));
    CCP_TRY(m->tmp.alloc(vec));
    CCP_TRY(m->state.alloc(1));
    m->uploaded = true;
    return start_device_upload(m);         // the device copy proceeds in the background
} CCP_ABI_CATCH

namespace {
int upload_rows_impl(ccp_csr *m, ccp_comm *c, int32_t first_row, int32_t n_rows, int32_t n_global, int64_t n_values,
                     const double *values, const int32_t *col_offset, const int32_t *row_begin, const int32_t *row_num_nze,
                     const int32_t *colour, int32_t n_colours)
{
    CCP_TRY(bind(m));
    if (!c || c->device != m->device) return CCP_ERR_BAD_ARG;
    const RcclApi *api = rccl_api();
    if (!api) return CCP_ERR_RCCL;
    wait_device_upload(m);
    hipStream_t s = m->stream;
    const int W = c->world, me = c->rank;
    const long lo = first_row, hi = (long)first_row + n_rows;

    // ---- 1. the block's own arguments; the columns outside it (the ghosts) ----------------------------------------
    int status = CCP_OK;
    if (first_row < 0 || n_rows < 0 || n_global < 0 || hi > n_global || n_values < 0 || n_colours < 1 ||
        (n_rows > 0 && (!row_begin || !row_num_nze || !colour)) || (n_values > 0 && (!values || !col_offset)))
        status = CCP_ERR_BAD_ARG;
    std::vector<int> ghost;
    if (status == CCP_OK) {
        std::atomic<int> bad{0};
        std::mutex merge;
        parallel_ranges(n_rows, 1 << 15, [&](long a, long b) {
            std::vector<int> mine;
            for (long i = a; i < b; ++i) {
                const long nz = row_num_nze[i], at = row_begin[i];
                if (colour[i] < 0 || colour[i] >= n_colours || nz < 0 || (nz > 0 && (at < 0 || at + nz > n_values))) {
                    bad.store(1, std::memory_order_relaxed);
                    continue;
                }
                for (long k = at; k < at + nz; ++k) {
                    const long col = col_offset[k];
                    if (col < 0 || col >= n_global || (k > at && col <= col_offset[k - 1])) bad.store(1, std::memory_order_relaxed);
                    else if (col < lo || col >= hi) mine.push_back((int)col);
                }
            }
            std::sort(mine.begin(), mine.end());
            mine.erase(std::unique(mine.begin(), mine.end()), mine.end());
            std::lock_guard<std::mutex> lk(merge);
            ghost.insert(ghost.end(), mine.begin(), mine.end());
        });
        if (bad.load()) status = CCP_ERR_BAD_ARG;
        std::sort(ghost.begin(), ghost.end());
        ghost.erase(std::unique(ghost.begin(), ghost.end()), ghost.end());
    }

    // ---- 2. every rank learns every block: contiguous blocks of one matrix, in rank order -------------------------
    std::vector<int> part((size_t)W + 1, 0);
    {
        const int mine[8] = {first_row, n_rows, n_global, n_colours, status, 0, 0, 0};
        int *dev = reinterpret_cast<int *>(c->scratch.p);
        std::vector<int> all((size_t)8 * W);
        CCP_HIP(hipMemcpyAsync(dev + 8 * me, mine, sizeof(mine), hipMemcpyHostToDevice, s));
        CCP_RCCL(api->AllGather(dev + 8 * me, dev, 8, ncclInt32, c->comm, s));
        CCP_HIP(hipMemcpyAsync(all.data(), dev, sizeof(int) * all.size(), hipMemcpyDeviceToHost, s));
        CCP_HIP(hipStreamSynchronize(s));
        if (status != CCP_OK) return status;
        long next = 0;
        for (int r = 0; r < W; ++r) {
            if (all[8 * r + 4] != CCP_OK) return CCP_ERR_STATE;              // a peer's arguments were refused
            if (all[8 * r] != next || all[8 * r + 2] != n_global || all[8 * r + 3] != n_colours) return CCP_ERR_BAD_ARG;
            part[(size_t)r] = (int)next;
            next += all[8 * r + 1];
        }
        if (next != n_global) return CCP_ERR_BAD_ARG;
        part[(size_t)W] = n_global;
    }

    // ---- 3. who asks whom for how many values ---------------------------------------------------------------------
    const int n_ghost = (int)ghost.size();
    const int n_lo = (int)(std::lower_bound(ghost.begin(), ghost.end(), (int)lo) - ghost.begin());
    std::vector<int> req_cnt((size_t)W, 0), req_off((size_t)W, 0);
    for (int o = 0; o < W; ++o) {
        const int a = (int)(std::lower_bound(ghost.begin(), ghost.end(), part[(size_t)o]) - ghost.begin());
        const int b = (int)(std::lower_bound(ghost.begin(), ghost.end(), part[(size_t)o + 1]) - ghost.begin());
        req_off[(size_t)o] = a;
        req_cnt[(size_t)o] = b - a;
    }
    DevBuf<int> d_cnt, d_req, d_serve;
    CCP_TRY(d_cnt.alloc((size_t)W * W));
    std::vector<int> cnt((size_t)W * W);
    CCP_HIP(hipMemcpyAsync(d_cnt.p + (size_t)me * W, req_cnt.data(), sizeof(int) * W, hipMemcpyHostToDevice, s));
    CCP_RCCL(api->AllGather(d_cnt.p + (size_t)me * W, d_cnt.p, (size_t)W, ncclInt32, c->comm, s));
    CCP_HIP(hipMemcpyAsync(cnt.data(), d_cnt.p, sizeof(int) * cnt.size(), hipMemcpyDeviceToHost, s));
    CCP_HIP(hipStreamSynchronize(s));
    std::vector<int> serve_cnt((size_t)W, 0), serve_off((size_t)W, 0);
    long serve_total = 0;
    for (int r = 0; r < W; ++r) {
        serve_cnt[(size_t)r] = cnt[(size_t)r * W + me];
        serve_off[(size_t)r] = (int)serve_total;
        serve_total += serve_cnt[(size_t)r];
    }
    if (serve_total > INT32_MAX) status = CCP_ERR_UNSUPPORTED;
    // every rank or none enters the exchanges of step 4 (the offsets above are ints)
    CCP_TRY(rb_agree(c, s, status));

    // ---- 4. the halo index lists travel to the owners, the colours of those rows travel back ----------------------
    CCP_TRY(d_req.alloc((size_t)std::max(n_ghost, 1)));
    CCP_TRY(d_serve.alloc((size_t)std::max<long>(serve_total, 1)));
    auto int_exchange = [&](const int *send_base, const int *scnt, const int *soff, int *recv_base, const int *rcnt, const int *roff) -> int {
        bool any = false;
        for (int r = 0; r < W; ++r) any |= scnt[r] > 0 || rcnt[r] > 0;
        if (!any) return CCP_OK;
        CCP_RCCL(api->GroupStart());
        ncclResult_t res = ncclSuccess;
        for (int r = 0; r < W && res == ncclSuccess; ++r) {
            if (scnt[r] > 0) res = api->Send(send_base + soff[r], (size_t)scnt[r], ncclInt32, r, c->comm, s);
            if (rcnt[r] > 0 && res == ncclSuccess) res = api->Recv(recv_base + roff[r], (size_t)rcnt[r], ncclInt32, r, c->comm, s);
        }
        const ncclResult_t e = api->GroupEnd();
        if (res != ncclSuccess) return rccl_fail(res, "ncclSend/ncclRecv", __FILE__, __LINE__);
        CCP_RCCL(e);
        return CCP_OK;
    };
    if (n_ghost) CCP_HIP(hipMemcpyAsync(d_req.p, ghost.data(), sizeof(int) * (size_t)n_ghost, hipMemcpyHostToDevice, s));
    CCP_TRY(int_exchange(d_req.p, req_cnt.data(), req_off.data(), d_serve.p, serve_cnt.data(), serve_off.data()));
    std::vector<int> serve((size_t)serve_total), serve_colour((size_t)serve_total, 0), ghost_colour((size_t)n_ghost, 0);
    if (serve_total) CCP_HIP(hipMemcpyAsync(serve.data(), d_serve.p, sizeof(int) * (size_t)serve_total, hipMemcpyDeviceToHost, s));
    CCP_HIP(hipStreamSynchronize(s));
    for (int r = 0; r < W; ++r)
        for (int k = serve_off[(size_t)r]; k < serve_off[(size_t)r] + serve_cnt[(size_t)r]; ++k) {
            const long g = serve[(size_t)k];
            if (g < lo || g >= hi || (k > serve_off[(size_t)r] && g <= serve[(size_t)k - 1])) status = CCP_ERR_STATE;   // not a list of this block's rows
            else serve_colour[(size_t)k] = colour[g - lo];
        }
    if (serve_total) CCP_HIP(hipMemcpyAsync(d_serve.p, serve_colour.data(), sizeof(int) * (size_t)serve_total, hipMemcpyHostToDevice, s));
    CCP_TRY(int_exchange(d_serve.p, serve_cnt.data(), serve_off.data(), d_req.p, req_cnt.data(), req_off.data()));
    if (n_ghost) CCP_HIP(hipMemcpyAsync(ghost_colour.data(), d_req.p, sizeof(int) * (size_t)n_ghost, hipMemcpyDeviceToHost, s));
    CCP_HIP(hipStreamSynchronize(s));
    for (int k = 0; k < n_ghost; ++k)
        if (ghost_colour[(size_t)k] < 0 || ghost_colour[(size_t)k] >= n_colours) status = CCP_ERR_STATE;
    CCP_TRY(rb_agree(c, s, status));

    // ---- 5. the extended system through the ordinary upload: ghosts are empty rows among the owned ones ------------
    const long n_ext = (long)n_rows + n_ghost;
    if (n_ext > INT32_MAX) status = CCP_ERR_UNSUPPORTED;
    if (status == CCP_OK) {
        try {
            std::vector<int> ext_begin((size_t)n_ext, 0), ext_nnz((size_t)n_ext, 0), ext_colour((size_t)n_ext, 0);
            std::unique_ptr<int[]> ext_col(new int[(size_t)std::max<int64_t>(n_values, 1)]);
            for (int k = 0; k < n_ghost; ++k) ext_colour[(size_t)(k < n_lo ? k : n_rows + k)] = ghost_colour[(size_t)k];
            parallel_ranges(n_rows, 1 << 15, [&](long a, long b) {
                for (long i = a; i < b; ++i) {
                    const long nz = row_num_nze[i], at = row_begin[i];
                    ext_begin[(size_t)(n_lo + i)] = (int)at;
                    ext_nnz[(size_t)(n_lo + i)] = (int)nz;
                    ext_colour[(size_t)(n_lo + i)] = colour[i];
                    for (long k = at; k < at + nz; ++k) {
                        const long col = col_offset[k];
                        if (col >= lo && col < hi) {
                            ext_col[(size_t)k] = (int)(n_lo + (col - lo));
                        } else {
                            const int pos = (int)(std::lower_bound(ghost.begin(), ghost.end(), (int)col) - ghost.begin());
                            ext_col[(size_t)k] = pos < n_lo ? pos : n_rows + pos;
                        }
                    }
                }
            });
            // (slack entries between the rows keep whatever the caller's array holds there: never read, like the reference's)
            status = ccp_csr_upload(m, (int)n_ext, (int)n_ext, n_values, values, ext_col.get(), ext_begin.data(), ext_nnz.data());
            if (status == CCP_OK) status = ccp_csr_set_colouring(m, ext_colour.data(), n_colours);
        } catch (const std::bad_alloc &) {
            status = CCP_ERR_ALLOC;
        }
    }
    if (status == CCP_OK) {
        m->allow_structured = false;           // the grid twins know nothing of ghosts
        m->allow_one_block = false;            // the exchange sits between the colour launches
        status = ensure_multicolour(m);        // (checks the colouring: CCP_ERR_UNSUPPORTED when two coupled rows share a colour)
    }
    status = rb_agree(c, s, status);
    if (status != CCP_OK) {
        m->uploaded = false;                   // no half-set-up block: the handle waits for the next upload
        return status;
    }

    // ---- 6. the message tables --------------------------------------------------------------------------------------
    ccp_csr::RowBlock &rb = m->rb;
    rb.comm = c;
    rb.row_begin = first_row;
    rb.n_local = n_rows;
    rb.n_global = n_global;
    rb.n_lo = n_lo;
    rb.n_ghost = n_ghost;
    rb.n_colours = n_colours;
    rb.ghost = ghost;
    auto ext_of_ghost = [&](int k) { return k < n_lo ? k : n_rows + k; };
    rb.nat_send_cnt = serve_cnt;
    rb.nat_send_off = serve_off;
    rb.nat_recv_cnt = req_cnt;
    rb.nat_recv_pos.assign((size_t)W, 0);
    rb.peers = 0;
    for (int r = 0; r < W; ++r) {
        if (req_cnt[(size_t)r] > 0) rb.nat_recv_pos[(size_t)r] = ext_of_ghost(req_off[(size_t)r]);
        rb.peers += (req_cnt[(size_t)r] > 0 || serve_cnt[(size_t)r] > 0) ? 1 : 0;
    }
    std::vector<int> nat_pos((size_t)serve_total);
    for (long k = 0; k < serve_total; ++k) nat_pos[(size_t)k] = n_lo + (serve[(size_t)k] - first_row);
    rb.nat_total = serve_total;
    const Schedule &sc = m->multicolour;
    rb.col_send_cnt.assign((size_t)n_colours * W, 0);
    rb.col_send_off.assign((size_t)n_colours * W, 0);
    rb.col_recv_cnt.assign((size_t)n_colours * W, 0);
    rb.col_recv_pos.assign((size_t)n_colours * W, 0);
    rb.col_seg_off.assign((size_t)n_colours + 1, 0);
    std::vector<int> col_pos;
    col_pos.reserve((size_t)serve_total);
    int table_ok = 1;
    for (int g = 0; g < n_colours; ++g) {
        rb.col_seg_off[(size_t)g] = (long)col_pos.size();
        for (int r = 0; r < W; ++r) {
            const size_t at = (size_t)g * W + r;
            rb.col_send_off[at] = (int)col_pos.size();
            for (int k = serve_off[(size_t)r]; k < serve_off[(size_t)r] + serve_cnt[(size_t)r]; ++k)
                if (serve_colour[(size_t)k] == g) col_pos.push_back(sc.inv[(size_t)(n_lo + (serve[(size_t)k] - first_row))]);
            rb.col_send_cnt[at] = (int)col_pos.size() - rb.col_send_off[at];
            // what rank r owns of my ghosts in colour g: one contiguous range of the colour-major order
            int first = -1, count = 0;
            for (int k = req_off[(size_t)r]; k < req_off[(size_t)r] + req_cnt[(size_t)r]; ++k) {
                if (ghost_colour[(size_t)k] != g) continue;
                const int pos = sc.inv[(size_t)ext_of_ghost(k)];
                if (first < 0) first = pos;
                if (pos != first + count) table_ok = 0;
                ++count;
            }
            rb.col_recv_cnt[at] = count;
            rb.col_recv_pos[at] = std::max(first, 0);
        }
    }
    rb.col_seg_off[(size_t)n_colours] = (long)col_pos.size();
    status = table_ok ? CCP_OK : CCP_ERR_STATE;
    if (status == CCP_OK && serve_total) {
        status = rb.nat_send_pos.alloc((size_t)serve_total);
        if (status == CCP_OK) status = rb.col_send_pos.alloc((size_t)serve_total);
        if (status == CCP_OK) status = rb.sendbuf.alloc((size_t)serve_total);
        if (status == CCP_OK) {
            CCP_HIP(hipMemcpyAsync(rb.nat_send_pos.p, nat_pos.data(), sizeof(int) * (size_t)serve_total, hipMemcpyHostToDevice, s));
            CCP_HIP(hipMemcpyAsync(rb.col_send_pos.p, col_pos.data(), sizeof(int) * (size_t)serve_total, hipMemcpyHostToDevice, s));
            CCP_HIP(hipStreamSynchronize(s));
        }
    }
    // the slices of every colour that hold a row a peer references, and the others
    if (status == CCP_OK && sc.n_slices > 0) {
        std::vector<int> row0((size_t)sc.n_slices);
        CCP_HIP(hipMemcpyAsync(row0.data(), sc.slice_row0.p, sizeof(int) * (size_t)sc.n_slices, hipMemcpyDeviceToHost, s));
        CCP_HIP(hipStreamSynchronize(s));
        std::vector<unsigned char> is_edge((size_t)sc.n_slices, 0);
        long n_edge = 0;
        for (int pos : col_pos) {
            const int sl = (int)(std::upper_bound(row0.begin(), row0.end(), pos) - row0.begin()) - 1;
            if (sl >= 0 && !is_edge[(size_t)sl]) {
                is_edge[(size_t)sl] = 1;
                ++n_edge;
            }
        }
        bool want = n_edge > 0 && n_edge * 4 <= (long)sc.n_slices;
        if (const char *e = getenv("CCP_GS_ROWS_OVERLAP")) want = want && atoi(e) != 0;
        if (want) {
            const int waves = kBlock / kWave;
            std::vector<int> list;
            list.reserve((size_t)sc.n_slices);
            rb.edge_off.assign((size_t)n_colours, 0); rb.edge_cnt.assign((size_t)n_colours, 0);
            rb.inner_off.assign((size_t)n_colours, 0); rb.inner_cnt.assign((size_t)n_colours, 0);
            rb.part_off.assign((size_t)n_colours + 1, 0);
            for (int g = 0; g < n_colours; ++g) {
                const int s0 = g < sc.n_groups ? sc.group_slice_ptr[(size_t)g] : 0, s1 = g < sc.n_groups ? sc.group_slice_ptr[(size_t)g + 1] : 0;
                rb.edge_off[(size_t)g] = (int)list.size();
                for (int sl = s0; sl < s1; ++sl)
                    if (is_edge[(size_t)sl]) list.push_back(sl);
                rb.edge_cnt[(size_t)g] = (int)list.size() - rb.edge_off[(size_t)g];
                rb.inner_off[(size_t)g] = (int)list.size();
                for (int sl = s0; sl < s1; ++sl)
                    if (!is_edge[(size_t)sl]) list.push_back(sl);
                rb.inner_cnt[(size_t)g] = (int)list.size() - rb.inner_off[(size_t)g];
                rb.part_off[(size_t)g + 1] = rb.part_off[(size_t)g] + (rb.edge_cnt[(size_t)g] + waves - 1) / waves + (rb.inner_cnt[(size_t)g] + waves - 1) / waves;
            }
            status = rb.slice_list.alloc(list.size());
            if (status == CCP_OK) {
                CCP_HIP(hipMemcpyAsync(rb.slice_list.p, list.data(), sizeof(int) * list.size(), hipMemcpyHostToDevice, s));
                CCP_HIP(hipStreamSynchronize(s));
                int lo_p = 0, hi_p = 0;                          // hi_p = greatest priority (numerically lowest)
                (void)hipDeviceGetStreamPriorityRange(&lo_p, &hi_p);
                if (hipStreamCreateWithPriority(&rb.stream_comm, hipStreamNonBlocking, hi_p) != hipSuccess ||
                    hipEventCreateWithFlags(&rb.ev_edge, hipEventDisableTiming) != hipSuccess ||
                    hipEventCreateWithFlags(&rb.ev_comm, hipEventDisableTiming) != hipSuccess)
                    status = CCP_ERR_HIP;
                else
                    rb.overlap = true;
            }
        }
    }
    status = rb_agree(c, s, status);
    if (status != CCP_OK) {
        m->uploaded = false;
        rb.reset();
        return status;
    }
    rb.on = true;
    return CCP_OK;
}
}  // namespace

int ccp_csr_upload_rows(ccp_csr *m, ccp_comm *c, int32_t first_row, int32_t n_rows, int32_t n_global, int64_t n_values,
                        const double *values, const int32_t *col_offset, const int32_t *row_begin, const int32_t *row_num_nze,
                        const int32_t *colour, int32_t n_colours)
try {
    int status = CCP_ERR_STATE;
    try {
        status = upload_rows_impl(m, c, first_row, n_rows, n_global, n_values, values, col_offset, row_begin, row_num_nze, colour, n_colours);
    } catch (...) {
        if (m) {
            m->uploaded = false;
            m->rb.reset();
        }
        throw;                                  // (CCP_ABI_CATCH turns it into a status)
    }
    if (status != CCP_OK && m) {
        // after ANY refusal the handle holds no matrix (ccp_gs.h): a caller that ignores the error must not solve an
        // earlier block against peers that have dropped theirs
        m->uploaded = false;
        m->rb.reset();
    }
    return status;
} CCP_ABI_CATCH

int ccp_csr_rows_info(ccp_csr *m, int32_t *first_row, int32_t *n_rows, int32_t *n_ghost, int32_t *n_peers, int32_t *edge_slices,
                      int64_t *values_sent, int64_t *exchanges)
try {
    if (!m) return CCP_ERR_BAD_ARG;
    if (!m->uploaded || !m->rb.on) return CCP_ERR_STATE;
    if (first_row) *first_row = m->rb.row_begin;
    if (n_rows) *n_rows = m->rb.n_local;
    if (n_ghost) *n_ghost = m->rb.n_ghost;
    if (n_peers) *n_peers = m->rb.peers;
    if (edge_slices) {
        *edge_slices = 0;
        if (m->rb.overlap)
            for (int v : m->rb.edge_cnt) *edge_slices += v;
    }
    if (values_sent) *values_sent = m->rb.values_sent;
    if (exchanges) *exchanges = m->rb.exchanges;
    return CCP_OK;
} CCP_ABI_CATCH

int ccp_csr_set_colouring(ccp_csr *m, const int32_t *colour, int32_t n_colours)
try {
    if (!m) return CCP_ERR_BAD_ARG;
    if (!m->uploaded) return CCP_ERR_STATE;
    if (m->rb.on) return CCP_ERR_UNSUPPORTED;      // a row block: its colouring and pattern were agreed with the peers at ccp_csr_upload_rows
    m->multicolour.reset();
    m->user_colour.clear();
    m->user_n_colours = 0;
    m->auto_colour.clear();
    m->greedy_colour.clear();
    m->used_colour.clear();
    m->used_n_colours = 0;
    m->region_state = m->edited ? 0 : -1;          // the embedding's parity follows the colouring
    if (!colour) return CCP_OK;
    if (n_colours < 1) return CCP_ERR_BAD_ARG;
    for (int i = 0; i < m->n_rows; ++i)
        if (colour[i] < 0 || colour[i] >= n_colours) return CCP_ERR_BAD_ARG;
    m->user_colour.assign(colour, colour + m->n_rows);
    m->user_n_colours = n_colours;
    return CCP_OK;
} CCP_ABI_CATCH

int ccp_csr_insert(ccp_csr *m, int32_t row, int32_t col, double val)
try {
    if (!m) return CCP_ERR_BAD_ARG;
    if (!m->uploaded) return CCP_ERR_STATE;
    if (m->rb.on) return CCP_ERR_UNSUPPORTED;      // a row block: its colouring and pattern were agreed with the peers at ccp_csr_upload_rows
    if (row < 0 || row >= m->n_rows || col < 0 || col >= m->n_cols) return CCP_ERR_BAD_ARG;
    // the row's current content, copied into the overlay on its first edit
    auto it = m->overlay.find(row);
    if (it == m->overlay.end()) {
        const RowView r = row_view(m, row);
        ccp_csr::RowContent rc;
        rc.col.assign(r.col, r.col + r.len);
        rc.val.assign(r.val, r.val + r.len);
        it = m->overlay.emplace(row, std::move(rc)).first;
    }
    std::vector<int> &c = it->second.col;
    std::vector<double> &v = it->second.val;
    const auto pos = std::lower_bound(c.begin(), c.end(), col);      // live entries of a row are sorted by column
    const size_t k = (size_t)(pos - c.begin());
    const bool present = pos != c.end() && *pos == col;
    bool changed = false;
    if (val == 0.0) {                                                // insertZero (:183-200): the entry becomes slack
        if (present) {
            c.erase(pos);
            v.erase(v.begin() + (long)k);
            changed = true;
        }
    } else if (present) {                                            // insertNoneZero (:206-212): overwrite in place
        changed = v[k] != val;
        v[k] = val;
    } else {                                                         // (:214-233): a new live entry, kept in column order
        c.insert(pos, col);
        v.insert(v.begin() + (long)k, val);
        changed = true;
    }
    m->stat_edits++;
    if (changed) {
        m->touched.push_back(row);
        m->poisson_w = -1;                                           // the structured twin must be recognised again
        m->region_state = 0;                                         // an edited matrix stays on the general path
        m->greedy_colour.clear();                                    // (a new coupling may make a cached colouring improper)
        m->edited = true;
    }
    return CCP_OK;
} CCP_ABI_CATCH

int ccp_csr_insert_many(ccp_csr *m, int64_t count, const int32_t *rows, const int32_t *cols, const double *vals)
try {
    if (!m || count < 0 || (count > 0 && (!rows || !cols || !vals))) return CCP_ERR_BAD_ARG;
    if (count < 64) {
        for (int64_t k = 0; k < count; ++k) CCP_TRY(ccp_csr_insert(m, rows[k], cols[k], vals[k]));
        return CCP_OK;
    }
    // A batch (a brush stroke; the reference's lab3 benchmark zeroes 200,000 entries of a 1000 x 1000 matrix, main6.cc:
    // 150-182): applied ROW BY ROW — the edits of a row, in the order given, folded into one merge with the row's sorted
    // content instead of one shift of the row per edit (sparse-matrix.h:183-233 moves the tail of the row for every
    // single insert).  Same result as the edits applied one by one: the last edit of a (row, column) wins.
    if (!m->uploaded) return CCP_ERR_STATE;
    if (m->rb.on) return CCP_ERR_UNSUPPORTED;
    for (int64_t k = 0; k < count; ++k)
        if (rows[k] < 0 || rows[k] >= m->n_rows || cols[k] < 0 || cols[k] >= m->n_cols) return CCP_ERR_BAD_ARG;   // (nothing applied)
    std::vector<int64_t> order((size_t)count);
    for (int64_t k = 0; k < count; ++k) order[(size_t)k] = k;
    bool by_row = true;
    for (int64_t k = 1; k < count && by_row; ++k) by_row = rows[k - 1] <= rows[k];
    if (!by_row) std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return rows[a] < rows[b]; });
    std::vector<std::pair<int, int64_t>> ed;                         // (column, position in the batch) of one row's edits
    std::vector<int> nc;
    std::vector<double> nv;
    bool any_changed = false;
    for (int64_t g0 = 0; g0 < count;) {
        const int row = rows[order[(size_t)g0]];
        int64_t g1 = g0;
        while (g1 < count && rows[order[(size_t)g1]] == row) ++g1;
        auto it = m->overlay.find(row);
        if (it == m->overlay.end()) {
            const RowView r = row_view(m, row);
            ccp_csr::RowContent rc;
            rc.col.assign(r.col, r.col + r.len);
            rc.val.assign(r.val, r.val + r.len);
            it = m->overlay.emplace(row, std::move(rc)).first;
        }
        std::vector<int> &c = it->second.col;
        std::vector<double> &v = it->second.val;
        // the row's edits by column, the LAST one of a column kept
        ed.clear();
        for (int64_t k = g0; k < g1; ++k) ed.emplace_back(cols[order[(size_t)k]], order[(size_t)k]);
        std::stable_sort(ed.begin(), ed.end(), [](const std::pair<int, int64_t> &a, const std::pair<int, int64_t> &b) { return a.first < b.first; });
        nc.clear();
        nv.clear();
        nc.reserve(c.size() + ed.size());
        nv.reserve(c.size() + ed.size());
        bool changed = false;
        size_t i = 0, e = 0;
        while (i < c.size() || e < ed.size()) {
            if (e == ed.size() || (i < c.size() && c[i] < ed[e].first)) {            // an entry no edit touches
                nc.push_back(c[i]);
                nv.push_back(v[i]);
                ++i;
                continue;
            }
            size_t last = e;                                                            // the last edit of this column
            while (last + 1 < ed.size() && ed[last + 1].first == ed[e].first) ++last;
            const int col = ed[e].first;
            const double val = vals[ed[last].second];
            const bool present = i < c.size() && c[i] == col;
            if (val == 0.0) {
                changed |= present;                                                     // insertZero: the entry becomes slack
            } else {
                changed |= !present || v[i] != val;
                nc.push_back(col);
                nv.push_back(val);
            }
            if (present) ++i;
            e = last + 1;
        }
        m->stat_edits += g1 - g0;
        if (changed) {
            c.swap(nc);
            v.swap(nv);
            m->touched.push_back(row);
            any_changed = true;
        }
        g0 = g1;
    }
    if (any_changed) {
        m->poisson_w = -1;                                           // (as ccp_csr_insert)
        m->region_state = 0;
        m->greedy_colour.clear();
        m->edited = true;
    }
    return CCP_OK;
} CCP_ABI_CATCH

// Host-only diagnostic (no device needed): run the raster-region recognition on a compressed CSR matrix with
// a 2-colouring and return the reconstructed pixel coordinates.  *recognised = 0 when the matrix is not such
// a Laplacian (or the reconstruction does not verify).  tests/test_region_embedding.py.
int ccp_csr_embed_region_host(int32_t n, const int32_t *row_offset, const int32_t *col, const double *val, const int32_t *colour,
                              int32_t *recognised, int32_t *canvas_width, int32_t *canvas_height, int32_t *x_out, int32_t *y_out)
try {
    if (n < 0 || !row_offset || !recognised || (n > 0 && (!col || !val || !colour))) return CCP_ERR_BAD_ARG;
    ccp_csr tmp;
    tmp.n_rows = tmp.n_cols = n;
    tmp.row_ptr.assign(row_offset, row_offset + n + 1);
    tmp.col.assign(col, col + row_offset[n]);
    tmp.val.assign(val, val + row_offset[n]);
    const std::vector<int> colours(colour, colour + n);
    RegionEmbedding E;
    const bool ok = n >= 1 && embed_region(&tmp, colours, E);
    *recognised = ok ? 1 : 0;
    if (ok) {
        if (canvas_width) *canvas_width = E.W;
        if (canvas_height) *canvas_height = E.H;
        if (x_out) std::memcpy(x_out, E.x.data(), sizeof(int32_t) * (size_t)n);
        if (y_out) std::memcpy(y_out, E.y.data(), sizeof(int32_t) * (size_t)n);
    }
    return CCP_OK;
} CCP_ABI_CATCH

int ccp_csr_last_path(ccp_csr *m, int32_t *path, int32_t *canvas_width, int32_t *canvas_height, int64_t *sweep_launches)
try {
    if (!m) return CCP_ERR_BAD_ARG;
    if (path) *path = m->last_path;
    if (sweep_launches) *sweep_launches = m->last_path == CCP_PATH_REGION_GRID ? m->last_launches : 0;
    if (canvas_width) *canvas_width = m->last_path == CCP_PATH_REGION_GRID ? m->region_w : (m->last_path == CCP_PATH_POISSON_GRID ? m->poisson_w : 0);
    if (canvas_height) *canvas_height = m->last_path == CCP_PATH_REGION_GRID ? m->region_h : (m->last_path == CCP_PATH_POISSON_GRID ? m->poisson_h : 0);
    return CCP_OK;
} CCP_ABI_CATCH

int ccp_csr_edit_stats(ccp_csr *m, int64_t *edits, int64_t *image_uploads, int64_t *rows_patched, int64_t *slices_relocated,
                       int64_t *image_rebuilds)
try {
    if (!m) return CCP_ERR_BAD_ARG;
    if (edits) *edits = m->stat_edits;
    if (image_uploads) *image_uploads = m->stat_uploads;
    if (rows_patched) *rows_patched = m->stat_rows_patched;
    if (slices_relocated) *slices_relocated = m->stat_slices_relocated;
    if (image_rebuilds) *image_rebuilds = m->stat_schedule_rebuilds;
    return CCP_OK;
} CCP_ABI_CATCH

int ccp_csr_device_footprint(ccp_csr *m, int64_t *stored_matrix_bytes, int64_t *eager_copies_skipped)
try {
    if (!m) return CCP_ERR_BAD_ARG;
    wait_device_upload(m);
    if (stored_matrix_bytes)
        *stored_matrix_bytes = (int64_t)(m->d_row_ptr.p ? m->d_row_ptr.n * sizeof(long) : 0) + (int64_t)(m->d_col.p ? m->d_col.n * sizeof(int) : 0) +
                               (int64_t)(m->d_val.p ? m->d_val.n * sizeof(double) : 0);
    if (eager_copies_skipped) *eager_copies_skipped = m->stat_eager_skipped;
    return CCP_OK;
} CCP_ABI_CATCH

int ccp_csr_get_colouring(ccp_csr *m, int32_t *colour, int32_t *n_colours)
try {
    CCP_TRY(bind(m));
    if (!m->uploaded) return CCP_ERR_STATE;
    CCP_TRY(flush_edits(m));
    if (!n_colours) return CCP_ERR_BAD_ARG;
    // (no colouring from the caller, nothing resolved yet: a raster region takes its canvas parity — detect_region)
    if (m->user_colour.empty() && m->allow_structured && m->allow_region && m->region_state < 0 && !m->multicolour.built && !m->rb.on)
        CCP_TRY(detect_region(m));
    if (m->region_state == 1 && !m->multicolour.built) {          // the raster-region twin sweeps (x + y) & 1 = the resolved colouring
        *n_colours = 2;
        if (colour && m->n_rows) std::memcpy(colour, m->region_colour.data(), sizeof(int32_t) * (size_t)m->n_rows);
        return CCP_OK;
    }
    CCP_TRY(ensure_multicolour(m));          // colours the rows now if no solve has done so yet
    *n_colours = m->used_n_colours;
    if (m->rb.on) {                          // a row block: the colours of the block's own rows
        *n_colours = m->rb.n_colours;        // (the whole matrix's count, also on a block that holds none of some colour)
        if (colour && m->rb.n_local) std::memcpy(colour, m->used_colour.data() + m->rb.n_lo, sizeof(int32_t) * (size_t)m->rb.n_local);
        return CCP_OK;
    }
    if (colour && m->n_rows) std::memcpy(colour, m->used_colour.data(), sizeof(int32_t) * (size_t)m->n_rows);
    return CCP_OK;
} CCP_ABI_CATCH

int ccp_csr_gauss_seidel(ccp_csr *m, const double *b, const double *x0, double *x_out, double epsilon,
                         int32_t max_iteration, int32_t check_every, int32_t ordering, ccp_gs_report *report)
try {
    CCP_TRY(bind(m));
    if (!m->uploaded) return CCP_ERR_STATE;
    CCP_TRY(flush_edits(m));
    if (!b || !x_out || check_every < 0) return CCP_ERR_BAD_ARG;
    if (m->n_rows != m->n_cols) return CCP_ERR_UNSUPPORTED;   // the reference asserts len(b) == n_cols and sizes x from b
    if (ordering != CCP_ORDER_LEXICOGRAPHIC && ordering != CCP_ORDER_MULTICOLOUR) return CCP_ERR_BAD_ARG;
    // a row block: the index-order sweep is one dependence chain through all the blocks — only the colour order shards
    if (m->rb.on && ordering != CCP_ORDER_MULTICOLOUR) return CCP_ERR_UNSUPPORTED;
    const bool rowblock = m->rb.on;
    if (ordering == CCP_ORDER_MULTICOLOUR && m->allow_structured) {
        // The matrix SolveChannel builds (via Eigen, ConvertFromEigen) is recognised and solved
        // matrix-free by the grid kernels: same sweep order, same arithmetic, same bits as the
        // sliced-ELL path, ~20x its speed (temporally blocked sweep).
        detect_poisson(m);
        if (m->poisson_w > 0 && colouring_is_checkerboard(m)) {
            if (!m->grid) {
                ccp_grid_desc d{m->poisson_w, m->poisson_h, 1, 0, m->poisson_h, 0, m->device, 0};
                CCP_TRY(ccp_grid_create(&d, &m->grid));
                (void)grid_set_allow_swap(m->grid, true);  // (results leave through ccp_grid_get_x_host)
                release_device_csr(m);                     // nothing on the device reads the stored matrix any more
            }
            CCP_TRY(ccp_grid_set_stream(m->grid, m->stream));
            CCP_TRY(ccp_grid_set_b_host(m->grid, 0, b, 0, m->poisson_h));
            if (x0) CCP_TRY(ccp_grid_set_x_host(m->grid, 0, x0, 0, m->poisson_h));
            else CCP_TRY(ccp_grid_fill_x(m->grid, 1.0));                       // sparse-matrix.h:352
            CCP_TRY(ccp_grid_gauss_seidel(m->grid, epsilon, max_iteration, check_every, report));
            m->last_path = CCP_PATH_POISSON_GRID;
            return ccp_grid_get_x_host(m->grid, 0, x_out, 0, m->poisson_h);
        }
    }
    if (m->allow_structured && m->allow_region) {
        // The 5-point Laplacian of a raster region (a brush / label region of a blend; BASELINE configs[4]):
        // swept matrix-free by the Dirichlet-mask grid kernels on a canvas the region is embedded in —
        // 24 B per pixel per PASS of several iterations instead of 92 B per row per iteration.  Same colour
        // order, same arithmetic, same bits as the sliced-ELL sweep (the step sums to summation order).
        // In the reference's own order the canvas is swept in raster order (k_lex_wg, Dirichlet-mask variant):
        // the unknowns are numbered in raster order and every piece of the region keeps its shape on the canvas,
        // so two coupled unknowns are met in the order of their indices — the index-order sweep, bit for bit.
        CCP_TRY(detect_region(m, ordering == CCP_ORDER_LEXICOGRAPHIC));
        if (m->region_state == 1 || (m->region_state == 2 && ordering == CCP_ORDER_LEXICOGRAPHIC)) {
            const long n = m->n_rows;
            hipStream_t s = m->stream;
            ccp_grid *g = m->region_grid;
            CCP_TRY(ccp_grid_set_stream(g, s));
            // the first solve runs on the default tiling of a mask grid; a matrix that is solved again gets its tiling
            // timed once (speed only: ~0.1 s at 41.75 M unknowns, forty times a 50-sweep solve)
            if (!m->region_tuned && ++m->region_solves >= 2) {
                CCP_TRY(ccp_grid_tune(g, 8, nullptr, nullptr, nullptr));
                m->region_tuned = true;
            }
            ccp_grid_layout lay{};
            CCP_TRY(ccp_grid_get_layout(g, &lay));
            double *gx = static_cast<double *>(lay.x_dev), *gb = static_cast<double *>(lay.b_dev);
            CCP_HIP(hipMemcpyAsync(m->tmp.p, b, sizeof(double) * n, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL((k_canvas_move<0>), dim3(blocks_for(n)), dim3(kBlock), 0, s, gb, m->tmp.p, m->region_where.p, n, 0.0);
            CCP_HIP(hipGetLastError());
            if (x0) {
                CCP_HIP(hipStreamSynchronize(s));
                CCP_HIP(hipMemcpyAsync(m->tmp.p, x0, sizeof(double) * n, hipMemcpyHostToDevice, s));
                hipLaunchKernelGGL((k_canvas_move<0>), dim3(blocks_for(n)), dim3(kBlock), 0, s, gx, m->tmp.p, m->region_where.p, n, 0.0);
            } else {
                hipLaunchKernelGGL((k_canvas_move<2>), dim3(blocks_for(n)), dim3(kBlock), 0, s, gx, m->tmp.p, m->region_where.p, n, 1.0);   // sparse-matrix.h:352
            }
            CCP_HIP(hipGetLastError());
            bool swept = true;
            if (ordering == CCP_ORDER_LEXICOGRAPHIC) {
                const int st = ccp_grid_gauss_seidel_lexicographic(g, epsilon, max_iteration, check_every, report);
                if (st == CCP_ERR_UNSUPPORTED) swept = false;     // (an engine chosen by CCP_GS_LEX_MODE that knows no masks)
                else CCP_TRY(st);
                m->last_launches = 0;
            } else {
                CCP_TRY(ccp_grid_region_begin(g));
                CCP_TRY(ccp_grid_gauss_seidel(g, epsilon, max_iteration, check_every, report));
                float region_ms = 0.f;
                int64_t launches = 0, pass_iters = 0;
                CCP_TRY(ccp_grid_region_end(g, &region_ms, &launches, &pass_iters));
                m->last_launches = launches;
            }
            if (swept) {
                CCP_TRY(ccp_grid_get_layout(g, &lay));             // (x and its ping-pong partner may have swapped roles)
                gx = static_cast<double *>(lay.x_dev);
                hipLaunchKernelGGL((k_canvas_move<1>), dim3(blocks_for(n)), dim3(kBlock), 0, s, gx, m->tmp.p, m->region_where.p, n, 0.0);
                CCP_HIP(hipGetLastError());
                CCP_HIP(hipMemcpyAsync(x_out, m->tmp.p, sizeof(double) * n, hipMemcpyDeviceToHost, s));
                CCP_HIP(hipStreamSynchronize(s));
                m->last_path = CCP_PATH_REGION_GRID;
                return CCP_OK;
            }
        }
    }
    if (ordering == CCP_ORDER_LEXICOGRAPHIC && m->allow_structured) {
        // The reference's own order on the recognised Poisson matrix: the hyperplane pipeline of
        // ccp_grid_lex.hpp — the same iterates as the level schedule below, without a launch per level
        // and per sweep.
        detect_poisson(m);
        if (m->poisson_w > 1 && m->poisson_h > 1) {
            if (!m->grid) {
                ccp_grid_desc d{m->poisson_w, m->poisson_h, 1, 0, m->poisson_h, 0, m->device, 0};
                CCP_TRY(ccp_grid_create(&d, &m->grid));
                (void)grid_set_allow_swap(m->grid, true);  // (results leave through ccp_grid_get_x_host)
                release_device_csr(m);                     // nothing on the device reads the stored matrix any more
            }
            CCP_TRY(ccp_grid_set_stream(m->grid, m->stream));
            CCP_TRY(ccp_grid_set_b_host(m->grid, 0, b, 0, m->poisson_h));
            if (x0) CCP_TRY(ccp_grid_set_x_host(m->grid, 0, x0, 0, m->poisson_h));
            else CCP_TRY(ccp_grid_fill_x(m->grid, 1.0));                       // sparse-matrix.h:352
            CCP_TRY(ccp_grid_gauss_seidel_lexicographic(m->grid, epsilon, max_iteration, check_every, report));
            m->last_path = CCP_PATH_POISSON_GRID;
            return ccp_grid_get_x_host(m->grid, 0, x_out, 0, m->poisson_h);
        }
    }
    m->last_path = CCP_PATH_SLICED_ELL;
    Schedule &sc = ordering == CCP_ORDER_MULTICOLOUR ? m->multicolour : m->lexicographic;
    CCP_TRY(ordering == CCP_ORDER_MULTICOLOUR ? ensure_multicolour(m) : ensure_lexicographic(m));
    const long n = m->n_rows;
    hipStream_t s = m->stream;
    CCP_TRY(ensure_partial(m, std::max<long>(sc.group_block_off[sc.n_groups], (rowblock && m->rb.overlap) ? m->rb.part_off.back() : 0)));
    // stage b (and x0) in natural order, gather into schedule order (a row block: b, x0 and x_out hold the block's
    // own rows; the ghosts sit around them in the extended order and get their first values from their owners)
    if (n) {
        if (rowblock) CCP_TRY(rb_stage(m, m->tmp.p, b));
        else CCP_HIP(hipMemcpyAsync(m->tmp.p, b, sizeof(double) * n, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL((k_permute<true>), dim3(blocks_for(n)), dim3(kBlock), 0, s, m->b.p, m->tmp.p, sc.perm.p, n);
        CCP_HIP(hipGetLastError());
        if (x0) {
            CCP_HIP(hipStreamSynchronize(s));
            if (rowblock) CCP_TRY(rb_stage(m, m->tmp.p, x0));
            else CCP_HIP(hipMemcpyAsync(m->tmp.p, x0, sizeof(double) * n, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL((k_permute<true>), dim3(blocks_for(n)), dim3(kBlock), 0, s, m->x.p, m->tmp.p, sc.perm.p, n);
        } else {
            hipLaunchKernelGGL(k_fill_n, dim3(blocks_for(n)), dim3(kBlock), 0, s, m->x.p, n, 1.0);   // sparse-matrix.h:352
        }
        CCP_HIP(hipGetLastError());
    }
    if (rowblock && x0)
        for (int g = 0; g < m->rb.n_colours; ++g) CCP_TRY(rb_exchange_colour(m, g, s));
    CsrSolveState host{};
    host.active = 1;
    host.last_eps = 10.0;                    // sparse-matrix.h:354
    CCP_HIP(hipMemcpyAsync(m->state.p, &host, sizeof(host), hipMemcpyHostToDevice, s));
    CCP_HIP(hipStreamSynchronize(s));
    CCP_HIP(hipEventRecord(m->ev0, s));
    const SellView view = sc.view();
    const int *active = reinterpret_cast<const int *>(m->state.p);
    double *eps_accum = reinterpret_cast<double *>(reinterpret_cast<char *>(m->state.p) + offsetof(CsrSolveState, eps_accum));
    int issued = 0;
    bool any_active = (10.0 > epsilon) && max_iteration > 0 && (n > 0 || rowblock);   // (an empty block still takes part in the stop rule)
    // narrow groups (<= 1024 rows: level schedules of grids up to ~1000 px wide, small matrices): the
    // whole solve in one workgroup; wider groups keep one launch per group (one CU cannot feed them)
    const bool one_block = m->allow_one_block && sc.n_groups > 0 && sc.max_group_slices <= 16;
    if (any_active && one_block) {
        hipLaunchKernelGGL(k_sell_gs_one_block, dim3(1), dim3(kSerialBlock), 0, s, view, sc.group_ptr_dev.p, sc.n_groups, m->x.p,
                           m->b.p, m->state.p, epsilon, max_iteration, check_every);
        CCP_HIP(hipGetLastError());
        issued = max_iteration;
        any_active = false;
    }
    if (any_active && ordering == CCP_ORDER_LEXICOGRAPHIC && m->allow_pipeline && sc.level_span >= 0 && sc.n_groups > 1) {
        // level schedule pipelined over sweeps (k_sell_gs_pipe): n_levels + stride*K launches per batch
        const int stride = sc.level_span + 1, n_levels = sc.n_groups;
        unsigned max_blocks = 1;
        for (int g = 0; g < n_levels; ++g)
            max_blocks = std::max(max_blocks, (unsigned)(sc.group_block_off[g + 1] - sc.group_block_off[g]));
        const long per_sweep = sc.group_block_off[n_levels];
        auto run = [&](int sweeps, double *partial) -> int {
            for (long tau = 0; tau <= (long)(n_levels - 1) + (long)stride * (sweeps - 1); ++tau) {
                const long k_hi = std::min<long>(sweeps - 1, tau / stride);
                const long k_lo = std::max<long>(0, (tau - (n_levels - 1) + stride - 1) / stride);
                if (k_hi < k_lo) continue;
                dim3 grid(max_blocks, (unsigned)(k_hi - k_lo + 1));
                if (partial)
                    hipLaunchKernelGGL((k_sell_gs_pipe<true>), grid, dim3(kBlock), 0, s, view, sc.group_ptr_dev.p,
                                       sc.group_block_off_dev.p, (int)tau, stride, (int)k_lo, m->x.p, m->b.p, partial, per_sweep);
                else
                    hipLaunchKernelGGL((k_sell_gs_pipe<false>), grid, dim3(kBlock), 0, s, view, sc.group_ptr_dev.p,
                                       sc.group_block_off_dev.p, (int)tau, stride, (int)k_lo, m->x.p, m->b.p,
                                       static_cast<double *>(nullptr), per_sweep);
            }
            CCP_HIP(hipGetLastError());
            return CCP_OK;
        };
        if (check_every == 0) {
            while (issued < max_iteration) {               // gridDim.y carries the sweeps in flight: keep it small
                const int kb = std::min(32768, max_iteration - issued);
                CCP_TRY(run(kb, nullptr));
                issued += kb;
            }
        } else {
            const int batch_max = 128;
            if (m->pipe_partial.n != (size_t)per_sweep * batch_max) CCP_TRY(m->pipe_partial.alloc((size_t)per_sweep * batch_max));
            if (m->pipe_eps.n != (size_t)batch_max) CCP_TRY(m->pipe_eps.alloc(batch_max));
            if (m->pipe_snap.n != (size_t)std::max<long>(n, 2)) CCP_TRY(m->pipe_snap.alloc((size_t)std::max<long>(n, 2)));
            std::vector<double> eps_host(batch_max);
            bool stopped = false;
            while (!stopped && issued < max_iteration) {
                const int kb = std::min(batch_max, max_iteration - issued);
                CCP_HIP(hipMemcpyAsync(m->pipe_snap.p, m->x.p, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
                CCP_TRY(run(kb, m->pipe_partial.p));
                hipLaunchKernelGGL(k_reduce_sweeps, dim3((unsigned)kb), dim3(kBlock), 0, s, m->pipe_partial.p, per_sweep, m->pipe_eps.p);
                CCP_HIP(hipGetLastError());
                CCP_HIP(hipMemcpyAsync(eps_host.data(), m->pipe_eps.p, sizeof(double) * kb, hipMemcpyDeviceToHost, s));
                CCP_HIP(hipStreamSynchronize(s));
                int stop = -1;
                for (int k = 0; k < kb; ++k) {
                    if ((issued + k + 1) % check_every != 0) continue;
                    host.last_eps = eps_host[k];
                    if (!(host.last_eps > epsilon)) {
                        stop = k;
                        break;
                    }
                }
                if (stop >= 0) {
                    stopped = true;
                    host.converged = 1;
                    host.iterations = issued + stop + 1;
                    if (stop < kb - 1) {      // the pipeline ran past the stop sweep: redo exactly stop+1 sweeps from the snapshot
                        CCP_HIP(hipMemcpyAsync(m->x.p, m->pipe_snap.p, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
                        CCP_TRY(run(stop + 1, nullptr));
                    }
                }
                issued += kb;
            }
            host.active = stopped ? 0 : 1;
            CCP_HIP(hipMemcpyAsync(m->state.p, &host, sizeof(host), hipMemcpyHostToDevice, s));
        }
        any_active = false;
    }
    while (any_active && issued < max_iteration) {
        int checks = 0;
        while (issued < max_iteration && checks < 8) {
            const int k = issued + 1;
            const bool check = check_every > 0 && (k % check_every == 0);
            for (int g = 0; g < (rowblock ? m->rb.n_colours : sc.n_groups); ++g) {
                const int s0 = g < sc.n_groups ? sc.group_slice_ptr[g] : 0, s1 = g < sc.n_groups ? sc.group_slice_ptr[g + 1] : 0;
                if (rowblock && m->rb.overlap) {
                    // edge slices, then their values on the way while the rest of the colour is swept
                    ccp_csr::RowBlock &rb = m->rb;
                    const int waves = kBlock / kWave;
                    const int ne = rb.edge_cnt[(size_t)g], ni = rb.inner_cnt[(size_t)g];
                    const unsigned be = (unsigned)((ne + waves - 1) / waves), bi = (unsigned)((ni + waves - 1) / waves);
                    double *part = m->partial.p + rb.part_off[(size_t)g];
                    if (ne) {
                        if (check) hipLaunchKernelGGL((k_sell_gs_list<true>), dim3(be), dim3(kBlock), 0, s, view, rb.slice_list.p + rb.edge_off[(size_t)g], ne, m->x.p, m->b.p, part, active);
                        else hipLaunchKernelGGL((k_sell_gs_list<false>), dim3(be), dim3(kBlock), 0, s, view, rb.slice_list.p + rb.edge_off[(size_t)g], ne, m->x.p, m->b.p, part, active);
                    }
                    CCP_HIP(hipEventRecord(rb.ev_edge, s));
                    CCP_HIP(hipStreamWaitEvent(rb.stream_comm, rb.ev_edge, 0));
                    CCP_TRY(rb_exchange_colour(m, g, rb.stream_comm));
                    CCP_HIP(hipEventRecord(rb.ev_comm, rb.stream_comm));
                    if (ni) {
                        if (check) hipLaunchKernelGGL((k_sell_gs_list<true>), dim3(bi), dim3(kBlock), 0, s, view, rb.slice_list.p + rb.inner_off[(size_t)g], ni, m->x.p, m->b.p, part + be, active);
                        else hipLaunchKernelGGL((k_sell_gs_list<false>), dim3(bi), dim3(kBlock), 0, s, view, rb.slice_list.p + rb.inner_off[(size_t)g], ni, m->x.p, m->b.p, part + be, active);
                    }
                    CCP_HIP(hipStreamWaitEvent(s, rb.ev_comm, 0));       // the next colour reads the ghosts
                    continue;
                }
                if (s1 > s0) {
                    const unsigned blocks = (unsigned)(sc.group_block_off[g + 1] - sc.group_block_off[g]);
                    if (check)
                        hipLaunchKernelGGL((k_sell_gs<true>), dim3(blocks), dim3(kBlock), 0, s, view, s0, s1, m->x.p, m->b.p,
                                           m->partial.p + sc.group_block_off[g], active);
                    else
                        hipLaunchKernelGGL((k_sell_gs<false>), dim3(blocks), dim3(kBlock), 0, s, view, s0, s1, m->x.p, m->b.p,
                                           m->partial.p, active);
                }
                // a row block: the colour's new values reach the blocks that reference them before the next colour runs
                if (rowblock) CCP_TRY(rb_exchange_colour(m, g, s));
            }
            CCP_HIP(hipGetLastError());
            if (check) {
                const long n_part = (rowblock && m->rb.overlap) ? m->rb.part_off.back() : sc.group_block_off[sc.n_groups];
                hipLaunchKernelGGL(k_reduce_to, dim3(1), dim3(kBlock), 0, s, m->partial.p, n_part, 1L, eps_accum, 0);
                if (rowblock) {
                    // sparse-matrix.h:376 over the whole matrix: the blocks' step sums added up, on every rank alike
                    const RcclApi *api = rccl_api();
                    if (!api) return CCP_ERR_RCCL;
                    CCP_RCCL(api->AllReduce(eps_accum, eps_accum, 1, ncclDouble, ncclSum, m->rb.comm->comm, s));
                }
                hipLaunchKernelGGL(k_csr_check, dim3(1), dim3(64), 0, s, m->state.p, epsilon, k);
                CCP_HIP(hipGetLastError());
                ++checks;
            }
            ++issued;
        }
        if (check_every > 0) {
            CCP_HIP(hipMemcpyAsync(&host, m->state.p, sizeof(host), hipMemcpyDeviceToHost, s));
            CCP_HIP(hipStreamSynchronize(s));
            any_active = host.active != 0;
        }
    }
    CCP_HIP(hipEventRecord(m->ev1, s));
    if (n) {
        hipLaunchKernelGGL((k_permute<false>), dim3(blocks_for(n)), dim3(kBlock), 0, s, m->tmp.p, m->x.p, sc.perm.p, n);
        CCP_HIP(hipGetLastError());
        if (!rowblock) CCP_HIP(hipMemcpyAsync(x_out, m->tmp.p, sizeof(double) * n, hipMemcpyDeviceToHost, s));
        else if (m->rb.n_local) CCP_HIP(hipMemcpyAsync(x_out, m->tmp.p + m->rb.n_lo, sizeof(double) * (size_t)m->rb.n_local, hipMemcpyDeviceToHost, s));
    }
    CCP_HIP(hipMemcpyAsync(&host, m->state.p, sizeof(host), hipMemcpyDeviceToHost, s));
    CCP_HIP(hipStreamSynchronize(s));
    if (report) {
        float ms = 0.f;
        CCP_HIP(hipEventElapsedTime(&ms, m->ev0, m->ev1));
        report->converged = host.converged;
        report->iterations = host.converged ? host.iterations : issued;
        report->last_l1_step = host.last_eps;
        report->seconds = ms * 1e-3;
    }
    return CCP_OK;
} CCP_ABI_CATCH

int ccp_csr_conjugate_gradient(ccp_csr *m, const double *b, const double *init, double *x_out, double epsilon,
                               int32_t max_iteration, ccp_gs_report *report)
try {
    CCP_TRY(bind(m));
    if (!m->uploaded) return CCP_ERR_STATE;
    CCP_TRY(flush_edits(m));
    const bool rowblock = m->rb.on;
    if ((!b || !x_out) && !(rowblock && m->rb.n_local == 0)) return CCP_ERR_BAD_ARG;
    if (m->n_rows != m->n_cols) return CCP_ERR_UNSUPPORTED;
    if (m->allow_structured) {
        // SolveChannel's matrix at the unchanged call site (PhotoMontage.cpp:613): matrix-free SpMV of the
        // grid path (16 B per pixel instead of ~92 B per row of the sliced-ELL image); same row order in
        // the products, reductions tree-ordered either way
        detect_poisson(m);
        if (m->poisson_w > 1 && m->poisson_h > 1) {
            if (!m->grid) {
                ccp_grid_desc d{m->poisson_w, m->poisson_h, 1, 0, m->poisson_h, 0, m->device, 0};
                CCP_TRY(ccp_grid_create(&d, &m->grid));
                (void)grid_set_allow_swap(m->grid, true);  // (results leave through ccp_grid_get_x_host)
                release_device_csr(m);                     // nothing on the device reads the stored matrix any more
            }
            CCP_TRY(ccp_grid_set_stream(m->grid, m->stream));
            CCP_TRY(ccp_grid_set_b_host(m->grid, 0, b, 0, m->poisson_h));
            if (init) CCP_TRY(ccp_grid_set_x_host(m->grid, 0, init, 0, m->poisson_h));
            else CCP_TRY(ccp_grid_fill_x(m->grid, 0.0));                       // sparse-matrix.h:397
            CCP_TRY(ccp_grid_conjugate_gradient(m->grid, epsilon, max_iteration, report));
            return ccp_grid_get_x_host(m->grid, 0, x_out, 0, m->poisson_h);
        }
    }
    CCP_TRY(ensure_natural(m));
    const long n = m->n_rows;
    hipStream_t s = m->stream;
    if (!m->cg_p.p || m->cg_p.n < (size_t)std::max<long>(n, 2)) {
        CCP_TRY(m->cg_p.alloc((size_t)std::max<long>(n, 2)));
        CCP_TRY(m->cg_ap.alloc((size_t)std::max<long>(n, 2)));
        CCP_TRY(m->cg_state.alloc(1));
    }
    const unsigned spmv_blocks = (unsigned)std::min<long>(2048, (m->natural.n_slices + kBlock / kWave - 1) / (kBlock / kWave));   // (k_sell_apply strides)
    CCP_TRY(ensure_partial(m, std::max<long>(2048, spmv_blocks)));
    if (rowblock) {
        // b, init, x_out: the block's own rows.  The loop runs on the extended system: a ghost is an empty row, so its
        // entries of A p and r are zero and take no part in the dot products; its entry of the vector A is applied to
        // comes from its owner before every product, and every dot product is added up over the ranks.
        CCP_TRY(rb_stage(m, m->b.p, b));
        if (init) CCP_TRY(rb_stage(m, m->x.p, init));
        else CCP_HIP(hipMemsetAsync(m->x.p, 0, sizeof(double) * (size_t)std::max<long>(n, 1), s));
    } else if (n) {
        CCP_HIP(hipMemcpyAsync(m->b.p, b, sizeof(double) * n, hipMemcpyHostToDevice, s));
        if (init) CCP_HIP(hipMemcpyAsync(m->x.p, init, sizeof(double) * n, hipMemcpyHostToDevice, s));
        else CCP_HIP(hipMemsetAsync(m->x.p, 0, sizeof(double) * n, s));                 // sparse-matrix.h:397
    }
    const SellView view = m->natural.view();
    const int n_slices = m->natural.n_slices;
    auto spmv = [&](const double *in, double *out) -> int {
        if (rowblock) CCP_TRY(rb_exchange_natural(m, const_cast<double *>(in)));
        if (n_slices == 0) return CCP_OK;
        hipLaunchKernelGGL((k_sell_apply<0>), dim3(spmv_blocks), dim3(kBlock), 0, s, view, n_slices, in, out, in, m->partial.p);
        return hipGetLastError() == hipSuccess ? CCP_OK : CCP_ERR_HIP;
    };
    auto spmv_dot = [&](const double *in, double *out, int *n_partials) -> int {
        *n_partials = 0;
        if (rowblock) CCP_TRY(rb_exchange_natural(m, const_cast<double *>(in)));
        if (n_slices == 0) return CCP_OK;
        hipLaunchKernelGGL((k_sell_apply<2>), dim3(spmv_blocks), dim3(kBlock), 0, s, view, n_slices, in, out, in, m->partial.p);
        *n_partials = (int)spmv_blocks;
        return hipGetLastError() == hipSuccess ? CCP_OK : CCP_ERR_HIP;
    };
    if (rowblock) {
        const RcclApi *api = rccl_api();
        if (!api) return CCP_ERR_RCCL;
        double *total = reinterpret_cast<double *>(m->rb.comm->scratch.p);     // (the communicator's own scratch words)
        auto sums = [&](double *partial, int *count, const double **sum_at, int slot) -> int {
            hipLaunchKernelGGL(k_reduce_to, dim3(1), dim3(kBlock), 0, s, partial, (long)*count, 1L, total + slot, 0);
            CCP_HIP(hipGetLastError());
            CCP_RCCL(api->AllReduce(total + slot, total + slot, 1, ncclDouble, ncclSum, m->rb.comm->comm, s));
            *count = 1;
            *sum_at = total + slot;
            return CCP_OK;
        };
        CCP_TRY(cg_solve(spmv, spmv_dot, m->b.p, m->x.p, m->tmp.p, m->cg_p.p, m->cg_ap.p, n, epsilon, max_iteration, m->cg_state.p,
                         m->partial.p, s, m->ev0, m->ev1, report, sums, true));
        if (m->rb.n_local) CCP_HIP(hipMemcpyAsync(x_out, m->x.p + m->rb.n_lo, sizeof(double) * (size_t)m->rb.n_local, hipMemcpyDeviceToHost, s));
        CCP_HIP(hipStreamSynchronize(s));
        return CCP_OK;
    }
    // One GPU: the fused loop (ccp_cg.hpp: 12 nnz + 72 B per row and iteration instead of 12 nnz + 88; the same iterates
    // bit for bit).  CCP_GS_CG_FUSED=0: the three-pass loop (the tests run both).  A row block keeps the three-pass loop:
    // the fused pass would need the ghost rows' r as well as their p before every product (two exchanges, not one).
    const bool fused = !(getenv("CCP_GS_CG_FUSED") && atoi(getenv("CCP_GS_CG_FUSED")) == 0);
    if (fused && n_slices > 0) {
        if (m->cg_p2.n < (size_t)std::max<long>(n, 2)) CCP_TRY(m->cg_p2.alloc((size_t)std::max<long>(n, 2)));
        CCP_HIP(hipMemsetAsync(m->cg_p2.p, 0, sizeof(double) * (size_t)std::max<long>(n, 2), s));
        auto apply = [&](double *xv, const double *rv, const double *p_in, double *p_out, double *apv, int *n_partials) -> int {
            hipLaunchKernelGGL(k_sell_cg_apply, dim3(spmv_blocks), dim3(kBlock), 0, s, view, n_slices, xv, rv, p_in, p_out, apv, m->partial.p,
                               m->cg_state.p);
            *n_partials = (int)spmv_blocks;
            return hipGetLastError() == hipSuccess ? CCP_OK : CCP_ERR_HIP;
        };
        CCP_TRY(cg_solve_fused(spmv, apply, m->b.p, m->x.p, m->tmp.p, m->cg_p.p, m->cg_p2.p, m->cg_ap.p, n, epsilon, max_iteration,
                               m->cg_state.p, m->partial.p, s, m->ev0, m->ev1, report));
    } else {
        CCP_TRY(cg_solve(spmv, spmv_dot, m->b.p, m->x.p, m->tmp.p, m->cg_p.p, m->cg_ap.p, n, epsilon, max_iteration, m->cg_state.p,
                         m->partial.p, s, m->ev0, m->ev1, report));
    }
    if (n) CCP_HIP(hipMemcpyAsync(x_out, m->x.p, sizeof(double) * n, hipMemcpyDeviceToHost, s));
    CCP_HIP(hipStreamSynchronize(s));
    return CCP_OK;
} CCP_ABI_CATCH

int ccp_csr_conjugate_gradient_jacobi(ccp_csr *m, const double *b, double *x_out, double epsilon, int32_t max_iteration,
                                      ccp_gs_report *report)
try {
    CCP_TRY(bind(m));
    if (!m->uploaded) return CCP_ERR_STATE;
    CCP_TRY(flush_edits(m));
    const bool rowblock = m->rb.on;        // b, x_out: the block's own rows (see ccp_csr_conjugate_gradient)
    if ((!b || !x_out) && !(rowblock && m->rb.n_local == 0)) return CCP_ERR_BAD_ARG;
    if (m->n_rows != m->n_cols) return CCP_ERR_UNSUPPORTED;
    CCP_TRY(ensure_natural(m));
    const long n = m->n_rows;
    hipStream_t s = m->stream;
    if (!m->cg_p.p || m->cg_p.n < (size_t)std::max<long>(n, 2)) {
        CCP_TRY(m->cg_p.alloc((size_t)std::max<long>(n, 2)));
        CCP_TRY(m->cg_ap.alloc((size_t)std::max<long>(n, 2)));
        CCP_TRY(m->cg_state.alloc(1));
    }
    if (!m->cg_state.p) CCP_TRY(m->cg_state.alloc(1));
    const unsigned spmv_blocks = (unsigned)std::min<long>(2048, (m->natural.n_slices + kBlock / kWave - 1) / (kBlock / kWave));   // (k_sell_apply strides)
    CCP_TRY(ensure_partial(m, std::max<long>(2048, spmv_blocks)));
    if (m->cg_partial2.n != 4096) CCP_TRY(m->cg_partial2.alloc(4096));
    // extractDiagnolColInv (sparse-matrix.h:472-491): 1/a_ii of the first stored diagonal entry, 1 if absent or 0
    CCP_TRY(materialise(m));
    std::vector<double> inv((size_t)std::max<long>(n, 1), 1.0);
    parallel_ranges(n, 1 << 16, [&](long lo, long hi) {
        for (long i = lo; i < hi; ++i)
            for (long k = m->row_ptr[i]; k < m->row_ptr[i + 1]; ++k)
                if (m->col[k] == (int)i) {
                    if (m->val[k] != 0.0) inv[(size_t)i] = 1.0 / m->val[k];
                    break;
                }
    });
    if (m->cg_inv.n != inv.size()) CCP_TRY(m->cg_inv.alloc(inv.size()));
    if (n) {
        CCP_HIP(hipMemcpyAsync(m->cg_inv.p, inv.data(), sizeof(double) * n, hipMemcpyHostToDevice, s));
        if (rowblock) CCP_TRY(rb_stage(m, m->tmp.p, b));                                         // (a ghost: empty row, inverse diagonal 1, r = 0)
        else CCP_HIP(hipMemcpyAsync(m->tmp.p, b, sizeof(double) * n, hipMemcpyHostToDevice, s));     // r = b - A 0 = b (:500-501)
        CCP_HIP(hipMemsetAsync(m->x.p, 0, sizeof(double) * n, s));                               // :495
    }
    const SellView view = m->natural.view();
    const int n_slices = m->natural.n_slices;
    auto spmv_dot = [&](const double *in, double *out, int *n_partials) -> int {
        *n_partials = 0;
        if (rowblock) CCP_TRY(rb_exchange_natural(m, const_cast<double *>(in)));
        if (n_slices == 0) return CCP_OK;
        hipLaunchKernelGGL((k_sell_apply<2>), dim3(spmv_blocks), dim3(kBlock), 0, s, view, n_slices, in, out, in, m->partial.p);
        *n_partials = (int)spmv_blocks;
        return hipGetLastError() == hipSuccess ? CCP_OK : CCP_ERR_HIP;
    };
    if (rowblock) {
        const RcclApi *api = rccl_api();
        if (!api) return CCP_ERR_RCCL;
        double *total = reinterpret_cast<double *>(m->rb.comm->scratch.p);
        auto sums = [&](double *partial, int *count, const double **sum_at, int slot) -> int {
            hipLaunchKernelGGL(k_reduce_to, dim3(1), dim3(kBlock), 0, s, partial, (long)*count, 1L, total + slot, 0);
            CCP_HIP(hipGetLastError());
            CCP_RCCL(api->AllReduce(total + slot, total + slot, 1, ncclDouble, ncclSum, m->rb.comm->comm, s));
            *count = 1;
            *sum_at = total + slot;
            return CCP_OK;
        };
        CCP_TRY(pcg_solve(spmv_dot, m->x.p, m->tmp.p, m->cg_p.p, m->cg_ap.p, m->cg_inv.p, n, epsilon, max_iteration, m->cg_state.p,
                          m->partial.p, m->cg_partial2.p, s, m->ev0, m->ev1, report, sums, true));
        if (m->rb.n_local) CCP_HIP(hipMemcpyAsync(x_out, m->x.p + m->rb.n_lo, sizeof(double) * (size_t)m->rb.n_local, hipMemcpyDeviceToHost, s));
        CCP_HIP(hipStreamSynchronize(s));    // `inv` lives on this stack frame
        return CCP_OK;
    }
    CCP_TRY(pcg_solve(spmv_dot, m->x.p, m->tmp.p, m->cg_p.p, m->cg_ap.p, m->cg_inv.p, n, epsilon, max_iteration, m->cg_state.p,
                      m->partial.p, m->cg_partial2.p, s, m->ev0, m->ev1, report));
    if (n) CCP_HIP(hipMemcpyAsync(x_out, m->x.p, sizeof(double) * n, hipMemcpyDeviceToHost, s));
    CCP_HIP(hipStreamSynchronize(s));        // `inv` lives on this stack frame
    return CCP_OK;
} CCP_ABI_CATCH

int ccp_csr_apply_to_vector(ccp_csr *m, const double *in, double *out)
try {
    CCP_TRY(bind(m));
    if (!m->uploaded) return CCP_ERR_STATE;
    CCP_TRY(flush_edits(m));
    if (m->rb.on ? (m->rb.n_local && (!in || !out)) : ((m->n_cols && !in) || (m->n_rows && !out))) return CCP_ERR_BAD_ARG;
    if (!m->rb.on && m->allow_structured && m->n_rows == m->n_cols) {
        // SolveChannel's matrix: b := A x matrix-free on the grid twin (the row's products in the stored order: same bits)
        // instead of a sliced-ELL image of the whole matrix (16 GB at 16384^2) that only this call would need
        detect_poisson(m);
        if (m->poisson_w > 1 && m->poisson_h > 1) {
            if (!m->grid) {
                ccp_grid_desc d{m->poisson_w, m->poisson_h, 1, 0, m->poisson_h, 0, m->device, 0};
                CCP_TRY(ccp_grid_create(&d, &m->grid));
                (void)grid_set_allow_swap(m->grid, true);
            }
            CCP_TRY(ccp_grid_set_stream(m->grid, m->stream));
            CCP_TRY(ccp_grid_set_x_host(m->grid, 0, in, 0, m->poisson_h));
            CCP_TRY(ccp_grid_b_from_x(m->grid));
            return ccp_grid_get_b_host(m->grid, 0, out, 0, m->poisson_h);
        }
        if (m->allow_region && m->region_state > 0 && m->region_grid && !m->edited) {
            // a raster-region Laplacian a solve has already recognised: b := A x on its canvas (Dirichlet-mask grid)
            const long n = m->n_rows;
            hipStream_t s = m->stream;
            ccp_grid *g = m->region_grid;
            CCP_TRY(ccp_grid_set_stream(g, s));
            ccp_grid_layout lay{};
            CCP_TRY(ccp_grid_get_layout(g, &lay));
            CCP_HIP(hipMemcpyAsync(m->tmp.p, in, sizeof(double) * n, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL((k_canvas_move<0>), dim3(blocks_for(n)), dim3(kBlock), 0, s, static_cast<double *>(lay.x_dev), m->tmp.p, m->region_where.p, n, 0.0);
            CCP_HIP(hipGetLastError());
            CCP_TRY(ccp_grid_b_from_x(g));
            hipLaunchKernelGGL((k_canvas_move<1>), dim3(blocks_for(n)), dim3(kBlock), 0, s, static_cast<double *>(lay.b_dev), m->tmp.p, m->region_where.p, n, 0.0);
            CCP_HIP(hipGetLastError());
            CCP_HIP(hipMemcpyAsync(out, m->tmp.p, sizeof(double) * n, hipMemcpyDeviceToHost, s));
            CCP_HIP(hipStreamSynchronize(s));
            return CCP_OK;
        }
    }
    CCP_TRY(ensure_natural(m));
    hipStream_t s = m->stream;
    const bool rowblock = m->rb.on;        // in / out: the block's own rows; the ghosts of `in` come from their owners
    if (rowblock) {
        CCP_TRY(rb_stage(m, m->x.p, in));
        CCP_TRY(rb_exchange_natural(m, m->x.p));
    } else if (m->n_cols) {
        CCP_HIP(hipMemcpyAsync(m->x.p, in, sizeof(double) * m->n_cols, hipMemcpyHostToDevice, s));
    }
    if (m->n_rows) {
        const unsigned blocks = (unsigned)((m->natural.n_slices + kBlock / kWave - 1) / (kBlock / kWave));
        CCP_TRY(ensure_partial(m, blocks));
        hipLaunchKernelGGL((k_sell_apply<0>), dim3(blocks), dim3(kBlock), 0, s, m->natural.view(), m->natural.n_slices,
                           m->x.p, m->tmp.p, m->b.p, m->partial.p);
        CCP_HIP(hipGetLastError());
        if (!rowblock) CCP_HIP(hipMemcpyAsync(out, m->tmp.p, sizeof(double) * m->n_rows, hipMemcpyDeviceToHost, s));
        else if (m->rb.n_local) CCP_HIP(hipMemcpyAsync(out, m->tmp.p + m->rb.n_lo, sizeof(double) * (size_t)m->rb.n_local, hipMemcpyDeviceToHost, s));
    }
    CCP_HIP(hipStreamSynchronize(s));
    return CCP_OK;
} CCP_ABI_CATCH

int ccp_csr_residual_norm2(ccp_csr *m, const double *b, const double *x, double *rr, double *bb)
try {
    CCP_TRY(bind(m));
    if (!m->uploaded) return CCP_ERR_STATE;
    CCP_TRY(flush_edits(m));
    if (!rr || !bb || ((!b || !x) && !(m->rb.on && m->rb.n_local == 0))) return CCP_ERR_BAD_ARG;
    if (!m->rb.on && m->allow_structured && m->n_rows == m->n_cols && m->n_rows > 0) {
        detect_poisson(m);                   // (as ccp_csr_apply_to_vector: matrix-free on the grid twin)
        if (m->poisson_w > 1 && m->poisson_h > 1) {
            if (!m->grid) {
                ccp_grid_desc d{m->poisson_w, m->poisson_h, 1, 0, m->poisson_h, 0, m->device, 0};
                CCP_TRY(ccp_grid_create(&d, &m->grid));
                (void)grid_set_allow_swap(m->grid, true);
            }
            CCP_TRY(ccp_grid_set_stream(m->grid, m->stream));
            CCP_TRY(ccp_grid_set_b_host(m->grid, 0, b, 0, m->poisson_h));
            CCP_TRY(ccp_grid_set_x_host(m->grid, 0, x, 0, m->poisson_h));
            double both[2] = {0.0, 0.0};
            CCP_TRY(ccp_grid_residual_norm2(m->grid, both));
            *rr = both[0];
            *bb = both[1];
            return CCP_OK;
        }
        if (m->allow_region && m->region_state > 0 && m->region_grid && !m->edited) {
            const long n = m->n_rows;
            hipStream_t s = m->stream;
            ccp_grid *g = m->region_grid;
            CCP_TRY(ccp_grid_set_stream(g, s));
            ccp_grid_layout lay{};
            CCP_TRY(ccp_grid_get_layout(g, &lay));
            CCP_HIP(hipMemcpyAsync(m->tmp.p, b, sizeof(double) * n, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL((k_canvas_move<0>), dim3(blocks_for(n)), dim3(kBlock), 0, s, static_cast<double *>(lay.b_dev), m->tmp.p, m->region_where.p, n, 0.0);
            CCP_HIP(hipGetLastError());
            CCP_HIP(hipStreamSynchronize(s));            // (tmp is staged twice)
            CCP_HIP(hipMemcpyAsync(m->tmp.p, x, sizeof(double) * n, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL((k_canvas_move<0>), dim3(blocks_for(n)), dim3(kBlock), 0, s, static_cast<double *>(lay.x_dev), m->tmp.p, m->region_where.p, n, 0.0);
            CCP_HIP(hipGetLastError());
            double both[2] = {0.0, 0.0};
            CCP_TRY(ccp_grid_residual_norm2(g, both));
            *rr = both[0];
            *bb = both[1];
            return CCP_OK;
        }
    }
    CCP_TRY(ensure_natural(m));
    hipStream_t s = m->stream;
    *rr = 0.0;
    *bb = 0.0;
    const bool rowblock = m->rb.on;        // b, x: the block's own rows; rr, bb: sums over the whole matrix, on every rank
    if (!m->n_rows && !rowblock) return CCP_OK;
    if (rowblock) {
        CCP_TRY(rb_stage(m, m->x.p, x));
        CCP_TRY(rb_exchange_natural(m, m->x.p));
        CCP_TRY(rb_stage(m, m->b.p, b));
    } else {
        CCP_HIP(hipMemcpyAsync(m->x.p, x, sizeof(double) * m->n_cols, hipMemcpyHostToDevice, s));
        CCP_HIP(hipMemcpyAsync(m->b.p, b, sizeof(double) * m->n_rows, hipMemcpyHostToDevice, s));
    }
    const unsigned blocks = (unsigned)((m->natural.n_slices + kBlock / kWave - 1) / (kBlock / kWave));
    CCP_TRY(ensure_partial(m, std::max(blocks, 1u)));
    if (blocks)
        hipLaunchKernelGGL((k_sell_apply<1>), dim3(blocks), dim3(kBlock), 0, s, m->natural.view(), m->natural.n_slices, m->x.p,
                           m->tmp.p, m->b.p, m->partial.p);
    double *res = m->tmp.p;   // two doubles of scratch for the result
    hipLaunchKernelGGL(k_reduce_to, dim3(1), dim3(kBlock), 0, s, m->partial.p, (long)blocks, 2L, res, 0);
    hipLaunchKernelGGL(k_reduce_to, dim3(1), dim3(kBlock), 0, s, m->partial.p + 1, (long)blocks, 2L, res + 1, 0);
    CCP_HIP(hipGetLastError());
    if (rowblock) {
        const RcclApi *api = rccl_api();
        if (!api) return CCP_ERR_RCCL;
        CCP_RCCL(api->AllReduce(res, res, 2, ncclDouble, ncclSum, m->rb.comm->comm, s));
    }
    double host[2];
    CCP_HIP(hipMemcpyAsync(host, res, sizeof(host), hipMemcpyDeviceToHost, s));
    CCP_HIP(hipStreamSynchronize(s));
    *rr = host[0];
    *bb = host[1];
    return CCP_OK;
} CCP_ABI_CATCH

}  // extern "C"
