// ccp_csr_kernels.hpp — general sparse Gauss-Seidel / SpMV kernels for gfx950.
//
// The reference stores slack-CSR (project/src/PhotoMontage/sparse-matrix.h:670-676) and walks a
// row's live entries in storage order (:364-372, :384-390).  One thread per row over raw CSR
// would read `values_`/`col_offset_` with a stride of one row between lanes; instead the upload
// re-tiles the live entries into sliced ELL with 64-row slices (one slice = one wavefront):
//     entry k of slice-row r lives at  slice_off + k*64 + r
// so the k-th entries of 64 consecutive rows are one contiguous 512-byte (values) / 256-byte
// (columns) lane-coalesced access.  Rows are listed schedule-major (colour by colour, or level
// by level for the lexicographic order) and x/b live on the device in that permuted order, so
// b reads and x writes are coalesced too; only the x[col] gathers are indexed.
#pragma once

#include "ccp_common.hpp"
#include "ccp_cg.hpp"          // CgState (k_sell_cg_apply)

namespace ccp {

struct CsrSolveState {
    int active;
    int converged;
    int iterations;
    int pad;
    double eps_accum;     // running sum of the current checked sweep
    double last_eps;
};

struct SellView {
    const long *__restrict__ slice_off;    // [n_slices] first entry of the slice
    const int *__restrict__ slice_width;   // [n_slices] entries per row in the slice
    const int *__restrict__ slice_row0;    // [n_slices] first (permuted) row of the slice
    const int *__restrict__ slice_rows;    // [n_slices] rows in the slice (<= 64)
    const int *__restrict__ cols;          // permuted column index, -1 = padding
    const double *__restrict__ vals;
};

// The entries of a row in storage order, kSellBatch at a time: the column indices and values of a batch are all loaded
// before the first is looked at, then the vector elements they point to, then `each(col, value, vec[col])` runs over the
// batch in order (padding: col = -1, vec[0] fetched and ignored).  A rolled "load the column, wait, load value and
// vec[col], wait, accumulate" loop has one entry in flight per wave — two memory latencies per entry (ISA, round 4).
// The order of the calls — and so every sum built from them — is the rolled loop's.
constexpr int kSellBatch = 8;
template <typename Each>
__device__ __forceinline__ void sell_row_entries(const SellView &m, long off, int width, const double *__restrict__ vec, Each &&each)
{
    for (int k0 = 0; k0 < width; k0 += kSellBatch) {
        int c[kSellBatch];
        double v[kSellBatch], xv[kSellBatch];
#pragma unroll
        for (int j = 0; j < kSellBatch; ++j) {
            const bool in = k0 + j < width;                                  // (uniform: the slice's width)
            const long at = off + (long)(in ? k0 + j : k0) * kWave;
            const int cj = m.cols[at];
            c[j] = in ? cj : -1;
            v[j] = m.vals[at];
        }
#pragma unroll
        for (int j = 0; j < kSellBatch; ++j) xv[j] = vec[c[j] >= 0 ? c[j] : 0];
#pragma unroll
        for (int j = 0; j < kSellBatch; ++j)
            if (c[j] >= 0) each(c[j], v[j], xv[j]);
    }
}

// One Gauss-Seidel pass over the slices [s_first, s_last) — all rows of one colour / level —
// wave per slice, lane per row.  Row update as sparse-matrix.h:360-373: a_ii = at(i,i); skip
// when 0; sigma accumulates values*x over col != i in storage order; x = (b - sigma) / a_ii.
// L1: block partial of sum|x_new - x_old| to partial[blockIdx.x].  `active` (nullable) points at
// the device-side solve state: a converged solve turns the remaining queued passes into no-ops.
// The rows of slice s: lane per row.  Returns |x_new - x_old| of the lane's row (L1) or 0.
template <bool L1>
__device__ __forceinline__ double sell_gs_slice(const SellView &m, int s, int lane, double *__restrict__ x, const double *__restrict__ b)
{
    double acc = 0.0;
    if (lane < m.slice_rows[s]) {
        const int row = m.slice_row0[s] + lane;
        const long off = m.slice_off[s] + lane;
        const int width = m.slice_width[s];
        double a_ii = 0.0;
        double sigma = 0.0;
        sell_row_entries(m, off, width, x, [&](int c, double v, double xc) {
            if (c == row) a_ii = v;
            else sigma += v * xc;
        });
        if (a_ii != 0.0) {
            const double nv = (b[row] - sigma) / a_ii;
            if (L1) acc = fabs(nv - x[row]);
            x[row] = nv;
        }
    }
    return acc;
}

template <bool L1>
__global__ void __launch_bounds__(kBlock)
k_sell_gs(SellView m, int s_first, int s_last, double *__restrict__ x, const double *__restrict__ b,
          double *__restrict__ partial, const int *__restrict__ active)
{
    __shared__ double scratch[kBlock / kWave];
    const int lane = threadIdx.x & (kWave - 1);
    const int s = s_first + blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
    double acc = 0.0;
    const bool run = (active == nullptr) || (*active != 0);
    if (run && s < s_last) acc = sell_gs_slice<L1>(m, s, lane, x, b);
    if (L1) {
        const double t = block_sum(acc, scratch);
        if (threadIdx.x == 0) partial[blockIdx.x] = t;
    }
}

// The same pass over the slices list[0 .. count): a row block sweeps the slices whose rows other blocks reference
// first, so that their values travel while the rest of the colour is swept (ccp_csr.hip: RowBlock).
template <bool L1>
__global__ void __launch_bounds__(kBlock)
k_sell_gs_list(SellView m, const int *__restrict__ list, int count, double *__restrict__ x, const double *__restrict__ b,
               double *__restrict__ partial, const int *__restrict__ active)
{
    __shared__ double scratch[kBlock / kWave];
    const int lane = threadIdx.x & (kWave - 1);
    const int at = blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
    double acc = 0.0;
    const bool run = (active == nullptr) || (*active != 0);
    if (run && at < count) acc = sell_gs_slice<L1>(m, list[at], lane, x, b);
    if (L1) {
        const double t = block_sum(acc, scratch);
        if (threadIdx.x == 0) partial[blockIdx.x] = t;
    }
}

// The level schedule, pipelined over sweeps.  Row i of level l in sweep k reads the NEW values of
// its coupled rows j < i (lower levels, same sweep) and the OLD values of its coupled rows j > i
// (higher levels, previous sweep).  With span = max over coupled pairs of the level difference and
// stride = span + 1, every value row (l, k) needs was produced at a time tau' < tau = l + stride*k:
// lower level same sweep trivially; level l' <= l + span of sweep k-1 at l' + stride*(k-1) < tau.
// One launch per tau updates level tau - stride*k of every sweep k in flight: n_levels + stride*K
// launches instead of n_levels*K, same iterates.  Levels in flight in one launch differ by multiples
// of stride > span, so none of them is coupled to another: no hazard inside a launch.
// grid = (blocks of the widest level, sweeps in flight); blockIdx.y -> k = k_lo + blockIdx.y.
// L1: partial[k*partial_per_sweep + group_block_off[level] + blockIdx.x] (every slot written once per batch).
template <bool L1>
__global__ void __launch_bounds__(kBlock)
k_sell_gs_pipe(SellView m, const int *__restrict__ group_slice_ptr, const long *__restrict__ group_block_off, int tau,
               int stride, int k_lo, double *__restrict__ x, const double *__restrict__ b,
               double *__restrict__ partial, long partial_per_sweep)
{
    __shared__ double scratch[kBlock / kWave];
    const int lane = threadIdx.x & (kWave - 1);
    const int k = k_lo + blockIdx.y;
    const int level = tau - stride * k;
    const int s_first = group_slice_ptr[level], s_last = group_slice_ptr[level + 1];
    const int s = s_first + blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
    if (s_first + (int)blockIdx.x * (kBlock / kWave) >= s_last) return;        // block beyond this level (uniform)
    double acc = 0.0;
    if (s < s_last && lane < m.slice_rows[s]) {
        const int row = m.slice_row0[s] + lane;
        const long off = m.slice_off[s] + lane;
        const int width = m.slice_width[s];
        double a_ii = 0.0;
        double sigma = 0.0;
        sell_row_entries(m, off, width, x, [&](int c, double v, double xc) {
            if (c == row) a_ii = v;
            else sigma += v * xc;
        });
        if (a_ii != 0.0) {
            const double nv = (b[row] - sigma) / a_ii;
            if (L1) acc = fabs(nv - x[row]);
            x[row] = nv;
        }
    }
    if (L1) {
        const double t = block_sum(acc, scratch);
        if (threadIdx.x == 0) partial[(long)k * partial_per_sweep + group_block_off[level] + blockIdx.x] = t;
    }
}

// eps[k] = sum of the partials of sweep k in a fixed order.  grid = sweeps
__global__ void __launch_bounds__(kBlock)
k_reduce_sweeps(const double *__restrict__ partial, long per_sweep, double *__restrict__ eps)
{
    __shared__ double scratch[kBlock / kWave];
    const double *__restrict__ p = partial + (long)blockIdx.x * per_sweep;
    double acc = 0.0;
    for (long i = threadIdx.x; i < per_sweep; i += kBlock) acc += p[i];
    const double t = block_sum(acc, scratch);
    if (threadIdx.x == 0) eps[blockIdx.x] = t;
}

// Whole solve in ONE workgroup: all groups (levels / colours) of all sweeps, __syncthreads()
// between groups, stop rule evaluated in the kernel.  For schedules whose groups are narrow —
// the level schedule of the lexicographic order has W+H-1 levels of <= min(W,H) rows on a grid —
// one launch per group costs ~5 us each (4.8 ms per 512x512 sweep); here a group costs a barrier.
// Visibility: all waves of a workgroup share the CU's L1 and __syncthreads() is a workgroup-scope
// release/acquire, so rows written in one group are seen by the next.
constexpr int kSerialBlock = 1024;
__global__ void __launch_bounds__(kSerialBlock)
k_sell_gs_one_block(SellView m, const int *__restrict__ group_slice_ptr, int n_groups, double *__restrict__ x,
                    const double *__restrict__ b, CsrSolveState *__restrict__ st, double epsilon, int max_iteration,
                    int check_every);

// out[row] = sum values*in[col] in storage order (applyToVector, sparse-matrix.h:382-393).
// MODE 0: store; MODE 1: partial sums of (b - Ax)^2 and b^2; MODE 2: store and one partial sum of
// in'(A in) per block (p'Ap of conjugateGradient fused into the SpMV).
template <int MODE>
__global__ void __launch_bounds__(kBlock)
k_sell_apply(SellView m, int n_slices, const double *__restrict__ in, double *__restrict__ out,
             const double *__restrict__ b, double *__restrict__ partial)
{
    __shared__ double scratch[kBlock / kWave];
    const int lane = threadIdx.x & (kWave - 1);
    double rr = 0.0, bb = 0.0;
    // a wave takes slices blockIdx.x * 4 + wave, + gridDim.x * 4, ...: a launch with fewer blocks than slices / 4 leaves
    // fewer partial sums to add up (the conjugate-gradient loops: 2,048 instead of one per 256 rows — the one-block sum
    // of 40,000 partials was 54 us of a 390 us iteration at 10 M rows); one with a block per 4 slices is the plain form
    for (int s = blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave; s < n_slices; s += gridDim.x * (kBlock / kWave)) {
        if (lane >= m.slice_rows[s]) continue;
        const int row = m.slice_row0[s] + lane;
        const long off = m.slice_off[s] + lane;
        const int width = m.slice_width[s];
        double sum = 0.0;
        sell_row_entries(m, off, width, in, [&](int, double v, double xc) { sum += v * xc; });
        if (MODE == 0) {
            out[row] = sum;
        } else if (MODE == 2) {
            out[row] = sum;
            rr += in[row] * sum;
        } else {
            const double bv = b[row];
            const double r = bv - sum;
            rr += r * r;
            bb += bv * bv;
        }
    }
    if (MODE == 1) {
        const double t0 = block_sum(rr, scratch);
        const double t1 = block_sum(bb, scratch);
        if (threadIdx.x == 0) {
            partial[2 * (long)blockIdx.x] = t0;
            partial[2 * (long)blockIdx.x + 1] = t1;
        }
    }
    if (MODE == 2) {
        const double t0 = block_sum(rr, scratch);
        if (threadIdx.x == 0) partial[blockIdx.x] = t0;
    }
}

// Pass A of the fused conjugate-gradient iteration (ccp_cg.hpp, cg_solve_fused) on the stored matrix: for row i
//   x_i += alpha p_old_i;   p_new_j = r_j + beta p_old_j for j = i and for every column j of the row (recomputed where it
//   is used: a direction is one multiply-add of two values the gather brings anyway);   Ap_i = sum a_ij p_new_j in
//   storage order (applyToVector, sparse-matrix.h:382-393);   one partial sum of p_new'Ap per block
// — the operations of k_cg_direction, k_sell_apply<2> and the x half of k_cg_update on the same operands in the same
// order, block for block: the iterates are the three-pass loop's bit for bit.  Per row and iteration 12 nnz + 72 B
// instead of 12 nnz + 88 (p is no longer written and read back between the passes).  p is double-buffered: a row's
// neighbours still read p_old while it stores its p_new.
__global__ void __launch_bounds__(kBlock)
k_sell_cg_apply(SellView m, int n_slices, double *__restrict__ x, const double *__restrict__ r, const double *__restrict__ p_old,
                double *__restrict__ p_new, double *__restrict__ ap, double *__restrict__ partial, const CgState *__restrict__ st)
{
    __shared__ double scratch[kBlock / kWave];
    const int lane = threadIdx.x & (kWave - 1);
    double dot = 0.0;
    const bool active = st->active != 0;
    const double alpha = st->alpha, beta = st->beta;
    for (int s = blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave; active && s < n_slices; s += gridDim.x * (kBlock / kWave)) {   // (as k_sell_apply)
        if (lane >= m.slice_rows[s]) continue;
        const int row = m.slice_row0[s] + lane;
        const long off = m.slice_off[s] + lane;
        const int width = m.slice_width[s];
        double sum = 0.0;
        // (rolled: batching the entries as sell_row_entries does — sixteen gathers in flight — made this pass slower,
        // 58.2 -> 66.0 ms per 200 iterations at 10 M rows)
        for (int k = 0; k < width; ++k) {
            const int c = m.cols[off + (long)k * kWave];
            const double v = m.vals[off + (long)k * kWave];
            if (c >= 0) sum += v * (r[c] + beta * p_old[c]);
        }
        const double po = p_old[row];
        const double pn = r[row] + beta * po;
        x[row] = x[row] + alpha * po;
        p_new[row] = pn;
        ap[row] = sum;
        dot += pn * sum;
    }
    const double t0 = block_sum(dot, scratch);
    if (threadIdx.x == 0) partial[blockIdx.x] = t0;
}

// dst[i] = src[perm[i]] (gather into schedule order) / dst[perm[i]] = src[i] (scatter back).
template <bool GATHER>
__global__ void __launch_bounds__(kBlock)
k_permute(double *__restrict__ dst, const double *__restrict__ src, const int *__restrict__ perm, long n)
{
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock) {
        if (GATHER) dst[i] = src[perm[i]];
        else dst[perm[i]] = src[i];
    }
}

__global__ void __launch_bounds__(kBlock) k_fill_n(double *__restrict__ p, long n, double v)
{
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock) p[i] = v;
}

// Sum `count` doubles (stride 1) in a fixed order: out[0] (+= when accumulate).  One block.
__global__ void __launch_bounds__(kBlock)
k_reduce_to(const double *__restrict__ partial, long count, long stride, double *__restrict__ out, int accumulate)
{
    __shared__ double scratch[kBlock / kWave];
    double acc = 0.0;
    for (long i = threadIdx.x; i < count; i += kBlock) acc += partial[i * stride];
    const double t = block_sum(acc, scratch);
    if (threadIdx.x == 0) out[0] = accumulate ? out[0] + t : t;
}

// Raster-region dispatch: unknown i of the uploaded matrix lives at element where[i] of the canvas planes.
// canvas[where[i]] = v[i] (scatter) / v[i] = canvas[where[i]] (gather); fill: canvas[where[i]] = value.
template <int MODE>
__global__ void __launch_bounds__(kBlock)
k_canvas_move(double *__restrict__ canvas, double *__restrict__ v, const long *__restrict__ where, long n, double value)
{
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock) {
        if (MODE == 0) canvas[where[i]] = v[i];
        else if (MODE == 1) v[i] = canvas[where[i]];
        else canvas[where[i]] = value;
    }
}

// Incremental edits (ccp_csr_insert).  One block per patched row: write the row's `cap` entry slots
// (entry k at base + k*64; beyond the row's live entries the patch carries padding: column -1, value 0).
__global__ void __launch_bounds__(kWave)
k_patch_rows(int *__restrict__ cols, double *__restrict__ vals, const long *__restrict__ base, const int *__restrict__ cap,
             const long *__restrict__ off, const int *__restrict__ pcols, const double *__restrict__ pvals)
{
    const long b = base[blockIdx.x], o = off[blockIdx.x];
    for (int k = threadIdx.x; k < cap[blockIdx.x]; k += kWave) {
        cols[b + (long)k * kWave] = pcols[o + k];
        vals[b + (long)k * kWave] = pvals[o + k];
    }
}

// Move one slice (64 lanes x `width` live entry columns) to a block of `new_cap` columns at `dst`, padding
// the new spare columns.  One block of 64 threads.
__global__ void __launch_bounds__(kWave)
k_move_slice(int *__restrict__ cols, double *__restrict__ vals, long src, long dst, int width, int new_cap)
{
    const int lane = threadIdx.x;
    for (int k = 0; k < new_cap; ++k) {
        const bool live = k < width;
        cols[dst + (long)k * kWave + lane] = live ? cols[src + (long)k * kWave + lane] : -1;
        vals[dst + (long)k * kWave + lane] = live ? vals[src + (long)k * kWave + lane] : 0.0;
    }
}

__global__ void k_csr_check(CsrSolveState *__restrict__ st, double epsilon, int sweep_index)
{
    if (threadIdx.x == 0 && blockIdx.x == 0 && st->active) {
        const double eps = st->eps_accum;
        st->last_eps = eps;
        if (!(eps > epsilon)) {
            st->active = 0;
            st->converged = 1;
            st->iterations = sweep_index;
        }
    }
}

// What a lane needs for one row that does NOT depend on x: fetched for the NEXT group while the
// current one is still being swept, so after the barrier only the x gathers are on the
// critical path (one dependent memory latency per group instead of three).
constexpr int kPrefetchWidth = 8;
struct RowAhead {
    int row;            // -1: lane idle in this slice
    int width;          // entries per row in the slice; > kPrefetchWidth: cols/vals not prefetched
    long off;
    double bval;
    int c[kPrefetchWidth];
    double v[kPrefetchWidth];
};

__device__ __forceinline__ void fetch_row_ahead(const SellView &m, const double *__restrict__ b, int s, int lane, RowAhead &ra)
{
    ra.row = -1;
    ra.width = 0;
    if (lane < m.slice_rows[s]) {
        ra.row = m.slice_row0[s] + lane;
        ra.off = m.slice_off[s] + lane;
        ra.width = m.slice_width[s];
        ra.bval = b[ra.row];
        if (ra.width <= kPrefetchWidth) {
#pragma unroll
            for (int e = 0; e < kPrefetchWidth; ++e) {
                const bool on = e < ra.width;
                ra.c[e] = on ? m.cols[ra.off + (long)e * kWave] : -1;
                ra.v[e] = on ? m.vals[ra.off + (long)e * kWave] : 0.0;
            }
        }
    }
}

__global__ void __launch_bounds__(kSerialBlock)
k_sell_gs_one_block(SellView m, const int *__restrict__ group_slice_ptr, int n_groups, double *__restrict__ x,
                    const double *__restrict__ b, CsrSolveState *__restrict__ st, double epsilon, int max_iteration,
                    int check_every)
{
    __shared__ double scratch[kSerialBlock / kWave];
    __shared__ int s_go;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    constexpr int kWaves = kSerialBlock / kWave;
    int done = 0;
    // one row update; the x gathers are the only loads left when `ra` was prefetched
    auto update = [&](const RowAhead &ra, bool check, double &acc) {
        if (ra.row < 0) return;
        double a_ii = 0.0, sigma = 0.0;
        if (ra.width <= kPrefetchWidth) {
#pragma unroll
            for (int e = 0; e < kPrefetchWidth; ++e) {
                if (ra.c[e] == ra.row) a_ii = ra.v[e];
                else if (ra.c[e] >= 0) sigma += ra.v[e] * x[ra.c[e]];
            }
        } else {
            for (int e = 0; e < ra.width; ++e) {
                const int c = m.cols[ra.off + (long)e * kWave];
                const double v = m.vals[ra.off + (long)e * kWave];
                if (c == ra.row) a_ii = v;
                else if (c >= 0) sigma += v * x[c];
            }
        }
        if (a_ii != 0.0) {
            const double nv = (ra.bval - sigma) / a_ii;
            if (check) acc += fabs(nv - x[ra.row]);
            x[ra.row] = nv;
        }
    };
    RowAhead ahead;
    ahead.row = -1;
    ahead.width = 0;
    int ahead_slice = -1;
    {
        const int s = group_slice_ptr[0] + wave;
        if (n_groups > 0 && s < group_slice_ptr[1]) {
            fetch_row_ahead(m, b, s, lane, ahead);
            ahead_slice = s;
        }
    }
    for (int k = 1; k <= max_iteration; ++k) {
        const bool check = check_every > 0 && (k % check_every) == 0;
        double acc = 0.0;
        for (int g = 0; g < n_groups; ++g) {
            const int s0 = group_slice_ptr[g], s1 = group_slice_ptr[g + 1];
            RowAhead cur = ahead;
            const int cur_slice = ahead_slice;
            // prefetch this wave's first slice of the next group (next sweep's first group at the end)
            {
                const int gn = (g + 1 < n_groups) ? g + 1 : 0;
                const int s = group_slice_ptr[gn] + wave;
                ahead.row = -1;
                ahead_slice = -1;
                if (s < group_slice_ptr[gn + 1]) {
                    fetch_row_ahead(m, b, s, lane, ahead);
                    ahead_slice = s;
                }
            }
            for (int s = s0 + wave; s < s1; s += kWaves) {
                if (s == cur_slice) {
                    update(cur, check, acc);
                } else {
                    RowAhead now;
                    fetch_row_ahead(m, b, s, lane, now);
                    update(now, check, acc);
                }
            }
            __syncthreads();
        }
        done = k;
        if (check) {
            double v = wave_sum(acc);
            if (lane == 0) scratch[wave] = v;
            __syncthreads();
            if (threadIdx.x == 0) {
                double eps = 0.0;
                for (int w = 0; w < kWaves; ++w) eps += scratch[w];
                st->last_eps = eps;
                int go = 1;
                if (!(eps > epsilon)) {            // `while (eps > epsilon && ...)`, sparse-matrix.h:356
                    st->active = 0;
                    st->converged = 1;
                    st->iterations = k;
                    go = 0;
                }
                s_go = go;
            }
            __syncthreads();
            if (!s_go) break;
        }
    }
    if (threadIdx.x == 0 && st->active) st->iterations = done;
}

}  // namespace ccp
