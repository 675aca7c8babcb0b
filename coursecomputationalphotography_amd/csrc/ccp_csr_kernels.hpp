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

namespace ccp {

struct SellView {
    const long *__restrict__ slice_off;    // [n_slices] first entry of the slice
    const int *__restrict__ slice_width;   // [n_slices] entries per row in the slice
    const int *__restrict__ slice_row0;    // [n_slices] first (permuted) row of the slice
    const int *__restrict__ slice_rows;    // [n_slices] rows in the slice (<= 64)
    const int *__restrict__ cols;          // permuted column index, -1 = padding
    const double *__restrict__ vals;
};

// One Gauss-Seidel pass over the slices [s_first, s_last) — all rows of one colour / level —
// wave per slice, lane per row.  Row update as sparse-matrix.h:360-373: a_ii = at(i,i); skip
// when 0; sigma accumulates values*x over col != i in storage order; x = (b - sigma) / a_ii.
// L1: block partial of sum|x_new - x_old| to partial[blockIdx.x].  `active` (nullable) points at
// the device-side solve state: a converged solve turns the remaining queued passes into no-ops.
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
    if (run && s < s_last && lane < m.slice_rows[s]) {
        const int row = m.slice_row0[s] + lane;
        const long off = m.slice_off[s] + lane;
        const int width = m.slice_width[s];
        double a_ii = 0.0;
        double sigma = 0.0;
        for (int k = 0; k < width; ++k) {
            const int c = m.cols[off + (long)k * kWave];
            const double v = m.vals[off + (long)k * kWave];
            if (c == row) a_ii = v;
            else if (c >= 0) sigma += v * x[c];
        }
        if (a_ii != 0.0) {
            const double nv = (b[row] - sigma) / a_ii;
            if (L1) acc = fabs(nv - x[row]);
            x[row] = nv;
        }
    }
    if (L1) {
        const double t = block_sum(acc, scratch);
        if (threadIdx.x == 0) partial[blockIdx.x] = t;
    }
}

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
    const int s = blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
    double rr = 0.0, bb = 0.0;
    if (s < n_slices && lane < m.slice_rows[s]) {
        const int row = m.slice_row0[s] + lane;
        const long off = m.slice_off[s] + lane;
        const int width = m.slice_width[s];
        double sum = 0.0;
        for (int k = 0; k < width; ++k) {
            const int c = m.cols[off + (long)k * kWave];
            const double v = m.vals[off + (long)k * kWave];
            if (c >= 0) sum += v * in[c];
        }
        if (MODE == 0) {
            out[row] = sum;
        } else if (MODE == 2) {
            out[row] = sum;
            rr = in[row] * sum;
        } else {
            const double bv = b[row];
            const double r = bv - sum;
            rr = r * r;
            bb = bv * bv;
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

struct CsrSolveState {
    int active;
    int converged;
    int iterations;
    int pad;
    double eps_accum;     // running sum of the current checked sweep
    double last_eps;
};

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

}  // namespace ccp
