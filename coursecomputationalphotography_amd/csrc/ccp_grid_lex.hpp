// ccp_grid_lex.hpp — the reference's OWN sweep order (index order, sparse-matrix.h:357-370) on the
// structured Poisson grid, in parallel and bit for bit.
//
// In the lexicographic sweep pixel (x,y) of iteration k reads the NEW values of (x,y-1) and (x-1,y)
// and the OLD values of (x+1,y) and (x,y+1).  With tau = x + y + 2k every one of those four lies on
// hyperplane tau - 1:  (x,y-1,k) and (x-1,y,k) trivially, (x+1,y,k-1) and (x,y+1,k-1) because
// x+y+1 + 2(k-1) = tau - 1.  All points of one hyperplane are therefore independent, any order that
// walks tau upwards reproduces the sequential sweep exactly, and one hyperplane holds pixels of many
// iterations at once — the pipeline never drains between sweeps.  One launch per tau updates, in
// place, anti-diagonal d = tau - 2k of every iteration k in flight (the launch writes diagonals of
// one parity and reads the other, so there is no hazard inside a launch).
//
// Layout: "diagonal-major" — diagonal d = x + y is row d of a (W+H-1) x P array, pixel at column x:
//   (x,y-1) -> [d-1][x]   (x-1,y) -> [d-1][x-1]   (x+1,y) -> [d+1][x+1]   (x,y+1) -> [d+1][x]
// every access of a launch is unit-stride along x.  Twice the memory of the image, none of the traffic.
//
// Arithmetic per pixel: classify()/gs_update() (ccp_grid_kernels.hpp) — the reference's accumulation
// order and its true division; interior pixels take (b + (((up+left)+right)+down)) * 0.25, the same bits.
#pragma once

#include "ccp_grid_kernels.hpp"

namespace ccp {

constexpr int kLexPPT = 4;                         // pixels per thread along a diagonal
constexpr int kLexTile = kBlock * kLexPPT;         // pixels per block

struct LexGeom {
    int W, H;
    long P;          // doubles per diagonal row (>= W)
    long plane;      // doubles per channel = (W+H-1) * P
    int n_diag;      // W + H - 1
    int nbx;         // blocks along the longest diagonal
};

// split colour planes <-> diagonal-major.  grid = (ceil(W/kBlock), H, channels)
template <bool TO_DIAG>
__global__ void __launch_bounds__(kBlock)
k_lex_convert(double *__restrict__ split, double *__restrict__ diag, Geom g, LexGeom lg)
{
    const int x = blockIdx.x * kBlock + threadIdx.x;
    const int y = blockIdx.y, ch = blockIdx.z;
    if (x >= lg.W) return;
    const long s = (long)ch * g.ch_stride + row_off(g, y, (x + y) & 1) + (x >> 1);
    const long d = (long)ch * lg.plane + (long)(x + y) * lg.P + x;
    if (TO_DIAG) diag[d] = split[s];
    else split[s] = diag[d];
}

// One hyperplane.  grid = (nbx, iterations in flight, channels); blockIdx.y -> k = k_lo + blockIdx.y,
// diagonal d = tau - 2k (the host launches only k with 0 <= d < n_diag).
// CHECK: partial[((k*channels + ch)*n_diag + d)*nbx + blockIdx.x] = sum |x_new - x_old| of the block
// (the reference's per-sweep manhattonDist, reduced later in a fixed order).
template <bool CHECK>
__global__ void __launch_bounds__(kBlock)
k_lex_plane(double *__restrict__ xd, const double *__restrict__ bd, Geom g, LexGeom lg, int tau, int k_lo,
            unsigned active_mask, double *__restrict__ partial)
{
    __shared__ double scratch[kBlock / kWave];
    const int ch = blockIdx.z;
    const int k = k_lo + blockIdx.y;
    const int d = tau - 2 * k;
    const int x_lo = max(0, d - (lg.H - 1)), x_hi = min(lg.W - 1, d);
    double acc = 0.0;
    if ((active_mask >> ch) & 1u) {
        const long row = (long)ch * lg.plane + (long)d * lg.P;
#pragma unroll
        for (int q = 0; q < kLexPPT; ++q) {
            const int x = x_lo + (blockIdx.x * kLexPPT + q) * kBlock + (int)threadIdx.x;
            if (x > x_hi) continue;
            const int y = d - x;
            const Stencil s = classify(g, x, y, y);
            if (s.diag == 0) continue;                                   // empty row: skipped (sparse-matrix.h:361-363)
            const long i = row + x;
            const double up = s.up ? xd[i - lg.P] : 0.0;
            const double left = s.left ? xd[i - lg.P - 1] : 0.0;
            const double right = s.right ? xd[i + lg.P + 1] : 0.0;
            const double down = s.down ? xd[i + lg.P] : 0.0;
            const double bv = bd[i];
            double nv;
            if (s.up && s.left && s.right && s.down && s.diag == 4) nv = (bv + (((up + left) + right) + down)) * 0.25;
            else (void)gs_update(s, bv, up, left, right, down, nv);
            if (CHECK) acc += fabs(nv - xd[i]);
            xd[i] = nv;
        }
    }
    if (CHECK) {
        const double total = block_sum(acc, scratch);
        if (threadIdx.x == 0)
            partial[(((long)k * gridDim.z + ch) * lg.n_diag + d) * lg.nbx + blockIdx.x] = total;
    }
}

// eps[k*channels + ch] = sum of the partials of iteration k in a fixed order.  grid = (iterations, channels)
__global__ void __launch_bounds__(kBlock)
k_lex_reduce(const double *__restrict__ partial, long per_iteration_channel, double *__restrict__ eps)
{
    __shared__ double scratch[kBlock / kWave];
    const long slot = (long)blockIdx.x * gridDim.y + blockIdx.y;
    const double *__restrict__ p = partial + slot * per_iteration_channel;
    double acc = 0.0;
    for (long i = threadIdx.x; i < per_iteration_channel; i += kBlock) acc += p[i];
    const double t = block_sum(acc, scratch);
    if (threadIdx.x == 0) eps[slot] = t;
}

}  // namespace ccp
