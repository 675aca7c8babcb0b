// ccp_grid_fused.hpp — temporally blocked red-black Gauss-Seidel: T full iterations (2T colour
// half-sweeps) per pass over the grid, for gfx950.
//
// Why: a plain half-sweep moves 24 B per pixel update and the chip tops out near 5.5 TB/s for
// that 2-reads-1-write mix (tools/hbm_calib).  Red-black updates of one colour are mutually
// independent, so ANY schedule that respects the colour-to-colour dependences produces the
// same bits.  This kernel reads x and b once, applies 2T half-sweeps while the rows sit in
// registers, and writes x once: 24 B per pixel per T iterations (+ halo redundancy).
//
// Schedule (one wavefront = one worker; no LDS, no barriers):
//   * a wave owns a strip of 64 half-columns (128 pixel columns), one per lane, and marches
//     down the image rows of its chunk; lane i holds the red and the black pixel of its
//     half-column for every row inside a sliding window of N = 2T+4 rows (registers, the march
//     loop is unrolled N-fold so the window rotates by renaming);
//   * time skewing along the march: when row f is the newest row, row f-h receives half-sweep
//     h (h = 1..2T, red for odd h, black for even h).  Processing h in increasing order keeps
//     every neighbour at exactly the level the sequential red-black sweep would see;
//   * the row that just received half-sweep 2T is final and is stored; loads run D = 2 rows
//     ahead of the newest row;
//   * the horizontal neighbour in the adjacent half-column comes from the adjacent lane by
//     DPP (v_mov_b32_dpp wave_shr:1 / wave_shl:1), never from memory;
//   * strips overlap by 2T pixel columns per side and chunks by 2T rows per side: the values in
//     those halos go stale one column/row per half-sweep and are never stored (trapezoid
//     blocking).  Useful fraction: (128-4T)/128 in x, R/(R+4T) in y.
//   * x is read from one buffer and written to another (a neighbour strip still needs the old
//     values of the halo columns): the host ping-pongs an even number of launches.
//
// Arithmetic per update is exactly k_half_sweep's (and therefore the reference's on the
// colour-major matrix): interior  x = (b + (((up + left) + right) + down)) * 0.25,
// border pixels through classify()/gs_update().
#pragma once

#include "ccp_grid_kernels.hpp"

namespace ccp {

constexpr int kFusedMaxT = 4;
constexpr int kFusedD = 2;               // rows loaded ahead of the newest row
constexpr int kStripLanes = kWave;       // half-columns per strip

__host__ __device__ constexpr int fused_halo_px(int T) { return 2 * T; }               // per side
__host__ __device__ constexpr int fused_useful_px(int T) { return 2 * kStripLanes - 4 * T; }

// lane i <- lane i-1 (lane 0 keeps its own value) / lane i <- lane i+1 (lane 63 keeps its own)
__device__ __forceinline__ double lane_prev(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);   // wave_shr:1
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_next(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);   // wave_shl:1
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

struct FusedParams {
    const double *__restrict__ xin;
    double *__restrict__ xout;
    const double *__restrict__ b;
    Geom g;
    int st_lo, st_hi;          // local rows to finalise and store
    int rows_per_chunk;
    int n_strips;
    double *__restrict__ partial;   // L1: one double per block
};

// One wave: strip `sx`, rows [ra, rb) of channel data at xin/xout/b (already channel-offset).
template <int T, bool BORDER, bool L1>
__device__ __forceinline__ double fused_wave(const double *__restrict__ xin, double *__restrict__ xout,
                                             const double *__restrict__ bb, const Geom &g, int sx,
                                             int ra, int rb)
{
    constexpr int HS = 2 * T;
    constexpr int D = kFusedD;
    constexpr int N = HS + 2 + D;
    static_assert(N % 2 == 0, "window must hold an even number of rows (parity bookkeeping)");
    const int lane = threadIdx.x & (kWave - 1);
    const int U = fused_useful_px(T);
    const int px0 = sx * U - fused_halo_px(T);          // first pixel column of the strip (even)
    const int j = px0 / 2 + lane;                       // this lane's half-column (may be < 0)
    const bool col_ok = (j >= 0) && (j < g.pitch);
    const int ux0 = sx * U, ux1 = ux0 + U;              // pixel columns this strip stores
    const bool col_store = col_ok && (2 * j >= ux0) && (2 * j + 1 < ux1);

    const int m0 = max(ra - HS, 0);                     // rows this wave loads: [m0, m1)
    const int m1 = min(rb + HS, g.local_rows);
    // window slot of row r is (r - base) mod N with (y0 + base) even, so the colour parity of a
    // slot is a compile-time constant inside the unrolled march
    const int base = m0 - ((g.y0 + m0) & 1);
    const int f_end = rb - 1 + HS;                      // last step: row rb-1 gets half-sweep HS

    double wr[N], wk[N], br[N], bk[N];                  // x red/black, b red/black per window row
#pragma unroll
    for (int s = 0; s < N; ++s) wr[s] = wk[s] = br[s] = bk[s] = 0.0;
    double acc = 0.0;

    for (int fb = base - D; fb <= f_end; fb += N) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int f = fb + i;                       // newest row of this step
            const int u = (i - D + N) % N;              // slot of row f (compile-time after unroll)
            if (f <= f_end) {
                // ---- load row q = f + D into its slot -----------------------------------------
                {
                    const int q = f + D;
                    const int sq = (u + D) % N;
                    if (q >= m0 && q < m1) {
                        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
                        if (col_ok) {
                            const long o0 = row_off(g, q, 0) + j, o1 = row_off(g, q, 1) + j;
                            a0 = xin[o0];
                            a1 = xin[o1];
                            a2 = bb[o0];
                            a3 = bb[o1];
                        }
                        wr[sq] = a0; wk[sq] = a1; br[sq] = a2; bk[sq] = a3;
                    }
                }
                // ---- half-sweep h on row f - h, h = 1..HS ------------------------------------
#pragma unroll
                for (int h = 1; h <= HS; ++h) {
                    const int r = f - h;
                    const int sr = (u - h + 2 * N) % N;
                    const int su = (sr - 1 + N) % N, sd = (sr + 1) % N;
                    const int c = (h - 1) & 1;                       // 0 = red, 1 = black
                    const int p = ((u - h + 2 * N) + c) & 1;         // pixel column = 2j + p
                    if (r >= m0 && r < m1) {
                        // opposite colour: rows r-1, r, r+1
                        const double up = c ? wr[su] : wk[su];
                        const double dn = c ? wr[sd] : wk[sd];
                        const double same = c ? wr[sr] : wk[sr];
                        const double other = p ? lane_next(same) : lane_prev(same);
                        const double left = p ? same : other;
                        const double right = p ? other : same;
                        const double bv = c ? bk[sr] : br[sr];
                        const double old = c ? wk[sr] : wr[sr];
                        double nv = old;
                        if (!BORDER) {
                            nv = (bv + (((up + left) + right) + dn)) * 0.25;
                        } else {
                            const int x = 2 * j + p;
                            if (col_ok && x < g.W) {
                                const Stencil s = classify(g, x, g.y0 + r, r);
                                double t;
                                if (gs_update(s, bv, up, left, right, dn, t)) nv = t;
                            }
                        }
                        if (L1 && h >= HS - 1) {
                            const bool counted = col_store && r >= ra && r < rb && r >= g.own_lo && r < g.own_hi &&
                                                 (!BORDER || (2 * j + p) < g.W);
                            if (counted) acc += fabs(nv - old);
                        }
                        if (c) wk[sr] = nv; else wr[sr] = nv;
                    }
                }
                // ---- row f - HS is final: store it -------------------------------------------
                {
                    const int r = f - HS;
                    const int sr = (u - HS + 2 * N) % N;
                    if (r >= ra && r < rb && col_store) {
                        xout[row_off(g, r, 0) + j] = wr[sr];
                        xout[row_off(g, r, 1) + j] = wk[sr];
                    }
                }
            }
        }
    }
    return acc;
}

// grid = (ceil(n_strips / 4), n_chunks, channels); block = 256 threads = 4 waves = 4 adjacent
// strips of one chunk.
template <int T, bool L1>
__global__ void __launch_bounds__(kBlock)
k_fused_sweep(FusedParams P)
{
    __shared__ double scratch[kBlock / kWave];
    const int wave = threadIdx.x / kWave;
    const int sx = blockIdx.x * (kBlock / kWave) + wave;
    const int ch = blockIdx.z;
    const int ra = P.st_lo + blockIdx.y * P.rows_per_chunk;
    const int rb = min(ra + P.rows_per_chunk, P.st_hi);
    double acc = 0.0;
    if (sx < P.n_strips && ra < rb) {
        const Geom &g = P.g;
        const long off = (long)ch * g.ch_stride;
        constexpr int HS = 2 * T;
        // a wave needs the border-aware update if any pixel it can touch lacks a neighbour:
        // strip at the left/right image edge, or rows at the top/bottom of the local block
        // (image border, or the stale edge of the ghost zone)
        const int px0 = sx * fused_useful_px(T) - fused_halo_px(T);
        const int px1 = px0 + 2 * kStripLanes;                      // exclusive
        const bool border = (px0 <= 0) || (px1 >= g.W - 1) || (ra - HS <= 0) || (rb + HS >= g.local_rows) ||
                            (g.y0 + ra - HS <= 0) || (g.y0 + rb + HS >= g.H - 1);
        if (border) acc = fused_wave<T, true, L1>(P.xin + off, P.xout + off, P.b + off, g, sx, ra, rb);
        else acc = fused_wave<T, false, L1>(P.xin + off, P.xout + off, P.b + off, g, sx, ra, rb);
    }
    if (L1) {
        const double total = block_sum(acc, scratch);
        if (threadIdx.x == 0)
            P.partial[((long)ch * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = total;
    }
}

}  // namespace ccp
