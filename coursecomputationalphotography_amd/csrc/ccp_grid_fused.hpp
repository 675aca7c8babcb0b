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

constexpr int kFusedMaxT = 8;
constexpr int kFusedMaxCheckedT = 5;     // deepest pass that also reports the step of each of its sweeps (+2T VGPRs)
constexpr int kFusedUnroll = 4;          // march steps unrolled per loop trip (shifted window)
// rows loaded ahead of the newest row: enough bytes in flight per CU at the occupancy the
// register window of T allows (T<=5: 3 waves/SIMD, T>=6: 2 waves/SIMD)
#ifndef CCP_FUSED_D_LO
#define CCP_FUSED_D_LO 2
#endif
#ifndef CCP_FUSED_D_HI
#define CCP_FUSED_D_HI 2
#endif
__host__ __device__ constexpr int fused_prefetch(int T) { return T <= 5 ? CCP_FUSED_D_LO : CCP_FUSED_D_HI; }
constexpr int kStripLanes = kWave;       // half-columns per strip

__host__ __device__ constexpr int fused_halo_px(int T) { return 2 * T; }               // per side
__host__ __device__ constexpr int fused_useful_px(int T) { return 2 * kStripLanes - 4 * T; }

// lane i <- lane i-1 (lane 0 gets 0) / lane i <- lane i+1 (lane 63 gets 0): the strip-edge lanes
// have no neighbour inside the wave; their pixels are halo (never stored) or image-edge pixels
// whose stencil has no such neighbour.
__device__ __forceinline__ double lane_prev(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);   // wave_shr:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_next(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, true);   // wave_shl:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
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
    double *__restrict__ partial;   // L1 = 1: one double per block; L1 = 2: T doubles per block (per iteration)
    const int *__restrict__ active; // nullable: per-channel "still iterating" flags (device)
};

// How the row window maps onto registers.  G march steps are unrolled per loop trip.
//   G == N (UNR = 0): the window rotates by renaming, slot = (step - distance) mod N, no copies,
//       but the code grows as N*2T row updates (T=5: 74 KB, T=8: 177 KB — past the instruction
//       cache: T=8 measured 3.0 ms per pass against 2.0 ms for the shifted form).  Kept for
//       reference; the library instantiates UNR = kFusedUnroll only.
//   G == UNR (> 0)  : slot = 2T+1 + step - distance inside a trip, then the whole window is
//       shifted down by UNR slots ((2T+D+1)*4/UNR register moves per step); code ~ UNR*2T updates.
template <int T, int UNR>
struct FusedWindow {
    static constexpr int HS = 2 * T;
    static constexpr int D = fused_prefetch(T);
    static constexpr int NFULL = HS + 2 + D;
    static constexpr int G = UNR > 0 ? UNR : NFULL;
    static constexpr int NT = UNR > 0 ? HS + D + UNR + 1 : NFULL;
    static_assert(G % 2 == 0 && D % 2 == 0, "row parity must be a compile-time constant per unrolled step");
    // slot of the row at `dist` rows behind the newest row of unrolled step i (dist in [-D, HS+1])
    __host__ __device__ static constexpr int slot(int i, int dist)
    {
        return UNR > 0 ? (HS + 1 + i - dist) : ((i - D - dist) % NFULL + 2 * NFULL) % NFULL;
    }
};

// Wave-uniform march parameters (SGPRs) and the few per-lane predicates of a strip.
struct FusedCtx {
    const double *__restrict__ xin;
    double *__restrict__ xout;
    const double *__restrict__ bb;
    int jbase;          // half-column of lane 0 (may be negative in the first strip)
    int j;              // this lane's half-column
    unsigned lane;
    bool col_ok;        // lane's half-column exists
    bool col_store;     // lane's pixels belong to the columns this strip stores
    int ra, rb;         // rows to finalise and store
    int m0, m1;         // rows loaded
};

// One march step: newest row f, unrolled position i.  STEADY: every row the step touches is known
// to be inside [m0, m1), so loads and row updates are straight-line code (the only scalar branch
// left guards the store of the finished row): the s_waitcnt pass can then count the loads in
// flight instead of draining them, and that is what lets the D-rows-ahead prefetch overlap.
// L1: 0 = no step norm; 1 = accumulate sum|x_new - x_old| of the pass's LAST iteration in acc[0];
// 2 = of EVERY iteration t = 1..T in acc[t-1] (old is the previous level of the same colour, which
// the window still holds — the reference's per-sweep manhattonDist at no extra memory traffic).
template <int T, bool BORDER, int L1, int UNR, bool STEADY, int NT, int AN>
__device__ __forceinline__ void fused_step(double (&wr)[NT], double (&wk)[NT], double (&br)[NT], double (&bk)[NT],
                                           double (&acc)[AN], const FusedCtx &cx, const Geom &g, int f, int i)
{
    using Win = FusedWindow<T, UNR>;
    constexpr int HS = Win::HS, D = Win::D;
    // ---- load row q = f + D into its slot ---------------------------------------------------
    {
        const int q = f + D;
        const int sq = Win::slot(i, -D);
        if (STEADY || (q >= cx.m0 && q < cx.m1)) {
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            if (!BORDER || cx.col_ok) {
                // uniform row base + lane index
                const long o0 = row_off(g, q, 0) + cx.jbase, o1 = row_off(g, q, 1) + cx.jbase;
                a0 = (cx.xin + o0)[cx.lane];
                a1 = (cx.xin + o1)[cx.lane];
                a2 = (cx.bb + o0)[cx.lane];
                a3 = (cx.bb + o1)[cx.lane];
            }
            wr[sq] = a0; wk[sq] = a1; br[sq] = a2; bk[sq] = a3;
        }
    }
    // ---- half-sweep h on row f - h, h = 1..HS ----------------------------------------------
#pragma unroll
    for (int h = 1; h <= HS; ++h) {
        const int r = f - h;
        const int sr = Win::slot(i, h), su = Win::slot(i, h + 1), sd = Win::slot(i, h - 1);
        const int c = (h - 1) & 1;                       // 0 = red, 1 = black
        const int p = ((i - h + 2 * HS + 2) + c) & 1;    // pixel column = 2j + p
        if (STEADY || (r >= cx.m0 && r < cx.m1)) {
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
                const int x = 2 * cx.j + p;
                if (cx.col_ok && x < g.W) {
                    const Stencil s = classify(g, x, g.y0 + r, r);
                    double t;
                    if (gs_update(s, bv, up, left, right, dn, t)) nv = t;
                }
            }
            if ((L1 == 1 && h >= HS - 1) || L1 == 2) {
                const bool counted = cx.col_store && r >= cx.ra && r < cx.rb && r >= g.own_lo && r < g.own_hi &&
                                     (!BORDER || (2 * cx.j + p) < g.W);
                if (counted) acc[L1 == 2 ? (h - 1) / 2 : 0] += fabs(nv - old);
            }
            if (c) wk[sr] = nv; else wr[sr] = nv;
        }
    }
    // ---- row f - HS is final: store it -----------------------------------------------------
    {
        const int r = f - HS;
        const int sr = Win::slot(i, HS);
        if (r >= cx.ra && r < cx.rb && cx.col_store) {
            (cx.xout + (row_off(g, r, 0) + cx.jbase))[cx.lane] = wr[sr];
            (cx.xout + (row_off(g, r, 1) + cx.jbase))[cx.lane] = wk[sr];
        }
    }
    // keep the machine scheduler from pulling later steps' loads/updates up across this point:
    // unconstrained it hoists all G steps' work together and spills the register window
    __builtin_amdgcn_sched_barrier(0);
}

// One wave: strip `sx`, rows [ra, rb) of channel data at xin/xout/b (already channel-offset).
template <int T, bool BORDER, int L1, int UNR, int AN>
__device__ __forceinline__ void fused_wave(const double *__restrict__ xin, double *__restrict__ xout,
                                           const double *__restrict__ bb, const Geom &g, int sx,
                                           int ra, int rb, double (&acc)[AN])
{
    using Win = FusedWindow<T, UNR>;
    constexpr int HS = Win::HS, D = Win::D, G = Win::G, NT = Win::NT;
    FusedCtx cx;
    cx.xin = xin; cx.xout = xout; cx.bb = bb;
    cx.lane = threadIdx.x & (kWave - 1);
    const int U = fused_useful_px(T);
    const int px0 = sx * U - fused_halo_px(T);          // first pixel column of the strip (even)
    cx.jbase = px0 / 2;                                 // wave-uniform (sx is): row pointers stay scalar
    cx.j = cx.jbase + (int)cx.lane;                     // this lane's half-column (may be < 0)
    cx.col_ok = (cx.j >= 0) && (cx.j < g.pitch);
    const int ux0 = sx * U, ux1 = ux0 + U;              // pixel columns this strip stores
    cx.col_store = cx.col_ok && (2 * cx.j >= ux0) && (2 * cx.j + 1 < ux1);
    cx.ra = ra; cx.rb = rb;
    cx.m0 = max(ra - HS, 0);                            // rows this wave loads: [m0, m1)
    cx.m1 = min(rb + HS, g.local_rows);
    // the march starts on an even image row (y0 + base even) and advances G (even) rows per
    // trip, so the colour parity of every unrolled row update is a compile-time constant
    const int base = cx.m0 - ((g.y0 + cx.m0) & 1);
    const int f_end = rb - 1 + HS;                      // last step: row rb-1 gets half-sweep HS
    // steps f in [s_lo, s_hi] touch only existing rows: r = f-h >= m0 for h <= HS+1, f+D < m1
    const int s_lo = cx.m0 + HS + 1, s_hi = cx.m1 - 1 - D;

    double wr[NT], wk[NT], br[NT], bk[NT];              // x red/black, b red/black per window row
#pragma unroll
    for (int s = 0; s < NT; ++s) wr[s] = wk[s] = br[s] = bk[s] = 0.0;

    for (int fb = base - D; fb <= f_end; fb += G) {
        if (fb >= s_lo && fb + G - 1 <= s_hi) {
#pragma unroll
            for (int i = 0; i < G; ++i)
                fused_step<T, BORDER, L1, UNR, true, NT, AN>(wr, wk, br, bk, acc, cx, g, fb + i, i);
        } else {
#pragma unroll
            for (int i = 0; i < G; ++i)
                if (fb + i <= f_end)
                    fused_step<T, BORDER, L1, UNR, false, NT, AN>(wr, wk, br, bk, acc, cx, g, fb + i, i);
        }
        if (UNR > 0) {
#pragma unroll
            for (int s = 0; s + G < NT; ++s) {
                wr[s] = wr[s + G]; wk[s] = wk[s + G]; br[s] = br[s + G]; bk[s] = bk[s + G];
            }
        }
    }
}

// grid = (ceil(n_strips / 4), n_chunks, channels); block = 256 threads = 4 waves = 4 adjacent
// strips of one chunk.  Each wave picks one of two bodies: the branch-free one when all its
// pixels have four neighbours, the border-aware one when it touches an image edge or the stale
// edge of a ghost zone.  (Two separate launches were tried: the border launch ran alone at low
// occupancy and cost +0.25 ms per pass at 16384^2.)
// Waves per SIMD the register window of depth T is budgeted for (2nd __launch_bounds__ argument:
// it caps the allocator, so the straight-line steady-state code cannot trade occupancy for
// load hoisting).
__host__ __device__ constexpr int fused_waves_per_simd(int T, int L1 = 0) { return T <= 1 ? 4 : (T <= (L1 == 2 ? 2 : 3) ? 3 : 2); }

template <int T, int L1, int UNR>
__global__ void __launch_bounds__(kBlock, fused_waves_per_simd(T, L1))
k_fused_sweep(FusedParams P)
{
    __shared__ double scratch[kBlock / kWave];
    // readfirstlane: tell the compiler the wave index is uniform, so strip/row addressing is SALU
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int sx = blockIdx.x * (kBlock / kWave) + wave;
    const int ch = blockIdx.z;
    // Workgroups are dispatched in index order.  The waves of the first and the last chunk take
    // the border-aware body, which runs ~2.6x longer at T=8: give the last chunk index 1 instead
    // of the highest one so both start at the beginning and never form the tail of the launch.
    int chunk = blockIdx.y;
    if (gridDim.y > 2) chunk = blockIdx.y == 1 ? (int)gridDim.y - 1 : (blockIdx.y > 1 ? (int)blockIdx.y - 1 : 0);
    const int ra = P.st_lo + chunk * P.rows_per_chunk;
    const int rb = min(ra + P.rows_per_chunk, P.st_hi);
    constexpr int AN = L1 == 2 ? T : 1;
    double acc[AN];
#pragma unroll
    for (int t = 0; t < AN; ++t) acc[t] = 0.0;
    const bool run = (P.active == nullptr) || (P.active[ch] != 0);   // a converged channel is frozen
    if (run && sx < P.n_strips && ra < rb) {
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
        if (border) fused_wave<T, true, L1, UNR, AN>(P.xin + off, P.xout + off, P.b + off, g, sx, ra, rb, acc);
        else fused_wave<T, false, L1, UNR, AN>(P.xin + off, P.xout + off, P.b + off, g, sx, ra, rb, acc);
    }
    if (L1 != 0) {
        // partial[((t*channels + ch)*gridDim.y + by)*gridDim.x + bx], t = 0 for L1 = 1
#pragma unroll
        for (int t = 0; t < AN; ++t) {
            const double total = block_sum(acc[t], scratch);
            if (threadIdx.x == 0)
                P.partial[(((long)t * gridDim.z + ch) * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = total;
        }
    }
}

}  // namespace ccp
