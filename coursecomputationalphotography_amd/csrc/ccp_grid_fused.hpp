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
//     half-column (x and b) for every row inside a sliding window of 2T+3 rows in registers;
//   * time skewing along the march: when row f is the newest row, row f-h receives half-sweep
//     h (h = 1..2T, red for odd h, black for even h).  Processing h in increasing order keeps
//     every neighbour at exactly the level the sequential red-black sweep would see;
//   * the row that just received half-sweep 2T is final and is stored; the rows the next loop
//     trip starts with are loaded a whole trip ahead into landing registers (FusedWindow);
//   * the horizontal neighbour in the adjacent half-column comes from the adjacent lane by
//     DPP (v_mov_b32_dpp wave_shr:1 / wave_shl:1), never from memory;
//   * strips overlap by 2T pixel columns per side and chunks by 2T rows per side: the values in
//     those halos go stale one column/row per half-sweep and are never stored (trapezoid
//     blocking).  Useful fraction: (128-4T)/128 in x, R/(R+4T) in y;
//   * x is read from one buffer and written to another (a neighbour strip still needs the old
//     values of the halo columns): the host ping-pongs an even number of launches;
//   * all memory accesses are raw buffer instructions whose range check drops what must not be
//     read or written (row_rsrc below): the march has no branch around a memory instruction.
//
// Arithmetic per update is exactly k_half_sweep's (and therefore the reference's on the
// colour-major matrix): interior  x = (b + (((up + left) + right) + down)) * 0.25,
// border pixels as classify()/gs_update() do.
#pragma once

#include "ccp_grid_kernels.hpp"

namespace ccp {

#ifndef CCP_FUSED_MAX_T
#define CCP_FUSED_MAX_T 8
#endif
constexpr int kFusedMaxT = CCP_FUSED_MAX_T;          // deepest pass instantiated
constexpr int kFusedMaxCheckedT = 8;     // deepest pass that also reports the step of each of its sweeps (+2T+4 VGPRs)
#ifndef CCP_FUSED_UNROLL
#define CCP_FUSED_UNROLL 2
#endif
constexpr int kFusedUnroll = CCP_FUSED_UNROLL;          // march steps unrolled per loop trip (shifted window)
#ifndef CCP_FUSED_LAND
#define CCP_FUSED_LAND 2
#endif
// Trips of rows in flight from memory in the ordinary tiles of an unchecked, unmasked pass of depth 8: 2 (the checked and
// the masked kernels, and the shallower passes at their higher occupancy, have no registers left for a second landing
// pair and keep 1).  Measured (profiles/r03_window_forms.jsonl):
// +2-3 % at 16384^2 and 4096^2 x 3 over one trip.
constexpr int kFusedLand = CCP_FUSED_LAND;
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
    // A chunk row that touches the image top / bottom runs in k_fused_border, whose waves march
    // slower: it is kept short (first_rows / last_rows > 0) so those tiles end early; the chunks in
    // between have rows_per_chunk rows (fused_chunk_rows).
    int first_rows, last_rows;
    int n_strips;
    double *__restrict__ partial;   // L1 = 1: one double per block; L1 = 2: T doubles per block (per iteration)
    const int *__restrict__ active; // nullable: per-channel "still iterating" flags (device)
    // Tiles (chunk c, strip s) that can touch a pixel with a missing neighbour — image edges, the
    // stale edge of a ghost zone — are "border tiles": the first nb_top / last nb_bot chunks and the
    // first ns_left / last ns_right strips (host: fused_tile_counts).  k_fused_sweep skips them,
    // k_fused_border runs exactly them, concurrently on a second stream.
    int n_chunks;
    int nb_top, nb_bot, ns_left, ns_right;
    // the side strips of the middle chunks march slower per step (kStepSide), so each chunk of them
    // is cut into side_subs tiles of side_rows rows: the slow tiles end with the ordinary ones
    int side_rows, side_subs;
    int side_rows_edge;                    // sub-tile height in the edge chunks of an EDGE pass: short, so they are final early
    double *__restrict__ partial_border;   // L1 sums of the border launch (same layout, own region)
    // Row blocks with neighbours (EDGE kernels): chunk 0 / chunk n_chunks-1 hold exactly the owned rows the
    // upper / lower neighbour's ghost zone takes (first_edge / last_edge).  They are short (first_rows /
    // last_rows), dispatched first (fused_chunk_of), and every wave that finished one of their tiles counts
    // itself in *edge_counter; the wave that brings it to edge_target publishes edge_epoch in *edge_flag,
    // which the halo exchange's stream is waiting for (hipStreamWaitValue64) — so the messages leave while
    // the rest of the pass is still running.
    // Dirichlet-mask grids (MASKED kernels): one byte per pixel in the layout of x (colour half-rows of
    // `pitch` bytes): 1 = unknown, 0 = fixed at zero
    const unsigned char *__restrict__ mask;
    int first_edge, last_edge;
    // Workgroups are dealt round-robin over the 8 XCDs (MI355X_MICROARCH.md, Workgroup dispatch): with
    // xcd_swizzle every XCD walks a CONTIGUOUS run of tiles (strip groups fastest) in dispatch order, so the
    // workgroups that share an L2 at any moment are neighbours in x and the halo columns they both read are
    // fetched from HBM once.  Speed only: placement is never assumed for correctness.
    int xcd_swizzle;
    // Diagnostics (CCP_GS_TRACE_FILE): 4 words per wave — start and end time (100 MHz constant clock), HW_ID |
    // XCC_ID << 32, and chunk | strip << 16 | channel << 32 | kernel << 40 — written by lane 0; nullptr otherwise.
    unsigned long long *__restrict__ trace;
    unsigned long long *__restrict__ edge_counter;
    unsigned long long *__restrict__ edge_flag;
    unsigned long long edge_target, edge_epoch;
};

// blockIdx.y -> chunk with the edge chunks first (workgroups are dispatched in block-index order)
__host__ __device__ __forceinline__ int fused_chunk_of(const FusedParams &P, int y)
{
    const int n_e = P.first_edge + P.last_edge;
    if (y < n_e) return (P.first_edge && y == 0) ? 0 : P.n_chunks - 1;
    return y - n_e + P.first_edge;
}

__host__ __device__ __forceinline__ bool fused_is_edge_chunk(const FusedParams &P, int chunk)
{
    return (P.first_edge && chunk == 0) || (P.last_edge && chunk == P.n_chunks - 1);
}

// One wave reports a finished edge tile.  Producer side of the hand-off (MI355X_MICROARCH.md, correctness
// boundaries): the wave's stores are drained and written back at agent scope before it counts itself.
__device__ __forceinline__ void fused_signal_edge(const FusedParams &P)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if ((threadIdx.x & (kWave - 1)) == 0) {
        const unsigned long long old = __hip_atomic_fetch_add(P.edge_counter, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == P.edge_target)
            __hip_atomic_store(P.edge_flag, P.edge_epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// The tile coordinates of this workgroup: blockIdx, or — xcd_swizzle — the id-th tile of the run of tiles its
// XCD walks.  Block id = 8k + x goes to XCD x (observed); XCD x owns tiles [x q + min(x, r), ...) with
// q = n / 8, r = n % 8, of which this is the k-th: a bijection of [0, n) for every n.
__device__ __forceinline__ void fused_tile_coords(const FusedParams &P, unsigned &bx, unsigned &by, unsigned &bz)
{
    bx = blockIdx.x;
    by = blockIdx.y;
    bz = blockIdx.z;
    if (P.xcd_swizzle) {
        const unsigned gx = gridDim.x, gy = gridDim.y;
        const unsigned n = gx * gy * gridDim.z;
        const unsigned id = bx + gx * (by + gy * bz);
        const unsigned xcd = id & 7u, k = id >> 3;
        const unsigned q = n >> 3, r = n & 7u;
        const unsigned tile = xcd * q + (xcd < r ? xcd : r) + k;
        bx = tile % gx;
        const unsigned t = tile / gx;
        by = t % gy;
        bz = t / gy;
    }
}

__device__ __forceinline__ void fused_trace_begin(const FusedParams &P, unsigned long long &t0)
{
    if (P.trace) t0 = wall_clock64();
}
__device__ __forceinline__ void fused_trace_end(const FusedParams &P, unsigned long long t0, long slot, int chunk, int sx, int ch, int kernel)
{
    if (P.trace && (threadIdx.x & (kWave - 1)) == 0) {
        unsigned long long *r = P.trace + 4 * slot;
        r[0] = t0;
        r[1] = wall_clock64();
        r[2] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11))        // HW_REG_HW_ID, 32 bits
               | ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 32);   // HW_REG_XCC_ID[3:0]
        r[3] = (unsigned long long)(unsigned)chunk | ((unsigned long long)(unsigned)sx << 16) | ((unsigned long long)ch << 32) |
               ((unsigned long long)kernel << 40);
    }
}

// rows [ra, rb) of chunk c
__host__ __device__ __forceinline__ void fused_chunk_rows(const FusedParams &P, int c, int &ra, int &rb)
{
    if (P.first_rows > 0 && c == 0) {
        ra = P.st_lo;
        rb = P.st_lo + P.first_rows;
    } else if (P.last_rows > 0 && c == P.n_chunks - 1) {
        ra = P.st_hi - P.last_rows;
        rb = P.st_hi;
    } else {
        const int k = c - (P.first_rows > 0 ? 1 : 0);
        const int end = P.st_hi - P.last_rows;
        ra = P.st_lo + P.first_rows + k * P.rows_per_chunk;
        rb = ra + P.rows_per_chunk < end ? ra + P.rows_per_chunk : end;
    }
}

// How the row window maps onto registers.  G = UNR march steps are unrolled per loop trip: inside
// a trip the row at `dist` rows behind the newest row of step i sits in slot 2T+1 + i - dist;
// after the trip the whole window is shifted down by G slots ((2T+1)*4/G register moves per step;
// code ~ G*2T row updates — the fully renamed form, no moves but N*2T updates, is 177 KB at T=8
// and lost 1.5x to instruction-cache misses).
// Rows in flight from memory are not in the window: a trip first issues the loads of the G rows
// the NEXT trip will start with into landing registers, runs its G steps, shifts the window and
// only then moves the landed rows in.  Loads and their first use sit in the same loop iteration,
// a whole trip of arithmetic apart, and nothing that is shifted is the destination of a load in
// flight.
template <int T, int UNR>
struct FusedWindow {
    static_assert(UNR > 0 && UNR % 2 == 0, "row parity must be a compile-time constant per unrolled step");
    static constexpr int HS = 2 * T;
    static constexpr int G = UNR;
    static constexpr int NT = HS + G + 1;
    // slot of the row at `dist` rows behind the newest row of unrolled step i (dist in [0, HS+1])
    __host__ __device__ static constexpr int slot(int i, int dist) { return HS + 1 + i - dist; }
};

// (A RING form of the window — slot (i - dist) mod P with P = 2T + 2 + rows in flight steps unrolled per trip, nothing
// ever moved: a third fewer vector instructions — was built and measured in round 3: 7-20 % SLOWER, the waves parked on
// s_waitcnt twice as long; the pass is bound by memory, not by vector issue.  NOTES.md, profiles/r03_window_forms.jsonl.)

// Memory goes through raw buffer instructions with one descriptor per image row: a row that does
// not exist gets num_records = 0 and a lane whose half-column does not exist (or must not be
// stored) an offset past the row, so loads return 0 and stores are dropped by the address range
// check instead of by branches.  With no branch around any memory instruction the march is
// straight-line code and every s_waitcnt the compiler places is an exact count.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr unsigned kLaneOut = 0x80000000u;          // byte offset no row reaches

__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(const double *plane, const Geom &g, int row, bool exists)
{
    // base of the red half-row; the black half-row follows at +pitch doubles
    return __builtin_amdgcn_make_buffer_rsrc((void *)(plane + (long)row * 2 * g.pitch), 0,
                                             exists ? (int)(g.pitch * 16) : 0, 0x00020000);
}
// COH: the access carries sc1 (aux bit 4: system-coherence level 1 on gfx942/gfx950) — a store writes through to
// memory, a load does not trust the XCD's (non-coherent) L2.  What k_fused_multi hands from one pass to the next
// INSIDE a launch goes through these (MI355X_MICROARCH.md, correctness boundaries: every store of the handed-off
// bytes sc1 and drained before the counter, every load of them an sc1 load).
constexpr int kAuxSc1 = 16;
template <bool COH = false>
__device__ __forceinline__ double buf_load(__amdgpu_buffer_rsrc_t rs, unsigned voff)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)voff, 0, COH ? kAuxSc1 : 0));
}
template <bool COH = false>
__device__ __forceinline__ void buf_store(double v, __amdgpu_buffer_rsrc_t rs, unsigned voff)
{
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rs, (int)voff, 0, COH ? kAuxSc1 : 0);
}

// Wave-uniform march parameters (SGPRs) and the few per-lane values of a strip.
struct FusedCtx {
    const double *__restrict__ xin;
    double *__restrict__ xout;
    const double *__restrict__ bb;
    int j;              // this lane's half-column (may be negative in the first strip)
    unsigned ld_r, ld_k;   // byte offset of the lane's red / black pixel inside a row, kLaneOut if the half-column does not exist
    unsigned st_r, st_k;   // the same for stores: kLaneOut unless the strip stores this half-column
    const unsigned char *__restrict__ mask;   // MASKED: the mask plane (channel-independent)
    unsigned ldm_r, ldm_k; // MASKED: byte offset of the lane's red / black mask byte inside a row
    bool col_store;     // lane's pixels belong to the columns this strip stores
    int ra, rb;         // rows to finalise and store
    int m0, m1;         // rows loaded
    // border tiles only: what depends on the pixel column, per parity p (pixel x = 2j + p)
    bool col_interior;  // wave-uniform: every pixel column of the strip is in [1, W-2]
    bool px_ok[2];      // the pixel exists (x < W, half-column in range)
    bool px_left[2];    // x >= 1
    bool px_right[2];   // x < W-1
    bool px_first[2];   // x == 0
    bool has_first;     // wave-uniform: the strip holds pixel column 0 (the a_ii = 3 column)
};

// Row q of x (red, black) and b (red, black); rows outside [m0, m1) and lanes outside the image read 0.
template <bool COH = false>
__device__ __forceinline__ void fused_load_row(const FusedCtx &cx, const Geom &g, int q, double (&dst)[4])
{
    const bool exists = q >= cx.m0 && q < cx.m1;
    const __amdgpu_buffer_rsrc_t rx = row_rsrc(cx.xin, g, q, exists), rbb = row_rsrc(cx.bb, g, q, exists);
    dst[0] = buf_load<COH>(rx, cx.ld_r);
    dst[1] = buf_load<COH>(rx, cx.ld_k);
    dst[2] = buf_load(rbb, cx.ld_r);                   // (b never changes during a solve: plain loads)
    dst[3] = buf_load(rbb, cx.ld_k);
}

// MASKED: the two mask bytes of the lane's pixels in row q (0 outside the block / the image).
__device__ __forceinline__ void fused_load_mask(const FusedCtx &cx, const Geom &g, int q, int (&dst)[2])
{
    const bool exists = q >= cx.m0 && q < cx.m1;
    const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void *)(cx.mask + (long)q * 2 * g.pitch), 0,
                                                                        exists ? (int)(g.pitch * 2) : 0, 0x00020000);
    dst[0] = (int)__builtin_amdgcn_raw_buffer_load_b8(rm, (int)cx.ldm_r, 0, 0);
    dst[1] = (int)__builtin_amdgcn_raw_buffer_load_b8(rm, (int)cx.ldm_k, 0, 0);
}

// One march step: newest row f, unrolled position i: half-sweep h on row f - h for h = 1..2T, then
// the store of row f - 2T.  No range checks: rows that do not exist (pipeline fill and drain, the
// outside of the image) hold zeros or garbage, are updated like any other, are never stored, and
// by the trapezoid argument never reach a stored value.  Three bodies, chosen per loop trip:
//   kStepFast   : ordinary image rows of a strip away from the left/right image edge — the only
//                 body of k_fused_sweep;
//   kStepSide   : (border tiles) ordinary rows, strip at the left/right image edge: absent
//                 neighbours are selected to 0 per lane (adding 0 is exact, so the sum keeps the
//                 reference's order), a_ii is 4 inside, 1 in column W-1 and 3 in column 0 (the
//                 only true division);
//   kStepBorder : (border tiles) rows at an image edge or at the stale edge of a ghost zone: the
//                 border-aware arithmetic of SURVEY §8a-8 from per-lane column flags and
//                 wave-uniform row flags.
// L1: 0 = no step norm; 1 = accumulate sum|x_new - x_old| of the pass's LAST iteration in acc[0];
// 2 = of EVERY iteration t = 1..T in acc[t-1] (old is the previous level of the same colour, which
// the window still holds — the reference's per-sweep manhattonDist at no extra memory traffic).
constexpr int kStepFast = 0, kStepBorder = 2, kStepSide = 3;
typedef int QWord;
constexpr int kQuarterHi = 0x3FD00000;                  // high word of 0.25

// MASKED (Dirichlet-mask grid): the update is fma(sum, q, b/4) with q = 1/4 for an unknown and 0 for a pixel
// fixed at zero (whose b is 0).  The stencil is the uniform 5-point one — no degree logic anywhere.  Both values of q
// have a zero low word, so the window keeps the HIGH word only (QWord: one VGPR per pixel instead of two — what lets
// the masked pass go as deep as the plain one) and the factor is put together at its use (a register move or two).
template <int T, int MODE, int L1, int UNR, int NT, int AN, bool MASKED = false, int NQ = 1, bool COH = false, class Win = FusedWindow<T, UNR>>
__device__ __forceinline__ void fused_step(double (&wr)[NT], double (&wk)[NT], double (&br)[NT], double (&bk)[NT],
                                           double (&acc)[AN], const FusedCtx &cx, const Geom &g, int f, int i,
                                           QWord (&qr)[NQ], QWord (&qk)[NQ])
{
    static_assert(Win::NT == NT, "window policy and register arrays disagree");
    constexpr int HS = Win::HS;
#pragma unroll
    for (int h = 1; h <= HS; ++h) {
        const int r = f - h;
        const int sr = Win::slot(i, h), su = Win::slot(i, h + 1), sd = Win::slot(i, h - 1);
        const int c = (h - 1) & 1;                       // 0 = red, 1 = black
        const int p = ((i - h + 2 * HS + 2) + c) & 1;    // pixel column = 2j + p
        // opposite colour: rows r-1, r, r+1
        const double up = c ? wr[su] : wk[su];
        const double dn = c ? wr[sd] : wk[sd];
        const double same = c ? wr[sr] : wk[sr];
        const double other = p ? lane_next(same) : lane_prev(same);
        const double left = p ? same : other;
        const double right = p ? other : same;
        // the window holds b/4 (exact): (b + s)/4 == fma(s, 1/4, b/4) bit for bit — scaling by a power
        // of two commutes with the one rounding — and saves an instruction per update
        const double bq = c ? bk[sr] : br[sr];
        const double old = c ? wk[sr] : wr[sr];
        double nv = old;
        if (MODE == kStepFast) {
            if (MASKED) nv = __builtin_fma(((up + left) + right) + dn, __hiloint2double(c ? qk[MASKED ? sr : 0] : qr[MASKED ? sr : 0], 0), bq);
            else nv = __builtin_fma(((up + left) + right) + dn, 0.25, bq);
        } else if (MODE == kStepSide) {
            const double bv = bq * 4.0;
            // ordinary row y in [1, H-2]: cell(x,y-1), cell(x,y) exist iff x < W-1 (up, right,
            // down), cell(x-1,y) iff x >= 1 (left); a_ii = 3[x < W-1] + [x >= 1]
            const bool mr = cx.px_right[p], ml = cx.px_left[p];
            const double t = bv + ((((mr ? up : 0.0) + (ml ? left : 0.0)) + (mr ? right : 0.0)) + (mr ? dn : 0.0));
            double q = t * (ml && mr ? 0.25 : 1.0);
            if (p == 0 && cx.has_first) {
                if (cx.px_first[0] && mr) q = t / 3.0;
            }
            // a_ii = 0 (a one-pixel-wide image: no cell left or right of the column) is an empty row: skipped
            nv = (cx.px_ok[p] && (mr || ml)) ? q : old;
        } else {
            // Most rows of a border trip are still ordinary: an image row with both neighbour
            // rows present, in a strip away from the left/right image edge, takes the plain
            // update.  Otherwise the stencil of SURVEY §8a-8 is evaluated from the per-lane
            // column flags and the (wave-uniform) row flags; only a_ii = 3 (image row 0, image
            // column 0) needs a true division — 1, 2 and 4 are exact reciprocals.
            const int y = g.y0 + r;
            const bool row_plain = (y >= 1) && (y <= g.H - 2);
            const double bv = bq * 4.0;
            if (row_plain && cx.col_interior) {
                nv = __builtin_fma(((up + left) + right) + dn, 0.25, bq);
            } else if (cx.px_ok[p]) {
                const bool cellrow = y < g.H - 1;                        // cell(.,y) rows
                const bool cf_up = (y >= 1) && cx.px_right[p];           // cell(x, y-1)
                const bool s_left = cx.px_left[p] && cellrow;            // cell(x-1, y)
                const bool s_here = cx.px_right[p] && cellrow;           // cell(x, y)
                const int diag = (int)cf_up + (int)s_left + 2 * (int)s_here + (int)(cx.px_first[p] && y == 0);
                const bool s_up = cf_up && (r >= 1);
                const bool s_down = s_here && (r + 1 < g.local_rows);
                double sigma = 0.0;
                if (s_up) sigma += -1.0 * up;
                if (s_left) sigma += -1.0 * left;
                if (s_here) sigma += -1.0 * right;
                if (s_down) sigma += -1.0 * dn;
                const double t = bv - sigma;
                if (diag == 3) nv = t / 3.0;
                else if (diag != 0) nv = t * (diag == 4 ? 0.25 : (diag == 2 ? 0.5 : 1.0));
            }
        }
        if ((L1 == 1 && h >= HS - 1) || L1 == 2) {
            // Every lane adds its step on the rows that count; the lanes whose columns another strip stores are dropped
            // ONCE, at the end of fused_wave, instead of inside the test of every update (the checked pass is bound by
            // vector issue: 4 instructions per update for the step on top of 7, 2.0 ms a pass against 1.4 without them —
            // ISA and an experiment build, round 4).  A pixel that does not exist keeps its value (nv = old) and adds an
            // exact 0.  The sums are the same sums, lane for lane.
            // (The row test is wave-uniform, yet the compiler makes two selects per update of it.  Tried: a forced scalar
            // branch — slower, 1.06 against 1.10e12 at 16384^2; bodies specialised for "every row of the trip counts",
            // side by side in one loop or as three loops — the depth-8 kernel went from 210 VGPRs to 256 with 18-43 spills.)
            const bool rows_count = r >= cx.ra && r < cx.rb && r >= g.own_lo && r < g.own_hi;
            if (rows_count) acc[L1 == 2 ? (h - 1) / 2 : 0] += fabs(nv - old);
        }
        if (c) wk[sr] = nv; else wr[sr] = nv;
    }
    // ---- row f - HS is final: store it (dropped by the range check outside [ra, rb)) ----------
    {
        const int r = f - HS;
        const int sr = Win::slot(i, HS);
        const __amdgpu_buffer_rsrc_t ro = row_rsrc(cx.xout, g, r, r >= cx.ra && r < cx.rb);
        buf_store<COH>(wr[sr], ro, cx.st_r);
        buf_store<COH>(wk[sr], ro, cx.st_k);
    }
    // keep the machine scheduler from pulling later steps' work up across this point:
    // unconstrained it hoists all G steps together and spills the register window
    __builtin_amdgcn_sched_barrier(0);
}

template <int T>
__device__ __forceinline__ void fused_ctx_init(FusedCtx &cx, const double *__restrict__ xin, double *__restrict__ xout,
                                               const double *__restrict__ bb, const Geom &g, int sx, int ra, int rb,
                                               const unsigned char *__restrict__ mask)
{
    constexpr int HS = 2 * T;
    cx.xin = xin; cx.xout = xout; cx.bb = bb;
    const int lane = (int)(threadIdx.x & (kWave - 1));
    const int U = fused_useful_px(T);
    const int px0 = sx * U - fused_halo_px(T);          // first pixel column of the strip (even)
    cx.j = px0 / 2 + lane;                              // this lane's half-column (may be < 0)
    const bool col_ok = (cx.j >= 0) && (cx.j < g.pitch);
    const int ux0 = sx * U, ux1 = ux0 + U;              // pixel columns this strip stores
    cx.col_store = col_ok && (2 * cx.j >= ux0) && (2 * cx.j + 1 < ux1);
    cx.ld_r = col_ok ? (unsigned)cx.j * 8u : kLaneOut;
    cx.ld_k = col_ok ? (unsigned)(g.pitch + cx.j) * 8u : kLaneOut;
    cx.st_r = cx.col_store ? cx.ld_r : kLaneOut;
    cx.st_k = cx.col_store ? cx.ld_k : kLaneOut;
    cx.mask = mask;
    cx.ldm_r = col_ok ? (unsigned)cx.j : kLaneOut;
    cx.ldm_k = col_ok ? (unsigned)(g.pitch + cx.j) : kLaneOut;
    cx.col_interior = (px0 >= 1) && (px0 + 2 * kStripLanes <= g.W - 1);
    cx.has_first = (px0 <= 0) && (px0 + 2 * kStripLanes > 0);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int x = 2 * cx.j + p;
        cx.px_ok[p] = col_ok && x < g.W;
        cx.px_left[p] = x >= 1;
        cx.px_right[p] = x < g.W - 1;
        cx.px_first[p] = x == 0;
    }
    cx.ra = ra; cx.rb = rb;
    cx.m0 = max(ra - HS, 0);                            // rows this wave loads: [m0, m1)
    cx.m1 = min(rb + HS, g.local_rows);
}

// One wave: strip `sx`, rows [ra, rb) of channel data at xin/xout/b (already channel-offset).
// BORDERTILE = false: every pixel the wave can touch is ordinary: one straight-line loop of
// kStepFast trips.  BORDERTILE = true: each trip picks among the three bodies (force_border:
// debug, every trip takes kStepBorder).
template <int T, bool BORDERTILE, int L1, int UNR, int AN, bool MASKED = false, bool COH = false>
__device__ __forceinline__ void fused_wave(const double *__restrict__ xin, double *__restrict__ xout,
                                           const double *__restrict__ bb, const Geom &g, int sx,
                                           int ra, int rb, double (&acc)[AN], bool force_border = false,
                                           const unsigned char *__restrict__ mask = nullptr)
{
    static_assert(!(MASKED && BORDERTILE), "a Dirichlet-mask grid has no border tiles: everything outside is zero");
    using Win = FusedWindow<T, UNR>;
    constexpr int HS = Win::HS, G = Win::G, NT = Win::NT;
    FusedCtx cx;
    fused_ctx_init<T>(cx, xin, xout, bb, g, sx, ra, rb, mask);
    // the march starts on an even image row (y0 + base even) and advances G (even) rows per
    // trip, so the colour parity of every unrolled row update is a compile-time constant
    const int base = cx.m0 - ((g.y0 + cx.m0) & 1);
    const int f_end = rb - 1 + HS;                      // last step: row rb-1 gets half-sweep HS

    double wr[NT], wk[NT], br[NT], bk[NT];              // x red/black, b red/black per window row
    constexpr int NQ = MASKED ? NT : 1;
    QWord qr[NQ], qk[NQ];                               // MASKED: the factor 1/4 (an unknown) or 0 (a pixel fixed at zero), high word
    double land[G][4];                                  // rows in flight
    int landm[G][2];
#pragma unroll
    for (int s = 0; s < NT; ++s) wr[s] = wk[s] = br[s] = bk[s] = 0.0;
#pragma unroll
    for (int s = 0; s < NQ; ++s) qr[s] = qk[s] = 0;
#pragma unroll
    for (int i = 0; i < G; ++i) {
        fused_load_row<COH>(cx, g, base + i, land[i]);
        if (MASKED) fused_load_mask(cx, g, base + i, landm[i]);
    }
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const int s0 = Win::slot(i, 0);
        wr[s0] = land[i][0]; wk[s0] = land[i][1]; br[s0] = land[i][2] * 0.25; bk[s0] = land[i][3] * 0.25;
        if (MASKED) {
            qr[s0] = landm[i][0] ? kQuarterHi : 0;
            qk[s0] = landm[i][1] ? kQuarterHi : 0;
        }
    }

    if constexpr (kFusedLand == 2 && !BORDERTILE && L1 == 0 && !MASKED && T >= 8) {
        // Two trips of rows in flight instead of one: a row has 2G march steps to arrive.  The landing registers of
        // the two trips swap roles from trip to trip, so the trip loop is unrolled by two (PH = which pair lands now).
        double land2[G][4];                             // the other landing pair (`land` is pair 0)
        int landm2[G][2];
#pragma unroll
        for (int i = 0; i < G; ++i) {
            fused_load_row<COH>(cx, g, base + G + i, land2[i]);
            if (MASKED) fused_load_mask(cx, g, base + G + i, landm2[i]);
        }
#define CCP_FUSED_TRIP(LANDING, LANDINGM, ARRIVED, ARRIVEDM, FB)                                                              \
        {                                                                                                                     \
            _Pragma("unroll") for (int i = 0; i < G; ++i) {                                                                   \
                fused_load_row<COH>(cx, g, (FB) + 2 * G + i, LANDING[i]);                                                     \
                if (MASKED) fused_load_mask(cx, g, (FB) + 2 * G + i, LANDINGM[i]);                                            \
            }                                                                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                                \
            _Pragma("unroll") for (int i = 0; i < G; ++i)                                                                     \
                fused_step<T, kStepFast, L1, UNR, NT, AN, MASKED, NQ, COH>(wr, wk, br, bk, acc, cx, g, (FB) + i, i, qr, qk);   \
            _Pragma("unroll") for (int s = 0; s + G < NT; ++s) {                                                              \
                wr[s] = wr[s + G]; wk[s] = wk[s + G]; br[s] = br[s + G]; bk[s] = bk[s + G];                                   \
                if (MASKED) { qr[s] = qr[s + G]; qk[s] = qk[s + G]; }                                                         \
            }                                                                                                                 \
            _Pragma("unroll") for (int i = 0; i < G; ++i) {                                                                   \
                const int s0 = Win::slot(i, 0);                                                                               \
                wr[s0] = ARRIVED[i][0]; wk[s0] = ARRIVED[i][1]; br[s0] = ARRIVED[i][2] * 0.25; bk[s0] = ARRIVED[i][3] * 0.25;  \
                if (MASKED) { qr[s0] = ARRIVEDM[i][0] ? kQuarterHi : 0; qk[s0] = ARRIVEDM[i][1] ? kQuarterHi : 0; }                    \
            }                                                                                                                 \
        }
        for (int fb = base; fb <= f_end; fb += 2 * G) {
            CCP_FUSED_TRIP(land, landm, land2, landm2, fb)
            if (fb + G > f_end) break;
            CCP_FUSED_TRIP(land2, landm2, land, landm, fb + G)
        }
#undef CCP_FUSED_TRIP
        return;
    }
    for (int fb = base; fb <= f_end; fb += G) {
#pragma unroll
        for (int i = 0; i < G; ++i) {
            fused_load_row<COH>(cx, g, fb + G + i, land[i]);
            if (MASKED) fused_load_mask(cx, g, fb + G + i, landm[i]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!BORDERTILE) {
#pragma unroll
            for (int i = 0; i < G; ++i)
                fused_step<T, kStepFast, L1, UNR, NT, AN, MASKED, NQ, COH>(wr, wk, br, bk, acc, cx, g, fb + i, i, qr, qk);
        } else {
            // rows the trip updates that matter: fb-HS .. fb+G-2, clipped to the rows this wave holds
            const int r_first = max(fb - HS, cx.m0), r_last = min(fb + G - 2, cx.m1 - 1);
            const bool rows_plain = !force_border && (g.y0 + r_first >= 1) && (g.y0 + r_last <= g.H - 2);
            if (rows_plain && cx.col_interior) {
#pragma unroll
                for (int i = 0; i < G; ++i)
                    fused_step<T, kStepFast, L1, UNR, NT, AN, false, NQ, COH>(wr, wk, br, bk, acc, cx, g, fb + i, i, qr, qk);
            } else if (rows_plain) {
#pragma unroll
                for (int i = 0; i < G; ++i)
                    fused_step<T, kStepSide, L1, UNR, NT, AN, false, NQ, COH>(wr, wk, br, bk, acc, cx, g, fb + i, i, qr, qk);
            } else {
#pragma unroll
                for (int i = 0; i < G; ++i)
                    fused_step<T, kStepBorder, L1, UNR, NT, AN, false, NQ, COH>(wr, wk, br, bk, acc, cx, g, fb + i, i, qr, qk);
            }
        }
#pragma unroll
        for (int s = 0; s + G < NT; ++s) {
            wr[s] = wr[s + G]; wk[s] = wk[s + G]; br[s] = br[s + G]; bk[s] = bk[s + G];
            if (MASKED) {
                qr[s] = qr[s + G];
                qk[s] = qk[s + G];
            }
        }
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int s0 = Win::slot(i, 0);
            wr[s0] = land[i][0]; wk[s0] = land[i][1]; br[s0] = land[i][2] * 0.25; bk[s0] = land[i][3] * 0.25;
            if (MASKED) {
                qr[s0] = landm[i][0] ? kQuarterHi : 0;
                qk[s0] = landm[i][1] ? kQuarterHi : 0;
            }
        }
    }
    if (L1 != 0) {                                       // (fused_step: the lanes of columns another strip stores do not count)
#pragma unroll
        for (int t = 0; t < AN; ++t) acc[t] = cx.col_store ? acc[t] : 0.0;
    }
}

// The 2nd __launch_bounds__ argument (waves per SIMD) caps the register allocator at what the
// window of depth T needs anyway ((2T+5) rows x 4 values x 2 VGPRs + addresses), so the scheduler
// cannot trade occupancy for hoisted loads.  512 VGPRs per SIMD lane: 64 -> 8 waves, 102 -> 5,
// 128 -> 4, 168 -> 3, 256 -> 2.
__host__ __device__ constexpr int fused_waves_per_simd(int T, int L1 = 0)
{
    return L1 == 0 ? (T <= 2 ? 8 : T <= 3 ? 5 : T <= 5 ? 4 : T <= 7 ? 3 : 2)
                   : (T <= 1 ? 6 : T <= 3 ? 4 : T <= 5 ? 3 : 2);
}
__host__ __device__ constexpr int fused_border_waves_per_simd(int T, int L1 = 0) { return T <= 1 ? 4 : (T <= 4 ? 3 : (T <= (L1 == 2 ? 7 : 8) ? 2 : 1)); }

__device__ __forceinline__ bool fused_is_border_tile(const FusedParams &P, int chunk, int sx)
{
    return chunk < P.nb_top || chunk >= P.n_chunks - P.nb_bot || sx < P.ns_left || sx >= P.n_strips - P.ns_right;
}

template <int L1, int AN>
__device__ __forceinline__ void fused_write_partials(double (&acc)[AN], double *__restrict__ partial, int ch, double *scratch,
                                                     unsigned bx, unsigned by)
{
    if (L1 != 0) {
        // partial[((t*channels + ch)*gridDim.y + by)*gridDim.x + bx], t = 0 for L1 = 1
#pragma unroll
        for (int t = 0; t < AN; ++t) {
            const double total = block_sum(acc[t], scratch);
            if (threadIdx.x == 0)
                partial[(((long)t * gridDim.z + ch) * gridDim.y + by) * gridDim.x + bx] = total;
        }
    }
}

// Ordinary tiles.  grid = (ceil(n_strips / 4), n_chunks, channels); block = 256 threads = 4 waves =
// 4 adjacent strips of one chunk; waves of border tiles leave at once (k_fused_border runs them).
template <int T, int L1, int UNR, bool EDGE = false>
__global__ void __launch_bounds__(kBlock, fused_waves_per_simd(T, L1))
k_fused_sweep(FusedParams P)
{
    __shared__ double scratch[kBlock / kWave];
    // readfirstlane: tell the compiler the wave index is uniform, so strip/row addressing is SALU
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (!EDGE) fused_tile_coords(P, bx, by, bz);             // (an EDGE pass keeps dispatch order: its edge chunks go first)
    const int sx = (int)bx * (kBlock / kWave) + wave;
    const int ch = (int)bz;
    const int chunk = EDGE ? fused_chunk_of(P, (int)by) : (int)by;
    int ra, rb;
    fused_chunk_rows(P, chunk, ra, rb);
    constexpr int AN = L1 == 2 ? T : 1;
    double acc[AN];
#pragma unroll
    for (int t = 0; t < AN; ++t) acc[t] = 0.0;
    const bool run = (P.active == nullptr) || (P.active[ch] != 0);   // a converged channel is frozen
    if (run && sx < P.n_strips && ra < rb && !fused_is_border_tile(P, chunk, sx)) {
        const Geom &g = P.g;
        const long off = (long)ch * g.ch_stride;
        unsigned long long t0 = 0;
        fused_trace_begin(P, t0);
        fused_wave<T, false, L1, UNR, AN>(P.xin + off, P.xout + off, P.b + off, g, sx, ra, rb, acc);
        if (EDGE && fused_is_edge_chunk(P, chunk)) fused_signal_edge(P);
        fused_trace_end(P, t0, (((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (kBlock / kWave) + wave, chunk, sx, ch, 0);
    }
    fused_write_partials<L1, AN>(acc, P.partial, ch, scratch, bx, by);
}

// Dirichlet-mask grid: every tile is an ordinary tile (rows and columns outside the block read as zero
// through the buffer range check, and zero is what lies outside a Dirichlet region); a tile whose extended
// region holds no unknown at all leaves at once (its pixels are zero in both ping-pong buffers and stay so).
// tile_live: one byte per (chunk, strip), written by k_masked_tile_census for this tiling.
#ifndef CCP_MASKED_MAX_T
#define CCP_MASKED_MAX_T 8
#endif
constexpr int kMaskedMaxT = CCP_MASKED_MAX_T;   // the q window costs one VGPR per pixel kept (its high word)
#ifndef CCP_MASKED_MAX_CHECKED_T
#define CCP_MASKED_MAX_CHECKED_T 7
#endif
constexpr int kMaskedMaxCheckedT = CCP_MASKED_MAX_CHECKED_T;    // deepest masked pass that also reports the step of each of its sweeps
__host__ __device__ constexpr int masked_waves_per_simd(int T, int L1 = 0)
{
    return L1 == 0 ? (T <= 1 ? 6 : T <= 2 ? 5 : T <= 3 ? 3 : 2) : (T <= 1 ? 4 : T <= 2 ? 3 : 2);
}

// tile_rows (round 4): the rows of the tile that hold an unknown in the strip's columns, [2i] .. [2i + 1) — a tile on the
// rim of a region marches only those (rows without an unknown are zero in both ping-pong buffers and stay so, like the
// dead tiles; what the kept rows read of them is the zero that is there).
template <int T, int L1, int UNR>
__global__ void __launch_bounds__(kBlock, masked_waves_per_simd(T, L1))
k_fused_sweep_masked(FusedParams P, const unsigned char *__restrict__ tile_live, const int *__restrict__ tile_rows)
{
    __shared__ double scratch[kBlock / kWave];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    unsigned bx, by, bz;
    fused_tile_coords(P, bx, by, bz);
    const int sx = (int)bx * (kBlock / kWave) + wave;
    const int ch = (int)bz;
    const int chunk = (int)by;
    int ra, rb;
    fused_chunk_rows(P, chunk, ra, rb);
    constexpr int AN = L1 == 2 ? T : 1;
    double acc[AN];
#pragma unroll
    for (int t = 0; t < AN; ++t) acc[t] = 0.0;
    const bool run = (P.active == nullptr) || (P.active[ch] != 0);
    if (run && sx < P.n_strips && ra < rb && (tile_live == nullptr || tile_live[(long)chunk * P.n_strips + sx] != 0)) {
        if (tile_rows != nullptr) {
            ra = max(ra, tile_rows[2 * ((long)chunk * P.n_strips + sx)]);
            rb = min(rb, tile_rows[2 * ((long)chunk * P.n_strips + sx) + 1]);
        }
        if (ra < rb) {
            const Geom &g = P.g;
            const long off = (long)ch * g.ch_stride;
            fused_wave<T, false, L1, UNR, AN, true>(P.xin + off, P.xout + off, P.b + off, g, sx, ra, rb, acc, false, P.mask);
        }
    }
    fused_write_partials<L1, AN>(acc, P.partial, ch, scratch, bx, by);
}

// tile_live[chunk * n_strips + strip] = does the tile's extended region (its rows and columns plus the 2T
// halo) hold any unknown?  tile_rows[2 i], [2 i + 1]: first row, and one past the last, of the tile's OWN rows that hold
// an unknown anywhere in the strip's columns (halo columns included: conservative).  grid = (n_strips, n_chunks), one
// wave per tile.
__global__ void __launch_bounds__(kWave)
k_masked_tile_census(FusedParams P, int T, unsigned char *__restrict__ tile_live, int *__restrict__ tile_rows)
{
    const int sx = blockIdx.x, chunk = blockIdx.y, lane = threadIdx.x;
    int ra, rb;
    fused_chunk_rows(P, chunk, ra, rb);
    const Geom &g = P.g;
    const int HS = 2 * T;
    const int m0 = max(ra - HS, 0), m1 = min(rb + HS, g.local_rows);
    const int U = fused_useful_px(T);
    const int j = (sx * U - fused_halo_px(T)) / 2 + lane;
    int any = 0, first = INT_MAX, last = -1;
    if (j >= 0 && j < g.pitch)
        for (int q = m0; q < m1; ++q) {
            const int here = P.mask[(long)q * 2 * g.pitch + j] | P.mask[(long)q * 2 * g.pitch + g.pitch + j];
            any |= here;
            if (here && q >= ra && q < rb) {
                first = min(first, q);
                last = q;
            }
        }
    const unsigned long long vote = __ballot(any != 0);
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        first = min(first, __shfl_xor(first, o));
        last = max(last, __shfl_xor(last, o));
    }
    if (lane == 0) {
        tile_live[(long)chunk * P.n_strips + sx] = vote ? 1 : 0;
        tile_rows[2 * ((long)chunk * P.n_strips + sx)] = first & ~1;                 // (the march starts on an even row: keep the parity)
        tile_rows[2 * ((long)chunk * P.n_strips + sx) + 1] = last + 1;
    }
}

// Border tiles, compactly enumerated: first the top and bottom chunk rows (strips away from the
// left/right edge), then the left and right strips of every chunk, each cut into side_subs tiles.  grid = (ceil(n_border_tiles / 4), 1, channels).  Runs on
// a second stream beside k_fused_sweep: its waves take ~2.6x longer at T=8 when a whole trip is
// border work, and as part of one launch they used to be the tail every small grid waited for.
template <int T, int L1, int UNR, bool EDGE = false>
__global__ void __launch_bounds__(kBlock, fused_border_waves_per_simd(T, L1))
k_fused_border(FusedParams P, int force_border)
{
    __shared__ double scratch[kBlock / kWave];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int id = blockIdx.x * (kBlock / kWave) + wave;
    const int ch = blockIdx.z;
    constexpr int AN = L1 == 2 ? T : 1;
    double acc[AN];
#pragma unroll
    for (int t = 0; t < AN; ++t) acc[t] = 0.0;
    const int edge_chunks = min(P.nb_top + P.nb_bot, P.n_chunks);
    const int edge_strips = min(P.ns_left + P.ns_right, P.n_strips);
    const int inner = P.n_strips - edge_strips;                        // strips away from the left/right edge
    const int n_full = edge_chunks * inner;                            // top/bottom chunk rows, inner strips
    const int n_side = P.n_chunks * edge_strips * P.side_subs;         // edge strips of every chunk, cut in sub-tiles
    const bool run = (P.active == nullptr) || (P.active[ch] != 0);
    if (run && id < n_full + n_side) {
        int chunk, sx, ra, rb;
        if (id < n_full) {
            const int e = id / inner;
            sx = P.ns_left + id % inner;
            chunk = e < P.nb_top ? e : P.n_chunks - edge_chunks + e;
            fused_chunk_rows(P, chunk, ra, rb);
        } else {
            int k = id - n_full;
            const int sub = k % P.side_subs;
            k /= P.side_subs;
            const int e = k % edge_strips;
            chunk = EDGE ? fused_chunk_of(P, k / edge_strips) : k / edge_strips;
            sx = e < P.ns_left ? e : P.n_strips - edge_strips + e;
            int c0, c1;
            fused_chunk_rows(P, chunk, c0, c1);
            const int sr = (EDGE && fused_is_edge_chunk(P, chunk)) ? P.side_rows_edge : P.side_rows;
            ra = c0 + sub * sr;
            rb = min(ra + sr, c1);
        }
        if (ra < rb) {
            const Geom &g = P.g;
            const long off = (long)ch * g.ch_stride;
            unsigned long long t0 = 0;
            fused_trace_begin(P, t0);
            fused_wave<T, true, L1, UNR, AN>(P.xin + off, P.xout + off, P.b + off, g, sx, ra, rb, acc, force_border != 0);
            if (EDGE && fused_is_edge_chunk(P, chunk)) fused_signal_edge(P);
            // (the border launch's records follow the ordinary launch's: P.trace is offset by the host)
            fused_trace_end(P, t0, ((long)blockIdx.z * gridDim.x + blockIdx.x) * (kBlock / kWave) + wave, chunk, sx, ch, id < n_full ? 1 : 2);
        }
    }
    fused_write_partials<L1, AN>(acc, P.partial_border, ch, scratch, blockIdx.x, blockIdx.y);
}

// ---------------------------------------------------------------------------------------------
// Several passes of depth T in ONE launch (k_fused_multi): no drain, no fill and no idle chip between them.
//
// Between two dependent pass launches the chip idles ~18-26 us (profiles/r02_edge_pass_trace.json): the last tiles
// of a pass trickle out, the next launch starts from an empty chip.  That is 1.5 % of a 16384^2 pass but 8-14 % of a
// 4096^2 x 3 pass or of one 2048-row block of an 8-GPU run.  Here the tiles of ALL passes of a group are handed out
// by one ticket counter, pass-major (within a pass the slow border tiles first); a wave that takes tile (c, s) of
// pass q waits — polling fifteen counters — until the tiles (c-2..c+2, s-1..s+1) of pass q-1 are complete: they wrote
// everything its trapezoid reads, and they have read everything it overwrites (the passes ping-pong between two
// buffers).  A wave only ever waits for tickets smaller than its own, all of which are held by workgroups already
// started, so the grid always drains whatever the dispatch order (the ticket, not blockIdx, names the work).
//
// What one pass hands to the next travels through memory INSIDE the launch, between CUs of different XCDs whose L2s
// are not coherent: every x store is sc1 (write-through) and drained (s_waitcnt vmcnt(0)) before the wave counts its
// tile complete with an agent-scope atomic, and every x load is an sc1 load (COH in fused_wave) — the valid form of
// MI355X_MICROARCH.md's correctness boundaries.  b is read with plain loads (it does not change).
//
// All passes share the tiling of the first one (the widest row range); a later pass of a row block with ghost rows
// stores fewer rows per side (validity recedes 2T rows per pass) and simply clips its tiles' rows.  A tile cut into
// side sub-tiles is complete when all of them are.  Every spin is bounded (~1 s): a wave that gives up raises
// *error — the host turns that into CCP_ERR_STATE — and goes on, so a logic error costs a result, never the GPU.
constexpr int kMultiMaxPasses = 8;

struct FusedMultiParams {
    FusedParams P;                 // pass 0: xin -> xout with the group's tiling; odd passes run xout -> xin
    int n_passes;
    int st_lo[kMultiMaxPasses], st_hi[kMultiMaxPasses];   // rows pass q finalises and stores
    int gx, gy, bgx;               // workgroups of the ordinary tiles (gx strips-of-4 x gy chunks) and of the border tiles, per channel
    int channels;
    unsigned *__restrict__ ticket;
    unsigned *__restrict__ cells;  // [pass][channel][chunk][strip] waves that completed the tile
    unsigned *__restrict__ error;
    const unsigned char *__restrict__ tile_live;   // MASKED: which tiles hold any unknown (k_masked_tile_census); the others never run
};

// waves that make up tile (chunk, sx): 1, or the non-empty side sub-tiles of an edge strip
__device__ __forceinline__ unsigned fused_multi_expected(const FusedParams &P, int chunk, int sx)
{
    const bool side = sx < P.ns_left || sx >= P.n_strips - P.ns_right;
    if (!side) return 1u;
    int c0, c1;
    fused_chunk_rows(P, chunk, c0, c1);
    const int n = (c1 - c0 + P.side_rows - 1) / P.side_rows;
    return (unsigned)min(n, P.side_subs);
}

// MASKED (Dirichlet-mask grid): every tile is an ordinary one, tiles without an unknown in reach are complete without
// running (a waiter expects nothing from them).
template <int T, int UNR, bool MASKED = false>
__global__ void __launch_bounds__(kBlock, 2)
k_fused_multi(FusedMultiParams M)
{
    __shared__ unsigned s_ticket;
    if (threadIdx.x == 0) s_ticket = atomicAdd(M.ticket, 1u);
    __syncthreads();
    const FusedParams &P = M.P;
    const Geom &g = P.g;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int lane = (int)(threadIdx.x & (kWave - 1));
    const unsigned per_pass = (unsigned)(M.bgx + M.gx * M.gy) * (unsigned)M.channels;
    const unsigned tk = __builtin_amdgcn_readfirstlane(s_ticket);
    const int q = (int)(tk / per_pass);
    if (q >= M.n_passes) return;
    unsigned i = tk % per_pass;
    // ---- which tile -------------------------------------------------------------------------------------------
    int chunk = 0, sx = 0, ch = 0, ra = 0, rb = 0;
    bool border = false, have = false;
    const int edge_chunks = min(P.nb_top + P.nb_bot, P.n_chunks);
    const int edge_strips = min(P.ns_left + P.ns_right, P.n_strips);
    if (!MASKED && i < (unsigned)(M.bgx * M.channels)) {
        // a workgroup of k_fused_border's enumeration: top/bottom chunk rows (inner strips), then the side sub-tiles
        border = true;
        ch = (int)(i / (unsigned)M.bgx);
        const int id = (int)(i % (unsigned)M.bgx) * (kBlock / kWave) + wave;
        const int inner = P.n_strips - edge_strips;
        const int n_full = edge_chunks * inner;
        const int n_side = P.n_chunks * edge_strips * P.side_subs;
        if (id < n_full + n_side) {
            if (id < n_full) {
                const int e = id / inner;
                sx = P.ns_left + id % inner;
                chunk = e < P.nb_top ? e : P.n_chunks - edge_chunks + e;
                fused_chunk_rows(P, chunk, ra, rb);
            } else {
                int k = id - n_full;
                const int sub = k % P.side_subs;
                k /= P.side_subs;
                const int e = k % edge_strips;
                chunk = k / edge_strips;
                sx = e < P.ns_left ? e : P.n_strips - edge_strips + e;
                int c0, c1;
                fused_chunk_rows(P, chunk, c0, c1);
                ra = c0 + sub * P.side_rows;
                rb = min(ra + P.side_rows, c1);
            }
            have = ra < rb;
        }
    } else {
        i -= (unsigned)(M.bgx * M.channels);
        const unsigned per_ch = (unsigned)(M.gx * M.gy);
        ch = (int)(i / per_ch);
        const unsigned r = i % per_ch;
        chunk = (int)(r / (unsigned)M.gx);
        sx = (int)(r % (unsigned)M.gx) * (kBlock / kWave) + wave;
        fused_chunk_rows(P, chunk, ra, rb);
        have = sx < P.n_strips && ra < rb;
        if (have) have = MASKED ? M.tile_live[(long)chunk * P.n_strips + sx] != 0 : !fused_is_border_tile(P, chunk, sx);
    }
    if (!have) return;
    // ---- wait for the nine tiles of the previous pass around this one ------------------------------------------
    const long cells_per_pass = (long)M.channels * P.n_chunks * P.n_strips;
    bool gave_up = false;                                            // (uniform) the wait timed out: the inputs may not be final
    if (q > 0) {
        const unsigned *prev = M.cells + (long)(q - 1) * cells_per_pass + (long)ch * P.n_chunks * P.n_strips;
        // chunks c-2 .. c+2: the one chunk of a pass that may be shorter than the halo (the remainder of the rows) has
        // full-height neighbours, so two chunks either way always cover the 2T rows a trapezoid reaches
        const int dc = lane / 3 - 2, ds = lane % 3 - 1;
        const int c2 = chunk + dc, s2 = sx + ds;
        const bool mine = lane < 15 && c2 >= 0 && c2 < P.n_chunks && s2 >= 0 && s2 < P.n_strips;
        const unsigned need = !mine ? 0u : (MASKED ? (unsigned)(M.tile_live[(long)c2 * P.n_strips + s2] != 0) : fused_multi_expected(P, c2, s2));
        const unsigned *word = prev + (mine ? (long)c2 * P.n_strips + s2 : 0);
        bool ok = !mine;
        const unsigned long long t0 = wall_clock64();
        // The hand-off itself travels by ACCESS FORM, not by cache-wide fences: the producer's x stores are sc1 (write
        // through) and acknowledged (s_waitcnt) before its count, the consumer's x loads are sc1 (read through).  An
        // agent-scope acquire / release here — buffer_inv / buffer_wbl2 of a whole L2 per tile — was tried in round 4 as
        // the language-level statement of the same thing: the four-passes launch went from 0.95x to 1.7x the time of four
        // launches (profiles/r04_multi_fences.txt).  What the language DOES have to be told is not to move the tile's
        // loads above the poll loop: a wavefront-scope fence, which orders this wave's accesses and emits no cache
        // operation.
        while (!__all(ok)) {
            if (!ok) ok = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need;
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > 100000000ull) {              // 1 s of the 100 MHz clock: never in a correct run
                if (lane == 0) __hip_atomic_store(M.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                gave_up = true;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // ---- the tile -------------------------------------------------------------------------------------------
    // a wave that gave up stores nothing (the handle's error word is up: every later call returns CCP_ERR_STATE) but still
    // counts its tile, so that the waves behind it drain
    const int r0 = max(ra, M.st_lo[q]), r1 = gave_up ? r0 : min(rb, M.st_hi[q]);
    if (r0 < r1) {
        const long off = (long)ch * g.ch_stride;
        const double *xin = ((q & 1) ? P.xout : P.xin) + off;
        double *xout = ((q & 1) ? const_cast<double *>(P.xin) : P.xout) + off;
        double acc[1] = {0.0};
        if (MASKED) fused_wave<T, false, 0, UNR, 1, MASKED, true>(xin, xout, P.b + off, g, sx, r0, r1, acc, false, P.mask);
        else if (border) fused_wave<T, true, 0, UNR, 1, false, true>(xin, xout, P.b + off, g, sx, r0, r1, acc, false);
        else fused_wave<T, false, 0, UNR, 1, false, true>(xin, xout, P.b + off, g, sx, r0, r1, acc);
    }
    // ---- publish: the (write-through) stores acknowledged, then the count ----------------------------------------
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");           // (compiler ordering: the stores stay above the count)
    __builtin_amdgcn_s_waitcnt(0);                                   // every (write-through) store of the tile acknowledged
    if (lane == 0)
        __hip_atomic_fetch_add(M.cells + (long)q * cells_per_pass + ((long)ch * P.n_chunks + chunk) * P.n_strips + sx, 1u, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace ccp
