// ccp_grid_lex.hpp — the reference's OWN sweep order (index order, sparse-matrix.h:357-370) on the
// structured Poisson grid, in parallel and bit for bit.
//
// In the lexicographic sweep pixel (x,y) of iteration k reads the NEW values of (x,y-1) and (x-1,y)
// and the OLD values of (x+1,y) and (x,y+1).  With tau = x + y + 2k every one of those four lies on
// hyperplane tau - 1:  (x,y-1,k) and (x-1,y,k) trivially, (x+1,y,k-1) and (x,y+1,k-1) because
// x+y+1 + 2(k-1) = tau - 1.  All points of one hyperplane are therefore independent, any order that
// walks tau upwards reproduces the sequential sweep exactly, and one hyperplane holds pixels of many
// iterations at once — the pipeline never drains between sweeps.  Two engines walk it (ccp_grid.hip):
// k_lex_wg, the default — time-skewed strips (x' = x + 2t, y' = y + 2t), the T sweeps of a pass on the T waves of a
// workgroup, rows exchanged through LDS, all passes in one launch — and k_lex_plane (CCP_GS_LEX_MODE=planes), one
// launch per tau, anti-diagonal d = tau - 2k of every iteration k in flight (a launch writes diagonals of one parity
// and reads the other: no hazard inside a launch): the simple statement of the same order, kept as the independent
// engine the parity tests run beside the default.  (Rounds 2-3 also carried a wave-per-sweep strip engine, the skewed
// pass in one wave and a two-pixels-per-lane variant of k_lex_wg: all bit-identical, none faster — NOTES.md.)
//
// Layout: "diagonal-major" — diagonal d = x + y is row d of a (W+H-1) x P array, pixel at column x:
//   (x,y-1) -> [d-1][x]   (x-1,y) -> [d-1][x-1]   (x+1,y) -> [d+1][x+1]   (x,y+1) -> [d+1][x]
// every access of a launch is unit-stride along x.  Twice the memory of the image, none of the traffic.
//
// Arithmetic per pixel: classify()/gs_update() (ccp_grid_kernels.hpp) — the reference's accumulation
// order and its true division; interior pixels take (b + (((up+left)+right)+down)) * 0.25, the same bits.
#pragma once

#include "ccp_grid_kernels.hpp"
#include "ccp_grid_fused.hpp"      // lane_prev / lane_next (DPP)

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

// split colour planes <-> diagonal-major through a 64 x 64 tile in LDS, so that BOTH sides move in runs: image rows of the tile on the split side (two
// runs of 32 doubles, one per colour plane), pieces of anti-diagonals on the diagonal-major side (1 to 64 doubles).  (A
// thread per pixel scatters one side in single doubles a row pitch apart: 2.0 ms per array at 16384^2, 2.1 TB/s of
// useful traffic, three conversions per solve — rounds 1-3.)  Tile rows are 66 doubles apart: a diagonal walks the tile in steps of
// 65 doubles = 130 words, two banks on — conflict-free.  grid = (ceil(W/64), ceil(H/64), channels), block = 256.
constexpr int kLexCT = 64;
template <bool TO_DIAG>
__global__ void __launch_bounds__(kBlock)
k_lex_convert_tiled(double *__restrict__ split, double *__restrict__ diag, Geom g, LexGeom lg)
{
    __shared__ double tile[kLexCT][kLexCT + 2];
    const int x0 = blockIdx.x * kLexCT, y0 = blockIdx.y * kLexCT, ch = blockIdx.z;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    constexpr int kWaves = kBlock / kWave;
    double *__restrict__ sp = split + (long)ch * g.ch_stride;
    double *__restrict__ dg = diag + (long)ch * lg.plane;
    // Both phases are written as batches: every load of a wave goes out before the first value is used (a rolled
    // "load, wait, write to LDS" loop had ONE load in flight per wave: 1.1 ms per array at 16384^2 — ISA, round 4).
    // A pixel off the image clamps to the tile's first one: loaded, never stored.
    constexpr int kPer = kLexCT / kWaves;                    // image rows (anti-diagonal pairs) per wave
    auto split_at = [&](int r, bool &on) {
        const int x = x0 + lane, y = y0 + r;
        on = x < lg.W && y < lg.H;
        const int xc = on ? x : x0, yc = on ? y : y0;
        return row_off(g, yc, (xc + yc) & 1) + (xc >> 1);
    };
    // (the tile's anti-diagonal k holds columns 0 .. k, its anti-diagonal k + 64 columns k+1 .. 63: one wave moves both —
    // 64 full instructions a tile instead of 127 that are half empty on average)
    auto diag_at = [&](int k, int &yy, bool &on) {
        const int xx = lane, kk = xx <= k ? k : k + kLexCT;
        yy = kk - xx;
        on = x0 + xx < lg.W && y0 + yy < lg.H;
        return on ? (long)(x0 + y0 + kk) * lg.P + (x0 + xx) : (long)(x0 + y0) * lg.P + x0;
    };
    double v[kPer];
    if (TO_DIAG) {
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            bool on;
            v[i] = sp[split_at(wave + i * kWaves, on)];
        }
#pragma unroll
        for (int i = 0; i < kPer; ++i) tile[wave + i * kWaves][lane] = v[i];
    } else {
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            int yy;
            bool on;
            v[i] = dg[diag_at(wave + i * kWaves, yy, on)];
        }
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            int yy;
            bool on;
            (void)diag_at(wave + i * kWaves, yy, on);
            tile[yy][lane] = v[i];                           // (a pixel off the image: a slot nobody reads)
        }
    }
    __syncthreads();
    if (TO_DIAG) {
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            int yy;
            bool on;
            const long d = diag_at(wave + i * kWaves, yy, on);
            if (on) dg[d] = tile[yy][lane];
        }
    } else {
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            bool on;
            const long at = split_at(wave + i * kWaves, on);
            if (on) sp[at] = tile[wave + i * kWaves][lane];
        }
    }
}

// split colour planes -> diagonal-major with the tile cut the other way: 64 columns x 64 DIAGONALS (a sheared piece of the
// image, 127 image rows high), so that the WRITE side moves whole 512-byte runs of a diagonal row and the partial runs —
// pieces of image rows — are on the read side, where the neighbouring tile's re-read of a line comes from L2.  (With the
// square tile above the forward conversion wrote two partial runs per instruction: 1.07 ms per array at 16384^2 against
// 0.80 ms for the backward one, which reads them — kernel stats, round 4.)  Image row yb + j of the tile holds its columns
// 63-j .. 63 (j <= 63), row yb + j + 64 columns 0 .. 62-j: one wave instruction moves both.
// grid = (ceil(W/64), ceil((W+H-1)/64), channels), block = 256; tiles that miss the image leave at once.
__global__ void __launch_bounds__(kBlock)
k_lex_to_diag_sheared(const double *__restrict__ split, double *__restrict__ diag, Geom g, LexGeom lg)
{
    __shared__ double tile[kLexCT][kLexCT + 2];              // [diagonal][column]
    const int x0 = blockIdx.x * kLexCT, d0 = blockIdx.y * kLexCT, ch = blockIdx.z;
    const int yb = d0 - x0 - (kLexCT - 1);                   // the tile's first image row
    if (yb + 2 * kLexCT - 2 < 0 || yb > lg.H - 1) return;    // (uniform)
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    constexpr int kWaves = kBlock / kWave, kPer = kLexCT / kWaves;
    const double *__restrict__ sp = split + (long)ch * g.ch_stride;
    double *__restrict__ dg = diag + (long)ch * lg.plane;
    const int x = x0 + lane;
    auto row_of = [&](int k) { return lane >= kLexCT - 1 - k ? k : k + kLexCT; };
    double v[kPer];
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
        const int y = yb + row_of(wave + i * kWaves);
        const bool on = x < lg.W && y >= 0 && y < lg.H;
        const int xc = on ? x : x0, yc = on ? y : max(yb, 0) < lg.H ? max(yb, 0) : 0;
        v[i] = sp[row_off(g, yc, (xc + yc) & 1) + (xc >> 1)];
    }
#pragma unroll
    for (int i = 0; i < kPer; ++i) tile[row_of(wave + i * kWaves) - (kLexCT - 1) + lane][lane] = v[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
        const int dd = wave + i * kWaves, d = d0 + dd, y = d - x;
        if (x < lg.W && y >= 0 && y < lg.H) dg[(long)d * lg.P + x] = tile[dd][lane];
    }
}

// Dirichlet-mask grids (k_lex_wg<.., MASKED>): b in diagonal-major layout with a marker — a signalling NaN no
// arithmetic produces — wherever the pixel is not an unknown.  The sweep sees one array instead of two: a pixel
// whose b is the marker stays 0, every other pixel has the full row (diagonal 4; a neighbour that is not an
// unknown holds 0, and adding -1 * 0 to the row sum leaves its bits unchanged).
constexpr unsigned kLexFixedHi = 0x7ff4c0deu;
__device__ __forceinline__ double lex_fixed_marker() { return __hiloint2double((int)kLexFixedHi, 1); }
__device__ __forceinline__ bool lex_is_fixed(double b) { return (unsigned)__double2hiint(b) == kLexFixedHi; }

__global__ void __launch_bounds__(kBlock)
k_lex_convert_b_masked(const double *__restrict__ split, const unsigned char *__restrict__ mask, double *__restrict__ diag, Geom g, LexGeom lg)
{
    const int x = blockIdx.x * kBlock + threadIdx.x;
    const int y = blockIdx.y, ch = blockIdx.z;
    if (x >= lg.W) return;
    const long s1 = row_off(g, y, (x + y) & 1) + (x >> 1);
    diag[(long)ch * lg.plane + (long)(x + y) * lg.P + x] = mask[s1] ? split[(long)ch * g.ch_stride + s1] : lex_fixed_marker();
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

// ---------------------------------------------------------------------------------------------
// Hand-off through memory between workgroups of one launch (MI355X_MICROARCH.md, correctness boundaries: the per-XCD
// L2s are not coherent with each other): every access to x that another workgroup of the launch reads or wrote is an
// agent-scope (sc1) load or store — write-through, never served from a stale line — and a wave makes sure its stores
// are acknowledged before it publishes a progress counter.  Cache-wide release/acquire fences instead (buffer_wbl2 /
// buffer_inv per chunk and wave) were measured 4x slower.  Work items are handed out by a ticket counter: a workgroup
// only ever waits for tickets smaller than its own, which belong to workgroups that have already started — no
// assumption about dispatch order or co-residency, no deadlock.
constexpr unsigned kLexDone = 0xffffffffu;

__device__ __forceinline__ double lex_ld(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void lex_st(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Temporal blocking of the reference-order sweep: T sweeps per pass through memory.
// A red-black pass can recompute halos redundantly; the index-order sweep cannot (the left neighbour is the
// NEW value of the same sweep, which depends on the whole row to its left).  What it can do is SKEW: with
//     x' = x + 2t,  y' = y + 2t      (t = sweep inside the group of T)
// the four dependences of point (x', y', t)
//     (x-1, y, t) -> (x'-1, y',   t)      (x, y-1, t)   -> (x',   y'-1, t)
//     (x+1, y, t-1) -> (x'-1, y'-2, t-1)  (x, y+1, t-1) -> (x'-2, y'-1, t-1)
// all point to smaller x' (or equal x' and smaller y'): strips in x' depend on the strip to their LEFT only.
// A strip owns 64 skewed columns and marches down the skewed diagonals d' = x' + y'; lane l walks down skewed column
// xs0 + l, so `up` is the lane's own previous result and `left` the left lane's (DPP); sweep t takes `right` and `down`
// from sweep t-1's results three steps back, one and two lanes to the left.  Lanes 0 and 1 are GHOST lanes: they carry
// the results of the left strip's lanes 62 and 63 (2T doubles per step, written by that strip to `edges`, read back
// here) so every lane shift is uniform; a strip therefore advances 62 skewed columns.  Producers of strip (group, s):
// (group, s-1) through the same step; (group-1, s) and (group-1, s+1) through step + 1 + 4(T-1) (they wrote the x this
// group's sweep 0 reads).  Only sweep 0 reads x and only sweep T-1 writes it; b is read once per pass.
constexpr int kLexSkewCols = kWave - 2;
// Strip s of group k (the k-th pass of T sweeps of one launch) starts 2T columns further LEFT than strip s of group k-1:
// xs0 = kLexSkewCols s - 2 - 2T k.  The skew of a group (sweep t works 2t columns left of sweep 0) simply goes on into
// the next group, and so what sweep 0 of (k, s) reads — x columns xs0+2 .. xs0+64 — was written by (k-1, s) (its last
// sweep covers xs0+4 .. xs0+65 in the new strip's numbering) and, the first two columns, by (k-1, s-1), which is never
// behind (k-1, s).  With the strips of all groups in the same place (rounds 2-4) it was (k-1, s) AND its right neighbour
// (k-1, s+1), which starts a whole strip-to-strip stagger later: a group followed its predecessor 35 us behind instead
// of 17 (512^2 trace, round 4).
// Strips that hold no pixel of the image — on the left in late groups, on the right in early ones — get no ticket.
__host__ __device__ inline int lex_strip_first(int T, int k) { return (2 * T * k) / kLexSkewCols; }
__host__ __device__ inline int lex_strip_last(int W, int T, int k) { return (W - 1 + 2 * T * k + 2 * (T - 1)) / kLexSkewCols; }
__host__ __device__ inline bool lex_strip_exists(int W, int T, int k, int s)
{
    return k >= 0 && s >= lex_strip_first(T, k) && s <= lex_strip_last(W, T, k);
}
__host__ __device__ inline int lex_strip_count(int W, int T, int groups) { return lex_strip_last(W, T, groups - 1) + 1; }
// The edge values of a strip live in one of kLexEdgeSets buffers (group k: set k % kLexEdgeSets): (k, s) overwrites what
// (k - kLexEdgeSets, s) left for (k - kLexEdgeSets, s+1), and waits until that strip has taken it (lex_wg_body).
constexpr int kLexEdgeSets = 2;

// ---------------------------------------------------------------------------------------------
// The skewed pass with the T sweeps spread over the waves of a workgroup (k_lex_wg): a workgroup of T + 2 waves per
// strip:
//
//   waves 0 .. T-1   one sweep each.  Wave t reads its inputs from LDS only — the results of sweep t-1 three and
//                    steps back (one and two lanes to the left) from a ring of the last 4 result rows per
//                    sweep, b from a ring of 32 diagonal rows — and writes its result row into the ring.  No
//                    global memory operation, ~20 instructions per step.
//   wave T           the loader: everything the pass reads from memory, eight steps ahead in registers — the
//                    next row of b (read once per PASS: sweep t uses row d-4t at step d), the next row of x for
//                    sweep 0 (written into ring[0], so sweep 0 looks like every other sweep), the left strip's
//                    edge values for the ghost lanes (a batch per 8 steps, into spare columns of the b rows).
//                    It also watches the progress of the strips this one depends on, and so gates the others.
//   wave T+1         the storer: sweep T-1's row to x and the edge values of all sweeps, one step behind; it
//                    publishes the strip's progress.  Loads never queue behind stores (vmcnt is in order).
//
// The waves march in lock-step, one workgroup barrier per step (an isolated step of this shape costs 131 ns,
// tools/step_bench.hip).  Three bodies:
//   A  every real lane of the wave is an interior pixel (4 neighbours, diagonal 4);
//   B  every real lane has 1 <= y <= H-2 but the strip touches the left / right image border: the row of a
//      lane depends on its column alone and is classified once (the first and last strips must keep pace with
//      the rest — every strip waits on its left neighbour);
//   C  anything else (the first and last ~10 blocks of a strip): classify / gs_update at every step.
// All three read their inputs from the rings: no wave but the loader loads, no wave but the storer stores.
// grid = (min(G * S, resident workgroups), channels), block = (T + 2) * 64.  CHECK: partial[((group*T + t)*channels + ch)*partial_stride + s].
#ifndef CCP_LEX_SHIFT_DOWN
#define CCP_LEX_SHIFT_DOWN 1
#endif
constexpr bool kLexShiftDown = CCP_LEX_SHIFT_DOWN != 0;   // interior bodies of k_lex_wg: `down` from `right` by a lane shift
constexpr int kLexRing = 4;                        // result rows kept per sweep: written at step d, read at step d+3, free at d+4
constexpr int kLexBRows = 32;
// The ring of b rows in LDS carries kLexBMirror more rows: slot 32 + k is a copy of slot k (k < 4), kept by everybody who
// writes a row (lex_b_slot_mirror).  A compute wave reads the 8 rows of a block at slots sb .. sb + 7 with sb = (db - 4t) & 31
// a multiple of 4, i.e. up to slot 35: with the mirror the slot of step j is sb + j — a constant offset from the block's
// base address — instead of (sb + j) & 31, which cost every compute wave a scalar and, a scalar multiply and a vector add
// per step (round 4).
constexpr int kLexBMirror = 4;
__device__ __forceinline__ bool lex_b_slot_mirrored(int slot) { return slot < kLexBMirror; }
constexpr int kLexFrontRows = 64;                  // diagonal rows allocated BEFORE the first one (a strip that starts left of the image prefetches them; never used)
constexpr int kLexSlackRows = 320;                 // diagonal rows allocated beyond the last one (k_lex_wg prefetches past the image)
constexpr int kLexScratch = 32;                    // doubles per workgroup the storer of k_lex_wg may write to and nobody reads
constexpr int kLexStoresPerBlock = 8 * 2 + 1;      // k_lex_wg's storer: two stores per step (x row, edge values) + the publication, per 8-step block
constexpr int kLexPublishLagBlocks = 1;            // ... publishes the steps before block db - 8*1 ...
constexpr int kLexPublishVmcnt = 32;               // ... once at most this many of its stores are outstanding (< 2 blocks' worth)
constexpr int kLexWordStride = 32;                 // progress words of k_lex_wg: one per 128-byte line (the word a strip's
                                                   // storer writes is polled by its neighbours' loaders)

// Workgroup barrier that orders LDS traffic only: __syncthreads() would also drain the global loads a wave
// has in flight (the loader's prefetch ring) at every step.
__device__ __forceinline__ void lex_lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int T>
struct LexWgShape {
    static constexpr int kCols = kLexSkewCols + 2 * (T - 1);                 // image columns the T sweeps of a strip touch
    static constexpr int kGhost = (kCols + 1) / 2 * 2;                       // first of the 2T ghost columns of a row in LDS
    static constexpr int kRowW = kGhost + 2 * T;
};

// a / 3, correctly rounded, without the division sequence (body B does it at every step in the strip that
// holds column 0, and every other strip waits on that one).  y = RN(1/3) = (1/3)(1 - 2^-54); q0 = RN(a y) is
// within 1.5 ulp of a/3; r = a - 3 q0 is a small multiple of ulp(q0), exact in an fma; a/3 = q0 + r/3 exactly,
// and q0 + r y differs from it by |r/3| 2^-54.  a/3 is never closer than ulp/6 to the midpoint of two doubles
// (a - 3m is a non-zero multiple of ulp/2 for a midpoint m), so rounding q0 + r y — one rounding, in the fma —
// gives RN(a/3); r == 0 means q0 is the quotient itself (this keeps the sign of a zero).  Holds for every
// finite a, subnormals included (fp64 subnormals are not flushed); infinities and NaNs the caller divides.
// Checked on the host against the machine's division over every binade: tests/cpp/div3_check.cpp.
__device__ __forceinline__ double lex_div3(double a, bool &finite)
{
    const double y = 0x1.5555555555555p-2;
    finite = !__builtin_amdgcn_class(a, 0x207);               // not (NaN | +-inf)
    const double q0 = a * y;
    const double r = __builtin_fma(-3.0, q0, a);
    const double q1 = __builtin_fma(r, y, q0);
    return r == 0.0 ? q0 : q1;
}

// The left strip's edge values ("ghost values": its lanes 62 / 63 of every sweep and step, 2T doubles per step in `edges`)
// come in once per 8-step block, as one batch, into spare columns of the b rows in LDS: value (sweep g_t, step blk + g_k,
// edge g_e) goes to b row (blk + g_k - 4 g_t), column kGhost + 2 g_t + g_e, where that sweep's ghost lanes read it as
// their "b".  Compute wave 0 carries the batch beside its sweep (round 4; the loader did it before: it needed 82 VGPRs for
// an 80-VGPR budget, and in the persistent form of the kernel a spill landed inside its prefetch loop).  The batch of
// block blk + 8 is fetched behind the first barrier of block blk — the loader passes the block's gate, which vouches for
// the left strip's steps up to blk + 15, before it gets there — and written before the block's last barrier.
template <int T>
struct LexGhosts {
    static constexpr int kOps = (16 * T + kWave - 1) / kWave;               // 64-lane loads per batch
    static constexpr int kGhost = LexWgShape<T>::kGhost;
    const double *e_left;                                                    // nullptr: strip 0 (its ghost lanes lie off the image)
    int left_begin, left_end;
    bool on;                                                                 // this wave carries the batch (wave 0; uniform)
    double q[kOps];
    __device__ __forceinline__ bool lane_on(int lane, int k) const { return k * kWave + lane < 16 * T; }
    __device__ __forceinline__ int g_t(int lane, int k) const { return min((k * kWave + lane) >> 4, T - 1); }
    __device__ __forceinline__ int g_k(int lane) const { return (lane & 15) >> 1; }
    __device__ __forceinline__ int g_col(int lane, int k) const { return kGhost + 2 * g_t(lane, k) + (lane & 1); }
    // raw: whether the step exists is applied where the value is used — a select right behind the load would wait for it
    __device__ __forceinline__ void issue(int blk, int lane)
    {
        // (worked out from the lane at every use: hoisted out of the block loop, the per-lane addresses are two more
        // vector registers per load that live — and spill — across the step loops of every compute wave)
        asm volatile("" : "+v"(lane));
#pragma unroll
        for (int k = 0; k < kOps; ++k) {
            const int d = blk + g_k(lane);
            q[k] = e_left == nullptr ? 0.0
                                     : lex_ld(e_left + ((long)(min(max(d, left_begin), left_end) - left_begin) * T + g_t(lane, k)) * 2 + (lane & 1));
        }
    }
    __device__ __forceinline__ void write(double (*brow)[LexWgShape<T>::kRowW], int blk, int lane) const
    {
        asm volatile("" : "+v"(lane));
#pragma unroll
        for (int k = 0; k < kOps; ++k) {
            const int d = blk + g_k(lane);
            const bool valid = e_left != nullptr && d >= left_begin && d <= left_end;
            const int slot = (d - 4 * g_t(lane, k)) & (kLexBRows - 1);
            if (lane_on(lane, k)) {
                brow[slot][g_col(lane, k)] = valid ? q[k] : 0.0;
                if (lex_b_slot_mirrored(slot)) brow[kLexBRows + slot][g_col(lane, k)] = valid ? q[k] : 0.0;
            }
        }
    }
    // step j of block db
    __device__ __forceinline__ void step(double (*brow)[LexWgShape<T>::kRowW], int db, int j, int lane)
    {
        if (!on) return;
        if (j == 1) issue(db + 8, lane);
        if (j == 7) write(brow, db + 8, lane);
    }
};

// One block (8 steps from db) of compute wave t with the general body C: a lane's pixel and row are worked out
// at every step (classify / gs_update).  Inputs come from the rings like everywhere else.  A rolled loop: it runs
// for the ~20 blocks at the two ends of a strip only.
template <int T, bool CHECK>
__device__ __forceinline__ void lex_wg_general_block(double &h1, double &acc, double &old, double (*ring)[kLexRing][kWave],
                                                               double (*brow)[LexWgShape<T>::kRowW], Geom g, int W, int H, int t,
                                                               int lane, int db, int xp, LexGhosts<T> &gh)
{
    const bool ghost = lane < 2;
    const int lds2 = max(lane - 2, 0);
    const int col = ghost ? LexWgShape<T>::kGhost + 2 * t + lane : lane - 2 - 2 * t + 2 * (T - 1);
    const int xl = xp - 2 * t;
    const bool lane_on = !ghost && xl >= 0 && xl < W;
#pragma unroll 1
    for (int j = 0; j < 8; ++j) {
        const int d = db + j, y = d - xp - 2 * t;
        const double *in = &ring[t][(j + 1) & (kLexRing - 1)][lds2];
        const double down = in[0];
        const double right = in[1];
        const double vv = brow[(d - 4 * t) & (kLexBRows - 1)][col];
        const double up = h1;
        const double left = lane_prev(h1);
        double nv = ghost ? vv : 0.0;
        if (lane_on && y >= 0 && y < H) {
            const Stencil sc = classify(g, xl, y, y);
            if (sc.diag != 0) {
                (void)gs_update(sc, vv, up, left, right, down, nv);
                if (CHECK) acc += fabs(nv - old);
            }
        }
        old = down;
        ring[t + 1][j & (kLexRing - 1)][lane] = nv;
        h1 = nv;
        gh.step(brow, db, j, lane);
        lex_lds_barrier();
    }
}

// One block at the HEAD of a strip (EDGE 0: every pixel the wave's real lanes touch in these 8 steps lies on a row
// y <= H-2 — lanes come onto the image through row 0) or at its TAIL (EDGE 1: on a row y >= 1 — they leave through row
// H-1), for an image of at least 2 x 3 pixels.  The row of the matrix a pixel has then follows from its column (as in the
// inner blocks, lex_wg_compute: c_x0 / c_xl / c_off) and from ONE comparison of its y:
//   row 0      no `up`: diagonal 3 (left, right, down; at x = 0: right, down and the corner's + 1), at x = W-1 `left` alone
//   row H-1    `up` alone, diagonal 1; no row at x = W-1
// in gs_update()'s accumulation order, a / 3 by lex_div3.  A third of body C's vector instructions (~25 a step against
// ~60: classify() per pixel and step, the division sequence in diverged lanes) — and these blocks are what the start of
// the NEXT strip waits for: its first gate opens when this strip has done its first ~100 steps, most of them here
// (per-workgroup trace at 512^2: 36 us from strip to strip, 41 us from group to group, round 4).
template <int T, bool CHECK, int KIND, int EDGE>
__device__ __forceinline__ void lex_wg_edge_block(double &h1, double &acc, double &old, double (*ring)[kLexRing][kWave],
                                                  double (*brow)[LexWgShape<T>::kRowW], int H, int t, int lane, int db, int xp,
                                                  bool c_off, bool c_x0, bool c_xl, LexGhosts<T> &gh)
{
    const bool ghost = lane < 2;
    const int lds2 = max(lane - 2, 0);
    const int col = ghost ? LexWgShape<T>::kGhost + 2 * t + lane : lane - 2 - 2 * t + 2 * (T - 1);
    int y = db - xp - 2 * t;
#pragma unroll 1
    for (int j = 0; j < 8; ++j, ++y) {
        const double *in = &ring[t][(j + 1) & (kLexRing - 1)][lds2];
        const double down = in[0];
        const double right = in[1];
        const double vv = brow[(db + j - 4 * t) & (kLexBRows - 1)][col];
        const double up = h1;
        const double left = lane_prev(h1);
        double nv = (vv + (((up + left) + right) + down)) * 0.25;            // rows 1 .. H-2, by column: lex_wg_compute
        double a3 = vv + ((up + right) + down);                              // ... x = 0
        bool third = (KIND & 1) && c_x0;
        bool row = !c_off;
        if (EDGE == 0) {
            const bool top = y == 0;
            double at = vv + ((left + right) + down);
            if (KIND & 1) at = c_x0 ? vv + (right + down) : at;
            a3 = top ? at : a3;
            third = third || top;
            row = row && y >= 0;
        }
        if ((KIND & 1) || EDGE == 0) {
            bool finite;
            double q = lex_div3(a3, finite);
            if (__any(third && !finite)) {
                asm volatile("" ::: "memory");
                q = a3 / 3.0;
            }
            nv = third ? q : nv;
        }
        if (KIND & 2) nv = c_xl ? vv + left : nv;                            // x = W-1, rows 0 .. H-2: `left` alone
        if (EDGE == 1) {
            nv = y == H - 1 ? vv + up : nv;
            row = row && y < H && !((KIND & 2) && c_xl && y == H - 1);
        }
        nv = row ? nv : ghost ? vv : 0.0;
        if (CHECK) acc += row ? fabs(nv - old) : 0.0;
        old = down;
        ring[t + 1][j & (kLexRing - 1)][lane] = nv;
        h1 = nv;
        gh.step(brow, db, j, lane);
        lex_lds_barrier();
    }
}

// Blocks db0 .. db1 of compute wave t.  A block whose 8 steps have 1 <= y <= H-2 for every real lane of the wave
// (steps xs0+64+2t .. xs0+2t+H) takes body A (KIND 0) or B (1: the wave holds column 0 — only in strip 0, whose
// ghost lanes lie off the image; 2: it holds column W-1; 3: both, an image narrower than a strip): the row of a
// pixel then depends on its column alone — column 0 has no left neighbour (diagonal 3), column W-1 only its
// left one (diagonal 1), a 1-pixel-wide image and the lanes off the image have no row at all; those keep
// whatever the full-row formula gives, no row of the matrix reads them.  A group moves at the pace of its first
// strip (every strip waits on its left neighbour): body B is kept as short as body A allows.  The other blocks
// take body C.  ring[t] holds sweep t's INPUT rows (ring[0]: x, filled by the loader), ring[t+1] its results.
template <int T, bool CHECK, int KIND>
__device__ __forceinline__ void lex_wg_compute(double &h1, double &acc, double (*ring)[kLexRing][kWave],
                                               double (*brow)[LexWgShape<T>::kRowW], Geom g, int W, int H, int t, int lane,
                                               int db0, int db1, int xs0, bool lane_on, Stencil st_b, LexGhosts<T> &gh)
{
    const bool ghost = lane < 2;
    const int lds2 = max(lane - 2, 0), lds1 = max(lane - 1, 0);
    const int col = ghost ? LexWgShape<T>::kGhost + 2 * t + lane : lane - 2 - 2 * t + 2 * (T - 1);   // of a b row in LDS
    const bool c_off = !lane_on || st_b.diag == 0;           // (st_b: classify() of this lane's column at an interior y)
    const bool c_x0 = !c_off && !st_b.left, c_xl = !c_off && !st_b.right;
    const bool wrote = KIND == 0 ? !ghost : !c_off;
    const int in_lo = xs0 + 64 + 2 * t, in_hi = xs0 + 2 * t + H;
    double old = 0.0;
    for (int db = db0; db <= db1; db += 8) {
        if (db < in_lo || db + 7 > in_hi) {
            // rows of the real lanes (2 .. 63) in these 8 steps: y = d - (xs0 + lane) - 2t
            const int y_lo = db - (xs0 + 63) - 2 * t, y_hi = db + 7 - (xs0 + 2) - 2 * t;
            if (W >= 2 && H >= 3 && y_hi <= H - 2)
                lex_wg_edge_block<T, CHECK, KIND, 0>(h1, acc, old, ring, brow, H, t, lane, db, xs0 + lane, c_off, c_x0, c_xl, gh);
            else if (W >= 2 && H >= 3 && y_lo >= 1)
                lex_wg_edge_block<T, CHECK, KIND, 1>(h1, acc, old, ring, brow, H, t, lane, db, xs0 + lane, c_off, c_x0, c_xl, gh);
            else
                lex_wg_general_block<T, CHECK>(h1, acc, old, ring, brow, g, W, H, t, lane, db, xs0 + lane, gh);
            continue;
        }
        const int sb = (db - 4 * t) & (kLexBRows - 1);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // `right` is sweep t-1's value one lane to the left, `down` the one two lanes to the left: one LDS read and a
            // lane shift instead of two adjacent doubles per lane (half the LDS bytes of the step's biggest read; the
            // ghost lanes 0 and 1 never use either)
            const double right = ring[t][(j + 1) & (kLexRing - 1)][lds1];
            const double down = kLexShiftDown ? lane_prev(right) : ring[t][(j + 1) & (kLexRing - 1)][lds2];
            const double vv = brow[sb + j][col];                            // (sb + j <= 35: the mirror rows)
            const double up = h1;
            const double left = lane_prev(h1);
            double nv = (vv + (((up + left) + right) + down)) * 0.25;        // (sparse-matrix.h:361-376 on a full row)
            if (KIND & 1) {
                const double a = vv + ((up + right) + down);
                bool finite;
                double q = lex_div3(a, finite);
                if (__any(c_x0 && !finite)) {
                    asm volatile("" ::: "memory");                           // (keeps the division out of the common path)
                    q = a / 3.0;
                }
                nv = c_x0 ? q : nv;
            }
            if (KIND & 2) nv = c_xl ? vv + left : nv;
            if (KIND != 1) nv = ghost ? vv : nv;
            if (CHECK) acc += wrote ? fabs(nv - old) : 0.0;                  // old: this pixel in the previous sweep
            old = down;                                                      // ... is what was `down` one step (one row) earlier
            ring[t + 1][j & (kLexRing - 1)][lane] = nv;
            h1 = nv;
            gh.step(brow, db, j, lane);
            lex_lds_barrier();
        }
    }
}

// Dirichlet-mask grid: one body for every block.  An unknown has the full row (diagonal 4); a pixel whose b is
// the marker — not an unknown, or off the canvas (the loader puts the marker there) — stays 0.
template <int T, bool CHECK>
__device__ __forceinline__ void lex_wg_compute_masked(double &h1, double &acc, double (*ring)[kLexRing][kWave],
                                                      double (*brow)[LexWgShape<T>::kRowW], int t, int lane, int db0, int db1, LexGhosts<T> &gh)
{
    const bool ghost = lane < 2;
    const int lds2 = max(lane - 2, 0), lds1 = max(lane - 1, 0);
    const int col = ghost ? LexWgShape<T>::kGhost + 2 * t + lane : lane - 2 - 2 * t + 2 * (T - 1);
    double old = 0.0;
    for (int db = db0; db <= db1; db += 8) {
        const int sb = (db - 4 * t) & (kLexBRows - 1);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const double right = ring[t][(j + 1) & (kLexRing - 1)][lds1];                  // (one LDS read and a lane shift: lex_wg_compute)
            const double down = kLexShiftDown ? lane_prev(right) : ring[t][(j + 1) & (kLexRing - 1)][lds2];
            const double vv = brow[sb + j][col];
            const double up = h1;
            const double left = lane_prev(h1);
            double nv = (vv + (((up + left) + right) + down)) * 0.25;
            nv = lex_is_fixed(vv) ? 0.0 : nv;
            nv = ghost ? vv : nv;
            if (CHECK) acc += ghost ? 0.0 : fabs(nv - old);
            old = down;
            ring[t + 1][j & (kLexRing - 1)][lane] = nv;
            h1 = nv;
            gh.step(brow, db, j, lane);
            lex_lds_barrier();
        }
    }
}

// A sweep count that is not a multiple of T ends in a group whose last waves have nothing to do: wave t >= t_last hands
// sweep t-1's value of its pixel on unchanged — it is what was `down` one step earlier (lex_wg_compute) — so that the
// whole count runs as ONE pipeline of depth-T groups (round 4; before, the remainder went out as launches of depth 4, 2
// and 1, each paying the full ramp over the strips again: 0.3 ms of the 1.3 ms of 100 sweeps on a 512 x 512 grid).
// Ghost lanes take the left strip's value of the same (passed-through) sweep as always; the rings, the edge values
// and the storer see nothing unusual.
template <int T>
__device__ __forceinline__ void lex_wg_pass_through(double (*ring)[kLexRing][kWave], double (*brow)[LexWgShape<T>::kRowW], int t, int lane,
                                                    int db0, int db1, LexGhosts<T> &gh)
{
    const bool ghost = lane < 2;
    const int lds2 = max(lane - 2, 0);
    const int col = LexWgShape<T>::kGhost + 2 * t + (lane & 1);              // (the ghost lanes' column of a b row)
    double old = 0.0;
    for (int db = db0; db <= db1; db += 8) {
        const int sb = (db - 4 * t) & (kLexBRows - 1);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const double down = ring[t][(j + 1) & (kLexRing - 1)][lds2];
            const double vv = brow[sb + j][col];
            const double nv = ghost ? vv : old;
            old = down;
            ring[t + 1][j & (kLexRing - 1)][lane] = nv;
            gh.step(brow, db, j, lane);
            lex_lds_barrier();
        }
    }
}

// What the loader and the storer know about the strip.
struct LexWgStrip {
    const unsigned *words;               // the progress words of this launch (uniform)
    unsigned watch;                      // lanes 0..2 of the loader: index of the word to watch (own word: nothing to wait for)
    int need_off;                        // ... which has to reach block start + need_off
    unsigned known;
    unsigned *mine;                      // this strip's progress word: steps < value are complete and visible
};

__device__ __forceinline__ void lex_wg_gate(LexWgStrip &st, int db)
{
    const int need = db + st.need_off;
    bool ok = (int)min(st.known, 0x7fffffffu) >= need;
    while (!__all(ok)) {
        if (!ok) {
            st.known = __hip_atomic_load(st.words + st.watch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = (int)min(st.known, 0x7fffffffu) >= need;
        }
        if (!__all(ok)) __builtin_amdgcn_s_sleep(16);        // ~1000 cycles: the word is being written by the strip polled
    }
}

// The loader's side of blocks db0 .. db1 (the whole strip): the b rows, the x rows of sweep 0, and the gate — it is the
// wave that watches the strips this one depends on and so holds the others back at the step's barrier.  Rows and columns
// outside the arrays are clamped: what such a load fetches is never used, and no load is conditional.  Every slot of the
// register rings is refilled only after its old contents have been used (a load issued while the old value is live lands
// in another register and costs a copy and a full drain at the loop's back edge).
//   bp, xq     b and x of this channel (diagonal-major: element (row r, column c) at r*P + c)
//   cb         the leftmost image column any sweep of the strip touches (b row in LDS: column c - cb)
//   MASKED     Dirichlet-mask grid: what lies off the canvas is made "not an unknown" (b: the marker) holding 0 (x)
//              as it goes into LDS — there, not behind the load, where a select would wait for the load.
// (The left strip's edge values are compute wave 0's job since round 4: LexGhosts.)
template <int T, bool MASKED>
__device__ __forceinline__ void lex_wg_load(LexWgStrip &st, double (*ring)[kLexRing][kWave], double (*brow)[LexWgShape<T>::kRowW], int lane,
                                            int db0, int db1, const double *bp, const double *xq, long P, int n_diag, int W, int H, int cb,
                                            int xs0, unsigned long long *tr)
{
    constexpr int kCols = LexWgShape<T>::kCols;
    static_assert(kCols % 2 == 0 && kCols / 2 <= kWave && LexWgShape<T>::kRowW % 2 == 0, "a b row travels as pairs of columns");
    // A b row of the strip (kCols = 62 + 2(T - 1) columns from column cb, which is even) travels as PAIRS of columns: lane l
    // brings columns cb + 2l, cb + 2l + 1 in one 16-byte load and puts them into LDS in one 16-byte write (rows are 16-byte
    // aligned on both sides: P and kRowW are even).  One load and one write per step instead of two and two (64 + 12
    // columns; round 4).  A pair that starts left of the image is moved to column 0 and one that runs past the row's end
    // reads the padding or the next row — allocated memory either way, values never used.
    const bool b_lane = lane < kCols / 2;
    const unsigned c_b = (unsigned)max(cb + 2 * min(lane, kCols / 2 - 1), 0);
    const unsigned c_x = (unsigned)min(max(xs0 + 2 + min(lane, kWave - 2), 0), W - 1);   // as sweep 0's lanes 2.. read x: one and two places to their left
    auto b_row = [&](int r) { return bp + (long)min(max(r, 0), n_diag - 1) * P; };
    auto x_row = [&](int r) { return xq + (long)min(max(r, 0), n_diag - 1) * P; };
    auto b_pair = [&](const double *row) { return *reinterpret_cast<const double2 *>(row + c_b); };
    // MASKED: is (diagonal row r, this lane's column) a pixel of the canvas?
    const int k_b = cb + 2 * lane, k_x = xs0 + 2 + lane;
    auto on_canvas = [&](int r, int c) { return c >= 0 && c < W && (unsigned)(r - c) < (unsigned)H; };
    auto b_in = [&](double v, int r, int c) { return (!MASKED || on_canvas(r, c)) ? v : lex_fixed_marker(); };
    auto x_in = [&](double v, int r, int c) { return (!MASKED || (lane < kWave - 1 && on_canvas(r, c))) ? v : 0.0; };
    auto put_b = [&](int slot, double2 v, int r) {
        if (b_lane) {
            const double2 w = make_double2(b_in(v.x, r, k_b), b_in(v.y, r, k_b + 1));
            *reinterpret_cast<double2 *>(&brow[slot][2 * lane]) = w;
            if (lex_b_slot_mirrored(slot)) *reinterpret_cast<double2 *>(&brow[kLexBRows + slot][2 * lane]) = w;   // (uniform)
        }
    };
    lex_wg_gate(st, db0);
    if (tr && lane == 0) tr[1] = wall_clock64();
    lex_lds_barrier();                                                       // (wave 0 fetches the first ghost batch behind the gate)
    // what the first steps read before the rings are rolling: x rows db0, db0+1, db0+2 (the b rows up to db0: by all
    // waves, in the kernel)
    // (all nineteen loads go out before the first is waited for: one trip to memory between the gate and the strip's first
    // step, not two — the next strip's gate waits for this strip's first ~100 steps and for everything in front of them)
    double x3[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) x3[q] = lex_ld(x_row(db0 + q) + c_x);
    double2 qb[8];
    double qx[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        qb[j] = b_pair(b_row(db0 + 1 + j));
        qx[j] = lex_ld(x_row(db0 + 3 + j) + c_x);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) ring[0][(db0 + q) & (kLexRing - 1)][lane] = x_in(x3[q], db0 + q, k_x);
    const double *rb = bp + (long)(db0 + 9) * P, *rx = xq + (long)(db0 + 11) * P;   // (uniform row bases: scalar registers)
    lex_lds_barrier();                                                       // (every wave of the workgroup comes here)
    for (int db = db0; db <= db1; db += 8) {
        if (db > db0) lex_wg_gate(st, db);
        // issued here, looked at after the block's last step: no loaded value but the prefetch slots lives across
        // the loop's back edge
        const unsigned polled = __hip_atomic_load(st.words + st.watch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // The rows fetched from here on exist: b row d+9 >= -55 (a strip's first block starts at -64 or later: the
        // leftmost strip of a group reaches at most 63 columns beyond the image; kLexFrontRows), and
        // the arrays carry kLexSlackRows rows beyond the last diagonal for the prefetches that run past the image
        // at the end of the rightmost strips (never used).  Running pointers, nothing to clamp.
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            put_b((db + j + 1) & (kLexBRows - 1), qb[j], db + j + 1);           // b row d + 1
            ring[0][(j + 3) & (kLexRing - 1)][lane] = x_in(qx[j], db + j + 3, k_x);       // x row d + 3: read by sweep 0 at steps d+2, d+3
            asm volatile("" ::: "memory");
            qb[j] = b_pair(rb);                                              // b row d + 9: plain loads (b does not change) — with
            qx[j] = lex_ld(rx + c_x);                                        // sc1 here two workgroups sharing a CU run at half speed; x row d + 11
            rb += P;
            rx += P;
            lex_lds_barrier();
        }
        st.known = max(st.known, polled);
    }
}

// The storer's side of blocks db0 .. db1: after the barrier of step d, sweep T-1's row of that step goes to x
// (the pixels that have a row) and the 2T edge values of the step to the edge buffer.
template <int T, bool MASKED>
__device__ __forceinline__ void lex_wg_store(LexWgStrip &st, double (*ring)[kLexRing][kWave], Geom g, int W, int H, int lane, int db0, int db1,
                                             double *xq, long P, int xs0, int d_begin, int d_end, double *e_mine, double *scratch,
                                             bool strip_interior, Stencil st_b)
{
    constexpr int t = T - 1;
    const int xp = xs0 + lane, xl = xp - 2 * t;
    const bool ghost = lane < 2;
    const bool lane_on = !ghost && xl >= 0 && xl < W;
    const int in_lo = xs0 + 64 + 2 * t, in_hi = xs0 + 2 * t + H;
    const int e_t = min(lane >> 1, T - 1) + 1, e_lane = kWave - 2 + (lane & 1);
    lex_lds_barrier();                                                       // (the loader is through the strip's first gate)
    lex_lds_barrier();                                                       // (the priming barrier)
    for (int db = db0; db <= db1; db += 8) {
        // Exactly two stores per step, whatever the masks say (lane 0 and the edge lanes fall back to scratch slots
        // of this workgroup's own): the publication below counts on it.
        if (MASKED) {
            // Dirichlet-mask grid: every pixel of the canvas is stored (one that is not an unknown holds 0 and gets 0)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int d = db + j;
                lex_lds_barrier();
                const double v = ring[T][j & (kLexRing - 1)][lane];
                const double ev = ring[e_t][j & (kLexRing - 1)][e_lane];
                const bool wrote = lane_on && (unsigned)(d - xp - 2 * t) < (unsigned)H;
                if (wrote || lane == 0) lex_st(wrote ? xq + (long)(d - 4 * t) * P + xl : scratch + j, v);
                if (lane < 2 * T) lex_st((d >= d_begin && d <= d_end) ? e_mine + ((long)(d - d_begin) * T) * 2 + lane : scratch + 8 + lane, ev);
            }
        } else if (db >= in_lo && db + 7 <= in_hi) {
            // every real lane has 1 <= y <= H-2: which pixels have a row does not change from step to step.  In an
            // interior strip the ghost lanes store too: they carry the left strip's results of the same sweep for
            // exactly these pixels, so it is the value already there.
            const bool wrote = strip_interior || (lane_on && st_b.diag != 0);
            double *px = xq + (long)(db - 4 * t) * P + xl;
            double *pe = e_mine + ((long)(db - d_begin) * T) * 2 + lane;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                lex_lds_barrier();                                           // step d is in the rings
                const double v = ring[T][j & (kLexRing - 1)][lane];
                const double ev = ring[e_t][j & (kLexRing - 1)][e_lane];
                if (strip_interior) lex_st(px, v);
                else if (wrote || lane == 0) lex_st(wrote ? px : scratch + j, v);
                if (lane < 2 * T) lex_st(pe, ev);
                px += P;
                pe += 2 * T;
            }
        } else {
#pragma unroll 1
            for (int j = 0; j < 8; ++j) {
                const int d = db + j, y = d - xp - 2 * t;
                lex_lds_barrier();
                const double v = ring[T][j & (kLexRing - 1)][lane];
                const double ev = ring[e_t][j & (kLexRing - 1)][e_lane];
                const bool wrote = lane_on && y >= 0 && y < H && classify(g, xl, y, y).diag != 0;
                if (wrote || lane == 0) lex_st(wrote ? xq + (long)(d - 4 * t) * P + xl : scratch + j, v);
                if (lane < 2 * T) lex_st((d >= d_begin && d <= d_end) ? e_mine + ((long)(d - d_begin) * T) * 2 + lane : scratch + 8 + lane, ev);
            }
        }
        // Stores complete in issue order, and a block issues kLexStoresPerBlock of them (two per step + this
        // publication): once at most the kLexPublishVmcnt youngest are outstanding — this block's and all but one of the
        // block before — every store of the blocks before those two has been acknowledged.  Publish their steps
        // (everything before block db - 8 kLexPublishLagBlocks), without draining.  (Round 4: one block of lag, not two —
        // a store is two blocks = 4 us old by then; eight steps less between a strip and the strips that wait for it.)
        // REQUIRED CODEGEN: exactly kLexStoresPerBlock vector-memory instructions per block in this wave — the three
        // branches above each issue their two stores per step unconditionally (masked lanes go to scratch slots) so
        // that the compiler can neither merge nor drop one; a build whose storer issued FEWER would publish early.
        // The static_assert ties the count waited for to the lag published; tests/test_gpu_lex.py checks the bits.
        static_assert(kLexPublishVmcnt < (kLexPublishLagBlocks + 1) * kLexStoresPerBlock && kLexPublishVmcnt <= 63,
                      "the stores of block db - 8 (kLexPublishLagBlocks + 1) must all lie outside the youngest kLexPublishVmcnt");
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kLexPublishVmcnt) : "memory");
        if (lane == 0)
            __hip_atomic_store(st.mine, (unsigned)max(db - 8 * kLexPublishLagBlocks, 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Two workgroups share a CU only if 6 of their waves fit one SIMD (a workgroup's T + 2 = 10 waves go 3, 3, 2, 2): at
// most 80 VGPRs; and above ~53 KB of LDS per workgroup the second one is not placed (traced block start times,
// whatever the occupancy query says).
// Everything a launch is given.  The kernels take it BY VALUE and never touch the parameter: lex_wg_body reads the fields
// back from the kernel-argument segment at the top of every strip, through a pointer the compiler cannot see through, so
// that no argument is live across the persistent loop's back edge (kept live there, the arguments cost 67 scalar and 28
// vector spills at the 80-VGPR budget two workgroups per CU need).
struct LexWgArgs {
    double *xd;
    const double *bd;
    Geom g;
    LexGeom lg;
    int G, S;                       // groups; strip slots per group (lex_strip_count: not all of them exist in every group)
    unsigned n_tickets;             // strips that exist, over all groups
    unsigned *progress, *ticket;
    const unsigned *order;
    double *edges;
    long edge_steps;
    unsigned active_mask;
    double *partial;
    long partial_stride;
    unsigned long long *trace;
    int t_last;                     // sweeps the LAST group really performs (1 .. T): its waves t >= t_last pass their input through
};

template <int T, bool CHECK, bool MASKED>
__device__ __forceinline__ void lex_wg_body()
{
    static_assert(kLexRing == 4 && T >= 1, "the unrolled step index mod 4 is the ring slot");
    static_assert(4 * (T - 1) + 4 <= kLexBRows, "a b row stays in LDS from step r-1 to step r+4(T-1)");
    constexpr int kRowW = LexWgShape<T>::kRowW;
    __shared__ double ring[T + 1][kLexRing][kWave];
    __shared__ double brow[kLexBRows + kLexBMirror][kRowW];
    __shared__ unsigned s_ticket;
    const int ch = blockIdx.y;
    // Nothing of a strip may be carried around the persistent loop in VECTOR registers: the loader is held to 80 VGPRs
    // (two workgroups per CU), and whatever the compiler hoists out of the loop — the thread index, the zero the rings
    // are cleared with — is spilled inside the loader's prefetch loop, where a scratch load queues behind the prefetches
    // (measured: +24 % per step).  The wave's index in the workgroup lives in a scalar register, the lane comes from
    // mbcnt, the arguments are re-read from the kernel-argument segment per strip.
    const int wv0 = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    // PERSISTENT workgroups: the launch holds as many workgroups as the chip has room for (two per CU) and each takes
    // strip after strip from the ticket counter.  One workgroup per strip — 4,240 of them at 16384^2 and 128 sweeps —
    // left 40 % of the 512 slots empty in steady state (per-workgroup trace, profiles/r04_lex_trace.jsonl: 320 alive of
    // 512, starts in bursts of four): the hardware hands workgroups to its shader engines in launch order, and one that
    // has to wait for room on its engine holds back those behind it.  A ticket is still only ever waited for by larger
    // tickets, and its holder never waits for a larger one: no deadlock, whatever the residency.
    for (;;) {
    auto kernarg = __builtin_amdgcn_kernarg_segment_ptr();   // (constant address space: scalar loads)
    asm volatile("" : "+s"(kernarg));                        // (reloaded per strip, not hoisted: LexWgArgs)
    int wv = wv0;                                            // 0..T-1: sweeps, T: loader, T+1: storer
    asm volatile("" : "+s"(wv));
    int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(lane));
    const int tid = wv * kWave + lane;
    const int t = min(wv, T - 1);
    const LexWgArgs *ap = (const LexWgArgs *)kernarg;
    double *const xd = ap->xd;
    const double *const bd = ap->bd;
    const Geom g = ap->g;
    const LexGeom lg = ap->lg;
    const int G = ap->G, S = ap->S;
    const unsigned n_tickets = ap->n_tickets;
    unsigned *const progress = ap->progress, *const ticket = ap->ticket;
    const unsigned *const order = ap->order;
    double *const edges = ap->edges;
    const long edge_steps = ap->edge_steps;
    double *const partial = ap->partial;
    const long partial_stride = ap->partial_stride;
    unsigned long long *const trace = ap->trace;
    if (!((ap->active_mask >> ch) & 1u)) return;
    __syncthreads();                                         // the previous strip of this workgroup has left LDS
    if (tid == 0) s_ticket = atomicAdd(&ticket[ch], 1u);
    {
        double zero = 0.0;
        asm volatile("" : "+v"(zero));                       // (made here, dead after the clearing: not a loop invariant)
        if (wv <= T) {
#pragma unroll
            for (int q = 0; q < kLexRing; ++q) ring[wv][q][lane] = zero;
        }
        for (int i = tid; i < (kLexBRows + kLexBMirror) * kRowW; i += (T + 2) * kWave) (&brow[0][0])[i] = zero;
    }
    __syncthreads();
    const unsigned my_ticket = s_ticket;
    if (my_ticket >= n_tickets) break;                       // (uniform)
    const unsigned tk = order[my_ticket];                    // (group, strip) in wavefront order
    const int grp = (int)(tk / (unsigned)S), s = (int)(tk % (unsigned)S);
    // CCP_GS_TRACE_FILE (diagnostics): per strip — ticket taken, first gate passed, last step done, where it ran
    unsigned long long *tr = trace ? trace + 4 * ((long)ch * G * S + my_ticket) : nullptr;
    if (tr && tid == 0) {
        tr[0] = wall_clock64();
        tr[3] = (unsigned long long)(__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) & 0xffffffu)         // HW_REG_HW_ID
                | ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 24)                 // HW_REG_XCC_ID[3:0]
                | ((unsigned long long)tk << 32);
    }
    const int HS = lg.H + 2 * (T - 1);
    const int xs0 = kLexSkewCols * s - 2 - 2 * T * grp;      // (lex_strip_first: where the strips of group grp lie)
    const bool has_left = lex_strip_exists(lg.W, T, grp, s - 1);
    const int xl = xs0 + lane - 2 * t;                       // this lane's image column
    const bool lane_on = lane >= 2 && xl >= 0 && xl < lg.W;
    const int d_begin = xs0, d_end = xs0 + (kWave - 1) + HS - 1;
    const int db0 = d_begin & ~7, db1 = d_end & ~7;          // first and last block (floor to a multiple of 8, also when negative)
    const long plane = (long)ch * lg.plane;
    double *e_set = edges + (long)(grp % kLexEdgeSets) * gridDim.y * S * edge_steps * (2 * T);
    double *e_mine = e_set + ((long)ch * S + s) * edge_steps * (2 * T);
    const double *e_left = has_left ? e_set + ((long)ch * S + s - 1) * edge_steps * (2 * T) : nullptr;
    const int left_begin = xs0 - kLexSkewCols, left_end = left_begin + (kWave - 1) + HS - 1;
    const int cb = xs0 + 2 - 2 * (T - 1);                    // the leftmost image column any sweep of the strip touches

    // One progress word per strip, written by the storer.  The loader watches: lane 0 the left strip (edge
    // values), lane 1 the strip of the previous group sweep 0 reads x from, lane 2 the last reader of the edge buffer.
    LexWgStrip st;
    const unsigned my_word = (unsigned)((((long)ch * G + grp) * S + s) * kLexWordStride);
    st.words = progress;
    st.mine = progress + my_word;
    st.watch = my_word;
    st.need_off = INT_MIN / 2;
    st.known = 0;
    if (lane == 0 && has_left) {
        st.watch = my_word - kLexWordStride;
        st.need_off = 16;                                    // before block [db, db+7]: the ghost batch of block db+8
    }
    if (lane == 1 && lex_strip_exists(lg.W, T, grp - 1, s)) {
        // (k-1, s): the x columns sweep 0 reads (but for the first two: (k-1, s-1)'s, which is never behind (k-1, s) —
        // a strip passes a block's gate only when its left neighbour has PUBLISHED 16 steps beyond it).  What this strip
        // WRITES to x (columns xs0+2-2(T-1) .. xs0+63-2(T-1), row d-4(T-1) at step d) was read by those same two strips.
        st.watch = (unsigned)((((long)ch * G + grp - 1) * S + s) * kLexWordStride);
        st.need_off = 19 + 4 * (T - 1);                      // x row db+18, written by sweep T-1 at step db+18+4(T-1)
    }
    if (lane == 2 && lex_strip_exists(lg.W, T, grp - kLexEdgeSets, s + 1)) {
        // (k-kLexEdgeSets, s+1) took its ghost values from the buffer this strip's edge values go to: the entries of
        // this strip's block db are that strip's block db + 2T kLexEdgeSets (its xs0 lies that much further right),
        // fetched during its block db + 2T kLexEdgeSets - 8 (LexGhosts) — taken once that block is complete.  (It started
        // kLexEdgeSets group lags minus one strip lag before this strip AND a strip further along the diagonals: it is
        // ~120 steps past that point when this strip starts; the watch is the guarantee.)
        st.watch = (unsigned)((((long)ch * G + grp - kLexEdgeSets) * S + s + 1) * kLexWordStride);
        st.need_off = 2 * T * kLexEdgeSets + 8;
    }

    {   // b rows db0 - 4(T-1) .. db0 into the ring, a few per wave: sweep t reads row d - 4t at step d (the loader
        // brings row d + 1 at step d)
        constexpr int kCols = LexWgShape<T>::kCols, kPrime = 4 * (T - 1) + 1, kPer = (kPrime + T + 1) / (T + 2);
        const int c0 = min(max(cb + lane, 0), lg.W - 1), c1 = min(max(cb + kWave + min(lane, max(kCols - kWave - 1, 0)), 0), lg.W - 1);
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
            const int r = db0 - (kPrime - 1) + wv * kPer + q;
            if (r <= db0) {                                              // (uniform)
                const double *row = bd + plane + (long)min(max(r, 0), lg.n_diag - 1) * lg.P;
                const bool in0 = !MASKED || (cb + lane >= 0 && cb + lane < lg.W && (unsigned)(r - cb - lane) < (unsigned)lg.H);
                const bool in1 = !MASKED || (cb + kWave + lane < lg.W && (unsigned)(r - cb - kWave - lane) < (unsigned)lg.H);
                const int slot = r & (kLexBRows - 1);
                const double v0 = in0 ? row[c0] : lex_fixed_marker(), v1 = in1 ? row[c1] : lex_fixed_marker();
                if (kCols >= kWave || lane < kCols) brow[slot][lane] = v0;
                if (kCols > kWave && lane < kCols - kWave) brow[slot][kWave + lane] = v1;
                if (lex_b_slot_mirrored(slot)) {
                    if (kCols >= kWave || lane < kCols) brow[kLexBRows + slot][lane] = v0;
                    if (kCols > kWave && lane < kCols - kWave) brow[kLexBRows + slot][kWave + lane] = v1;
                }
            }
        }
    }
    const bool strip_interior = has_left && xs0 + 2 - 2 * t >= 1 && xs0 + 63 - 2 * t <= lg.W - 2;
    const Stencil st_b = classify(g, lane_on ? xl : 0, 1, 1);                // (meaningful where H >= 3: inner blocks only)
    if (wv < T) {
        double h1 = 0.0, acc = 0.0;
        // which borders this wave's columns xs0+2-2t .. xs0+63-2t hold
        const bool has_x0 = xs0 + 2 - 2 * t <= 0 && xs0 + 63 - 2 * t >= 0;
        const bool has_xl = xs0 + 63 - 2 * t >= lg.W - 1;                // (column W-1, or nothing on the image at all)
        LexGhosts<T> gh;
        gh.e_left = e_left;
        gh.left_begin = left_begin;
        gh.left_end = left_end;
        gh.on = wv == 0;
        lex_lds_barrier();                                                   // (the loader is through the strip's first gate)
        if (gh.on) {                                                         // the ghost values of block db0
            gh.issue(db0, lane);
            gh.write(brow, db0, lane);
        }
        lex_lds_barrier();                                                   // (the priming barrier)
        if (grp == G - 1 && t >= ap->t_last) lex_wg_pass_through<T>(ring, brow, t, lane, db0, db1, gh);   // (checked kernels: its step sum stays 0, a sweep nobody looks at)
        else if (MASKED) lex_wg_compute_masked<T, CHECK>(h1, acc, ring, brow, t, lane, db0, db1, gh);
        else if (strip_interior) lex_wg_compute<T, CHECK, 0>(h1, acc, ring, brow, g, lg.W, lg.H, t, lane, db0, db1, xs0, lane_on, st_b, gh);
        else if (has_x0 && !has_xl) lex_wg_compute<T, CHECK, 1>(h1, acc, ring, brow, g, lg.W, lg.H, t, lane, db0, db1, xs0, lane_on, st_b, gh);
        else if (!has_x0) lex_wg_compute<T, CHECK, 2>(h1, acc, ring, brow, g, lg.W, lg.H, t, lane, db0, db1, xs0, lane_on, st_b, gh);
        else lex_wg_compute<T, CHECK, 3>(h1, acc, ring, brow, g, lg.W, lg.H, t, lane, db0, db1, xs0, lane_on, st_b, gh);
        if (CHECK) {
            const double total = wave_sum(acc);
            if (lane == 0) partial[(((long)grp * T + t) * gridDim.y + ch) * partial_stride + s] = total;
        }
    } else if (wv == T) {
        lex_wg_load<T, MASKED>(st, ring, brow, lane, db0, db1, bd + plane, xd + plane, lg.P, lg.n_diag, lg.W, lg.H, cb, xs0, tr);
    } else {
        // (scratch: kLexScratch doubles per resident workgroup behind the edge values of all strips)
        double *scratch = edges + (long)kLexEdgeSets * gridDim.y * S * edge_steps * (2 * T) + ((long)ch * gridDim.x + blockIdx.x) * kLexScratch;
        lex_wg_store<T, MASKED>(st, ring, g, lg.W, lg.H, lane, db0, db1, xd + plane, lg.P, xs0, d_begin, d_end, e_mine, scratch, strip_interior, st_b);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                   // compiler ordering only
    __builtin_amdgcn_s_waitcnt(0);                                           // this wave's (write-through) stores acknowledged
    lex_lds_barrier();
    if (wv == T + 1 && lane == 0) __hip_atomic_store(st.mine, kLexDone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tr && tid == 0) tr[2] = wall_clock64();
    }
}

// The kernels.  Two workgroups share a CU only if 6 of their waves fit one SIMD (a workgroup's T + 2 = 10 waves go
// 3, 3, 2, 2): at most 80 VGPRs — the loader, which decides the count, needs 82, and is held to 80 (2 spills) where
// sharing pays: +5..11 % on plain grids from 4096^2 up.  (Above ~53 KB of LDS per workgroup the second one is not
// placed at all — traced block start times — whatever the occupancy query says; the rings take 42 KB.)  The
// Dirichlet-mask variant carried three more values in its loader and was built for 5 waves per SIMD (one workgroup per
// CU) until the loader's b rows travelled as pairs (round 4): it fits the 80 now, without spills.
template <int T, bool CHECK>
__global__ void __launch_bounds__((T + 2) * kWave) __attribute__((amdgpu_waves_per_eu(6, 8)))
k_lex_wg(LexWgArgs)
{
    lex_wg_body<T, CHECK, false>();
}

template <int T, bool CHECK>
__global__ void __launch_bounds__((T + 2) * kWave) __attribute__((amdgpu_waves_per_eu(6, 8)))
k_lex_wg_masked(LexWgArgs)
{
    lex_wg_body<T, CHECK, true>();
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
