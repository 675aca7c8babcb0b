// ccp_grid_lex2.hpp — k_lex_wg2: the time-skewed reference-order pass (ccp_grid_lex.hpp, k_lex_wg) with TWO pixels per
// lane and step.
//
// k_lex_wg is bound by its lock-step step, not by bytes: one workgroup barrier, one LDS read pair, one LDS write and two
// DPP moves per pixel update (0.42 wave-instructions per update, SIMDs 54 % instruction-active).
// Here a lane owns two ADJACENT skewed columns, x'_A = xs0 + 2 lane and x'_B = x'_A + 1, and at step d updates the two
// pixels of anti-diagonal d in them:  A = (x'_A, d - x'_A),  B = (x'_B, d - x'_B)  (B lies one row above A).  Of the new
// values they need
//     up(A) = A of the step before        left(A) = B of the lane to the left, the step before   (one DPP move pair)
//     up(B) = B of the step before        left(B) = A of THIS lane, the step before               (a register)
// and of the previous sweep's values, three steps back (the same skew as k_lex_wg: x' = x + 2t, y' = y + 2t):
//     down(A), right(A) = the pair (A, B) of the lane to the left      right(B) = A of this lane      down(B) = right(A).
// Per two updates: one barrier, one DPP pair, three LDS reads (the neighbour's pair, the own A, the b pair) and one
// 16-byte LDS write — half the synchronisation, LDS instructions and lane traffic per update.  A strip is 126 skewed
// columns (lane 0 is the one ghost lane: the left strip's lane 63), so a grid has half the strips and half the start lag
// of the strip pipeline.  The rings are twice as large (77 KB of LDS per workgroup: one workgroup per CU).
//
// Everything else is k_lex_wg's: T compute waves (one sweep each) + a loader + a storer per workgroup, tickets in
// wavefront order, one progress word per strip with the counted-store publication, the edge buffer (the last lane's pair
// of every sweep and step for the strip to the right), sc1 accesses to x, plain loads of b, the diagonal-major layout.
// Three bodies per 8-step block: A every real lane interior; B rows 1..H-2 with image column 0 and / or W-1 in the wave
// (classified once per lane and pixel); C anything else (classify / gs_update per step).  MASKED: Dirichlet-mask grids
// (one body; a pixel whose b is the marker stays 0).  Arithmetic per pixel is the reference's, bit for bit.
#pragma once

#include "ccp_grid_lex.hpp"

namespace ccp {

constexpr int kLex2Cols = 2 * (kWave - 1);         // skewed columns a strip advances (lane 0: ghost)

struct Pair {
    double a, b;
};

template <int T>
struct Lex2Shape {
    static constexpr int kCols = kLex2Cols + 2 * (T - 1);    // image columns the T sweeps of a strip touch (even)
    static constexpr int kGhost = kCols;                     // first of the 2T ghost columns of a b row in LDS
    static constexpr int kRowW = kGhost + 2 * T;             // (even: every pair is 16-byte aligned)
};

// Pairs of x travel as ONE 16-byte sc1 buffer access per lane (the row is the buffer: an offset of kLaneOut is beyond
// it, so a lane that must not take part reads zeros / has its store dropped by the range check — no branch, and the
// storer issues exactly its two stores per step whatever the masks say).
typedef unsigned int lex2_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t lex2_row_rsrc(const double *row, long row_doubles)
{
    return __builtin_amdgcn_make_buffer_rsrc((void *)row, 0, (int)(row_doubles * 8), 0x00020000);
}
__device__ __forceinline__ Pair lex2_ld_pair_sc1(const double *row, long row_doubles, unsigned byte_off)
{
    const lex2_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(lex2_row_rsrc(row, row_doubles), (int)byte_off, 0, kAuxSc1);
    return __builtin_bit_cast(Pair, v);
}
__device__ __forceinline__ void lex2_st_pair_sc1(Pair v, double *row, long row_doubles, unsigned byte_off)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(lex2_u32x4, v), lex2_row_rsrc(row, row_doubles), (int)byte_off, 0, kAuxSc1);
}

// One block (8 steps from db) of compute wave t, general body: pixel and row worked out at every step.
template <int T, bool CHECK>
__device__ __forceinline__ void lex2_general_block(Pair &h1, double &acc, Pair &old, Pair (*ring)[kLexRing][kWave],
                                                   const double (*brow)[Lex2Shape<T>::kRowW], Geom g, int W, int H, int t, int lane,
                                                   int db, int xs0)
{
    const bool ghost = lane == 0;
    const int nb = max(lane - 1, 0);
    const int col = ghost ? Lex2Shape<T>::kGhost + 2 * t : 2 * (lane - 1 - t + T - 1);
    const int xa = xs0 + 2 * lane - 2 * t, xb = xa + 1;                // image columns of A and B
#pragma unroll 1
    for (int j = 0; j < 8; ++j) {
        const int d = db + j;
        const int ya = d - (xs0 + 2 * lane) - 2 * t, yb = ya - 1;
        const Pair in = ring[t][(j + 1) & (kLexRing - 1)][nb];          // the left lane's pair, three steps back
        const double own_a = ring[t][(j + 1) & (kLexRing - 1)][lane].a;
        const Pair vv = *reinterpret_cast<const Pair *>(&brow[(d - 4 * t) & (kLexBRows - 1)][col]);
        const double left_a = lane_prev(h1.b);
        // a pixel off the image holds 0; a pixel whose row is empty (a_ii = 0: skipped, sparse-matrix.h:361-363) keeps the
        // value it had in the previous sweep — pairs are stored whole, so what the ring holds for it goes to memory
        Pair nv;
        nv.a = ghost ? vv.a : 0.0;
        nv.b = ghost ? vv.b : 0.0;
        if (!ghost && xa >= 0 && xa < W && ya >= 0 && ya < H) {
            const Stencil sc = classify(g, xa, ya, ya);
            nv.a = old.a;
            if (sc.diag != 0) {
                (void)gs_update(sc, vv.a, h1.a, left_a, in.b, in.a, nv.a);
                if (CHECK) acc += fabs(nv.a - old.a);
            }
        }
        if (!ghost && xb >= 0 && xb < W && yb >= 0 && yb < H) {
            const Stencil sc = classify(g, xb, yb, yb);
            nv.b = old.b;
            if (sc.diag != 0) {
                (void)gs_update(sc, vv.b, h1.b, h1.a, own_a, in.b, nv.b);
                if (CHECK) acc += fabs(nv.b - old.b);
            }
        }
        old.a = in.a;                                                   // this pixel in the previous sweep = `down` one step earlier
        old.b = in.b;
        ring[t + 1][j & (kLexRing - 1)][lane] = nv;
        h1 = nv;
        lex_lds_barrier();
    }
}

// Blocks db0 .. db1 of compute wave t.  KIND 0: every real lane interior in the inner blocks; bit 0: the wave holds image
// column 0 (strip 0 only), bit 1: it holds column W-1.  MASKED: Dirichlet-mask grid, one body for every block.
template <int T, bool CHECK, int KIND, bool MASKED>
__device__ __forceinline__ void lex2_compute(Pair &h1, double &acc, Pair (*ring)[kLexRing][kWave],
                                             const double (*brow)[Lex2Shape<T>::kRowW], Geom g, int W, int H, int t, int lane, int db0,
                                             int db1, int xs0)
{
    const bool ghost = lane == 0;
    const int nb = max(lane - 1, 0);
    const int col = ghost ? Lex2Shape<T>::kGhost + 2 * t : 2 * (lane - 1 - t + T - 1);
    const int xa = xs0 + 2 * lane - 2 * t, xb = xa + 1;
    // what a pixel's column says about its row where 1 <= y <= H-2 (classify() at an interior y)
    const bool a_on = !ghost && xa >= 0 && xa < W, b_on = !ghost && xb >= 0 && xb < W;
    const Stencil sa = classify(g, a_on ? xa : 0, 1, 1), sb = classify(g, b_on ? xb : 0, 1, 1);
    const bool a_off = !a_on || sa.diag == 0, b_off = !b_on || sb.diag == 0;
    const bool a_x0 = !a_off && !sa.left, a_xl = !a_off && !sa.right;
    const bool b_xl = !b_off && !sb.right;                               // (B is never column 0: its column is odd)
    const bool a_wrote = KIND == 0 ? !ghost : !a_off, b_wrote = KIND == 0 ? !ghost : !b_off;
    // inner blocks: every real pixel of the wave has 1 <= y <= H-2 at all 8 steps.  A: y = d - x'_A - 2t, B: one less;
    // x'_A ranges over xs0+2 .. xs0+126
    const int in_lo = xs0 + 2 * (kWave - 1) + 2 + 2 * t, in_hi = xs0 + 2 + 2 * t + H - 2;
    Pair old;
    old.a = old.b = 0.0;
    for (int db = db0; db <= db1; db += 8) {
        if (!MASKED && (db < in_lo || db + 7 > in_hi)) {
            lex2_general_block<T, CHECK>(h1, acc, old, ring, brow, g, W, H, t, lane, db, xs0);
            continue;
        }
        const int sb8 = (db - 4 * t) & (kLexBRows - 1);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const Pair in = ring[t][(j + 1) & (kLexRing - 1)][nb];
            const double own_a = ring[t][(j + 1) & (kLexRing - 1)][lane].a;
            const Pair vv = *reinterpret_cast<const Pair *>(&brow[(sb8 + j) & (kLexBRows - 1)][col]);
            const double up_a = h1.a, up_b = h1.b;
            const double left_a = lane_prev(h1.b), left_b = h1.a;
            const double right_a = in.b, down_a = in.a, right_b = own_a, down_b = in.b;
            Pair nv;
            nv.a = (vv.a + (((up_a + left_a) + right_a) + down_a)) * 0.25;      // (sparse-matrix.h:361-376 on a full row)
            nv.b = (vv.b + (((up_b + left_b) + right_b) + down_b)) * 0.25;
            if (MASKED) {
                nv.a = lex_is_fixed(vv.a) ? 0.0 : nv.a;
                nv.b = lex_is_fixed(vv.b) ? 0.0 : nv.b;
            } else {
                if (KIND & 1) {
                    // column 0: no left neighbour, diagonal 3 (only ever pixel A of a lane: column 0 is even and xs0 is even)
                    const double a3 = vv.a + ((up_a + right_a) + down_a);
                    bool finite;
                    double q = lex_div3(a3, finite);
                    if (__any(a_x0 && !finite)) {
                        asm volatile("" ::: "memory");                           // (keeps the division out of the common path)
                        q = a3 / 3.0;
                    }
                    nv.a = a_x0 ? q : nv.a;
                }
                if (KIND & 2) {
                    nv.a = a_xl ? vv.a + left_a : nv.a;                          // column W-1: only the left neighbour, diagonal 1
                    nv.b = b_xl ? vv.b + left_b : nv.b;
                }
                if (KIND != 0) {
                    // a column without a row here (off the image; the only column of a 1-pixel-wide image) keeps the
                    // previous sweep's value: pairs are stored whole, so the ring's value goes to memory
                    nv.a = a_off ? old.a : nv.a;
                    nv.b = b_off ? old.b : nv.b;
                }
            }
            nv.a = ghost ? vv.a : nv.a;
            nv.b = ghost ? vv.b : nv.b;
            if (CHECK) {
                acc += (MASKED ? !ghost : a_wrote) ? fabs(nv.a - old.a) : 0.0;
                acc += (MASKED ? !ghost : b_wrote) ? fabs(nv.b - old.b) : 0.0;
            }
            old.a = down_a;
            old.b = down_b;
            ring[t + 1][j & (kLexRing - 1)][lane] = nv;
            h1 = nv;
            lex_lds_barrier();
        }
    }
}

// lex_wg_gate with a bound: a wait that lasts seconds means a broken dependence — the kernel then runs on (wrong
// results, which every parity test catches) instead of holding the card.
__device__ __forceinline__ void lex2_gate(LexWgStrip &st, int db)
{
    const int need = db + st.need_off;
    bool ok = (int)min(st.known, 0x7fffffffu) >= need;
    if (__all(ok)) return;
    const unsigned long long t0 = wall_clock64();
    while (!__all(ok)) {
        if (!ok) {
            st.known = __hip_atomic_load(st.words + st.watch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = (int)min(st.known, 0x7fffffffu) >= need;
        }
        if (__all(ok)) break;
        __builtin_amdgcn_s_sleep(16);
        if (wall_clock64() - t0 > 400000000ull) break;               // 4 s of the 100 MHz clock
    }
}

// The loader's side of blocks db0 .. db1.
//   b row d+1 -> brow[(d+1) & 31][0 .. kCols)   (plain 16-byte loads, 9 steps ahead in registers)
//   x row d+3 -> ring[0][(d+3) & 3][m] = the pair of columns xs0 + 2m + 2, xs0 + 2m + 3   (sc1, 11 steps ahead)
//   the left strip's edge pairs of block db+8 -> the ghost columns of the b rows, one batch per block
// Rows outside the arrays are clamped and columns beyond a row read as zero: what such a load brings is never used.
template <int T, bool MASKED>
__device__ __forceinline__ void lex2_load(LexWgStrip &st, Pair (*ring)[kLexRing][kWave], double (*brow)[Lex2Shape<T>::kRowW], int lane, int db0,
                                          int db1, const double *bp, const double *xq, long P, int n_diag, int W, int H, int cb, int xs0,
                                          const double *e_left, long e_left_doubles, int left_begin, int left_end)
{
    constexpr int kCols = Lex2Shape<T>::kCols, kGhost = Lex2Shape<T>::kGhost;
    constexpr int kPairs = kCols / 2;                                        // pairs of a b row (T = 8: 70, T = 1: 63)
    constexpr int kExtra = kPairs > kWave ? kPairs - kWave : 0;              // ... beyond the first 64
    static_assert(kExtra <= kWave, "a b row is at most two pair loads wide");
    static_assert(8 * T <= kWave, "the ghost batch of a block is one pair per lane");
    const int k_b0 = cb + 2 * lane, k_b1 = cb + 2 * (kWave + min(lane, max(kExtra - 1, 0))), k_x = xs0 + 2 * lane + 2;   // first column of the lane's pairs
    auto off_of = [&](int c) { return c >= 0 ? (unsigned)c * 8u : kLaneOut; };       // (a negative column: beyond the row)
    const unsigned o_b0 = off_of(k_b0), o_b1 = off_of(k_b1), o_x = off_of(k_x);
    auto row_of = [&](const double *base, int r) { return base + (long)min(max(r, 0), n_diag - 1) * P; };
    auto on_canvas = [&](int r, int c) { return c >= 0 && c < W && (unsigned)(r - c) < (unsigned)H; };
    auto b_in = [&](double v, int r, int c) { return (!MASKED || on_canvas(r, c)) ? v : lex_fixed_marker(); };
    auto x_in = [&](double v, int r, int c) { return (!MASKED || on_canvas(r, c)) ? v : 0.0; };
    auto ld_b = [&](const double *row, unsigned off) {
        const lex2_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(lex2_row_rsrc(row, P), (int)off, 0, 0);
        return __builtin_bit_cast(Pair, v);
    };
    // lane's ghost pair of a block: sweep g_t, step g_k -> b row (step - 4 g_t), columns kGhost + 2 g_t, + 1
    const bool g_on = lane < 8 * T;
    const int g_t = min(lane >> 3, T - 1), g_k = lane & 7;
    auto ghost_load = [&](int blk) -> Pair {
        const int d = blk + g_k;
        const bool ok = e_left != nullptr && g_on && d >= left_begin && d <= left_end;
        const long at = ((long)(min(max(d, left_begin), left_end) - left_begin) * T + g_t) * 2;
        // (the whole edge buffer of the left strip is the buffer; a lane without a value reads zeros)
        const lex2_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(
            __builtin_amdgcn_make_buffer_rsrc((void *)(e_left ? e_left : bp), 0, ok ? (int)(e_left_doubles * 8) : 0, 0x00020000), (int)(at * 8), 0, kAuxSc1);
        return __builtin_bit_cast(Pair, v);
    };
    auto put_ghost = [&](int blk, Pair v) {
        if (g_on) *reinterpret_cast<Pair *>(&brow[(blk + g_k - 4 * g_t) & (kLexBRows - 1)][kGhost + 2 * g_t]) = v;
    };
    auto put_b = [&](int r, Pair q0, Pair q1) {
        double *row = brow[r & (kLexBRows - 1)];
        Pair w0;
        w0.a = b_in(q0.a, r, k_b0);
        w0.b = b_in(q0.b, r, k_b0 + 1);
        if (lane < kPairs) *reinterpret_cast<Pair *>(&row[2 * lane]) = w0;
        if (kExtra > 0 && lane < kExtra) {
            Pair w1;
            w1.a = b_in(q1.a, r, k_b1);
            w1.b = b_in(q1.b, r, k_b1 + 1);
            *reinterpret_cast<Pair *>(&row[2 * (kWave + lane)]) = w1;
        }
    };
    auto put_x = [&](int r, Pair q) {
        Pair w;
        w.a = x_in(q.a, r, k_x);
        w.b = x_in(q.b, r, k_x + 1);
        ring[0][r & (kLexRing - 1)][lane] = w;
    };
    lex2_gate(st, db0);
    {   // what the first steps read before the rings are rolling: x rows db0 .. db0+2 and the ghost pairs of block db0
#pragma unroll
        for (int q = 0; q < 3; ++q) put_x(db0 + q, lex2_ld_pair_sc1(row_of(xq, db0 + q), P, o_x));
        put_ghost(db0, ghost_load(db0));
    }
    Pair qb[8], qb1[8], qx[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        qb[j] = ld_b(row_of(bp, db0 + 1 + j), o_b0);
        qb1[j] = ld_b(row_of(bp, db0 + 1 + j), o_b1);
        qx[j] = lex2_ld_pair_sc1(row_of(xq, db0 + 3 + j), P, o_x);
    }
    const double *rb = bp + (long)(db0 + 9) * P, *rx = xq + (long)(db0 + 11) * P;   // (rows >= 1 from here on; the arrays carry slack rows beyond the last diagonal)
    lex_lds_barrier();                                                       // (every wave of the workgroup comes here)
    for (int db = db0; db <= db1; db += 8) {
        if (db > db0) lex2_gate(st, db);
        const unsigned polled = __hip_atomic_load(st.words + st.watch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const Pair qg = ghost_load(db + 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            put_b(db + j + 1, qb[j], qb1[j]);                                // b row d + 1
            put_x(db + j + 3, qx[j]);                                        // x row d + 3
            if (j == 7) put_ghost(db + 8, qg);
            asm volatile("" ::: "memory");
            qb[j] = ld_b(rb, o_b0);                                          // b row d + 9
            qb1[j] = ld_b(rb, o_b1);
            qx[j] = lex2_ld_pair_sc1(rx, P, o_x);                            // x row d + 11
            rb += P;
            rx += P;
            lex_lds_barrier();
        }
        st.known = max(st.known, polled);
    }
}

// The storer's side: after the barrier of step d, sweep T-1's pairs of that step go to x and the last lane's pair of every
// sweep to the edge buffer — exactly two 16-byte sc1 stores per lane and step, masked by the buffer range check
// (kLexStoresPerBlock and the counted publication are k_lex_wg's).  A pair is stored whole when its lane is a real one
// (a ghost lane too in an interior strip: it carries the left strip's results for exactly these pixels); the element
// of a pair that is no pixel lands in a slot of the diagonal-major array that no pixel owns.
template <int T>
__device__ __forceinline__ void lex2_store(LexWgStrip &st, Pair (*ring)[kLexRing][kWave], int lane, int db0, int db1, double *xq, long P,
                                           int n_diag, int xs0, int d_begin, int d_end, double *e_mine, long e_mine_doubles, bool strip_interior)
{
    constexpr int t = T - 1;
    const int xa = xs0 + 2 * lane - 2 * t;                                   // image column of the lane's A (even)
    const bool stores = (lane > 0 || strip_interior) && xa >= -1;            // (xa = -1 cannot happen: even)
    const unsigned o_x = (stores && xa >= 0) ? (unsigned)xa * 8u : kLaneOut;
    lex_lds_barrier();                                                       // (the loader's priming barrier)
    for (int db = db0; db <= db1; db += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int d = db + j;
            lex_lds_barrier();                                               // step d is in the rings
            const Pair v = ring[T][j & (kLexRing - 1)][lane];
            const Pair ev = ring[min(lane, T - 1) + 1][j & (kLexRing - 1)][kWave - 1];
            const int r = d - 4 * t;                                         // the diagonal the step's pixels of sweep T-1 lie on
            const bool row_ok = r >= 0 && r < n_diag && d >= d_begin && d <= d_end;
            lex2_st_pair_sc1(v, xq + (long)min(max(r, 0), n_diag - 1) * P, row_ok ? P : 0, o_x);
            const bool e_ok = lane < T && d >= d_begin && d <= d_end;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(lex2_u32x4, ev),
                                                   __builtin_amdgcn_make_buffer_rsrc((void *)e_mine, 0, e_ok ? (int)(e_mine_doubles * 8) : 0, 0x00020000),
                                                   (int)((((long)(d - d_begin) * T + min(lane, T - 1)) * 2) * 8), 0, kAuxSc1);
        }
        static_assert(kLexStoresPerBlock == 8 * 2 + 1, "two stores per step and the publication");
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kLexPublishVmcnt) : "memory");
        if (lane == 0)
            __hip_atomic_store(st.mine, (unsigned)max(db - 8 * kLexPublishLagBlocks, 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// A workgroup: T compute waves, the loader, the storer.  grid = (G * S, channels), block = (T + 2) * 64.
template <int T, bool CHECK, bool MASKED>
__device__ __forceinline__ void lex2_body(double *__restrict__ xd, const double *__restrict__ bd, Geom g, LexGeom lg, int G, int S,
                                          unsigned *__restrict__ progress, unsigned *__restrict__ ticket, const unsigned *__restrict__ order,
                                          double *__restrict__ edges, long edge_steps, unsigned active_mask, double *__restrict__ partial,
                                          long partial_stride)
{
    static_assert(kLexRing == 4 && T >= 1, "the unrolled step index mod 4 is the ring slot");
    static_assert(4 * (T - 1) + 4 <= kLexBRows, "a b row stays in LDS from step r-1 to step r+4(T-1)");
    constexpr int kRowW = Lex2Shape<T>::kRowW;
    __shared__ __attribute__((aligned(16))) Pair ring[T + 1][kLexRing][kWave];
    __shared__ __attribute__((aligned(16))) double brow[kLexBRows][kRowW];
    __shared__ unsigned s_ticket;
    const int ch = blockIdx.y;
    if (!((active_mask >> ch) & 1u)) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));     // 0..T-1: sweeps, T: loader, T+1: storer
    const int t = min(wv, T - 1);
    if (threadIdx.x == 0) s_ticket = atomicAdd(&ticket[ch], 1u);
    if (wv <= T) {
#pragma unroll
        for (int q = 0; q < kLexRing; ++q) ring[wv][q][lane].a = ring[wv][q][lane].b = 0.0;
    }
    for (int i = threadIdx.x; i < kLexBRows * kRowW; i += (T + 2) * kWave) (&brow[0][0])[i] = 0.0;
    __syncthreads();
    const unsigned tk = order[s_ticket];                     // (group, strip) in wavefront order
    const int grp = (int)(tk / (unsigned)S), s = (int)(tk % (unsigned)S);
    const int HS = lg.H + 2 * (T - 1);
    const int xs0 = kLex2Cols * s - 2;                       // skewed column of lane 0's A (the ghost lane)
    const int d_begin = xs0, d_end = xs0 + (2 * kWave - 1) + HS - 1;
    const int db0 = d_begin & ~7, db1 = d_end & ~7;
    const long plane = (long)ch * lg.plane;
    const long e_doubles = edge_steps * (2 * T);
    double *e_mine = edges + ((long)ch * S + s) * e_doubles;
    const double *e_left = s > 0 ? edges + ((long)ch * S + s - 1) * e_doubles : nullptr;
    const int left_begin = xs0 - kLex2Cols, left_end = left_begin + (2 * kWave - 1) + HS - 1;
    const int cb = xs0 + 2 - 2 * (T - 1);                    // the leftmost image column any sweep of the strip touches (even)

    LexWgStrip st;
    const unsigned my_word = (unsigned)((((long)ch * G + grp) * S + s) * kLexWordStride);
    st.words = progress;
    st.mine = progress + my_word;
    st.watch = my_word;
    st.need_off = INT_MIN / 2;
    st.known = 0;
    if (lane == 0 && s > 0) {
        st.watch = my_word - kLexWordStride;
        st.need_off = 16;                                    // before block [db, db+7]: the ghost batch of block db+8
    }
    if (grp > 0 && (lane == 1 || (lane == 2 && s + 1 < S))) {
        st.watch = (unsigned)((((long)ch * G + grp - 1) * S + s + (lane - 1)) * kLexWordStride);
        st.need_off = 19 + 4 * (T - 1);                      // x row db+18, written by sweep T-1 at step db+18+4(T-1)
    }
    {   // b rows db0 - 4(T-1) .. db0 into the ring, a few per wave (the loader brings row d + 1 at step d)
        constexpr int kCols = Lex2Shape<T>::kCols, kPrime = 4 * (T - 1) + 1, kPer = (kPrime + T + 1) / (T + 2);
        constexpr int kPairs = kCols / 2;
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
            const int r = db0 - (kPrime - 1) + wv * kPer + q;
            if (r <= db0) {                                              // (uniform)
                const double *row = bd + plane + (long)min(max(r, 0), lg.n_diag - 1) * lg.P;
                for (int half = 0; half < 2; ++half) {
                    const int pr = half * kWave + lane;                  // pair index in the row
                    if (pr >= kPairs) break;
                    const int c = cb + 2 * pr;
                    Pair w;
                    w.a = (c >= 0 && c < lg.W) ? row[c] : 0.0;
                    w.b = (c + 1 >= 0 && c + 1 < lg.W) ? row[c + 1] : 0.0;
                    if (MASKED) {
                        if (!(c >= 0 && c < lg.W && (unsigned)(r - c) < (unsigned)lg.H)) w.a = lex_fixed_marker();
                        if (!(c + 1 >= 0 && c + 1 < lg.W && (unsigned)(r - c - 1) < (unsigned)lg.H)) w.b = lex_fixed_marker();
                    }
                    *reinterpret_cast<Pair *>(&brow[r & (kLexBRows - 1)][2 * pr]) = w;
                }
            }
        }
    }
    // the columns this wave's real lanes hold: xs0 + 2 - 2t .. xs0 + 127 - 2t
    const bool strip_interior = s > 0 && xs0 + 2 - 2 * t >= 1 && xs0 + 2 * kWave - 1 - 2 * t <= lg.W - 2;
    if (wv < T) {
        Pair h1;
        h1.a = h1.b = 0.0;
        double acc = 0.0;
        const bool has_x0 = s == 0 && xs0 + 2 - 2 * t <= 0 && xs0 + 2 * kWave - 1 - 2 * t >= 0;
        const bool has_xl = xs0 + 2 * kWave - 1 - 2 * t >= lg.W - 1;     // (column W-1, or nothing on the image at all)
        lex_lds_barrier();                                                   // (the loader's priming barrier)
        if (MASKED) lex2_compute<T, CHECK, 0, true>(h1, acc, ring, brow, g, lg.W, lg.H, t, lane, db0, db1, xs0);
        else if (strip_interior) lex2_compute<T, CHECK, 0, false>(h1, acc, ring, brow, g, lg.W, lg.H, t, lane, db0, db1, xs0);
        else if (has_x0 && !has_xl) lex2_compute<T, CHECK, 1, false>(h1, acc, ring, brow, g, lg.W, lg.H, t, lane, db0, db1, xs0);
        else if (!has_x0) lex2_compute<T, CHECK, 2, false>(h1, acc, ring, brow, g, lg.W, lg.H, t, lane, db0, db1, xs0);
        else lex2_compute<T, CHECK, 3, false>(h1, acc, ring, brow, g, lg.W, lg.H, t, lane, db0, db1, xs0);
        if (CHECK) {
            const double total = wave_sum(acc);
            if (lane == 0) partial[(((long)grp * T + t) * gridDim.y + ch) * partial_stride + s] = total;
        }
    } else if (wv == T) {
        lex2_load<T, MASKED>(st, ring, brow, lane, db0, db1, bd + plane, xd + plane, lg.P, lg.n_diag, lg.W, lg.H, cb, xs0, e_left, e_doubles,
                             left_begin, left_end);
    } else {
        const bool store_ghost = s > 0 && xs0 - 2 * (T - 1) >= 1 && xs0 + 2 * kWave - 1 - 2 * (T - 1) <= lg.W - 2;   // sweep T-1's strip_interior
        lex2_store<T>(st, ring, lane, db0, db1, xd + plane, lg.P, lg.n_diag, xs0, d_begin, d_end, e_mine, e_doubles, store_ghost);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                   // compiler ordering only
    __builtin_amdgcn_s_waitcnt(0);                                           // this wave's (write-through) stores acknowledged
    lex_lds_barrier();
    if (wv == T + 1 && lane == 0) __hip_atomic_store(st.mine, kLexDone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int T, bool CHECK>
__global__ void __launch_bounds__((T + 2) * kWave)
k_lex_wg2(double *__restrict__ xd, const double *__restrict__ bd, Geom g, LexGeom lg, int G, int S, unsigned *__restrict__ progress,
          unsigned *__restrict__ ticket, const unsigned *__restrict__ order, double *__restrict__ edges, long edge_steps,
          unsigned active_mask, double *__restrict__ partial, long partial_stride)
{
    lex2_body<T, CHECK, false>(xd, bd, g, lg, G, S, progress, ticket, order, edges, edge_steps, active_mask, partial, partial_stride);
}

template <int T, bool CHECK>
__global__ void __launch_bounds__((T + 2) * kWave)
k_lex_wg2_masked(double *__restrict__ xd, const double *__restrict__ bd, Geom g, LexGeom lg, int G, int S, unsigned *__restrict__ progress,
                 unsigned *__restrict__ ticket, const unsigned *__restrict__ order, double *__restrict__ edges, long edge_steps,
                 unsigned active_mask, double *__restrict__ partial, long partial_stride)
{
    lex2_body<T, CHECK, true>(xd, bd, g, lg, G, S, progress, ticket, order, edges, edge_steps, active_mask, partial, partial_stride);
}

}  // namespace ccp
