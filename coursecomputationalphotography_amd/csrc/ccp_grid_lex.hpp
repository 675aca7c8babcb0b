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

// ---------------------------------------------------------------------------------------------
// The same sweep without a launch per hyperplane: STRIP WAVES.  One wavefront owns a strip of 64 image
// columns of one sweep k and marches down the diagonals d = x + y: lane l walks down column x0 + l, one row
// per step, so at step d the wave's 64 pixels are exactly the strip's piece of diagonal d — one 512-byte
// row of the diagonal-major arrays.  Of the four neighbours of a pixel
//   up    (x, y-1), new : the lane's own previous result          (register)
//   left  (x-1, y), new : the left lane's previous result         (DPP wave_shr; lane 0: the strip to the left)
//   down  (x, y+1), old : row d+1 of x, same column               (coalesced load)
//   right (x+1, y), old : the right lane's `down` value           (DPP wave_shl; lane 63: the strip to the right)
// only the two strip-edge values come from another wave, through memory: wave (k, s) at step d needs
//   (k, s-1) finished through diagonal d-1,  (k-1, s) and (k-1, s+1) finished through diagonal d+1,
// exactly the hyperplane order restricted to neighbours.  Every wave publishes its progress (diagonals
// finished) every `chunk` steps and checks its three producers once per chunk.  Hand-off through memory
// (MI355X_MICROARCH.md, correctness boundaries: the per-XCD L2s are not coherent with each other): every
// access to x is an agent-scope (sc1) load or store — write-through, never served from a stale line — and a
// wave drains its stores (s_waitcnt vmcnt(0)) before it publishes the counter.  Cache-wide release/acquire
// fences instead (buffer_wbl2 / buffer_inv per chunk and wave) were measured 4x slower: thousands of waves
// flushing and invalidating whole L2s serialise on the caches.  In place on the diagonal-major x: a value is
// overwritten only after every reader of the old one is past it (the readers are the producers this wave
// waits for, or this wave itself).
// Work items are handed out by a ticket counter in (sweep, strip) order: a wave only ever waits for tickets
// smaller than its own, which belong to waves that have already started — no assumption about dispatch
// order or co-residency, no deadlock.  The whole pipeline of K sweeps is ONE launch: the sweep count no
// longer multiplies launches, and the rate is the same whether 4 or 4000 sweeps are asked for.
// grid = (K * S, channels), block = 64.  CHECK: partial[(k*channels + ch)*S + s] = the wave's sum |new - old|.
constexpr int kLexStripCols = kWave;
constexpr int kLexSub = 8;               // steps per software-pipeline block
constexpr unsigned kLexDone = 0xffffffffu;

__device__ __forceinline__ double lex_ld(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void lex_st(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void lex_wait(const unsigned *progress, unsigned need)
{
    if (progress == nullptr) return;
    if ((threadIdx.x & (kWave - 1)) == 0) {
        while (__hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) __builtin_amdgcn_s_sleep(2);
    }
}

template <bool CHECK>
__global__ void __launch_bounds__(kWave)
k_lex_strips(double *__restrict__ xd, const double *__restrict__ bd, Geom g, LexGeom lg, int K, int S, int chunk,
             unsigned *__restrict__ progress, unsigned *__restrict__ ticket, unsigned active_mask, double *__restrict__ partial)
{
    const int ch = blockIdx.y;
    if (!((active_mask >> ch) & 1u)) return;
    const int lane = threadIdx.x;
    unsigned t = 0;
    if (lane == 0) t = atomicAdd(&ticket[ch], 1u);
    t = (unsigned)__builtin_amdgcn_readfirstlane((int)__shfl((int)t, 0, kWave));
    const int k = (int)(t / (unsigned)S), s = (int)(t % (unsigned)S);
    const int x0 = s * kLexStripCols, x = x0 + lane;
    const bool col_ok = x < lg.W;
    const int x_last = min(x0 + kLexStripCols - 1, lg.W - 1);
    const int d_begin = x0, d_end = x_last + lg.H - 1;
    unsigned *prog = progress + ((long)ch * K + k) * S;
    unsigned *mine = prog + s;
    const unsigned *left_p = s > 0 ? prog + (s - 1) : nullptr;
    const unsigned *prev0_p = k > 0 ? prog - S + s : nullptr;
    const unsigned *prev1_p = (k > 0 && s + 1 < S) ? prog - S + s + 1 : nullptr;
    const long plane = (long)ch * lg.plane;
    double prev_new = 0.0;                                   // the lane's latest result: (x, y-1) for itself, (x-1, y) for the next lane
    double acc = 0.0;
    for (int dc = d_begin; dc <= d_end; dc += chunk) {
        const int de = min(dc + chunk - 1, d_end);
        // producers: (k, s-1) through diagonal de-1; (k-1, s) and (k-1, s+1) through diagonal de+1
        lex_wait(left_p, (unsigned)de);
        lex_wait(prev0_p, (unsigned)(de + 2));
        if (de + 1 >= x0 + kLexStripCols) lex_wait(prev1_p, (unsigned)(de + 2));
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");       // compiler ordering only: no load of x moves above the waits
        // the two edge columns of this chunk, one step per lane (chunk <= 64):
        //   left  edge: pixel (x0-1, y) on diagonal d-1, new   -> element [d-1][x0-1]
        //   right edge: pixel (x_last+1, y) on diagonal d+1, old -> element [d+1][x_last+1]
        double edge_left = 0.0, edge_right = 0.0;
        {
            const int d = dc + lane;
            if (d <= de) {
                if (x0 > 0) {
                    const int yl = d - 1 - (x0 - 1);
                    if (yl >= 0 && yl < lg.H) edge_left = lex_ld(&xd[plane + (long)(d - 1) * lg.P + (x0 - 1)]);
                }
                if (x_last + 1 < lg.W) {
                    const int yr = d + 1 - (x_last + 1);
                    if (yr >= 0 && yr < lg.H) edge_right = lex_ld(&xd[plane + (long)(d + 1) * lg.P + (x_last + 1)]);
                }
            }
        }
        // The march, software-pipelined in blocks of kLexSub steps: the loads of a block (the row below, b, and
        // for the stop rule the old value) are issued a whole block ahead of their use, so a step never waits
        // for memory — only the two edge gathers above and the first block of a chunk are exposed.
        double q_down[kLexSub], q_b[kLexSub], q_old[kLexSub], n_down[kLexSub], n_b[kLexSub], n_old[kLexSub];
        auto fetch = [&](int d0, double (&fd)[kLexSub], double (&fb)[kLexSub], double (&fo)[kLexSub]) {
#pragma unroll
            for (int q = 0; q < kLexSub; ++q) {
                const int d = d0 + q, y = d - x;
                const long i = plane + (long)d * lg.P + x;
                const bool in_chunk = d <= de;
                const bool on = in_chunk && col_ok && y >= 0 && y < lg.H;
                fd[q] = 0.0;
                fb[q] = 0.0;
                fo[q] = 0.0;
                // (x, y+1) on diagonal d+1, same column: also the RIGHT neighbour of the lane to the left, which is
                // one row ahead — so it is fetched from the row before this lane's first (y = -1) as well
                if (in_chunk && col_ok && y + 1 >= 0 && y + 1 < lg.H) fd[q] = lex_ld(&xd[i + lg.P]);
                if (on) {
                    fb[q] = bd[i];
                    if (CHECK) fo[q] = lex_ld(&xd[i]);
                }
            }
        };
        fetch(dc, q_down, q_b, q_old);
        for (int sb = dc; sb <= de; sb += kLexSub) {
            if (sb + kLexSub <= de) fetch(sb + kLexSub, n_down, n_b, n_old);
#pragma unroll
            for (int q = 0; q < kLexSub; ++q) {
                const int d = sb + q;
                if (d <= de) {                                   // (wave-uniform)
                    const int y = d - x;
                    const bool on = col_ok && y >= 0 && y < lg.H;
                    const long i = plane + (long)d * lg.P + x;
                    const double down = q_down[q], bv = q_b[q], old = q_old[q];
                    const int j = d - dc;
                    double left = lane_prev(prev_new);
                    const double el = __shfl(edge_left, j, kWave), er = __shfl(edge_right, j, kWave);
                    if (lane == 0) left = el;
                    double right = lane_next(down);
                    if (x == x_last) right = er;
                    if (on) {
                        const Stencil st = classify(g, x, y, y);
                        if (st.diag != 0) {                      // empty row: skipped (sparse-matrix.h:361-363)
                            double nv;
                            if (st.up && st.left && st.right && st.down && st.diag == 4) nv = (bv + (((prev_new + left) + right) + down)) * 0.25;
                            else (void)gs_update(st, bv, prev_new, left, right, down, nv);
                            if (CHECK) acc += fabs(nv - old);
                            lex_st(&xd[i], nv);
                            prev_new = nv;
                        } else {
                            prev_new = lex_ld(&xd[i]);           // the value the row keeps is what its neighbours see
                        }
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < kLexSub; ++q) {
                q_down[q] = n_down[q];
                q_b[q] = n_b[q];
                q_old[q] = n_old[q];
            }
        }
        // publish: every lane's (write-through) stores acknowledged, then the counter
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // compiler ordering only
        __builtin_amdgcn_s_waitcnt(0);
        if (lane == 0) __hip_atomic_store(mine, de == d_end ? kLexDone : (unsigned)(de + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (CHECK) {
        const double total = wave_sum(acc);
        if (lane == 0) partial[((long)k * gridDim.y + ch) * S + s] = total;
    }
}

// ---------------------------------------------------------------------------------------------
// Temporal blocking of the reference-order sweep: T sweeps per pass through memory (k_lex_skew).
// A red-black pass can recompute halos redundantly; the index-order sweep cannot (the left neighbour is the
// NEW value of the same sweep, which depends on the whole row to its left).  What it can do is SKEW: with
//     x' = x + 2t,  y' = y + 2t      (t = sweep inside the group of T)
// the four dependences of point (x', y', t)
//     (x-1, y, t) -> (x'-1, y',   t)      (x, y-1, t)   -> (x',   y'-1, t)
//     (x+1, y, t-1) -> (x'-1, y'-2, t-1)  (x, y+1, t-1) -> (x'-2, y'-1, t-1)
// all point to smaller x' (or equal x' and smaller y'): strips in x' depend on the strip to their LEFT only.
// A wavefront owns 64 skewed columns and marches down the skewed diagonals d' = x' + y'; at every step each
// lane updates its pixel of ALL T sweeps (pixel (x'-2t, y'-2t) of sweep t) — T independent updates — from
// registers: its own results of the last three steps per sweep (h1..h3; h4 for the stop rule's old value)
// and its left neighbours' by DPP:
//     up = h1[t]   left = shr1(h1[t])   right = shr1(h3[t-1])   down = shr2(h3[t-1])   old = shr2(h4[t-1]).
// Only sweep 0 reads x (the previous group's result: rows d'+1 and, for the stop rule, d') and only sweep T-1
// writes it: 16/T B per update plus b (8 B, re-read by every sweep) instead of 32 B.
// Lanes 0 and 1 are GHOST lanes: they carry the results of the left strip's lanes 62 and 63 (2T doubles per
// step, written by that strip to `edges`, read back here) so every DPP shift is uniform; a strip therefore
// advances 62 skewed columns.  Producers of wave (group, s): (group, s-1) through the same step; (group-1, s)
// and (group-1, s+1) through step + 1 + 4(T-1) (they wrote the x this group's sweep 0 reads).  Progress
// counters, tickets, sc1 hand-off as in k_lex_strips.
// grid = (G * S, channels), block = 64.  CHECK: partial[((group*T + t)*channels + ch)*partial_stride + s].
constexpr int kLexSkewCols = kWave - 2;
constexpr int kLexSkewAhead = 4;         // steps the loads run ahead of the computation (ring of register slots)

__device__ __forceinline__ double lane_prev2(double v) { return lane_prev(lane_prev(v)); }

template <int T, bool CHECK>
__global__ void __launch_bounds__(kWave)
k_lex_skew(double *__restrict__ xd, const double *__restrict__ bd, Geom g, LexGeom lg, int G, int S, int chunk,
           unsigned *__restrict__ progress, unsigned *__restrict__ ticket, double *__restrict__ edges, long edge_steps,
           unsigned active_mask, double *__restrict__ partial, long partial_stride)
{
    const int ch = blockIdx.y;
    if (!((active_mask >> ch) & 1u)) return;
    const int lane = threadIdx.x;
    unsigned tk = 0;
    if (lane == 0) tk = atomicAdd(&ticket[ch], 1u);
    tk = (unsigned)__builtin_amdgcn_readfirstlane((int)__shfl((int)tk, 0, kWave));
    const int grp = (int)(tk / (unsigned)S), s = (int)(tk % (unsigned)S);
    const int HS = lg.H + 2 * (T - 1);                        // skewed rows
    const int xs0 = kLexSkewCols * s - 2;                     // skewed column of lane 0 (a ghost lane)
    const int xp = xs0 + lane;
    const bool ghost = lane < 2;
    const int d_begin = xs0, d_end = xs0 + (kWave - 1) + HS - 1;
    unsigned *prog = progress + ((long)ch * G + grp) * S;
    unsigned *mine = prog + s;
    const unsigned *left_p = s > 0 ? prog + (s - 1) : nullptr;
    const unsigned *prev0_p = grp > 0 ? prog - S + s : nullptr;
    const unsigned *prev1_p = (grp > 0 && s + 1 < S) ? prog - S + s + 1 : nullptr;
    const long plane = (long)ch * lg.plane;
    double *e_mine = edges + ((long)ch * S + s) * edge_steps * (2 * T);
    const double *e_left = s > 0 ? edges + ((long)ch * S + s - 1) * edge_steps * (2 * T) : nullptr;
    const int left_begin = xs0 - kLexSkewCols, left_end = left_begin + (kWave - 1) + HS - 1;
    double h1[T], h2[T], h3[T], h4[T], acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) h1[t] = h2[t] = h3[t] = h4[t] = acc[t] = 0.0;
    // Everything a step reads from memory is independent of the computation, so it is fetched kLexSkewAhead steps
    // early into a ring of registers (slot = step mod kLexSkewAhead, refilled as soon as it has been consumed):
    // sweep 0's two x values (and the old value for the stop rule), and per sweep one value that is b for a real
    // lane and the left strip's result for a ghost lane.
    constexpr int PF = kLexSkewAhead;
    double q_dn[PF], q_rt[PF], q_old[PF], q_v[PF][T];
    auto fetch = [&](int d, double &dn0, double &rt0, double &old0, double (&v)[T]) {
        const int yp = d - xp;
        dn0 = 0.0;
        rt0 = 0.0;
        old0 = 0.0;
        const bool live = d <= d_end;
        // sweep 0's inputs from x: (xp, yp+1) and (xp+1, yp) on diagonal d+1; for the stop rule (xp, yp) itself
        if (live && xp >= 0 && xp < lg.W && yp + 1 >= 0 && yp + 1 < lg.H) dn0 = lex_ld(&xd[plane + (long)(d + 1) * lg.P + xp]);
        if (live && xp + 1 >= 0 && xp + 1 < lg.W && yp >= 0 && yp < lg.H) rt0 = lex_ld(&xd[plane + (long)(d + 1) * lg.P + xp + 1]);
        if (CHECK && live && xp >= 0 && xp < lg.W && yp >= 0 && yp < lg.H) old0 = lex_ld(&xd[plane + (long)d * lg.P + xp]);
        const bool left_live = live && e_left != nullptr && d >= left_begin && d <= left_end;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int x = xp - 2 * t, y = yp - 2 * t;
            v[t] = 0.0;
            if (ghost) {                                                     // the left strip's lanes 62 / 63 of this step
                if (left_live) v[t] = lex_ld(&e_left[((long)(d - left_begin) * T + t) * 2 + lane]);
            } else if (live && x >= 0 && x < lg.W && y >= 0 && y < lg.H) {
                v[t] = bd[plane + (long)(x + y) * lg.P + x];
            }
        }
    };
    bool primed = false;
    for (int dc = d_begin; dc <= d_end; dc += chunk) {
        const int de = min(dc + chunk - 1, d_end);
        // producers, PF steps beyond the chunk (the ring is refilled that far ahead)
        lex_wait(left_p, (unsigned)(de + PF + 1));                         // the left strip through step de + PF
        lex_wait(prev0_p, (unsigned)(de + PF + 2 + 4 * (T - 1)));          // the previous group's x, rows up to de + PF + 1
        lex_wait(prev1_p, (unsigned)(de + PF + 2 + 4 * (T - 1)));
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");             // compiler ordering only
        if (!primed) {
#pragma unroll
            for (int j = 0; j < PF; ++j) fetch(d_begin + j, q_dn[j], q_rt[j], q_old[j], q_v[j]);
            primed = true;
        }
        for (int db = dc; db <= de; db += PF) {
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                const int d = db + j;
                if (d <= de) {                                               // (wave-uniform; chunk is a multiple of PF)
                    const int yp = d - xp;
                    const double dn0 = q_dn[j], rt0 = q_rt[j], old0 = q_old[j];
                    double vv[T];
#pragma unroll
                    for (int t = 0; t < T; ++t) vv[t] = q_v[j][t];
                    fetch(d + PF, q_dn[j], q_rt[j], q_old[j], q_v[j]);     // the slot is free again: refill it for step d + PF
                    double nh[T];
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        const int x = xp - 2 * t, y = yp - 2 * t;
                        const bool on = !ghost && x >= 0 && x < lg.W && y >= 0 && y < lg.H;
                        const double up = h1[t];
                        const double left = lane_prev(h1[t]);
                        const double right = t == 0 ? rt0 : lane_prev(h3[t > 0 ? t - 1 : 0]);
                        const double down = t == 0 ? dn0 : lane_prev2(h3[t > 0 ? t - 1 : 0]);
                        // (every DPP read happens here, with all lanes active: a source lane masked out by a branch reads as 0)
                        const double old = !CHECK ? 0.0 : (t == 0 ? old0 : lane_prev2(h4[t > 0 ? t - 1 : 0]));
                        double nv = ghost ? vv[t] : 0.0;
                        if (on) {
                            const double bv = vv[t];
                            const Stencil st = classify(g, x, y, y);
                            if (st.diag != 0) {
                                if (st.up && st.left && st.right && st.down && st.diag == 4) nv = (bv + (((up + left) + right) + down)) * 0.25;
                                else (void)gs_update(st, bv, up, left, right, down, nv);
                                if (CHECK) acc[t] += fabs(nv - old);
                                if (t == T - 1) lex_st(&xd[plane + (long)(x + y) * lg.P + x], nv);
                            }
                        }
                        nh[t] = nv;
                    }
                    if (lane >= kWave - 2) {
#pragma unroll
                        for (int t = 0; t < T; ++t) lex_st(&e_mine[((long)(d - d_begin) * T + t) * 2 + (lane - (kWave - 2))], nh[t]);
                    }
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        if (CHECK) h4[t] = h3[t];
                        h3[t] = h2[t];
                        h2[t] = h1[t];
                        h1[t] = nh[t];
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");               // compiler ordering only
        __builtin_amdgcn_s_waitcnt(0);
        if (lane == 0) __hip_atomic_store(mine, de == d_end ? kLexDone : (unsigned)(de + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (CHECK) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const double total = wave_sum(acc[t]);
            if (lane == 0) partial[(((long)grp * T + t) * gridDim.y + ch) * partial_stride + s] = total;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The skewed pass with the T sweeps spread over the waves of a workgroup (k_lex_wg).  Same skew, strips, ghost
// lanes, edge buffer, progress words and tickets as k_lex_skew, but a workgroup of T + 2 waves per strip:
//
//   waves 0 .. T-1   one sweep each.  Wave t reads its inputs from LDS only — the results of sweep t-1 three and
//                    four steps back (one and two lanes to the left) from a ring of the last 8 result rows per
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
//   C  anything else (the first and last ~80 steps of a strip): the general body of k_lex_skew, one step at a
//      time, every compute wave doing its own loads and stores.
// grid = (G * S, channels), block = (T + 2) * 64.  CHECK: partial[((group*T + t)*channels + ch)*partial_stride + s].
constexpr int kLexRing = 8;
constexpr int kLexBRows = 32;
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

// Blocks [db0, db1] (steps db0 .. db1+7) of compute wave t, every step in body A (KIND 0) or B (1: the wave holds
// column 0 — only in strip 0, whose ghost lanes lie off the image; 2: it holds column W-1; 3: both, an image
// narrower than a strip).  1 <= y <= H-2 for every lane, so the row of a pixel depends on its column alone:
// column 0 has no left neighbour (diagonal 3), column W-1 only its left one (diagonal 1), a 1-pixel-wide image
// and the lanes off the image have no row at all — those keep whatever the full-row formula gives, no row of the
// matrix reads them.  A group moves at the pace of its first strip (every strip waits on its left neighbour),
// and three of that strip's waves share a SIMD: body B is kept as short as body A allows.
// ring[t] holds sweep t's INPUT rows (ring[0]: x, filled by the loader), ring[t+1] its results.
template <int T, bool CHECK, int KIND>
__device__ __forceinline__ void lex_wg_compute(double &h1, double &acc, double (*ring)[kLexRing][kWave],
                                               const double (*brow)[LexWgShape<T>::kRowW], int t, int lane, int db0, int db1,
                                               bool lane_on, Stencil st_b)
{
    const bool ghost = lane < 2;
    const int lds1 = max(lane - 1, 0), lds2 = max(lane - 2, 0);
    const int col = ghost ? LexWgShape<T>::kGhost + 2 * t + lane : lane - 2 - 2 * t + 2 * (T - 1);   // of a b row in LDS
    const bool c_off = !lane_on || st_b.diag == 0;           // (st_b: classify() of this lane's column at an interior y)
    const bool c_x0 = !c_off && !st_b.left, c_xl = !c_off && !st_b.right;
    const bool wrote = KIND == 0 ? !ghost : !c_off;
    for (int db = db0; db <= db1; db += 8) {
        const int sb = (db - 4 * t) & (kLexBRows - 1);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const double right = ring[t][(j + 5) & 7][lds1];
            const double down = ring[t][(j + 5) & 7][lds2];
            const double vv = brow[(sb + j) & (kLexBRows - 1)][col];
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
            if (CHECK) {
                const double old = ring[t][(j + 4) & 7][lds2];
                acc += wrote ? fabs(nv - old) : 0.0;
            }
            ring[t + 1][j][lane] = nv;
            h1 = nv;
            lex_lds_barrier();
        }
    }
}

// What the loader and the storer know about the strip.
struct LexWgStrip {
    const unsigned *watch;               // lanes 0..2 of the loader: whose progress to watch (own word: nothing to wait for)
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
            st.known = __hip_atomic_load(st.watch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = (int)min(st.known, 0x7fffffffu) >= need;
        }
        if (!__all(ok)) __builtin_amdgcn_s_sleep(16);        // ~1000 cycles: the word is being written by the strip polled
    }
}

// The loader's side of blocks [db0, db1].  pb: this lane's column of b row db0+1 (pb1: the columns past 64),
// px: this lane's column of x row db0 (columns xs0+2 ..., as sweep 0's lanes 2.. read them one and two places to
// their left), pg: lanes 0 .. 16T-1: the left strip's edge value of sweep lane/16, step db0 + (lane%16)/2, edge
// lane%2 (a second load covers sweeps 4..7: 8 doubles further on).  Every slot of the register rings is refilled only after its old contents have been used (a load
// issued while the old value is live lands in another register and costs a copy and a full drain at the
// loop's back edge).
template <int T>
__device__ __forceinline__ void lex_wg_load(LexWgStrip &st, double (*ring)[kLexRing][kWave], double (*brow)[LexWgShape<T>::kRowW], int lane,
                                            int db0, int db1, const double *pb, const double *pb1, const double *px, const double *pg,
                                            bool ghost_live, long P)
{
    constexpr int kCols = LexWgShape<T>::kCols, kGhost = LexWgShape<T>::kGhost;
    constexpr int kGhostOps = (16 * T + kWave - 1) / kWave;                  // 64-lane loads per ghost batch
    // where lane's ghost value(s) of a block go: sweep gt, step gk/2, edge gk%2 -> row (d - 4 gt), column kGhost + 2 gt + e
    int g_t[kGhostOps], g_step[kGhostOps], g_col[kGhostOps];
    bool g_on[kGhostOps];
#pragma unroll
    for (int q = 0; q < kGhostOps; ++q) {
        const int idx = q * kWave + lane;
        g_on[q] = idx < 16 * T;
        g_t[q] = min(idx >> 4, T - 1);
        g_step[q] = (idx & 15) >> 1;
        g_col[q] = kGhost + 2 * g_t[q] + (idx & 1);
    }
    lex_wg_gate(st, db0);
    {   // what the first steps of the run read before the rings are rolling: x rows db0, db0+1, db0+2 and the
        // ghost values of block db0
#pragma unroll
        for (int q = 0; q < 3; ++q) ring[0][(db0 + q - 4) & 7][lane] = lex_ld(px + (long)q * P);
#pragma unroll
        for (int q = 0; q < kGhostOps; ++q) {
            const double v = ghost_live ? lex_ld(pg + 8 * q) : 0.0;
            if (g_on[q]) brow[(db0 + g_step[q] - 4 * g_t[q]) & (kLexBRows - 1)][g_col[q]] = v;
        }
    }
    px += 3 * P;
    pg += 16 * T;
    double qb[8], qb1[8], qx[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        qb[j] = lex_ld(pb);                                                  // b row db0 + 1 + j
        qb1[j] = kCols > kWave ? lex_ld(pb1) : 0.0;
        qx[j] = lex_ld(px);                                                  // x row db0 + 3 + j
        pb += P;
        pb1 += P;
        px += P;
    }
    lex_lds_barrier();                                                       // (every wave of the workgroup comes here)
    for (int db = db0; db <= db1; db += 8) {
        if (db > db0) lex_wg_gate(st, db);
        // issued here, looked at after the block's last step: no loaded value but the prefetch slots lives across
        // the loop's back edge
        const unsigned polled = __hip_atomic_load(st.watch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        double qg[kGhostOps];
#pragma unroll
        for (int q = 0; q < kGhostOps; ++q) qg[q] = ghost_live ? lex_ld(pg + 8 * q) : 0.0;       // block db + 8
        pg += 16 * T;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = (db + j + 1) & (kLexBRows - 1);
            brow[r][lane] = qb[j];                                           // b row d + 1
            if (kCols > kWave && lane < kCols - kWave) brow[r][kWave + lane] = qb1[j];
            ring[0][(j + 7) & 7][lane] = qx[j];                              // x row d + 3: read by sweep 0 at steps d+2, d+3
            if (j == 7) {
#pragma unroll
                for (int q = 0; q < kGhostOps; ++q)
                    if (g_on[q]) brow[(db + 8 + g_step[q] - 4 * g_t[q]) & (kLexBRows - 1)][g_col[q]] = qg[q];
            }
            asm volatile("" ::: "memory");
            qb[j] = lex_ld(pb);                                              // b row d + 9
            if (kCols > kWave) qb1[j] = lex_ld(pb1);
            qx[j] = lex_ld(px);                                              // x row d + 11
            pb += P;
            pb1 += P;
            px += P;
            lex_lds_barrier();
        }
        st.known = max(st.known, polled);
    }
}

// The storer's side of blocks [db0, db1]: after the barrier of step d, sweep T-1's row of that step goes to x
// and the 2T edge values of the step to the edge buffer.  ps: this lane's pixel of sweep T-1 at step db0,
// pe: lanes 0..2T-1: this strip's edge slot (step db0, sweep lane/2, edge lane%2).
template <int T, bool BORDER>
__device__ __forceinline__ void lex_wg_store(LexWgStrip &st, double (*ring)[kLexRing][kWave], int lane, int db0, int db1, double *ps,
                                             double *pe, long P, bool wrote)
{
    const int e_t = min(lane >> 1, T - 1) + 1, e_lane = kWave - 2 + (lane & 1);
    lex_lds_barrier();                                                       // (the loader's priming barrier)
    for (int db = db0; db <= db1; db += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            lex_lds_barrier();                                               // step d is in the rings
            const double v = ring[T][j][lane];
            const double ev = ring[e_t][j][e_lane];
            // body A stores from the ghost lanes too: they carry the left strip's results of the same sweep for
            // exactly these pixels, so it is the value already there
            if (!BORDER || wrote) lex_st(ps, v);
            if (lane < 2 * T) lex_st(pe, ev);
            ps += P;
            pe += 2 * T;
        }
        // Stores complete in issue order: once at most the 48 youngest are outstanding (16 per block and a
        // publication: this block, the one before, most of a third), every store of the blocks before those
        // has been acknowledged — publish their steps, without draining.
        asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
        if (lane == 0 && db - 16 > db0) __hip_atomic_store(st.mine, (unsigned)(db - 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int T, bool CHECK>
__global__ void __launch_bounds__((T + 2) * kWave)
k_lex_wg(double *__restrict__ xd, const double *__restrict__ bd, Geom g, LexGeom lg, int G, int S,
         unsigned *__restrict__ progress, unsigned *__restrict__ ticket, const unsigned *__restrict__ order,
         double *__restrict__ edges, long edge_steps, unsigned active_mask, double *__restrict__ partial, long partial_stride)
{
    static_assert(kLexRing == 8 && T >= 2, "the unrolled step index is the ring slot");
    static_assert(4 * (T - 1) + 4 <= kLexBRows, "a b row stays in LDS from step r-1 to step r+4(T-1)");
    constexpr int kCols = LexWgShape<T>::kCols, kRowW = LexWgShape<T>::kRowW;
    __shared__ double ring[T + 1][kLexRing][kWave];
    __shared__ double brow[kLexBRows][kRowW];
    __shared__ unsigned s_ticket;
    const int ch = blockIdx.y;
    if (!((active_mask >> ch) & 1u)) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));     // 0..T-1: sweeps, T: loader, T+1: storer
    const bool is_compute = wv < T, is_loader = wv == T, is_storer = wv == T + 1;
    const int t = min(wv, T - 1);                            // (the storer works on sweep T-1's geometry)
    if (threadIdx.x == 0) s_ticket = atomicAdd(&ticket[ch], 1u);
    if (wv <= T) {
#pragma unroll
        for (int q = 0; q < kLexRing; ++q) ring[wv][q][lane] = 0.0;
    }
    __syncthreads();
    const unsigned tk = order[s_ticket];                     // (group, strip) in wavefront order
    const int grp = (int)(tk / (unsigned)S), s = (int)(tk % (unsigned)S);
    const int HS = lg.H + 2 * (T - 1);
    const int xs0 = kLexSkewCols * s - 2;
    const int xp = xs0 + lane;
    const int xl = xp - 2 * t;                               // this lane's image column
    const bool ghost = lane < 2;
    const int d_begin = xs0, d_end = xs0 + (kWave - 1) + HS - 1;
    const int d_base = d_begin & ~7;                         // (floor to a multiple of 8, also when negative)
    const long plane = (long)ch * lg.plane;
    double *e_mine = edges + ((long)ch * S + s) * edge_steps * (2 * T);
    const double *e_left = s > 0 ? edges + ((long)ch * S + s - 1) * edge_steps * (2 * T) : nullptr;
    const int left_begin = xs0 - kLexSkewCols, left_end = left_begin + (kWave - 1) + HS - 1;
    const bool lane_on = !ghost && xl >= 0 && xl < lg.W;
    const bool ok_dn = xp >= 0 && xp < lg.W, ok_rt = xp + 1 >= 0 && xp + 1 < lg.W;
    const bool ghost_live = ghost && e_left != nullptr;
    const int lds1 = max(lane - 1, 0), lds2 = max(lane - 2, 0);
    const int cb = xs0 + 2 - 2 * (T - 1);                    // the leftmost image column any sweep of the strip touches

    // One progress word per strip, written by the storer.  The loader watches: lane 0 the left strip (edge
    // values), lanes 1 and 2 the two strips of the previous group sweep 0 reads x from.
    LexWgStrip st;
    st.mine = progress + (((long)ch * G + grp) * S + s) * kLexWordStride;
    st.watch = st.mine;
    st.need_off = INT_MIN / 2;
    st.known = 0;
    if (lane == 0 && s > 0) {
        st.watch = st.mine - kLexWordStride;
        st.need_off = 16;                                    // before block [db, db+7]: the ghost batch of block db+8
    }
    if (grp > 0 && (lane == 1 || (lane == 2 && s + 1 < S))) {
        st.watch = progress + (((long)ch * G + grp - 1) * S + s + (lane - 1)) * kLexWordStride;
        st.need_off = 19 + 4 * (T - 1);                      // x row db+18, written by sweep T-1 at step db+18+4(T-1)
    }

    // blocks in which every real lane of every sweep has 1 <= y <= H-2 at every step (sweep t: steps
    // xs0+64+2t .. xs0+2t+H), the same blocks for all waves
    const int run0 = (max(xs0 + 64 + 2 * (T - 1), d_begin) + 7) & ~7;
    const int run1 = (min(xs0 + lg.H, d_end) - 7) & ~7;                      // last block of the run (< run0: none)
    const bool strip_interior = s > 0 && xs0 + 2 - 2 * t >= 1 && xs0 + 63 - 2 * t <= lg.W - 2;
    double h1 = 0.0, acc = 0.0;

    auto general_block = [&](int db) {                                       // C: one step at a time, nothing in flight
        if (is_loader) lex_wg_gate(st, db);
        lex_lds_barrier();
#pragma unroll 1
        for (int d = db; d < db + 8; ++d) {
            if (d < d_begin || d > d_end) continue;                          // (uniform over the workgroup)
            if (is_loader) {                                                 // b row d + 1 into the ring
                const int r = d + 1;
                const bool r_ok = r >= 0 && r < lg.n_diag;
                const int c0 = cb + lane, c1 = cb + kWave + lane;
                brow[r & (kLexBRows - 1)][lane] = (r_ok && c0 >= 0 && c0 < lg.W) ? bd[plane + (long)r * lg.P + c0] : 0.0;
                if (kCols > kWave && lane < kCols - kWave)
                    brow[r & (kLexBRows - 1)][kWave + lane] = (r_ok && c1 >= 0 && c1 < lg.W) ? bd[plane + (long)r * lg.P + c1] : 0.0;
            }
            if (is_compute) {
                const int yp = d - xp, y = yp - 2 * t;
                double right = 0.0, down = 0.0, old = 0.0, vv = 0.0;
                if (t == 0) {                                 // sweep 0's inputs from x (the previous group's result)
                    if (ok_dn && yp + 1 >= 0 && yp + 1 < lg.H) down = lex_ld(&xd[plane + (long)(d + 1) * lg.P + xp]);
                    if (ok_rt && yp >= 0 && yp < lg.H) right = lex_ld(&xd[plane + (long)(d + 1) * lg.P + xp + 1]);
                    if (CHECK && ok_dn && yp >= 0 && yp < lg.H) old = lex_ld(&xd[plane + (long)d * lg.P + xp]);
                } else {
                    right = ring[t][(d - 3) & 7][lds1];
                    down = ring[t][(d - 3) & 7][lds2];
                    if (CHECK) old = ring[t][(d - 4) & 7][lds2];
                }
                const bool on = lane_on && y >= 0 && y < lg.H;
                if (ghost) {
                    if (ghost_live && d >= left_begin && d <= left_end) vv = lex_ld(&e_left[((long)(d - left_begin) * T + t) * 2 + lane]);
                } else if (on) {
                    vv = bd[plane + (long)(xl + y) * lg.P + xl];
                }
                const double up = h1;
                const double left = lane_prev(h1);
                double nv = ghost ? vv : 0.0;
                if (on) {
                    const Stencil sc = classify(g, xl, y, y);
                    if (sc.diag != 0) {
                        (void)gs_update(sc, vv, up, left, right, down, nv);
                        if (CHECK) acc += fabs(nv - old);
                        if (t == T - 1) lex_st(&xd[plane + (long)(xl + y) * lg.P + xl], nv);
                    }
                }
                ring[t + 1][d & 7][lane] = nv;
                if (lane >= kWave - 2) lex_st(&e_mine[((long)(d - d_begin) * T + t) * 2 + (lane - (kWave - 2))], nv);
                h1 = nv;
            }
            lex_lds_barrier();
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");               // compiler ordering only
        __builtin_amdgcn_s_waitcnt(0);                                       // this wave's (write-through) stores acknowledged
        lex_lds_barrier();                                                   // ... and every other wave's
        if (is_storer && lane == 0 && db + 7 < d_end)
            __hip_atomic_store(st.mine, (unsigned)max(db + 8, 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };

    int db = d_base;
    for (; db <= d_end && (db < run0 || run1 < run0); db += 8) general_block(db);
    if (run1 >= run0) {
        if (is_compute) {
            const Stencil st_b = classify(g, lane_on ? xl : 0, 1, 1);
            lex_lds_barrier();                                               // (the loader's priming barrier)
            // which borders this wave's columns xs0+2-2t .. xs0+63-2t hold
            const bool has_x0 = s == 0 && xs0 + 2 - 2 * t <= 0 && xs0 + 63 - 2 * t >= 0;
            const bool has_xl = xs0 + 63 - 2 * t >= lg.W - 1;            // (column W-1, or nothing on the image at all)
            if (strip_interior) lex_wg_compute<T, CHECK, 0>(h1, acc, ring, brow, t, lane, run0, run1, lane_on, st_b);
            else if (has_x0 && !has_xl) lex_wg_compute<T, CHECK, 1>(h1, acc, ring, brow, t, lane, run0, run1, lane_on, st_b);
            else if (!has_x0) lex_wg_compute<T, CHECK, 2>(h1, acc, ring, brow, t, lane, run0, run1, lane_on, st_b);
            else lex_wg_compute<T, CHECK, 3>(h1, acc, ring, brow, t, lane, run0, run1, lane_on, st_b);
        } else if (is_loader) {
            // lanes whose column lies outside the image load a clamped one: no load of the run is conditional,
            // and what they fetch is never used
            const double *pb = bd + plane + (long)(run0 + 1) * lg.P + min(max(cb + lane, 0), lg.W - 1);
            const double *pb1 = bd + plane + (long)(run0 + 1) * lg.P + min(max(cb + kWave + min(lane, max(kCols - kWave - 1, 0)), 0), lg.W - 1);
            const double *px = xd + plane + (long)run0 * lg.P + min(max(xs0 + 2 + min(lane, kWave - 2), 0), lg.W - 1);
            const int gi = min(lane, 16 * T - 1);            // (second 64 lanes of the batch: + 64 doubles)
            const double *pg = s > 0 ? e_left + ((long)(run0 - left_begin + ((gi & 15) >> 1)) * T + (gi >> 4)) * 2 + (gi & 1) : e_mine;
            lex_wg_load<T>(st, ring, brow, lane, run0, run1, pb, pb1, px, pg, s > 0, lg.P);
        } else {
            const Stencil st_b = classify(g, lane_on ? xl : 0, 1, 1);
            double *ps = xd + plane + (long)(run0 - 4 * t) * lg.P + xl;
            double *pe = e_mine + ((long)(run0 - d_begin) * T) * 2 + min(lane, 2 * T - 1);
            const bool wrote = lane_on && st_b.diag != 0;
            if (strip_interior) lex_wg_store<T, false>(st, ring, lane, run0, run1, ps, pe, lg.P, wrote);
            else lex_wg_store<T, true>(st, ring, lane, run0, run1, ps, pe, lg.P, wrote);
        }
        for (db = run1 + 8; db <= d_end; db += 8) general_block(db);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                   // compiler ordering only
    __builtin_amdgcn_s_waitcnt(0);
    lex_lds_barrier();
    if (is_storer && lane == 0) __hip_atomic_store(st.mine, kLexDone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (CHECK && is_compute) {
        const double total = wave_sum(acc);
        if (lane == 0) partial[(((long)grp * T + t) * gridDim.y + ch) * partial_stride + s] = total;
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
