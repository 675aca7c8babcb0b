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
// The skewed pass with the T sweeps spread over the T WAVES of a workgroup (k_lex_wg).  Same skew, same
// strips, ghost lanes, edge buffer, progress counters and tickets as k_lex_skew — but wave t owns sweep t
// alone: a handful of registers per lane instead of a 255-VGPR window, T times as many waves in flight, and
// the sweep-to-sweep hand-off (the results of sweep t-1 three and four steps back, one and two lanes to the
// left) goes through a ring of the last 8 result rows per sweep in LDS instead of DPP over register history.
// The waves march in lock-step, one workgroup barrier per step: what wave t reads at step d was written by
// wave t-1 at step d-3 / d-4, at least three barriers earlier, and is overwritten at step d+4 at the
// earliest.  Only wave 0 reads x, only wave T-1 writes it.
//
// Three bodies per step, chosen per wave (uniformly):
//   A  every real lane of the wave is an interior pixel (4 neighbours, diagonal 4): no classification, no
//      bounds tests, row bases in scalar registers — the body nearly every step of nearly every strip takes;
//   B  every real lane has 1 <= y <= H-2 but the strip touches the left / right image border: the stencil of a
//      lane does not change from step to step, so it is classified once, outside the loop (the first and last
//      strips must keep pace with the rest — every strip waits on its left neighbour);
//   C  anything else (the first and last ~64 steps of a strip): the general body of k_lex_skew.
// grid = (G * S, channels), block = T * 64.  CHECK: partial[((group*T + t)*channels + ch)*partial_stride + s].
constexpr int kLexRing = 8;
constexpr int kLexWgAhead = 8;                     // prefetch distance in steps (= the unroll, = the ring)

// Workgroup barrier that orders LDS traffic only: __syncthreads() would also drain the global loads this
// wave has in flight (the prefetch ring) at every step.
__device__ __forceinline__ void lex_lds_barrier()
{
#if defined(CCP_EXP) && CCP_EXP == 3
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// What a wave of k_lex_wg carries from block to block.
struct LexWgWave {
    double h1 = 0.0, acc = 0.0;          // last result of this lane (its `up`; the next lane's `left`), sum |dx|
    unsigned known = 0;                  // lanes 0..2: the newest progress value seen of the watched wave
    const unsigned *watch = nullptr;     // lanes 0..2: whose progress to watch (own cell when there is nothing to wait for)
    int need_off = 0;                    // ... which has to reach block start + need_off (INT_MIN/2: nothing to wait for)
    unsigned *mine = nullptr;
};

// b goes through LDS as well: sweep t needs diagonal row d - 4t of b at step d, so the T waves read the same
// row four steps apart, each two columns further left.  Every wave loads a 1/T slice of row d+9 at step d and
// writes the slice it loaded eight steps earlier (row d+1) into a ring of 32 rows; after the barrier of step d
// rows d-30 .. d+1 are there.  b is read from memory once per PASS, not once per sweep.
constexpr int kLexBRows = 32;
template <int T>
struct LexWgShape {
    static constexpr int kCols = kLexSkewCols + 2 * (T - 1);                 // image columns the T sweeps of a strip touch
    static constexpr int kSlice = (kCols + T - 1) / T;                       // ... of which every wave loads this many
    static constexpr int kBW = kSlice * T;
};

__device__ __forceinline__ double lane_next_rot(double v)                    // lane i <- lane i+1, lane 63 <- lane 0
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x134, 0xf, 0xf, true);          // wave_rol:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x134, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// a / 3, correctly rounded, without the division sequence (body B does it at every step in the strip that
// holds column 0, and every other strip waits on that one).  y = RN(1/3) = (1/3)(1 - 2^-54); q0 = RN(a y) is
// within 1.5 ulp of a/3; r = a - 3 q0 is exact in an fma; a/3 = q0 + r/3 exactly, and q0 + r y differs from
// it by |r/3| 2^-54 < 2^-106 |a|.  a/3 is never closer than ulp/6 to the midpoint of two doubles (a - 3m is a
// non-zero multiple of ulp/2 for a midpoint m), so rounding q0 + r y — one rounding, in the fma — gives
// RN(a/3).  r == 0 means q0 is the quotient itself (this keeps the sign of a zero).  Outside the exponent
// range where none of this can underflow or overflow the caller divides (`safe` false).
__device__ __forceinline__ double lex_div3(double a, bool &safe)
{
    const double y = 0x1.5555555555555p-2;
    const double aa = fabs(a);
    safe = (aa >= 0x1p-900 && aa <= 0x1p1000) || a == 0.0;
    const double q0 = a * y;
    const double r = __builtin_fma(-3.0, q0, a);
    const double q1 = __builtin_fma(r, y, q0);
    return r == 0.0 ? q0 : q1;
}

// Blocks [db0, db1] (steps db0 .. db1+7) of one wave with every step in body A (BORDER = false) or B (true).
// ROLE 0: the group's first sweep (reads x), 1: a middle one, 2: its last (writes x).  Straight-line steps:
// nothing conditional around the loads, every address a running pointer, the prefetch rings in registers with
// static indices — the 8 steps of a block are the 8 slots of the LDS result ring and of the prefetch rings.
//   pb     this lane's element of the b slice of row db0+1 (lanes beyond the slice repeat its last element)
//   pg     lanes 0..15: the left strip's edge value of step db0 + lane/2, edge lane%2 (the ghost lanes' input)
//   px_dn  ROLE 0: x one row below this lane's pixel — lane 0 (a ghost lane) fetches the column right of lane 63
//          instead, so `right` is `down` rotated by one lane
template <int T, bool CHECK, int ROLE, bool BORDER>
__device__ __forceinline__ void lex_wg_run(LexWgWave &w, double (*ring)[kLexRing][kWave], double (*brow)[LexWgShape<T>::kBW],
                                           double (*gring)[8][2], int t, int lane, int db0, int db1, const double *pb, const double *pg,
                                           bool ghost_live, const double *px_dn, long dn_stride, const double *px_old, long old_stride, double *ps,
                                           double *pe, long P, bool lane_on, Stencil st_b)
{
    constexpr int SL = LexWgShape<T>::kSlice;
    const bool ghost = lane < 2;
    const int lds1 = max(lane - 1, 0), lds2 = max(lane - 2, 0);
    const int ci = max(lane - 2 - 2 * t + 2 * (T - 1), 0);                   // this lane's column of a b row in LDS
    const int ld_col = t * SL + min(lane, SL - 1);
    // body B: this lane's kind of row (st_b: classify() of its column at an interior y)
    const bool c_off = !lane_on || st_b.diag == 0;
    const bool c_x0 = !c_off && !st_b.left, c_xl = !c_off && !st_b.right;
    const bool any_x0 = BORDER && __any(c_x0), any_xl = BORDER && __any(c_xl);
    double qb[8], qd[8], qo[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        qb[j] = lex_ld(pb);
        pb += P;
        qd[j] = qo[j] = 0.0;
        if (ROLE == 0) {
            qd[j] = lex_ld(px_dn);
            if (CHECK) qo[j] = lex_ld(px_old);
            px_dn += dn_stride;
            px_old += old_stride;
        }
    }
    {                                                                        // the ghost lanes' values of block db0
        const double g0 = lex_ld(pg);
        pg += 8 * 2 * T;
        if (lane < 16) gring[t][lane >> 1][lane & 1] = ghost_live ? g0 : 0.0;
    }
    for (int db = db0; db <= db1; db += 8) {
        {                                                                    // inputs of this block published?
            const int need = db + w.need_off;
            bool ok = (int)min(w.known, 0x7fffffffu) >= need;
            while (!__all(ok)) {
                if (!ok) {
                    w.known = __hip_atomic_load(w.watch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = (int)min(w.known, 0x7fffffffu) >= need;
                }
                if (!__all(ok)) __builtin_amdgcn_s_sleep(4);
            }
        }
        // issued here, looked at after the block's last step: no loaded value but the prefetch slots lives across
        // the loop's back edge (one that does is waited for there with vmcnt(0), draining the slots with it)
        const unsigned polled = __hip_atomic_load(w.watch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double qg = lex_ld(pg);                                        // the ghost lanes' values of block db + 8
        pg += 8 * 2 * T;
        const int sb = (db - 4 * t) & (kLexBRows - 1);
        int tag = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const double bl = qb[j];                                         // slice of row d + 1
            if (lane < SL) brow[(db + j + 1) & (kLexBRows - 1)][ld_col] = bl;
            if (j == 7) asm volatile("v_mov_b32 %0, %1" : "=v"(tag) : "v"(__double2loint(bl)));   // (a copy, made before the slot is refilled)
            double down = qd[j], old = qo[j], right;
            if (ROLE == 0) {
                right = lane_next_rot(down);
            } else {
                right = ring[t - 1][(j + 5) & 7][lds1];
                down = ring[t - 1][(j + 5) & 7][lds2];
                if (CHECK) old = ring[t - 1][(j + 4) & 7][lds2];
            }
            const double *src = ghost ? &gring[t][j][lane & 1] : &brow[(sb + j) & (kLexBRows - 1)][ci];
            const double vv = *src;
            const double up = w.h1;
            const double left = lane_prev(w.h1);
            double nv;
            bool wrote;
            nv = (vv + (((up + left) + right) + down)) * 0.25;               // (sparse-matrix.h:361-376 on a full row)
            wrote = !ghost;
            if (BORDER) {
                // 1 <= y <= H-2 for every lane: the row of a pixel depends on its column alone — column 0 has no
                // left neighbour (diagonal 3), column W-1 only its left one (diagonal 1), a 1-pixel-wide image
                // and the lanes off the image have no row at all
                if (any_x0) {
                    const double a = vv + ((up + right) + down);
                    bool safe;
                    double q = lex_div3(a, safe);
                    if (__any(c_x0 && !safe)) q = a / 3.0;
                    nv = c_x0 ? q : nv;
                }
                if (any_xl) nv = c_xl ? vv + left : nv;
                nv = c_off ? 0.0 : nv;
                wrote = !ghost && !c_off;
            }
            nv = ghost ? vv : nv;
            if (CHECK) w.acc += wrote ? fabs(nv - old) : 0.0;
            if (ROLE == 2) {
                // body A stores from the ghost lanes too: they carry the left strip's results of the same sweep
                // for exactly these pixels, so it is the value already there — and a store that is not
                // conditional keeps the compiler's count of operations in flight (the vmcnt waits) exact
                if (!BORDER || wrote) lex_st(ps, nv);
                ps += P;
            }
            ring[t][j][lane] = nv;
            w.h1 = nv;
            // the slots are refilled only now, when their old contents are dead: a load issued while the old
            // value is still live lands in another register and costs a copy — and a full vmcnt drain — at the
            // loop's back edge
            asm volatile("" ::: "memory");
            qb[j] = lex_ld(pb);                                              // ... of row d + 9
            pb += P;
            if (ROLE == 0) {
                qd[j] = lex_ld(px_dn);
                if (CHECK) qo[j] = lex_ld(px_old);
                px_dn += dn_stride;
                px_old += old_stride;
            }
            lex_lds_barrier();
        }
        if (lane < 16) lex_st(pe, ring[t][lane >> 1][kWave - 2 + (lane & 1)]);            // the block's 8 x 2 edge values
        pe += 8 * 2 * T;
        if (lane < 16) gring[t][lane >> 1][lane & 1] = ghost_live ? qg : 0.0;              // (this block's were read in its steps)
        w.known = max(w.known, polled);
        // Vector-memory operations complete in issue order: the slice written in the last step was loaded in
        // step db-1, so every store of the steps up to db-2 has been acknowledged — publish those, no drain.
        asm volatile("" ::"v"(tag) : "memory");
        if (lane == 0 && db > db0 && db - 1 > 0) __hip_atomic_store(w.mine, (unsigned)(db - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int T, bool CHECK>
__global__ void __launch_bounds__(T * kWave)
k_lex_wg(double *__restrict__ xd, const double *__restrict__ bd, Geom g, LexGeom lg, int G, int S,
         unsigned *__restrict__ progress, unsigned *__restrict__ ticket, const unsigned *__restrict__ order,
         double *__restrict__ edges, long edge_steps, unsigned active_mask, double *__restrict__ partial, long partial_stride)
{
    static_assert(kLexRing == 8 && T >= 2, "the unrolled step index is the ring slot; wave 0 reads x, wave T-1 writes it");
    static_assert(4 * (T - 1) + 4 <= kLexBRows, "a b row stays in LDS from step r-1 to step r+4(T-1)");
    constexpr int SL = LexWgShape<T>::kSlice, BW = LexWgShape<T>::kBW;
    __shared__ double ring[T][kLexRing][kWave];
    __shared__ double brow[kLexBRows][BW];
    __shared__ double gring[T][8][2];
    __shared__ unsigned s_ticket;
    const int ch = blockIdx.y;
    if (!((active_mask >> ch) & 1u)) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int t = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));      // this wave's sweep inside the group
    if (threadIdx.x == 0) s_ticket = atomicAdd(&ticket[ch], 1u);
#pragma unroll
    for (int q = 0; q < kLexRing; ++q) ring[t][q][lane] = 0.0;
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
    // the b slice this wave loads: columns cb + t*SL + lane of a row (cb: the leftmost column any sweep touches)
    const int b_col = xs0 + 2 - 2 * (T - 1) + t * SL + min(lane, SL - 1);
    const int b_col_c = min(max(b_col, 0), lg.W - 1);        // (clamped: what lies outside the image is never used)

    // progress is kept per wave: wave t of a strip feeds wave t of the strip to its right (edge values) and
    // wave T-1 feeds wave 0 of the next group (x).  Lane 0 watches the left strip, lanes 1 and 2 of wave 0 the
    // two strips of the previous group this one reads x from.
    LexWgWave w;
    w.mine = progress + ((((long)ch * G + grp) * S + s) * T + t);
    w.watch = w.mine;
    w.need_off = INT_MIN / 2;
    if (lane == 0 && s > 0) {
        w.watch = w.mine - T;
        w.need_off = 16;                                     // before block [db, db+7], prefetching to db+15
    }
    if (t == 0 && grp > 0 && (lane == 1 || (lane == 2 && s + 1 < S))) {
        w.watch = progress + ((((long)ch * G + grp - 1) * S + s + (lane - 1)) * T + (T - 1));
        w.need_off = 17 + 4 * (T - 1);
    }

    // blocks in which every real lane of this wave has 1 <= y <= H-2 at every step, prefetches included
    const int d_in_lo = xs0 + 64 + 2 * t, d_in_hi = xs0 + 2 * t + lg.H;
    const bool strip_interior = s > 0 && xs0 + 2 - 2 * t >= 1 && xs0 + 63 - 2 * t <= lg.W - 2;
    const int run0 = (max(d_in_lo, d_begin) + 7) & ~7;
    const int run1 = (min(d_in_hi - 15, d_end - 7)) & ~7;                   // last block of the run (may be < run0: none)

    auto general_block = [&](int db) {                                       // C: one step at a time, nothing in flight
        {
            const int need = db + w.need_off;
            bool ok = (int)min(w.known, 0x7fffffffu) >= need;
            while (!__all(ok)) {
                if (!ok) {
                    w.known = __hip_atomic_load(w.watch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = (int)min(w.known, 0x7fffffffu) >= need;
                }
                if (!__all(ok)) __builtin_amdgcn_s_sleep(4);
            }
        }
#pragma unroll 1
        for (int d = db; d < db + 8; ++d) {
            if (d < d_begin || d > d_end) continue;                          // (uniform over the workgroup)
            if (lane < SL) {                                                 // this wave's share of b row d + 1
                const int r = d + 1;
                brow[r & (kLexBRows - 1)][t * SL + lane] = (r >= 0 && r < lg.n_diag && b_col >= 0 && b_col < lg.W) ? bd[plane + (long)r * lg.P + b_col] : 0.0;
            }
            const int yp = d - xp, y = yp - 2 * t;
            double right = 0.0, down = 0.0, old = 0.0, vv = 0.0;
            if (t == 0) {                                     // sweep 0's inputs from x (the previous group's result)
                if (ok_dn && yp + 1 >= 0 && yp + 1 < lg.H) down = lex_ld(&xd[plane + (long)(d + 1) * lg.P + xp]);
                if (ok_rt && yp >= 0 && yp < lg.H) right = lex_ld(&xd[plane + (long)(d + 1) * lg.P + xp + 1]);
                if (CHECK && ok_dn && yp >= 0 && yp < lg.H) old = lex_ld(&xd[plane + (long)d * lg.P + xp]);
            } else {
                right = ring[t - 1][(d - 3) & 7][lds1];
                down = ring[t - 1][(d - 3) & 7][lds2];
                if (CHECK) old = ring[t - 1][(d - 4) & 7][lds2];
            }
            const bool on = lane_on && y >= 0 && y < lg.H;
            if (ghost) {
                if (ghost_live && d >= left_begin && d <= left_end) vv = lex_ld(&e_left[((long)(d - left_begin) * T + t) * 2 + lane]);
            } else if (on) {
                vv = bd[plane + (long)(xl + y) * lg.P + xl];
            }
            const double up = w.h1;
            const double left = lane_prev(w.h1);
            double nv = ghost ? vv : 0.0;
            if (on) {
                const Stencil st = classify(g, xl, y, y);
                if (st.diag != 0) {
                    (void)gs_update(st, vv, up, left, right, down, nv);
                    if (CHECK) w.acc += fabs(nv - old);
                    if (t == T - 1) lex_st(&xd[plane + (long)(xl + y) * lg.P + xl], nv);
                }
            }
            ring[t][d & 7][lane] = nv;
            if (lane >= kWave - 2) lex_st(&e_mine[((long)(d - d_begin) * T + t) * 2 + (lane - (kWave - 2))], nv);
            w.h1 = nv;
            lex_lds_barrier();
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");               // compiler ordering only
        __builtin_amdgcn_s_waitcnt(0);                                       // this wave's (write-through) stores acknowledged
        if (lane == 0 && db + 7 < d_end) __hip_atomic_store(w.mine, (unsigned)max(db + 8, 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };

    int db = d_base;
    for (; db <= d_end && (db < run0 || run1 < run0); db += 8) general_block(db);
    if (run1 >= run0) {
        // running pointers of bodies A / B at step run0; lanes with nothing to load read their own edge slot
        // (stride 0, value unused) so that no load of the run is conditional
        const double *pb = bd + plane + (long)(run0 + 1) * lg.P + b_col_c;
        const int gl = min(lane, 15);
        const double *pg = s > 0 ? e_left + ((long)(run0 - left_begin + (gl >> 1)) * T + t) * 2 + (gl & 1) : e_mine;
        const bool dn_live = lane == 0 ? xs0 + 64 < lg.W : ok_dn;            // lane 0: the column right of lane 63's
        const double *px_dn = !dn_live ? e_mine : xd + plane + (long)(run0 + 1) * lg.P + (lane == 0 ? xs0 + 64 : xp);
        const double *px_old = ok_dn ? xd + plane + (long)run0 * lg.P + xp : e_mine;
        const long dn_stride = dn_live ? (long)lg.P : 0, old_stride = ok_dn ? (long)lg.P : 0;
        double *ps = xd + plane + (long)(run0 - 4 * t) * lg.P + xl;
        double *pe = e_mine + ((long)(run0 - d_begin + (lane >> 1)) * T + t) * 2 + (lane & 1);    // lanes 0..15: step lane/2, edge lane%2
        const Stencil st_b = classify(g, lane_on ? xl : 0, 1, 1);
        const bool gl_live = s > 0;
#define CCP_LEX_WG_RUN(ROLE, BORDER) \
    lex_wg_run<T, CHECK, ROLE, BORDER>(w, ring, brow, gring, t, lane, run0, run1, pb, pg, gl_live, px_dn, dn_stride, px_old, old_stride, ps, pe, lg.P, lane_on, st_b)
        if (strip_interior) {
            if (t == 0) CCP_LEX_WG_RUN(0, false);
            else if (t == T - 1) CCP_LEX_WG_RUN(2, false);
            else CCP_LEX_WG_RUN(1, false);
        } else {
            if (t == 0) CCP_LEX_WG_RUN(0, true);
            else if (t == T - 1) CCP_LEX_WG_RUN(2, true);
            else CCP_LEX_WG_RUN(1, true);
        }
#undef CCP_LEX_WG_RUN
        for (db = run1 + 8; db <= d_end; db += 8) general_block(db);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                   // compiler ordering only
    __builtin_amdgcn_s_waitcnt(0);                                           // this wave's (write-through) stores acknowledged
    if (lane == 0) __hip_atomic_store(w.mine, kLexDone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (CHECK) {
        const double total = wave_sum(w.acc);
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
