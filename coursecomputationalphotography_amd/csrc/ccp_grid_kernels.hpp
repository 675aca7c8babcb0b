// ccp_grid_kernels.hpp — matrix-free red-black Gauss-Seidel / SpMV kernels for the structured
// 5-point Poisson system of SolveChannel (reference: project/src/PhotoMontage/
// PhotoMontage.cpp:541-597; closed form in SURVEY.md §8a-8), hand-written for gfx950.
//
// Data layout ("row-split colour planes"): every image row is stored de-interleaved, its
// colour-0 pixels first and its colour-1 pixels second, each half padded to `pitch` doubles:
//     elem(ch, l, c, j) = base[((ch*local_rows + l)*2 + c)*pitch + j],   c=(x+y)&1, j=x>>1.
// A half-sweep of colour c streams the opposite colour's plane rows l-1,l,l+1 plus b's colour
// c row and writes x's colour c row with fully coalesced 16-byte lane accesses: 24 algorithmic
// bytes per pixel update (b 8 + x neighbour plane 8 + x write 8).
//
// Arithmetic follows the reference sweep (project/src/PhotoMontage/sparse-matrix.h:359-374)
// operation for operation on the colour-major permuted matrix: sigma accumulates the
// neighbours in storage order up, left, right, down; x = (b - sigma) / a_ii.  Because every
// off-diagonal is exactly -1, b - sigma == b + (((xu + xl) + xr) + xd) bit for bit, and
// division by 4/2/1 equals multiplication by the exact reciprocal.  Compile with
// -ffp-contract=off (no FMA contraction) — the build enforces it.
#pragma once

#include "ccp_common.hpp"

namespace ccp {

struct Geom {
    int W, H;            // whole image
    int y0;              // image row of local row 0
    int local_rows;      // ghost_top + owned + ghost_bottom
    int own_lo, own_hi;  // owned local rows [own_lo, own_hi)
    long pitch;          // doubles per colour half-row
    long ch_stride;      // doubles per channel = local_rows*2*pitch
};

__device__ __forceinline__ long row_off(const Geom &g, int l, int c)
{
    return ((long)l * 2 + c) * g.pitch;
}

template <int CPT>
__device__ __forceinline__ void ld_vec(const double *__restrict__ p, double (&v)[CPT])
{
    static_assert(CPT % 2 == 0, "CPT must be even (16-byte lane accesses)");
#pragma unroll
    for (int k = 0; k < CPT; k += 2) {
        const double2 t = *reinterpret_cast<const double2 *>(p + k);
        v[k] = t.x;
        v[k + 1] = t.y;
    }
}

template <int CPT>
__device__ __forceinline__ void st_vec(double *__restrict__ p, const double (&v)[CPT])
{
#pragma unroll
    for (int k = 0; k < CPT; k += 2) {
        double2 t;
        t.x = v[k];
        t.y = v[k + 1];
        *reinterpret_cast<double2 *>(p + k) = t;
    }
}

template <int CPT>
__device__ __forceinline__ void zero_vec(double (&v)[CPT])
{
#pragma unroll
    for (int k = 0; k < CPT; ++k) v[k] = 0.0;
}

// Which neighbours pixel (x,y) has in the reference matrix and its diagonal:
// cell(x,y) <=> x < W-1 && y < H-1 (the forward-difference loop bounds, PhotoMontage.cpp:551-554).
struct Stencil {
    bool up, left, right, down;
    int diag;
};

__device__ __forceinline__ Stencil classify(const Geom &g, int x, int y, int l)
{
    Stencil s;
    const bool here = (x < g.W - 1) && (y < g.H - 1);
    const bool cf_up = (y >= 1) && (x < g.W - 1);
    s.left = (x >= 1) && (y < g.H - 1);
    s.right = here;
    s.diag = (int)cf_up + (int)s.left + 2 * (int)here + (int)((x | y) == 0);
    // a neighbour row outside the local block (beyond the ghosts) is treated as absent; such
    // rows are never inside the sweep range of a correctly driven handle.
    s.up = cf_up && (l >= 1);
    s.down = here && (l + 1 < g.local_rows);
    return s;
}

// (b - sigma) / a_ii with the reference's accumulation order.  Returns false when the row is
// skipped (a_ii == 0, sparse-matrix.h:361-363).
__device__ __forceinline__ bool gs_update(const Stencil &s, double bv, double xu, double xl,
                                          double xr, double xd, double &out)
{
    if (s.diag == 0) return false;
    double sigma = 0.0;
    if (s.up) sigma += -1.0 * xu;
    if (s.left) sigma += -1.0 * xl;
    if (s.right) sigma += -1.0 * xr;
    if (s.down) sigma += -1.0 * xd;
    out = (bv - sigma) / (double)s.diag;
    return true;
}

// A x for one pixel in applyToVector's order (sparse-matrix.h:382-393): up, left, diagonal,
// right, down; empty rows give 0.
__device__ __forceinline__ double apply_row(const Stencil &s, double xi, double xu, double xl,
                                            double xr, double xd)
{
    double sum = 0.0;
    if (s.up) sum += -1.0 * xu;
    if (s.left) sum += -1.0 * xl;
    if (s.diag != 0) sum += (double)s.diag * xi;
    if (s.right) sum += -1.0 * xr;
    if (s.down) sum += -1.0 * xd;
    return sum;
}

// ---------------------------------------------------------------------------------------------
// Half-sweep of colour c over local rows [l_lo, l_hi).  Each thread owns CPT adjacent
// half-columns and marches down `rows_per_block` rows, keeping the three opposite-colour rows
// it needs in registers so every neighbour value is fetched from memory once per thread.
// grid = (ceil(pitch / (kBlock*CPT)), ceil(rows / rows_per_block), channels).
// L1: also accumulate sum |x_new - x_old| over the OWNED rows into partial[] (one double per
// block at [(ch*gridDim.y + by)*gridDim.x + bx]; the host passes a per-colour region) — the
// reference's manhattonDist(x, prev) (sparse-matrix.h:376), in a deterministic order.
//
// The one horizontal neighbour outside the thread's own CPT half-columns comes from the
// adjacent lane's registers (SHFL: __shfl_up/__shfl_down, no memory traffic); only the two edge
// lanes of a wave fetch it, together with the row's vector load, one row ahead.  (The first
// version re-loaded it per lane a row later: by then the line had left the 4 MiB L2 and the
// kernel fetched 1.43x its algorithmic read bytes — profiles/r01_half_sweep_v0_*.)
// MASKED (Dirichlet-mask grid, ccp_grid_set_mask_host): the uniform 5-point stencil with a_ii = 4 on the
// pixels whose mask byte is 1; every other pixel — and everything outside the image — is fixed at zero, so
// every pixel takes the interior arithmetic and a masked-out one is written back as 0.
template <int CPT, bool L1, bool SHFL, bool MASKED = false>
__global__ void __launch_bounds__(kBlock)
k_half_sweep(const double *__restrict__ xr, double *__restrict__ xw, const double *__restrict__ b,
             Geom g, int c, int l_lo, int l_hi, int rows_per_block,
             double *__restrict__ partial, const int *__restrict__ active,
             const unsigned char *__restrict__ mask = nullptr)
{
    __shared__ double scratch[kBlock / kWave];
    const int ch = blockIdx.z;
    double acc = 0.0;
    const bool run = (active == nullptr) || (active[ch] != 0);
    const int j0 = (blockIdx.x * kBlock + threadIdx.x) * CPT;
    const int la = l_lo + blockIdx.y * rows_per_block;
    const int lb = min(la + rows_per_block, l_hi);
    if (run && j0 < g.pitch && la < lb) {
        const int o = 1 - c;
        const double *__restrict__ xo = xr + (long)ch * g.ch_stride;
        double *__restrict__ xc = xw + (long)ch * g.ch_stride;
        const double *__restrict__ bc = b + (long)ch * g.ch_stride;
        double up[CPT], mid[CPT], dn[CPT];
        if (la >= 1) ld_vec<CPT>(xo + row_off(g, la - 1, o) + j0, up);
        else zero_vec<CPT>(up);
        ld_vec<CPT>(xo + row_off(g, la, o) + j0, mid);
        const int lane = threadIdx.x & (kWave - 1);
        // edge value of the wave for the row held in `mid` / `dn` (SHFL only)
        double edge_mid = 0.0, edge_dn = 0.0;
        if (SHFL) {
            const int p0 = (g.y0 + la + c) & 1;
            const int je = p0 ? j0 + CPT : j0 - 1;
            if (lane == (p0 ? kWave - 1 : 0) && je >= 0 && je < g.pitch) edge_mid = xo[row_off(g, la, o) + je];
        }
#pragma unroll 2
        for (int l = la; l < lb; ++l) {
            if (l + 1 < g.local_rows) ld_vec<CPT>(xo + row_off(g, l + 1, o) + j0, dn);
            else zero_vec<CPT>(dn);
            const int y = g.y0 + l;
            const int p = (y + c) & 1;                 // own pixels sit at image x = 2j + p
            double side = 0.0;
            if (SHFL) {
                const int pn = p ^ 1;                  // parity of the next row
                const int je = pn ? j0 + CPT : j0 - 1;
                edge_dn = 0.0;
                if (l + 1 < g.local_rows && lane == (pn ? kWave - 1 : 0) && je >= 0 && je < g.pitch)
                    edge_dn = xo[row_off(g, l + 1, o) + je];
                const double from_right = __shfl_down(mid[0], 1, kWave);
                const double from_left = __shfl_up(mid[CPT - 1], 1, kWave);
                side = p ? from_right : from_left;
                if (lane == (p ? kWave - 1 : 0)) side = edge_mid;
            } else {
                const int js = p ? j0 + CPT : j0 - 1;  // the one neighbour outside [j0, j0+CPT)
                if (js >= 0 && js < g.pitch) side = xo[row_off(g, l, o) + js];
            }
            double bv[CPT], old[CPT], nv[CPT];
            const long own = row_off(g, l, c) + j0;
            ld_vec<CPT>(bc + own, bv);
            if (L1) ld_vec<CPT>(xc + own, old);
            const int x_first = 2 * j0 + p;
            const int x_last = x_first + 2 * (CPT - 1);
            const bool owned = (l >= g.own_lo) && (l < g.own_hi);
            const bool interior = MASKED || ((y >= 1) && (y <= g.H - 2) && (l >= 1) && (l + 1 < g.local_rows) &&
                                             (x_first >= 1) && (x_last <= g.W - 2));
            if (interior) {
#pragma unroll
                for (int k = 0; k < CPT; ++k) {
                    const double left = p ? mid[k] : (k == 0 ? side : mid[k > 0 ? k - 1 : 0]);
                    const double right = p ? (k == CPT - 1 ? side : mid[k < CPT - 1 ? k + 1 : k]) : mid[k];
                    // b - sigma with sigma = (((-xu) + (-xl)) + (-xr)) + (-xd); a_ii = 4
                    nv[k] = (bv[k] + (((up[k] + left) + right) + dn[k])) * 0.25;
                    if (MASKED && mask[row_off(g, l, c) + j0 + k] == 0) nv[k] = 0.0;
                    if (L1 && owned) acc += fabs(nv[k] - old[k]);
                }
                st_vec<CPT>(xc + own, nv);
            } else {
#pragma unroll
                for (int k = 0; k < CPT; ++k) {
                    const int x = x_first + 2 * k;
                    if (x < g.W) {
                        const double left = p ? mid[k] : (k == 0 ? side : mid[k > 0 ? k - 1 : 0]);
                        const double right = p ? (k == CPT - 1 ? side : mid[k < CPT - 1 ? k + 1 : k]) : mid[k];
                        const Stencil s = classify(g, x, y, l);
                        double r;
                        if (gs_update(s, bv[k], up[k], left, right, dn[k], r)) {
                            xc[own + k] = r;
                            if (L1 && owned) acc += fabs(r - old[k]);
                        }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                up[k] = mid[k];
                mid[k] = dn[k];
            }
            edge_mid = edge_dn;
        }
    }
    if (L1) {
        const double total = block_sum(acc, scratch);
        if (threadIdx.x == 0)
            partial[((long)ch * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = total;
    }
}

// Per-channel solver state kept on the device so the stop rule needs no host round trip.
struct SolveState {
    int active[kMaxChannels];       // 1 while the channel still iterates
    int converged[kMaxChannels];
    int iterations[kMaxChannels];   // sweep index at which the stop rule fired
    double last_eps[kMaxChannels];
};

// eps = sum of the red and black partials of one checked sweep; apply the reference stop rule
// `while (eps > epsilon && cnt < max_iteration)` (sparse-matrix.h:356).  grid = channels.
// partial0/partial1: the red / black regions, blocks0/blocks1 block results per channel in each.
// st == nullptr: only store the sums to out[ch] (row-blocked callers all-reduce them).
__global__ void __launch_bounds__(kBlock)
k_check(const double *__restrict__ partial0, long blocks0, const double *__restrict__ partial1,
        long blocks1, double epsilon, int sweep_index, SolveState *__restrict__ st,
        double *__restrict__ out)
{
    __shared__ double scratch[kBlock / kWave];
    const int ch = blockIdx.x;
    double acc = 0.0;
    for (long i = threadIdx.x; i < blocks0; i += kBlock) acc += partial0[(long)ch * blocks0 + i];
    for (long i = threadIdx.x; i < blocks1; i += kBlock) acc += partial1[(long)ch * blocks1 + i];
    const double eps = block_sum(acc, scratch);
    if (threadIdx.x == 0 && out) out[ch] = eps;
    if (threadIdx.x == 0 && st && st->active[ch]) {
        st->last_eps[ch] = eps;
        if (!(eps > epsilon)) {
            st->active[ch] = 0;
            st->converged[ch] = 1;
            st->iterations[ch] = sweep_index;
        }
    }
}

// A temporally blocked pass that reported the step of EACH of its T sweeps (partial[(t*channels + ch)*blocks + i], one
// region per launch of the pass: the ordinary and the border launch): k_sweep_sums_wide adds up the step of every sweep
// (sums[t*channels + ch]; a row block all-reduces them over the ranks), k_decide_sums applies the rule in sweep order — the
// first sweep whose step is not above epsilon stops the channel, exactly where the reference loop would have stopped.
// A block per (channel, sweep): grid = (channels, T) — the T sweeps side by side (one block doing them one after the other
// took 85 us per pass of 2 ms at 16384^2: kernel stats, round 4).
__global__ void __launch_bounds__(kBlock)
k_sweep_sums_wide(const double *__restrict__ partial0, long blocks0, const double *__restrict__ partial1, long blocks1,
                  double *__restrict__ sums)
{
    __shared__ double scratch[kBlock / kWave];
    const int ch = blockIdx.x, t = blockIdx.y;
    const int channels = gridDim.x;
    double acc = 0.0;
    const double *__restrict__ p = partial0 + ((long)t * channels + ch) * blocks0;
    for (long i = threadIdx.x; i < blocks0; i += kBlock) acc += p[i];
    const double *__restrict__ q = partial1 + ((long)t * channels + ch) * blocks1;
    for (long i = threadIdx.x; i < blocks1; i += kBlock) acc += q[i];
    const double eps = block_sum(acc, scratch);
    if (threadIdx.x == 0) sums[t * channels + ch] = eps;
}

__global__ void k_decide_sums(const double *__restrict__ sums, int channels, int T, int first_sweep_index, int every, double epsilon,
                              SolveState *__restrict__ st)
{
    const int ch = threadIdx.x;
    if (ch >= channels) return;
    for (int t = 0; t < T; ++t) {
        if ((first_sweep_index + t) % every != 0) continue;
        if (!st->active[ch]) break;
        const double eps = sums[t * channels + ch];
        st->last_eps[ch] = eps;
        if (!(eps > epsilon)) {
            st->active[ch] = 0;
            st->converged[ch] = 1;
            st->iterations[ch] = first_sweep_index + t;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// SpMV-shaped kernels on the same layout.  One thread per CPT half-columns of one (row, colour).
// MODE 0: b := A x (applyToVector).  MODE 1: partial sums of (b - A x)^2 and b^2 (residual).
// MODE 2: b := A x and one partial sum of x'(A x) per block (the p'Ap of conjugateGradient,
// sparse-matrix.h:419-420, fused into the SpMV pass).
// grid = (ceil(pitch/(kBlock*CPT)), rows, channels*2); rows l in [l_lo, l_hi).
template <int CPT, int MODE, bool MASKED = false>
__global__ void __launch_bounds__(kBlock)
k_apply(const double *__restrict__ x, double *__restrict__ bw, const double *__restrict__ br,
        Geom g, int l_lo, double *__restrict__ partial, const unsigned char *__restrict__ mask = nullptr)
{
    __shared__ double scratch[kBlock / kWave];
    const int ch = blockIdx.z >> 1;
    const int c = blockIdx.z & 1;
    const int o = 1 - c;
    const int l = l_lo + blockIdx.y;
    const int j0 = (blockIdx.x * kBlock + threadIdx.x) * CPT;
    double rr = 0.0, bb = 0.0;
    if (j0 < g.pitch) {
        const double *__restrict__ xc = x + (long)ch * g.ch_stride;
        double up[CPT], mid[CPT], dn[CPT], own[CPT];
        if (l >= 1) ld_vec<CPT>(xc + row_off(g, l - 1, o) + j0, up);
        else zero_vec<CPT>(up);
        ld_vec<CPT>(xc + row_off(g, l, o) + j0, mid);
        if (l + 1 < g.local_rows) ld_vec<CPT>(xc + row_off(g, l + 1, o) + j0, dn);
        else zero_vec<CPT>(dn);
        ld_vec<CPT>(xc + row_off(g, l, c) + j0, own);
        const int y = g.y0 + l;
        const int p = (y + c) & 1;
        const int js = p ? j0 + CPT : j0 - 1;
        double side = 0.0;
        if (js >= 0 && js < g.pitch) side = xc[row_off(g, l, o) + js];
        const long at = (long)ch * g.ch_stride + row_off(g, l, c) + j0;
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int xi = 2 * (j0 + k) + p;
            if (xi < g.W) {
                const double left = p ? mid[k] : (k == 0 ? side : mid[k > 0 ? k - 1 : 0]);
                const double right = p ? (k == CPT - 1 ? side : mid[k < CPT - 1 ? k + 1 : k]) : mid[k];
                double ax;
                if (MASKED) {
                    // the row of an unknown: -1 to its four neighbours (zero where there is no unknown), 4 on the
                    // diagonal, applyToVector's order; a pixel fixed at zero has an empty row
                    ax = 0.0;
                    if (mask[row_off(g, l, c) + j0 + k] != 0) {
                        ax += -1.0 * up[k];
                        ax += -1.0 * left;
                        ax += 4.0 * own[k];
                        ax += -1.0 * right;
                        ax += -1.0 * dn[k];
                    }
                } else {
                    const Stencil s = classify(g, xi, y, l);
                    ax = apply_row(s, own[k], up[k], left, right, dn[k]);
                }
                if (MODE == 0) {
                    bw[at + k] = ax;
                } else if (MODE == 2) {
                    bw[at + k] = ax;
                    rr += own[k] * ax;
                } else {
                    const double bv = br[at + k];
                    const double r = bv - ax;          // vecsub(b, Ax) (sparse-matrix.h:75-79)
                    rr += r * r;                       // veclen2 (sparse-matrix.h:51-55)
                    bb += bv * bv;
                }
            }
        }
    }
    if (MODE == 1) {
        const double t0 = block_sum(rr, scratch);
        const double t1 = block_sum(bb, scratch);
        if (threadIdx.x == 0) {
            const long blk = ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
            partial[2 * blk] = t0;
            partial[2 * blk + 1] = t1;
        }
    }
    if (MODE == 2) {
        const double t0 = block_sum(rr, scratch);
        if (threadIdx.x == 0)
            partial[((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = t0;
    }
}

// Sum `count` strided pairs per (channel, colour) group into out[channel*2 + {0,1}]:
// out[2*ch] = sum rr, out[2*ch+1] = sum bb.  grid = channels.
__global__ void __launch_bounds__(kBlock)
k_pair_reduce(const double *__restrict__ partial, long blocks_per_group, double *__restrict__ out)
{
    __shared__ double scratch[kBlock / kWave];
    const int ch = blockIdx.x;
    double a0 = 0.0, a1 = 0.0;
    // groups 2*ch (colour 0) and 2*ch+1 (colour 1) are adjacent in the partial array
    const double *__restrict__ p = partial + 2 * ((long)2 * ch) * blocks_per_group;
    for (long i = threadIdx.x; i < 2 * blocks_per_group; i += kBlock) {
        a0 += p[2 * i];
        a1 += p[2 * i + 1];
    }
    const double t0 = block_sum(a0, scratch);
    const double t1 = block_sum(a1, scratch);
    if (threadIdx.x == 0) {
        out[2 * ch] = t0;
        out[2 * ch + 1] = t1;
    }
}

// sum |x| over owned rows, per channel (checksum helper).  grid = (blocks, 1, channels).
__global__ void __launch_bounds__(kBlock)
k_abs_sum(const double *__restrict__ x, Geom g, double *__restrict__ partial)
{
    __shared__ double scratch[kBlock / kWave];
    const int ch = blockIdx.z;
    const long per_row = 2 * g.pitch;
    const long total = (long)(g.own_hi - g.own_lo) * per_row;
    double acc = 0.0;
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long)gridDim.x * kBlock) {
        const int l = g.own_lo + (int)(i / per_row);
        const long rem = i % per_row;
        const int c = (int)(rem / g.pitch);
        const int j = (int)(rem % g.pitch);
        const int xi = 2 * j + ((g.y0 + l + c) & 1);
        if (xi < g.W) acc += fabs(x[(long)ch * g.ch_stride + (long)l * per_row + rem]);
    }
    const double t = block_sum(acc, scratch);
    if (threadIdx.x == 0) partial[(long)ch * gridDim.x + blockIdx.x] = t;
}

__global__ void __launch_bounds__(kBlock)
k_sum_reduce(const double *__restrict__ partial, long count, double *__restrict__ out)
{
    __shared__ double scratch[kBlock / kWave];
    const int ch = blockIdx.x;
    double acc = 0.0;
    for (long i = threadIdx.x; i < count; i += kBlock) acc += partial[(long)ch * count + i];
    const double t = block_sum(acc, scratch);
    if (threadIdx.x == 0) out[ch] = t;
}

// ---------------------------------------------------------------------------------------------
// Layout conversion between natural raster rows (what std::vector<double> holds in the
// reference) and the row-split colour planes.  grid = (ceil(W/kBlock), rows).
template <bool TO_SPLIT>
__global__ void __launch_bounds__(kBlock)
k_convert(double *__restrict__ split, double *__restrict__ natural, Geom g, int ch, int l_first,
          int W)
{
    const int x = blockIdx.x * kBlock + threadIdx.x;
    const int r = blockIdx.y;
    if (x >= W) return;
    const int l = l_first + r;
    const int c = (x + g.y0 + l) & 1;
    const long s = (long)ch * g.ch_stride + row_off(g, l, c) + (x >> 1);
    const long n = (long)r * W + x;
    if (TO_SPLIT) split[s] = natural[n];
    else natural[n] = split[s];
}

__global__ void __launch_bounds__(kBlock) k_fill(double *__restrict__ p, long n, double v)
{
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock) p[i] = v;
}

// splitmix64 finaliser: value depends only on (seed, channel, x, y) -> identical for any row
// partition of the image.
__device__ __forceinline__ double hash_uniform(uint64_t seed, int ch, int x, int y)
{
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * ((((uint64_t)ch << 56) ^ ((uint64_t)(uint32_t)y << 28)) + (uint64_t)(uint32_t)x + 1ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

// grid = (ceil(pitch/kBlock), local_rows, channels*2)
__global__ void __launch_bounds__(kBlock)
k_randomize(double *__restrict__ x, Geom g, uint64_t seed, double lo, double hi)
{
    const int j = blockIdx.x * kBlock + threadIdx.x;
    const int l = blockIdx.y;
    const int ch = blockIdx.z >> 1, c = blockIdx.z & 1;
    if (j >= g.pitch) return;
    const int y = g.y0 + l;
    const int xi = 2 * j + ((y + c) & 1);
    double v = 0.0;
    if (xi < g.W) v = lo + (hi - lo) * hash_uniform(seed, ch, xi, y);
    x[(long)ch * g.ch_stride + row_off(g, l, c) + j] = v;
}

// ---------------------------------------------------------------------------------------------
// Poisson right-hand side ATb for all channels (PhotoMontage.cpp:563-572,579-581,592), Eigen's
// row-major sparse*dense accumulation order: gy above, gx left, -gx here, -gy here, pin.
// gx, gy: packed H x W x C float32 on device.  grid = (ceil(W/kBlock), H, C).
__global__ void __launch_bounds__(kBlock)
k_assemble_rhs(double *__restrict__ b, Geom g, const float *__restrict__ gx,
               const float *__restrict__ gy, int C, const int *__restrict__ constraint)
{
    const int x = blockIdx.x * kBlock + threadIdx.x;
    const int y = blockIdx.y;
    const int ch = blockIdx.z;
    if (x >= g.W) return;
    const long px = ((long)y * g.W + x) * C + ch;
    double acc = 0.0;
    if (y >= 1 && x < g.W - 1) acc += 1.0 * (double)gy[px - (long)g.W * C];
    if (x >= 1 && y < g.H - 1) acc += 1.0 * (double)gx[px - C];
    if (x < g.W - 1 && y < g.H - 1) {
        acc += -1.0 * (double)gx[px];
        acc += -1.0 * (double)gy[px];
    }
    if ((x | y) == 0) acc += 1.0 * (double)constraint[ch];
    const int l = y - g.y0;
    b[(long)ch * g.ch_stride + row_off(g, l, (x + y) & 1) + (x >> 1)] = acc;
}

// Gradient field + right-hand side in one pass, straight from the source images and the label
// map (BuildSolveGradientFusion, PhotoMontage.cpp:419-433): the forward differences of
// GradientAt (:399-408: integer subtraction of the label-selected image, then float) feed the
// same ATb accumulation as k_assemble_rhs, with the pin value Images[0](0,0)[ch] (:428).
// INIT: also write the composite image as the start vector (PhotoMontage.cpp:599-610).
// images: K stacked H x W x 3 u8 images, label: H x W u8.  grid = (ceil(W/kBlock), H, 3).
template <bool INIT>
__global__ void __launch_bounds__(kBlock)
k_assemble_from_images(double *__restrict__ b, double *__restrict__ x, Geom g,
                       const uint8_t *__restrict__ images, const uint8_t *__restrict__ label)
{
    const int xi = blockIdx.x * kBlock + threadIdx.x;
    const int y = blockIdx.y;
    const int ch = blockIdx.z;
    if (xi >= g.W) return;
    const long plane = (long)g.W * g.H * 3;
    auto pix = [&](int k, int yy, int xx) -> int { return (int)images[(long)k * plane + ((long)yy * g.W + xx) * 3 + ch]; };
    auto lab = [&](int yy, int xx) -> int { return (int)label[(long)yy * g.W + xx]; };
    double acc = 0.0;
    if (y >= 1 && xi < g.W - 1) {                       // gy(y-1, x) of the cell above
        const int k = lab(y - 1, xi);
        acc += 1.0 * (double)(float)(pix(k, y, xi) - pix(k, y - 1, xi));
    }
    if (xi >= 1 && y < g.H - 1) {                       // gx(y, x-1) of the cell to the left
        const int k = lab(y, xi - 1);
        acc += 1.0 * (double)(float)(pix(k, y, xi) - pix(k, y, xi - 1));
    }
    if (xi < g.W - 1 && y < g.H - 1) {                  // -gx(y,x) - gy(y,x) of this cell
        const int k = lab(y, xi);
        const int here = pix(k, y, xi);
        acc += -1.0 * (double)(float)(pix(k, y, xi + 1) - here);
        acc += -1.0 * (double)(float)(pix(k, y + 1, xi) - here);
    }
    if ((xi | y) == 0) acc += 1.0 * (double)pix(0, 0, 0);
    const long at = (long)ch * g.ch_stride + row_off(g, y - g.y0, (xi + y) & 1) + (xi >> 1);
    b[at] = acc;
    if (INIT) x[at] = (double)pix(lab(y, xi), y, xi);
}

// Solve epilogue: out(y,x)[ch] = uchar(max(min(sol,255),0)) (PhotoMontage.cpp:617-626).
__global__ void __launch_bounds__(kBlock)
k_store_u8(const double *__restrict__ x, Geom g, uint8_t *__restrict__ out, int C)
{
    const int xi = blockIdx.x * kBlock + threadIdx.x;
    const int y = blockIdx.y;
    const int ch = blockIdx.z;
    if (xi >= g.W) return;
    const int l = y - g.y0;
    double v = x[(long)ch * g.ch_stride + row_off(g, l, (xi + y) & 1) + (xi >> 1)];
    v = v < 255.0 ? v : 255.0;
    v = v > 0.0 ? v : 0.0;
    out[((long)y * g.W + xi) * C + ch] = (uint8_t)v;
}

// Composite initial guess: x(y,x)[ch] = image(y,x)[ch] (PhotoMontage.cpp:599-610).
__global__ void __launch_bounds__(kBlock)
k_load_u8(double *__restrict__ x, Geom g, const uint8_t *__restrict__ img, int C)
{
    const int xi = blockIdx.x * kBlock + threadIdx.x;
    const int y = blockIdx.y;
    const int ch = blockIdx.z;
    if (xi >= g.W) return;
    const int l = y - g.y0;
    x[(long)ch * g.ch_stride + row_off(g, l, (xi + y) & 1) + (xi >> 1)] =
        (double)img[((long)y * g.W + xi) * C + ch];
}

}  // namespace ccp
