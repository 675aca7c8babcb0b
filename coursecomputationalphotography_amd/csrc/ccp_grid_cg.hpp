// ccp_grid_cg.hpp — pass A of the fused conjugate-gradient iteration on the structured grid (ccp_cg.hpp:
// cg_solve_fused; reference loop: project/src/PhotoMontage/sparse-matrix.h:396-434).
//
//   x += alpha p_old          (:421 of the PREVIOUS iteration, deferred to this pass over p)
//   p_new = r + beta p_old    (:427 of the previous iteration)
//   Ap = A p_new              (:419; applyToVector's order, :382-393) with partial sums of p_new'Ap (:420)
//
// A thread owns CPT adjacent half-columns of colour c in row l (k_apply's tiling: one block per row tile and
// colour).  The p_new of its four neighbours — the opposite colour in rows l-1, l, l+1 — is recomputed from r and
// p_old with the same two roundings (mul, add; no contraction) as the thread that stores them, so every copy of a
// p_new value is the same double.  p is double-buffered: p_old stays readable while other workgroups store p_new.
// HBM bytes per unknown: x 16 + r 8 + p_old 8 + p_new 8 + Ap 8 = 48 (the neighbour reads hit L2).
#pragma once

#include "ccp_cg.hpp"
#include "ccp_grid_kernels.hpp"

namespace ccp {

// grid = (ceil(pitch/(kBlock*CPT)), rows, 2); one channel (the caller offsets the pointers).
template <int CPT, bool MASKED>
__global__ void __launch_bounds__(kBlock)
k_cg_apply_fused(double *__restrict__ x, const double *__restrict__ r, const double *__restrict__ p_old, double *__restrict__ p_new,
                 double *__restrict__ ap, Geom g, double *__restrict__ partial, const unsigned char *__restrict__ mask,
                 const CgState *__restrict__ st)
{
    __shared__ double scratch[kBlock / kWave];
    const int c = blockIdx.z & 1;
    const int o = 1 - c;
    const int l = blockIdx.y;
    const int j0 = (blockIdx.x * kBlock + threadIdx.x) * CPT;
    double dot = 0.0;
    if (st->active && j0 < g.pitch) {
        const double alpha = st->alpha, beta = st->beta;
        // direction of the opposite colour in rows l-1, l, l+1 (+ the one half-column beside the thread's own)
        auto dir = [&](long at, double (&out)[CPT]) {
            double rv[CPT], pv[CPT];
            ld_vec<CPT>(r + at, rv);
            ld_vec<CPT>(p_old + at, pv);
#pragma unroll
            for (int k = 0; k < CPT; ++k) out[k] = rv[k] + beta * pv[k];
        };
        double up[CPT], mid[CPT], dn[CPT], own[CPT];
        if (l >= 1) dir(row_off(g, l - 1, o) + j0, up);
        else zero_vec<CPT>(up);
        dir(row_off(g, l, o) + j0, mid);
        if (l + 1 < g.local_rows) dir(row_off(g, l + 1, o) + j0, dn);
        else zero_vec<CPT>(dn);
        const long at = row_off(g, l, c) + j0;
        double po[CPT], xo[CPT];
        ld_vec<CPT>(p_old + at, po);
        ld_vec<CPT>(x + at, xo);
        {
            double rv[CPT];
            ld_vec<CPT>(r + at, rv);
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                xo[k] = xo[k] + alpha * po[k];
                own[k] = rv[k] + beta * po[k];
            }
        }
        const int y = g.y0 + l;
        const int p = (y + c) & 1;
        const int js = p ? j0 + CPT : j0 - 1;
        double side = 0.0;
        if (js >= 0 && js < g.pitch) side = r[row_off(g, l, o) + js] + beta * p_old[row_off(g, l, o) + js];
        double ax[CPT];
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int xi = 2 * (j0 + k) + p;
            ax[k] = 0.0;
            if (xi < g.W) {
                const double left = p ? mid[k] : (k == 0 ? side : mid[k > 0 ? k - 1 : 0]);
                const double right = p ? (k == CPT - 1 ? side : mid[k < CPT - 1 ? k + 1 : k]) : mid[k];
                if (MASKED) {
                    if (mask[at + k] != 0) {
                        ax[k] += -1.0 * up[k];
                        ax[k] += -1.0 * left;
                        ax[k] += 4.0 * own[k];
                        ax[k] += -1.0 * right;
                        ax[k] += -1.0 * dn[k];
                    }
                } else {
                    const Stencil s = classify(g, xi, y, l);
                    ax[k] = apply_row(s, own[k], up[k], left, right, dn[k]);
                }
                dot += own[k] * ax[k];
            }
        }
        // the pad half-columns of a row hold zeros in x, r and p (and get 0 + alpha 0, 0 + beta 0, 0 back), so whole
        // vectors are stored
        st_vec<CPT>(x + at, xo);
        st_vec<CPT>(p_new + at, own);
        st_vec<CPT>(ap + at, ax);
    }
    const double t0 = block_sum(dot, scratch);
    if (threadIdx.x == 0) partial[((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = t0;
}

// The same pass as a MARCH (default): a thread owns CPT adjacent half-columns of BOTH colours and walks down
// `rows_per_block` image rows with the directions of rows l-1, l, l+1 in registers, so every r, p and x value is
// loaded exactly once (k_cg_apply_fused above re-reads each opposite-colour value from three rows and relies on L2):
// 24 B read + 24 B written per unknown is also what L2 sees.  The one horizontal neighbour outside a thread's own
// half-columns comes from the adjacent lane (one shuffle per side and row); the two edge lanes of a wave compute it
// from r and p_old themselves.  Same arithmetic per element; the partial sums of p'Ap are grouped differently, so
// alpha may differ from the row-per-block kernel's in the last bit.
// grid = (ceil(pitch / (kBlock*2)), ceil((row_end - row_first) / rows_per_block)); one channel (the caller offsets the pointers).
template <bool MASKED>
__global__ void __launch_bounds__(kBlock)
k_cg_apply_march(double *__restrict__ x, const double *__restrict__ r, const double *__restrict__ p_old, double *__restrict__ p_new,
                 double *__restrict__ ap, Geom g, int rows_per_block, double *__restrict__ partial, const unsigned char *__restrict__ mask,
                 const CgState *__restrict__ st, int row_first, int row_end)
{
    // rows [row_first, row_end) are updated (a row block: its owned rows; the rows next to them are read from the ghost
    // rows, which the caller has fetched from the neighbours for r and p_old)
    constexpr int CPT = 2;
    __shared__ double scratch[kBlock / kWave];
    const int j0 = (blockIdx.x * kBlock + threadIdx.x) * CPT;
    const int lane = (int)(threadIdx.x & (kWave - 1));
    const int l_lo = row_first + blockIdx.y * rows_per_block, l_hi = min(l_lo + rows_per_block, row_end);
    double dot = 0.0;
    if (st->active) {                                             // (uniform)
        const double alpha = st->alpha, beta = st->beta;
        const bool col_ok = j0 < g.pitch;
        // row l, both colours: direction d[c][k] = r + beta p_old, and p_old itself (for x's update)
        auto load_row = [&](int l, double (&d)[2][CPT], double (&po)[2][CPT]) {
            const bool on = col_ok && l >= 0 && l < g.local_rows;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if (on) {
                    double rv[CPT];
                    ld_vec<CPT>(r + row_off(g, l, c) + j0, rv);
                    ld_vec<CPT>(p_old + row_off(g, l, c) + j0, po[c]);
#pragma unroll
                    for (int k = 0; k < CPT; ++k) d[c][k] = rv[k] + beta * po[c][k];
                } else {
#pragma unroll
                    for (int k = 0; k < CPT; ++k) d[c][k] = po[c][k] = 0.0;
                }
            }
        };
        // the direction of colour c at half-column js of row l (one element), for a wave's edge lanes
        auto dir_at = [&](int l, int c, int js) -> double {
            if (js < 0 || js >= g.pitch) return 0.0;
            const long at = row_off(g, l, c) + js;
            return r[at] + beta * p_old[at];
        };
        double dm[2][CPT], d0[2][CPT], dp[2][CPT], po0[2][CPT], pop[2][CPT], pom[2][CPT];
        load_row(l_lo - 1, dm, pom);
        load_row(l_lo, d0, po0);
        for (int l = l_lo; l < l_hi; ++l) {
            load_row(l + 1, dp, pop);
            const int y = g.y0 + l;
            const int pr = y & 1;                                  // red pixel of half-column j: x = 2j + pr; black: x = 2j + 1 - pr
            // from the lane to the left: its last half-column of the colour whose pixels sit at x = 2j + 1 in this row
            // (the left neighbour of the pixels at x = 2j); from the lane to the right: its first half-column of the
            // colour at x = 2j (the right neighbour of the pixels at x = 2j + 1)
            const int c_odd = pr ? 0 : 1, c_even = 1 - c_odd;      // colour at odd / even x in this row
            double from_left = __shfl_up(d0[c_odd][CPT - 1], 1, kWave);
            double from_right = __shfl_down(d0[c_even][0], 1, kWave);
            if (lane == 0) from_left = col_ok ? dir_at(l, c_odd, j0 - 1) : 0.0;
            if (lane == kWave - 1) from_right = col_ok ? dir_at(l, c_even, j0 + CPT) : 0.0;
            if (col_ok) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int o = 1 - c;
                    const int pp = (y + c) & 1;                    // this colour's pixels sit at x = 2j + pp
                    const long at = row_off(g, l, c) + j0;
                    double xo[CPT], ax[CPT];
                    ld_vec<CPT>(x + at, xo);
#pragma unroll
                    for (int k = 0; k < CPT; ++k) {
                        xo[k] = xo[k] + alpha * po0[c][k];
                        const int xi = 2 * (j0 + k) + pp;
                        ax[k] = 0.0;
                        if (xi < g.W) {
                            const double left = pp ? d0[o][k] : (k == 0 ? from_left : d0[o][k > 0 ? k - 1 : 0]);
                            const double right = pp ? (k == CPT - 1 ? from_right : d0[o][k < CPT - 1 ? k + 1 : k]) : d0[o][k];
                            if (MASKED) {
                                if (mask[at + k] != 0) {
                                    ax[k] += -1.0 * dm[o][k];
                                    ax[k] += -1.0 * left;
                                    ax[k] += 4.0 * d0[c][k];
                                    ax[k] += -1.0 * right;
                                    ax[k] += -1.0 * dp[o][k];
                                }
                            } else {
                                const Stencil sc = classify(g, xi, y, l);
                                ax[k] = apply_row(sc, d0[c][k], dm[o][k], left, right, dp[o][k]);
                            }
                            dot += d0[c][k] * ax[k];
                        }
                    }
                    st_vec<CPT>(x + at, xo);
                    st_vec<CPT>(p_new + at, d0[c]);
                    st_vec<CPT>(ap + at, ax);
                }
            }
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int k = 0; k < CPT; ++k) {
                    dm[c][k] = d0[c][k];
                    d0[c][k] = dp[c][k];
                    po0[c][k] = pop[c][k];
                }
        }
    }
    const double t0 = block_sum(dot, scratch);
    if (threadIdx.x == 0) partial[(long)blockIdx.y * gridDim.x + blockIdx.x] = t0;
}

}  // namespace ccp
