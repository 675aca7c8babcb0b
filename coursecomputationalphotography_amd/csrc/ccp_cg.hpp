// ccp_cg.hpp — device-side conjugate gradient loop shared by the CSR and the grid paths.
//
// Reference: SparseMatrix::conjugateGradient with an optional initial guess
// (project/src/PhotoMontage/sparse-matrix.h:396-434) — the solver the blend call sites use today
// (PhotoMontage.cpp:613, hw8_pa.cc:972).  Per iteration: one SpMV, x += alpha p,
// r -= alpha Ap, p = r + beta p, with alpha = r'r / p'Ap and beta = r1'r1 / r'r.
//
// The scalars live in device memory and are produced by single-block kernels, so the host never
// waits inside the loop; a converged solve turns the remaining queued kernels into no-ops
// (same scheme as the Gauss-Seidel stop rule).  Reductions are deterministic (fixed shuffle tree,
// fixed block order) but not the reference's serial left-to-right order, so iterates agree with
// the reference to rounding (~1e-13 relative per iteration), not bit for bit.
#pragma once

#include "ccp_common.hpp"

namespace ccp {

struct CgState {
    int active;
    int converged;
    int iterations;      // the reference's `cnt`
    int pad;
    double rlen;         // r'r of the current residual
    double alpha;
    double beta;
    double r1norm;       // sqrt(r1'r1) of the last update
};

// partial[blockIdx.x] = sum over this block's grid-stride range of a[i]*b[i]
static __global__ void __launch_bounds__(kBlock)
k_cg_dot(const double *__restrict__ a, const double *__restrict__ b, long n, double *__restrict__ partial,
         const CgState *__restrict__ st)
{
    __shared__ double scratch[kBlock / kWave];
    double acc = 0.0;
    if (st->active)
        for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock) acc += a[i] * b[i];
    const double t = block_sum(acc, scratch);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// r := b - r  (r holds A x on entry; vecsub(b, r, r), sparse-matrix.h:407) and p := r (:410);
// partial sums of r'r.
static __global__ void __launch_bounds__(kBlock)
k_cg_init(const double *__restrict__ b, double *__restrict__ r, double *__restrict__ p, long n,
          double *__restrict__ partial)
{
    __shared__ double scratch[kBlock / kWave];
    double acc = 0.0;
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock) {
        const double v = b[i] - r[i];
        r[i] = v;
        p[i] = v;
        acc += v * v;
    }
    const double t = block_sum(acc, scratch);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// one block: sum the partials in a fixed order
__device__ __forceinline__ double reduce_partials(const double *__restrict__ partial, int count, double *scratch)
{
    double acc = 0.0;
    for (int i = threadIdx.x; i < count; i += kBlock) acc += partial[i];
    return block_sum(acc, scratch);
}

static __global__ void __launch_bounds__(kBlock)
k_cg_set_rlen(const double *__restrict__ partial, int count, CgState *__restrict__ st)
{
    __shared__ double scratch[kBlock / kWave];
    const double t = reduce_partials(partial, count, scratch);
    if (threadIdx.x == 0) st->rlen = t;
}

// alpha = r'r / p'Ap (sparse-matrix.h:420)
static __global__ void __launch_bounds__(kBlock)
k_cg_alpha(const double *__restrict__ partial, int count, CgState *__restrict__ st)
{
    __shared__ double scratch[kBlock / kWave];
    const double pap = reduce_partials(partial, count, scratch);
    if (threadIdx.x == 0 && st->active) st->alpha = st->rlen / pap;
}

// x += alpha p (:421); r := r + (-alpha) Ap (:422, r1 stored over r); partial sums of r1'r1 (:423)
static __global__ void __launch_bounds__(kBlock)
k_cg_update(double *__restrict__ x, const double *__restrict__ p, double *__restrict__ r,
            const double *__restrict__ ap, long n, double *__restrict__ partial, const CgState *__restrict__ st)
{
    __shared__ double scratch[kBlock / kWave];
    double acc = 0.0;
    if (st->active) {
        const double alpha = st->alpha;
        const double nalpha = -alpha;
        for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock) {
            x[i] = x[i] + alpha * p[i];
            const double v = r[i] + nalpha * ap[i];
            r[i] = v;
            acc += v * v;
        }
    }
    const double t = block_sum(acc, scratch);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// r1len; `if (sqrt(r1len) < epsilon) break;` (:425); beta = r1len / rlen (:426); ++cnt (:430)
static __global__ void __launch_bounds__(kBlock)
k_cg_beta(const double *__restrict__ partial, int count, double epsilon, CgState *__restrict__ st)
{
    __shared__ double scratch[kBlock / kWave];
    const double r1len = reduce_partials(partial, count, scratch);
    if (threadIdx.x == 0 && st->active) {
        st->r1norm = sqrt(r1len);
        if (st->r1norm < epsilon) {
            st->active = 0;
            st->converged = 1;
        } else {
            st->beta = r1len / st->rlen;
            st->rlen = r1len;
            st->iterations += 1;
        }
    }
}

// p := r1 + beta p (:427)
static __global__ void __launch_bounds__(kBlock)
k_cg_direction(double *__restrict__ p, const double *__restrict__ r, long n, const CgState *__restrict__ st)
{
    if (!st->active) return;
    const double beta = st->beta;
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock)
        p[i] = r[i] + beta * p[i];
}

// The loop.  `spmv(in, out)` enqueues out := A in on `stream` (all device pointers);
// `spmv_dot(in, out)` does the same and leaves block partials of in'(A in) in `partial`,
// returning how many (0 = not fused: a separate dot pass runs).
// x: initial guess on entry, solution on exit.  r, p, ap: scratch vectors of n doubles.
template <typename Spmv, typename SpmvDot>
int cg_solve(Spmv &&spmv, SpmvDot &&spmv_dot, const double *b, double *x, double *r, double *p, double *ap, long n, double epsilon,
             int max_iteration, CgState *st_dev, double *partial, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1,
             ccp_gs_report *report)
{
    const int blocks = (int)std::max<long>(1, std::min<long>(2048, (n + kBlock - 1) / kBlock));
    CgState host{};
    host.active = 1;
    CCP_HIP(hipMemcpyAsync(st_dev, &host, sizeof(host), hipMemcpyHostToDevice, stream));
    CCP_HIP(hipEventRecord(ev0, stream));
    CCP_TRY(spmv(x, r));                                                     // r = A x      (:406)
    hipLaunchKernelGGL(k_cg_init, dim3(blocks), dim3(kBlock), 0, stream, b, r, p, n, partial);   // r = b - r, p = r
    hipLaunchKernelGGL(k_cg_set_rlen, dim3(1), dim3(kBlock), 0, stream, partial, blocks, st_dev);
    CCP_HIP(hipGetLastError());
    int issued = 0;
    bool active = max_iteration > 0 && n > 0;
    while (active && issued < max_iteration) {
        const int batch = std::min(16, max_iteration - issued);
        for (int k = 0; k < batch; ++k) {
            int dot_blocks = 0;
            CCP_TRY(spmv_dot(p, ap, &dot_blocks));                           // Ap = A p (:419) [+ p'Ap partials]
            if (dot_blocks == 0) {
                hipLaunchKernelGGL(k_cg_dot, dim3(blocks), dim3(kBlock), 0, stream, p, ap, n, partial, st_dev);
                dot_blocks = blocks;
            }
            hipLaunchKernelGGL(k_cg_alpha, dim3(1), dim3(kBlock), 0, stream, partial, dot_blocks, st_dev);
            hipLaunchKernelGGL(k_cg_update, dim3(blocks), dim3(kBlock), 0, stream, x, p, r, ap, n, partial, st_dev);
            hipLaunchKernelGGL(k_cg_beta, dim3(1), dim3(kBlock), 0, stream, partial, blocks, epsilon, st_dev);
            hipLaunchKernelGGL(k_cg_direction, dim3(blocks), dim3(kBlock), 0, stream, p, r, n, st_dev);
        }
        CCP_HIP(hipGetLastError());
        issued += batch;
        CCP_HIP(hipMemcpyAsync(&host, st_dev, sizeof(host), hipMemcpyDeviceToHost, stream));
        CCP_HIP(hipStreamSynchronize(stream));
        active = host.active != 0;
    }
    CCP_HIP(hipEventRecord(ev1, stream));
    CCP_HIP(hipMemcpyAsync(&host, st_dev, sizeof(host), hipMemcpyDeviceToHost, stream));
    CCP_HIP(hipStreamSynchronize(stream));
    if (report) {
        float ms = 0.f;
        CCP_HIP(hipEventElapsedTime(&ms, ev0, ev1));
        report->iterations = host.iterations;
        report->converged = host.converged;
        report->last_l1_step = host.r1norm;       // CG: sqrt(r'r) of the last update
        report->seconds = ms * 1e-3;
    }
    return CCP_OK;
}

}  // namespace ccp
