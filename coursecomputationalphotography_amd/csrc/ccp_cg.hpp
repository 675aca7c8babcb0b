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
    int pcur;            // fused loop: which of the two direction buffers holds the current p
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

// out[0] = the partials added up in the fixed order (what a row block all-reduces over its communicator)
static __global__ void __launch_bounds__(kBlock)
k_reduce_to_one(const double *__restrict__ partial, long count, double *__restrict__ out)
{
    __shared__ double scratch[kBlock / kWave];
    const double t = reduce_partials(partial, (int)count, scratch);
    if (threadIdx.x == 0) out[0] = t;
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

// ---- fused loop (matrix-free grid path): 72 B per unknown and iteration instead of 88 -------------------------
// The three vector passes of an iteration are folded into the two that have to exist:
//   A(k)  [apply(...), supplied by the caller]   x += alpha_{k-1} p_{k-1};  p_k = r + beta_{k-1} p_{k-1};  Ap = A p_k
//                                                (p_k of the four neighbours recomputed from r and p_{k-1});
//                                                partial sums of p_k'Ap.   x rw 16 + r 8 + p r/w 16 + Ap w 8 = 48 B
//   B(k)  k_cg_residual                          r += (-alpha_k) Ap;  partial sums of r'r.          r rw 16 + Ap 8 = 24 B
// The same operations on the same operands in the same order as the unfused loop — x's update merely waits for the
// next pass over p, the last one is applied by k_cg_axpy_final — so the iterates are the unfused loop's bit for bit.
// p is double-buffered: a neighbour's p_{k-1} must still be readable while another workgroup stores its p_k.
static __global__ void __launch_bounds__(kBlock)
k_cg_alpha_fused(const double *__restrict__ partial, int count, int k, CgState *__restrict__ st)
{
    __shared__ double scratch[kBlock / kWave];
    const double pap = reduce_partials(partial, count, scratch);
    if (threadIdx.x == 0 && st->active) {
        st->alpha = st->rlen / pap;                       // (sparse-matrix.h:420)
        st->pcur = k & 1;
    }
}

static __global__ void __launch_bounds__(kBlock)
k_cg_residual(double *__restrict__ r, const double *__restrict__ ap, long n, double *__restrict__ partial, const CgState *__restrict__ st)
{
    __shared__ double scratch[kBlock / kWave];
    double acc = 0.0;
    if (st->active) {
        const double nalpha = -st->alpha;
        // (k_cg_update's traversal: which thread sums which elements decides the bits of r'r)
        for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock) {
            const double v = r[i] + nalpha * ap[i];      // (:422)
            r[i] = v;
            acc += v * v;                                 // (:423)
        }
    }
    const double t = block_sum(acc, scratch);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// the x update the loop still owes: x += alpha_K p_K (:421 of the last iteration that ran)
static __global__ void __launch_bounds__(kBlock)
k_cg_axpy_final(double *__restrict__ x, const double *__restrict__ p0, const double *__restrict__ p1, long n, const CgState *__restrict__ st)
{
    const double alpha = st->alpha;
    const double *__restrict__ p = st->pcur ? p1 : p0;
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock) x[i] = x[i] + alpha * p[i];
}

// `apply(x, r, p_in, p_out, ap, &n_partials)` enqueues A(k); n is even (whole colour half-rows).
template <typename Spmv, typename Apply, typename Sums>
int cg_solve_fused(Spmv &&spmv, Apply &&apply, const double *b, double *x, double *r, double *p0, double *p1, double *ap, long n,
                   double epsilon, int max_iteration, CgState *st_dev, double *partial, hipStream_t stream, hipEvent_t ev0,
                   hipEvent_t ev1, ccp_gs_report *report, Sums &&sums, bool every_rank_iterates = false)
{
    const int blocks = (int)std::max<long>(1, std::min<long>(2048, (n + kBlock - 1) / kBlock));
    CgState host{};
    host.active = 1;                                                         // alpha = beta = 0: A(1) leaves x alone, p_1 = r
    CCP_HIP(hipMemcpyAsync(st_dev, &host, sizeof(host), hipMemcpyHostToDevice, stream));
    CCP_HIP(hipEventRecord(ev0, stream));
    CCP_TRY(spmv(x, r));                                                     // r = A x      (:406)
    hipLaunchKernelGGL(k_cg_init, dim3(blocks), dim3(kBlock), 0, stream, b, r, p0, n, partial);   // r = b - r, p = r
    int count = blocks;
    const double *sum_at = partial;
    CCP_TRY(sums(partial, &count, &sum_at, 0));
    hipLaunchKernelGGL(k_cg_set_rlen, dim3(1), dim3(kBlock), 0, stream, sum_at, count, st_dev);
    CCP_HIP(hipGetLastError());
    int issued = 0;
    bool active = max_iteration > 0 && (n > 0 || every_rank_iterates);
    while (active && issued < max_iteration) {
        const int batch = std::min(16, max_iteration - issued);
        for (int k = issued + 1; k <= issued + batch; ++k) {
            int dot_blocks = 0;
            double *p_in = (k & 1) ? p0 : p1, *p_out = (k & 1) ? p1 : p0;    // iteration k reads buffer (k-1)&1, writes k&1
            CCP_TRY(apply(x, r, p_in, p_out, ap, &dot_blocks));
            sum_at = partial;
            CCP_TRY(sums(partial, &dot_blocks, &sum_at, 0));
            hipLaunchKernelGGL(k_cg_alpha_fused, dim3(1), dim3(kBlock), 0, stream, sum_at, dot_blocks, k, st_dev);
            hipLaunchKernelGGL(k_cg_residual, dim3(blocks), dim3(kBlock), 0, stream, r, ap, n, partial, st_dev);
            count = blocks;
            sum_at = partial;
            CCP_TRY(sums(partial, &count, &sum_at, 0));
            hipLaunchKernelGGL(k_cg_beta, dim3(1), dim3(kBlock), 0, stream, sum_at, count, epsilon, st_dev);
        }
        CCP_HIP(hipGetLastError());
        issued += batch;
        CCP_HIP(hipMemcpyAsync(&host, st_dev, sizeof(host), hipMemcpyDeviceToHost, stream));
        CCP_HIP(hipStreamSynchronize(stream));
        active = host.active != 0;
    }
    hipLaunchKernelGGL(k_cg_axpy_final, dim3(blocks), dim3(kBlock), 0, stream, x, p0, p1, n, st_dev);
    CCP_HIP(hipGetLastError());
    CCP_HIP(hipEventRecord(ev1, stream));
    CCP_HIP(hipMemcpyAsync(&host, st_dev, sizeof(host), hipMemcpyDeviceToHost, stream));
    CCP_HIP(hipStreamSynchronize(stream));
    if (report) {
        float ms = 0.f;
        CCP_HIP(hipEventElapsedTime(&ms, ev0, ev1));
        report->iterations = host.iterations;
        report->converged = host.converged;
        report->last_l1_step = host.r1norm;
        report->seconds = ms * 1e-3;
    }
    return CCP_OK;
}

template <typename Spmv, typename Apply>
int cg_solve_fused(Spmv &&spmv, Apply &&apply, const double *b, double *x, double *r, double *p0, double *p1, double *ap, long n,
                   double epsilon, int max_iteration, CgState *st_dev, double *partial, hipStream_t stream, hipEvent_t ev0,
                   hipEvent_t ev1, ccp_gs_report *report)
{
    return cg_solve_fused(spmv, apply, b, x, r, p0, p1, ap, n, epsilon, max_iteration, st_dev, partial, stream, ev0, ev1, report,
                          [](double *, int *, const double **, int) { return (int)CCP_OK; });
}

// The loop.  `spmv(in, out)` enqueues out := A in on `stream` (all device pointers);
// `spmv_dot(in, out)` does the same and leaves block partials of in'(A in) in `partial`,
// returning how many (0 = not fused: a separate dot pass runs).
// x: initial guess on entry, solution on exit.  r, p, ap: scratch vectors of n doubles.
// `sums(partial, &count, &sum_at, slot)` is called after every kernel that left `count` block partials of a dot product in
// `partial` and before the kernel that adds them up: a row block of a distributed matrix adds its partials up itself,
// all-reduces the one value over the ranks and returns where it is (`slot`: 0 or 1, two sums may be alive at once), with
// count = 1 (ccp_csr.hip); the one-GPU callers pass nothing.
template <typename Spmv, typename SpmvDot, typename Sums>
int cg_solve(Spmv &&spmv, SpmvDot &&spmv_dot, const double *b, double *x, double *r, double *p, double *ap, long n, double epsilon,
             int max_iteration, CgState *st_dev, double *partial, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1,
             ccp_gs_report *report, Sums &&sums, bool every_rank_iterates = false)
{
    const int blocks = (int)std::max<long>(1, std::min<long>(2048, (n + kBlock - 1) / kBlock));
    CgState host{};
    host.active = 1;
    CCP_HIP(hipMemcpyAsync(st_dev, &host, sizeof(host), hipMemcpyHostToDevice, stream));
    CCP_HIP(hipEventRecord(ev0, stream));
    CCP_TRY(spmv(x, r));                                                     // r = A x      (:406)
    hipLaunchKernelGGL(k_cg_init, dim3(blocks), dim3(kBlock), 0, stream, b, r, p, n, partial);   // r = b - r, p = r
    int count = blocks;
    const double *sum_at = partial;
    CCP_TRY(sums(partial, &count, &sum_at, 0));
    hipLaunchKernelGGL(k_cg_set_rlen, dim3(1), dim3(kBlock), 0, stream, sum_at, count, st_dev);
    CCP_HIP(hipGetLastError());
    int issued = 0;
    bool active = max_iteration > 0 && (n > 0 || every_rank_iterates);
    while (active && issued < max_iteration) {
        const int batch = std::min(16, max_iteration - issued);
        for (int k = 0; k < batch; ++k) {
            int dot_blocks = 0;
            CCP_TRY(spmv_dot(p, ap, &dot_blocks));                           // Ap = A p (:419) [+ p'Ap partials]
            if (dot_blocks == 0) {
                hipLaunchKernelGGL(k_cg_dot, dim3(blocks), dim3(kBlock), 0, stream, p, ap, n, partial, st_dev);
                dot_blocks = blocks;
            }
            sum_at = partial;
            CCP_TRY(sums(partial, &dot_blocks, &sum_at, 0));
            hipLaunchKernelGGL(k_cg_alpha, dim3(1), dim3(kBlock), 0, stream, sum_at, dot_blocks, st_dev);
            hipLaunchKernelGGL(k_cg_update, dim3(blocks), dim3(kBlock), 0, stream, x, p, r, ap, n, partial, st_dev);
            count = blocks;
            sum_at = partial;
            CCP_TRY(sums(partial, &count, &sum_at, 0));
            hipLaunchKernelGGL(k_cg_beta, dim3(1), dim3(kBlock), 0, stream, sum_at, count, epsilon, st_dev);
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

template <typename Spmv, typename SpmvDot>
int cg_solve(Spmv &&spmv, SpmvDot &&spmv_dot, const double *b, double *x, double *r, double *p, double *ap, long n, double epsilon,
             int max_iteration, CgState *st_dev, double *partial, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1,
             ccp_gs_report *report)
{
    return cg_solve(spmv, spmv_dot, b, x, r, p, ap, n, epsilon, max_iteration, st_dev, partial, stream, ev0, ev1, report,
                    [](double *, int *, const double **, int) { return (int)CCP_OK; });
}

// ---------------------------------------------------------------------------------------------
// Jacobi-preconditioned variant: SparseMatrix::conjugateGradientEigen (sparse-matrix.h:494-535, used by
// RunTest, utils.cc:99).  x0 = 0, invdiag = extractDiagnolColInv() (:472-491: 1/a_ii, 1 where the
// diagonal is absent or 0):  p = r.*invdiag, dist = p'r;  per iteration alpha = dist / p'Ap,
// x += alpha p, r -= alpha Ap, stop when sqrt(r'r) < epsilon, z = r.*invdiag, beta = z'r / dist,
// p = z + beta p.  Same device-side scalar scheme as cg_solve; CgState::rlen carries `olddist`.

// p := r .* inv (:503), partial sums of p'r (:508).  r holds b on entry (x0 = 0: r = b - A 0, :500-501).
static __global__ void __launch_bounds__(kBlock)
k_pcg_init(const double *__restrict__ r, const double *__restrict__ inv, double *__restrict__ p, long n,
           double *__restrict__ partial)
{
    __shared__ double scratch[kBlock / kWave];
    double acc = 0.0;
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock) {
        const double v = r[i] * inv[i];
        p[i] = v;
        acc += v * r[i];
    }
    const double t = block_sum(acc, scratch);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// x += alpha p (:518); r += (-alpha) Ap (:519); partial sums of r'r (:520) and of (r.*inv)'r (:522-523)
static __global__ void __launch_bounds__(kBlock)
k_pcg_update(double *__restrict__ x, const double *__restrict__ p, double *__restrict__ r, const double *__restrict__ ap,
             const double *__restrict__ inv, long n, double *__restrict__ partial_rr, double *__restrict__ partial_zr,
             const CgState *__restrict__ st)
{
    __shared__ double scratch[kBlock / kWave];
    double rr = 0.0, zr = 0.0;
    if (st->active) {
        const double alpha = st->alpha;
        const double nalpha = -alpha;
        for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock) {
            x[i] = x[i] + alpha * p[i];
            const double v = r[i] + nalpha * ap[i];
            r[i] = v;
            rr += v * v;
            const double z = v * inv[i];
            zr += z * v;
        }
    }
    const double t0 = block_sum(rr, scratch);
    if (threadIdx.x == 0) partial_rr[blockIdx.x] = t0;
    const double t1 = block_sum(zr, scratch);
    if (threadIdx.x == 0) partial_zr[blockIdx.x] = t1;
}

// error = r'r; `if (sqrt(error) < epsilon) break;` (:520-521); beta = newdist / olddist (:524-525); ++cnt (:528)
static __global__ void __launch_bounds__(kBlock)
k_pcg_beta(const double *__restrict__ partial_rr, const double *__restrict__ partial_zr, int count, double epsilon,
           CgState *__restrict__ st)
{
    __shared__ double scratch[kBlock / kWave];
    const double error = reduce_partials(partial_rr, count, scratch);
    const double newdist = reduce_partials(partial_zr, count, scratch);
    if (threadIdx.x == 0 && st->active) {
        st->r1norm = sqrt(error);
        if (st->r1norm < epsilon) {
            st->active = 0;
            st->converged = 1;
        } else {
            st->beta = newdist / st->rlen;
            st->rlen = newdist;
            st->iterations += 1;
        }
    }
}

// p := r .* inv + beta p (:522,526)
static __global__ void __launch_bounds__(kBlock)
k_pcg_direction(double *__restrict__ p, const double *__restrict__ r, const double *__restrict__ inv, long n,
                const CgState *__restrict__ st)
{
    if (!st->active) return;
    const double beta = st->beta;
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock)
        p[i] = r[i] * inv[i] + beta * p[i];
}

// x: zeroed by the caller; r: holds b on entry.  partial: 2 * 2048 doubles at least (beyond what spmv_dot uses).
template <typename SpmvDot, typename Sums>
int pcg_solve(SpmvDot &&spmv_dot, double *x, double *r, double *p, double *ap, const double *inv, long n, double epsilon,
              int max_iteration, CgState *st_dev, double *partial, double *partial2, hipStream_t stream, hipEvent_t ev0,
              hipEvent_t ev1, ccp_gs_report *report, Sums &&sums, bool every_rank_iterates = false)
{
    const int blocks = (int)std::max<long>(1, std::min<long>(2048, (n + kBlock - 1) / kBlock));
    CgState host{};
    host.active = 1;
    CCP_HIP(hipMemcpyAsync(st_dev, &host, sizeof(host), hipMemcpyHostToDevice, stream));
    CCP_HIP(hipEventRecord(ev0, stream));
    hipLaunchKernelGGL(k_pcg_init, dim3(blocks), dim3(kBlock), 0, stream, r, inv, p, n, partial);
    int count = blocks;
    const double *sum_at = partial, *sum2_at = partial2;
    CCP_TRY(sums(partial, &count, &sum_at, 0));
    hipLaunchKernelGGL(k_cg_set_rlen, dim3(1), dim3(kBlock), 0, stream, sum_at, count, st_dev);      // olddist = p'r
    CCP_HIP(hipGetLastError());
    int issued = 0;
    bool active = max_iteration > 0 && (n > 0 || every_rank_iterates);
    while (active && issued < max_iteration) {
        const int batch = std::min(16, max_iteration - issued);
        for (int k = 0; k < batch; ++k) {
            int dot_blocks = 0;
            CCP_TRY(spmv_dot(p, ap, &dot_blocks));                            // Ap = A p (:516) [+ p'Ap partials]
            if (dot_blocks == 0) {
                hipLaunchKernelGGL(k_cg_dot, dim3(blocks), dim3(kBlock), 0, stream, p, ap, n, partial, st_dev);
                dot_blocks = blocks;
            }
            sum_at = partial;
            CCP_TRY(sums(partial, &dot_blocks, &sum_at, 0));
            hipLaunchKernelGGL(k_cg_alpha, dim3(1), dim3(kBlock), 0, stream, sum_at, dot_blocks, st_dev);   // olddist / p'Ap (:517)
            hipLaunchKernelGGL(k_pcg_update, dim3(blocks), dim3(kBlock), 0, stream, x, p, r, ap, inv, n, partial2, partial2 + 2048,
                               st_dev);
            count = blocks;
            int count2 = blocks;
            sum_at = partial2;
            sum2_at = partial2 + 2048;
            CCP_TRY(sums(partial2, &count, &sum_at, 0));
            CCP_TRY(sums(partial2 + 2048, &count2, &sum2_at, 1));
            hipLaunchKernelGGL(k_pcg_beta, dim3(1), dim3(kBlock), 0, stream, sum_at, sum2_at, count, epsilon, st_dev);
            hipLaunchKernelGGL(k_pcg_direction, dim3(blocks), dim3(kBlock), 0, stream, p, r, inv, n, st_dev);
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
        report->last_l1_step = host.r1norm;
        report->seconds = ms * 1e-3;
    }
    return CCP_OK;
}

template <typename SpmvDot>
int pcg_solve(SpmvDot &&spmv_dot, double *x, double *r, double *p, double *ap, const double *inv, long n, double epsilon,
              int max_iteration, CgState *st_dev, double *partial, double *partial2, hipStream_t stream, hipEvent_t ev0,
              hipEvent_t ev1, ccp_gs_report *report)
{
    return pcg_solve(spmv_dot, x, r, p, ap, inv, n, epsilon, max_iteration, st_dev, partial, partial2, stream, ev0, ev1, report,
                     [](double *, int *, const double **, int) { return (int)CCP_OK; });
}

}  // namespace ccp
