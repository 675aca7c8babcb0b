// step_bench_io.hip — the workgroup shape of k_lex_wg in isolation: 8 compute waves (LDS ring reads, DPP, four
// dependent adds, LDS write, barrier per step), a loader wave (three global loads per step, eight steps ahead, into
// LDS) and a storer wave (two write-through stores per step), with and without the b-row ring, at 1 workgroup, 1 per
// CU and 2 per CU.  Wall time per lock-step step.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_=(x); if(e_!=hipSuccess){fprintf(stderr,"%s: %s\n",#x,hipGetErrorString(e_)); return 1;} } while(0)
__device__ __forceinline__ void bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ double prev(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double ld(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// MODE bit0: loader wave does global loads, bit1: storer wave does global stores, bit2: big LDS (60 KB)
template <int MODE, int ROWS>
__global__ void __launch_bounds__(640) k(double *buf, long stride, int n)
{
    __shared__ double ring[9][8][64];
    __shared__ double brow[ROWS][92];
    const int lane = threadIdx.x & 63, t = threadIdx.x >> 6;
    if (t < 9) for (int q = 0; q < 8; ++q) ring[t][q][lane] = 1.0 + lane * 1e-3 + q;
    if ((MODE & 4) && threadIdx.x < 92) for (int q = 0; q < ROWS; ++q) brow[q][threadIdx.x] = 0.5;
    __syncthreads();
    double *mine = buf + (long)blockIdx.x * stride;
    if (t < 8) {
        double h = 1.0 + lane;
        const int l2 = lane > 1 ? lane - 2 : 0;
        for (int i = 0; i < n; i += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const double *in = &ring[t][(j + 5) & 7][l2];
                const double d = in[0], r = in[1];
                const double v = (MODE & 4) ? brow[(i + j) % ROWS][lane] : ring[t][(j + 4) & 7][lane];
                const double left = prev(h);
                double nv = (v + (((h + left) + r) + d)) * 0.25;
                nv = lane < 2 ? v : nv;
                ring[t + 1][j][lane] = nv;
                h = nv;
                bar();
            }
        }
        mine[threadIdx.x] = h;
    } else if (t == 8) {
        double q0[8], q1[8], q2[8];
        const double *p = mine + 1024 + lane;
        for (int j = 0; j < 8; ++j) { q0[j] = 1.0; q1[j] = 2.0; q2[j] = 3.0; }
        for (int i = 0; i < n; i += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (MODE & 4) brow[(i + j + 1) % ROWS][lane] = q0[j];
                if ((MODE & 4) && lane < 12) brow[(i + j + 1) % ROWS][64 + lane] = q1[j];
                ring[0][(j + 7) & 7][lane] = q2[j];
                asm volatile("" ::: "memory");
                if (MODE & 1) { q0[j] = ld(p); q1[j] = ld(p + 64); q2[j] = ld(p + 128); p += 256; if (p > mine + stride - 1024) p = mine + 1024 + lane; }
                bar();
            }
        }
        mine[512 + lane] = q0[0] + q1[1] + q2[2];
    } else {
        double *p = mine + 2048 + lane;
        for (int i = 0; i < n; i += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                bar();
                const double v = ring[8][j][lane];
                const double e = ring[(lane >> 1 & 7) + 1][j][62 + (lane & 1)];
                if (MODE & 2) { st(p, v); if (lane < 16) st(p + 64, e); p += 128; if (p > mine + stride - 1024) p = mine + 2048 + lane; }
                else if (v == 12345.678 && e == 1.5) mine[600] = v;
            }
            if (MODE & 2) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
        }
    }
}
int main()
{
    const long stride = 1 << 20;
    double *buf;
    CK(hipMalloc(&buf, 8 * stride * 512));
    CK(hipMemset(buf, 0, 8 * stride * 512));
    const int n = 200000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *what[] = {"rings only (37 KB)", "+ loader loads", "+ storer stores", "+ loads and stores", "+ b ring, 16 rows (48 KB)", "+ b ring, 32 rows (59 KB)",
                          "+ b ring 16 rows, loads and stores", "+ b ring 32 rows, loads and stores"};
    for (int blocks : {1, 256, 512})
        for (int m = 0; m < 8; ++m) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                switch (m) {
                case 0: hipLaunchKernelGGL((k<0, 1>), dim3(blocks), dim3(640), 0, 0, buf, stride, n); break;
                case 1: hipLaunchKernelGGL((k<1, 1>), dim3(blocks), dim3(640), 0, 0, buf, stride, n); break;
                case 2: hipLaunchKernelGGL((k<2, 1>), dim3(blocks), dim3(640), 0, 0, buf, stride, n); break;
                case 3: hipLaunchKernelGGL((k<3, 1>), dim3(blocks), dim3(640), 0, 0, buf, stride, n); break;
                case 4: hipLaunchKernelGGL((k<4, 16>), dim3(blocks), dim3(640), 0, 0, buf, stride, n); break;
                case 5: hipLaunchKernelGGL((k<4, 32>), dim3(blocks), dim3(640), 0, 0, buf, stride, n); break;
                case 6: hipLaunchKernelGGL((k<7, 16>), dim3(blocks), dim3(640), 0, 0, buf, stride, n); break;
                case 7: hipLaunchKernelGGL((k<7, 32>), dim3(blocks), dim3(640), 0, 0, buf, stride, n); break;
                }
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
            }
            printf("workgroups=%3d  %-40s %.1f ns per step\n", blocks, what[m], ms * 1e6 / n);
        }
    return 0;
}
