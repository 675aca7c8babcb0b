// step_bench.hip — what one lock-step step of k_lex_wg costs, piece by piece: one workgroup of 8 waves on an
// otherwise idle chip (and the same with every CU holding 1 or 2 of them), n steps, wall time per step.
//   0 barrier alone            1 LDS write + barrier          2 LDS reads -> 4 dependent adds -> write -> barrier
//   3 (2) + the DPP shift      4 the arithmetic alone (no LDS, no barrier)   5 (3) with s_barrier removed
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_=(x); if(e_!=hipSuccess){fprintf(stderr,"%s: %s\n",#x,hipGetErrorString(e_)); return 1;} } while(0)
__device__ __forceinline__ void bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void nobar() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ double prev(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int MODE>
__global__ void __launch_bounds__(512) k(double *out, int n)
{
    __shared__ double ring[8][8][64];
    const int lane = threadIdx.x & 63, t = threadIdx.x >> 6;
    for (int q = 0; q < 8; ++q) ring[t][q][lane] = 1.0 + lane * 1e-3 + q;
    __syncthreads();
    double h = 1.0 + lane;
    const int tp = (t + 7) & 7, l1 = lane > 0 ? lane - 1 : 0, l2 = lane > 1 ? lane - 2 : 0;
    for (int i = 0; i < n; i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MODE == 0) bar();
            if (MODE == 1) { ring[t][j][lane] = h; bar(); }
            if (MODE == 2 || MODE == 3 || MODE == 5) {
                const double r = ring[tp][(j + 5) & 7][l1], d = ring[tp][(j + 5) & 7][l2], v = ring[tp][(j + 4) & 7][lane];
                const double left = MODE == 2 ? h * 0.5 : prev(h);
                double nv = (v + (((h + left) + r) + d)) * 0.25;
                nv = lane < 2 ? v : nv;
                ring[t][j][lane] = nv;
                h = nv;
                if (MODE == 5) nobar(); else bar();
            }
            if (MODE == 4) {
                const double left = prev(h);
                double nv = (1.5 + (((h + left) + 0.25) + 0.125)) * 0.25;
                nv = lane < 2 ? 0.5 : nv;
                h = nv;
            }
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = h;
}
int main()
{
    double *out;
    CK(hipMalloc(&out, 8 * 512 * 1024));
    const int n = 400000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[] = {"barrier", "lds write + barrier", "reads, 4 adds, write, barrier", "... + dpp", "arithmetic only", "... + dpp, no s_barrier"};
    for (int blocks : {1, 256, 512})
        for (int m = 0; m < 6; ++m) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                switch (m) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), 0, 0, out, n); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), 0, 0, out, n); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(512), 0, 0, out, n); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(512), 0, 0, out, n); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(512), 0, 0, out, n); break;
                case 5: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(512), 0, 0, out, n); break;
                }
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
            }
            printf("blocks=%3d  %-32s %.1f ns per step\n", blocks, names[m], ms * 1e6 / n);
        }
    return 0;
}
