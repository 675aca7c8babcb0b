// valu_rate.hip — issue rate of the VALU ops the fused sweep is made of, per SIMD, with 1..4 waves per
// SIMD: wall-clock over a grid that fills every SIMD (256 CUs x 4 SIMDs), independent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_=(x); if(e_!=hipSuccess){fprintf(stderr,"%s: %s\n",#x,hipGetErrorString(e_)); return 1;} } while(0)

template <int MODE>
__global__ void __launch_bounds__(256) k(double *out, double b, int n)
{
    double x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = b * (j + 1) + threadIdx.x;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MODE == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[j]) : "v"(b));
            if (MODE == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[j]) : "v"(b));
            if (MODE == 2) asm volatile("v_ldexp_f64 %0, %0, -2" : "+v"(x[j]));
            if (MODE == 3) asm volatile("v_mov_b64 %0, %1" : "=v"(x[j]) : "v"(x[(j + 1) & 7]));
            if (MODE == 4) {
                int lo = __double2loint(x[j]);
                asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(lo) : "v"(lo));
                x[j] = __hiloint2double(__double2hiint(x[j]), lo);
            }
            if (MODE == 5) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x[j]) : "v"(b));
        }
    }
    double s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += x[j];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    double *out;
    CK(hipMalloc(&out, sizeof(double) * 256 * 256 * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[] = {"v_add_f64", "v_mul_f64", "v_ldexp_f64", "v_mov_b64", "v_mov_b32_dpp", "v_fma_f64"};
    const int n = 20000;
    for (int wps = 1; wps <= 4; ++wps)
        for (int m = 0; m < 6; ++m) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                const dim3 grid(256 * wps), block(256);          // one 4-wave block per CU and per wave-per-SIMD
                CK(hipEventRecord(e0));
                switch (m) {
                case 0: hipLaunchKernelGGL(k<0>, grid, block, 0, 0, out, 1.0000001, n); break;
                case 1: hipLaunchKernelGGL(k<1>, grid, block, 0, 0, out, 1.0000001, n); break;
                case 2: hipLaunchKernelGGL(k<2>, grid, block, 0, 0, out, 1.0000001, n); break;
                case 3: hipLaunchKernelGGL(k<3>, grid, block, 0, 0, out, 1.0000001, n); break;
                case 4: hipLaunchKernelGGL(k<4>, grid, block, 0, 0, out, 1.0000001, n); break;
                case 5: hipLaunchKernelGGL(k<5>, grid, block, 0, 0, out, 1.0000001, n); break;
                }
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
            }
            const double ops_per_simd = (double)n * 8 * wps;      // wave-instructions issued on one SIMD
            printf("waves/SIMD=%d %-14s %.2f ns per wave-instruction per SIMD (%.1f cycles at 2.4 GHz)\n", wps, names[m],
                   ms * 1e6 / ops_per_simd, ms * 1e6 / ops_per_simd * 2.4);
        }
    return 0;
}
