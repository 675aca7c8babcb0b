// lat_bench.hip — dependent-chain latency of the fp64 ops the fused sweep's update chain uses.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_=(x); if(e_!=hipSuccess){fprintf(stderr,"%s: %s\n",#x,hipGetErrorString(e_)); return 1;} } while(0)
template <int MODE>
__global__ void k(double *out, double a, double b, int n, long long *cyc)
{
    double x0 = a + threadIdx.x, x1 = a * 2 + threadIdx.x, x2 = a * 3, x3 = a * 4;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        if (MODE == 0) { x0 = x0 + b; x0 = x0 + b; x0 = x0 + b; x0 = x0 + b; }                       // 4 dependent adds
        if (MODE == 1) { x0 = x0 + b; x1 = x1 + b; x2 = x2 + b; x3 = x3 + b; }                       // 4 independent adds
        if (MODE == 2) { x0 = __builtin_ldexp(x0 + b, -2); x0 = __builtin_ldexp(x0 + b, -2); }       // add->ldexp chain x2
        if (MODE == 3) { x0 = (x0 + b) * 0.25; x0 = (x0 + b) * 0.25; }                               // add->mul chain x2
        if (MODE == 4) {                                                                             // dpp + add chain
            int lo = __double2loint(x0), hi = __double2hiint(x0);
            lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);
            hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
            x0 = __hiloint2double(hi, lo) + b;
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main()
{
    double *out; long long *cyc, h;
    CK(hipMalloc(&out, 8 * 64 * 4096)); CK(hipMalloc(&cyc, 8));
    const int n = 4000;
    const char *names[] = {"4 dependent v_add_f64", "4 independent v_add_f64", "2x (add -> ldexp) chain", "2x (add -> mul) chain", "dpp pair + add chain"};
    const int ops[] = {4, 4, 4, 4, 3};
    for (int waves = 1; waves <= 2; ++waves)
    for (int m = 0; m < 5; ++m) {
        for (int rep = 0; rep < 2; ++rep) {
            switch (m) {
            case 0: hipLaunchKernelGGL(k<0>, dim3(1), dim3(64 * 4 * waves), 0, 0, out, 1.5, 0.25, n, cyc); break;
            case 1: hipLaunchKernelGGL(k<1>, dim3(1), dim3(64 * 4 * waves), 0, 0, out, 1.5, 0.25, n, cyc); break;
            case 2: hipLaunchKernelGGL(k<2>, dim3(1), dim3(64 * 4 * waves), 0, 0, out, 1.5, 0.25, n, cyc); break;
            case 3: hipLaunchKernelGGL(k<3>, dim3(1), dim3(64 * 4 * waves), 0, 0, out, 1.5, 0.25, n, cyc); break;
            case 4: hipLaunchKernelGGL(k<4>, dim3(1), dim3(64 * 4 * waves), 0, 0, out, 1.5, 0.25, n, cyc); break;
            }
            CK(hipDeviceSynchronize());
        }
        CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
        printf("waves/SIMD=%d  %-28s %.1f cycles per op (s_memtime ticks)\n", waves, names[m], (double)h / n / ops[m]);
    }
    return 0;
}
