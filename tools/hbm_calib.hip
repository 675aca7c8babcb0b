// hbm_calib.hip — device STREAM-style ceilings and PMC calibration kernels for MI355X.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/hbm_calib tools/hbm_calib.hip
// Prints achieved GB/s for read/copy/triad-like kernels with 16-byte and 8-byte lane accesses
// over buffers far larger than the 256 MiB Infinity Cache; run under
//   rocprofv3 --pmc FETCH_SIZE   and   rocprofv3 --pmc WRITE_SIZE
// to calibrate the counters against these known byte counts (MI355X_MICROARCH.md §HBM).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) copy16(const double2 *__restrict__ a, double2 *__restrict__ b, long n)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) b[i] = a[i];
}
__global__ void __launch_bounds__(256) read16(const double2 *__restrict__ a, double *__restrict__ out, long n)
{
    double acc = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) { double2 v = a[i]; acc += v.x + v.y; }
    if (acc == 1.2345e-300) out[0] = acc;
}
__global__ void __launch_bounds__(256) read8(const double *__restrict__ a, double *__restrict__ out, long n)
{
    double acc = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) acc += a[i];
    if (acc == 1.2345e-300) out[0] = acc;
}
__global__ void __launch_bounds__(256) write16(double2 *__restrict__ b, long n)
{
    double2 v; v.x = 1.0; v.y = 2.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) b[i] = v;
}
// two read streams + one write stream, 16 B per lane each: the byte mix of a GS half-sweep
__global__ void __launch_bounds__(256) add16(const double2 *__restrict__ a, const double2 *__restrict__ c, double2 *__restrict__ b, long n)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        double2 x = a[i], y = c[i]; x.x += y.x; x.y += y.y; b[i] = x;
    }
}
// same, but each block walks a contiguous chunk (row-marching order) instead of grid-striding
__global__ void __launch_bounds__(256) add16_chunk(const double2 *__restrict__ a, const double2 *__restrict__ c, double2 *__restrict__ b, long n, long per_block)
{
    const long lo = (long)blockIdx.x * per_block, hi = lo + per_block < n ? lo + per_block : n;
    for (long i = lo + threadIdx.x; i < hi; i += 256) {
        double2 x = a[i], y = c[i]; x.x += y.x; x.y += y.y; b[i] = x;
    }
}

// nontemporal variants (global_load/store ... nt)
__global__ void __launch_bounds__(256) add16_chunk_nt(const double *__restrict__ a, const double *__restrict__ c, double *__restrict__ b, long n, long per_block, int nt_load, int nt_store)
{
    const long lo = (long)blockIdx.x * per_block, hi = lo + per_block < n ? lo + per_block : n;
    for (long i = lo + threadIdx.x; i < hi; i += 256) {
        double x0, x1, y0, y1;
        if (nt_load) {
            x0 = __builtin_nontemporal_load(a + 2 * i); x1 = __builtin_nontemporal_load(a + 2 * i + 1);
            y0 = __builtin_nontemporal_load(c + 2 * i); y1 = __builtin_nontemporal_load(c + 2 * i + 1);
        } else {
            x0 = a[2 * i]; x1 = a[2 * i + 1]; y0 = c[2 * i]; y1 = c[2 * i + 1];
        }
        x0 += y0; x1 += y1;
        if (nt_store) { __builtin_nontemporal_store(x0, b + 2 * i); __builtin_nontemporal_store(x1, b + 2 * i + 1); }
        else { b[2 * i] = x0; b[2 * i + 1] = x1; }
    }
}
__global__ void __launch_bounds__(256) write16_nt(double *__restrict__ b, long n)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        __builtin_nontemporal_store(1.0, b + 2 * i); __builtin_nontemporal_store(2.0, b + 2 * i + 1);
    }
}

int main(int argc, char **argv)
{
    const long bytes = (argc > 1 ? atol(argv[1]) : 2048L) << 20;   // MiB per buffer
    const int reps = argc > 2 ? atoi(argv[2]) : 10;
    const long n16 = bytes / 16, n8 = bytes / 8;
    double2 *a, *b, *c; double *out;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&c, bytes)); CK(hipMalloc(&out, 64));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes)); CK(hipMemset(c, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grids[] = {2048, 8192};
    for (int gi = 0; gi < 2; ++gi) {
        const int grid = grids[gi];
        struct { const char *name; double moved; } tests[] = {
            {"read16", (double)bytes}, {"read8", (double)bytes}, {"write16", (double)bytes},
            {"copy16", 2.0 * bytes}, {"add16", 3.0 * bytes}, {"add16_chunk", 3.0 * bytes},
            {"write16_nt", (double)bytes}, {"add16c_ntS", 3.0 * bytes}, {"add16c_ntL", 3.0 * bytes}, {"add16c_ntLS", 3.0 * bytes}};
        for (int t = 0; t < 10; ++t) {
            float best = 1e30f;
            for (int r = 0; r < reps; ++r) {
                CK(hipEventRecord(e0));
                switch (t) {
                case 0: hipLaunchKernelGGL(read16, dim3(grid), dim3(256), 0, 0, a, out, n16); break;
                case 1: hipLaunchKernelGGL(read8, dim3(grid), dim3(256), 0, 0, (const double *)a, out, n8); break;
                case 2: hipLaunchKernelGGL(write16, dim3(grid), dim3(256), 0, 0, b, n16); break;
                case 3: hipLaunchKernelGGL(copy16, dim3(grid), dim3(256), 0, 0, a, b, n16); break;
                case 4: hipLaunchKernelGGL(add16, dim3(grid), dim3(256), 0, 0, a, c, b, n16); break;
                case 5: hipLaunchKernelGGL(add16_chunk, dim3(grid), dim3(256), 0, 0, a, c, b, n16, (n16 + grid - 1) / grid); break;
                case 6: hipLaunchKernelGGL(write16_nt, dim3(grid), dim3(256), 0, 0, (double *)b, n16); break;
                case 7: hipLaunchKernelGGL(add16_chunk_nt, dim3(grid), dim3(256), 0, 0, (const double *)a, (const double *)c, (double *)b, n16, (n16 + grid - 1) / grid, 0, 1); break;
                case 8: hipLaunchKernelGGL(add16_chunk_nt, dim3(grid), dim3(256), 0, 0, (const double *)a, (const double *)c, (double *)b, n16, (n16 + grid - 1) / grid, 1, 0); break;
                case 9: hipLaunchKernelGGL(add16_chunk_nt, dim3(grid), dim3(256), 0, 0, (const double *)a, (const double *)c, (double *)b, n16, (n16 + grid - 1) / grid, 1, 1); break;
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            printf("grid=%d %-12s %8.3f ms  %8.1f GB/s  (%.3f GB moved)\n", grid, tests[t].name, best, tests[t].moved / best / 1e6, tests[t].moved / 1e9);
        }
    }
    return 0;
}
