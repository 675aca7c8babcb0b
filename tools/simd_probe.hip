// Which SIMD does wave w of a 640-thread workgroup land on?  (k_lex_wg's shape: 10 waves, 42 KB of LDS, <= 80 VGPRs, two
// workgroups per CU.)  Every wave reports HW_ID; the host prints, per workgroup, the SIMD of waves 0..9 and how the two
// workgroups of a CU interleave.   hipcc --offload-arch=gfx950 -O2 -o tools/simd_probe tools/simd_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

__global__ void __launch_bounds__(640) __attribute__((amdgpu_waves_per_eu(6, 8)))
probe(unsigned *out, unsigned long long spin)
{
    __shared__ double pad[5200];                       // ~42 KB, as k_lex_wg
    pad[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
    const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 10 + threadIdx.x / 64) * 2] = hw;
        out[(blockIdx.x * 10 + threadIdx.x / 64) * 2 + 1] = xcc;
    }
    const unsigned long long t0 = wall_clock64();       // stay resident so that both workgroups of a CU are there together
    while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
    if (pad[(threadIdx.x * 7) % 5200] < 0) out[0] = 0;
}

int main()
{
    const int wgs = 512;
    unsigned *d = nullptr;
    hipMalloc(&d, sizeof(unsigned) * wgs * 20);
    hipMemset(d, 0, sizeof(unsigned) * wgs * 20);
    hipLaunchKernelGGL(probe, dim3(wgs), dim3(640), 0, 0, d, 200000ull);       // 2 ms
    hipDeviceSynchronize();
    std::vector<unsigned> h(wgs * 20);
    hipMemcpy(h.data(), d, sizeof(unsigned) * h.size(), hipMemcpyDeviceToHost);
    std::map<std::vector<int>, int> patterns;
    std::map<unsigned, std::vector<int>> per_cu;        // CU key -> SIMD histogram over both workgroups
    for (int w = 0; w < wgs; ++w) {
        std::vector<int> p;
        for (int k = 0; k < 10; ++k) {
            const unsigned hw = h[(w * 10 + k) * 2], xcc = h[(w * 10 + k) * 2 + 1];
            const int simd = (hw >> 4) & 3;
            p.push_back(simd);
            const unsigned key = (xcc << 16) | ((hw >> 8) & 0xff);
            auto &v = per_cu[key];
            v.resize(4);
            v[simd]++;
        }
        patterns[p]++;
    }
    printf("SIMD of waves 0..9 of a workgroup (pattern: count)\n");
    for (auto &kv : patterns) {
        for (int s : kv.first) printf("%d", s);
        printf(": %d\n", kv.second);
    }
    std::map<std::vector<int>, int> cu_hist;
    for (auto &kv : per_cu) cu_hist[kv.second]++;
    printf("waves per SIMD of a CU, both workgroups (histogram: CUs)\n");
    for (auto &kv : cu_hist) printf("%d %d %d %d: %d\n", kv.first[0], kv.first[1], kv.first[2], kv.first[3], kv.second);
    printf("CUs used: %zu\n", per_cu.size());
    return 0;
}
