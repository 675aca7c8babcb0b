// ccp_common.hpp — shared helpers of libccp_gs.so (MI355X / gfx950 only; wave = 64 lanes).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <new>

#include "ccp_gs.h"

namespace ccp {

constexpr int kWave = 64;          // CDNA wavefront width
constexpr int kBlock = 256;        // 4 waves: one per SIMD of a CU
constexpr int kMaxChannels = 8;

// HIP error -> status.  The first failing call is remembered for ccp_last_error_string-style
// diagnostics on stderr when CCP_GS_DEBUG is set.
inline int hip_fail(hipError_t e, const char *what, const char *file, int line)
{
    if (getenv("CCP_GS_DEBUG"))
        fprintf(stderr, "[ccp_gs] %s failed: %s (%s:%d)\n", what, hipGetErrorString(e), file, line);
    if (e == hipErrorOutOfMemory) return CCP_ERR_ALLOC;
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice) return CCP_ERR_NO_DEVICE;
    return CCP_ERR_HIP;
}

#define CCP_HIP(call)                                                        \
    do {                                                                     \
        hipError_t e_ = (call);                                              \
        if (e_ != hipSuccess) return ::ccp::hip_fail(e_, #call, __FILE__, __LINE__); \
    } while (0)

#define CCP_TRY(expr)                    \
    do {                                 \
        int s_ = (expr);                 \
        if (s_ != CCP_OK) return s_;     \
    } while (0)

// No C++ exception may cross the extern "C" boundary: every entry point is a function-try-block
// ending in this handler (host allocations of multi-GB schedules, std::thread construction, ...).
#define CCP_ABI_CATCH                                          \
    catch (const std::bad_alloc &) { return CCP_ERR_ALLOC; }   \
    catch (...) { return CCP_ERR_STATE; }

// Owning device buffer.
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    int alloc(size_t count)
    {
        release();
        if (count == 0) count = 1;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T));
        if (e != hipSuccess) {
            p = nullptr;
            return hip_fail(e, "hipMalloc", __FILE__, __LINE__);
        }
        n = count;
        return CCP_OK;
    }
};

}  // namespace ccp
struct ccp_grid;
namespace ccp {
// Library-internal (not part of the C ABI): install the mask of a Dirichlet-mask grid from a DEVICE buffer that is
// already in the grid's layout (one byte per element of a channel plane, colour half-rows of `pitch` bytes).
int grid_set_mask_split_device(ccp_grid *g, const unsigned char *split_mask_dev, long unknowns);
// Library-internal: the handle's owner asks for the layout (ccp_grid_get_layout) at every use, so x and its ping-pong
// partner may swap roles after a run of passes (ccp_grid.hip: run_unchecked).
int grid_set_allow_swap(ccp_grid *g, bool on);

inline int select_device(int device)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return CCP_ERR_NO_DEVICE;
    if (device < 0 || device >= count) return CCP_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return CCP_ERR_NO_DEVICE;
    return CCP_OK;
}

// ---- wave / block reductions (deterministic: fixed shuffle tree, fixed wave order) ---------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;
}

// Sum over a block of kBlock threads; result valid in thread 0.  `scratch` holds kBlock/kWave
// doubles of LDS.
__device__ __forceinline__ double block_sum(double v, double *scratch)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    double total = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < kBlock / kWave; ++w) total += scratch[w];
    }
    __syncthreads();
    return total;
}

}  // namespace ccp
