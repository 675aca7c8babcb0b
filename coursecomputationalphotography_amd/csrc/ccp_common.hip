// ccp_common.hip — status strings and device probing of libccp_gs.so.
#include "ccp_common.hpp"

extern "C" {

const char *ccp_status_string(int status)
{
    switch (status) {
    case CCP_OK: return "ok";
    case CCP_ERR_BAD_ARG: return "bad argument";
    case CCP_ERR_NO_DEVICE: return "no usable HIP device (this library has no CPU fallback)";
    case CCP_ERR_HIP: return "HIP runtime error";
    case CCP_ERR_ALLOC: return "allocation failed";
    case CCP_ERR_STATE: return "invalid call sequence for this handle";
    case CCP_ERR_UNSUPPORTED: return "unsupported input";
    case CCP_ERR_RCCL: return "RCCL unavailable or a collective call failed";
    default: return "unknown status";
    }
}

int ccp_abi_version(void) { return CCP_GS_ABI_VERSION; }

int ccp_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n < 0 ? 0 : n;
}

}  // extern "C"
