// ccp_comm.hip — RCCL communicator handles of the C ABI (ccp_comm_*).  See ccp_comm.hpp.
#include "ccp_comm.hpp"

#include <dlfcn.h>
#include <link.h>

#include <cstring>
#include <mutex>
#include <set>
#include <string>

namespace ccp {

namespace {

// Distinct files mapped in this process whose name contains `needle` (dl_iterate_phdr).
struct LoadedCopies {
    const char *needle;
    std::set<std::string> paths;
};
int collect_copies(struct dl_phdr_info *info, size_t, void *data)
{
    auto *lc = static_cast<LoadedCopies *>(data);
    if (info->dlpi_name && std::strstr(info->dlpi_name, lc->needle)) {
        char real[4096];
        lc->paths.insert(realpath(info->dlpi_name, real) ? std::string(real) : std::string(info->dlpi_name));
    }
    return 0;
}
std::set<std::string> loaded_copies(const char *needle)
{
    LoadedCopies lc{needle, {}};
    dl_iterate_phdr(collect_copies, &lc);
    return lc.paths;
}

}  // namespace

// Why the communicator layer is unusable ("" while it is usable); ccp_comm_probe prints it under CCP_GS_DEBUG.
static std::string g_rccl_refusal;

// One HIP runtime and one collective library per process.  Two copies of either (say /opt/rocm's next to the ones a
// Python package bundles) each keep their own device state and their own exit handlers: at best the communicator runs
// on a runtime the grid handles do not live on, at worst the process aborts in the teardown.  Looked at when RCCL is
// bound AND again at every ccp_comm_probe / ccp_comm_create: a second copy mapped later is refused as well, whatever
// order the host program loaded things in.
static std::string two_copies_refusal()
{
    for (const char *needle : {"libamdhip64.so", "librccl.so"}) {
        const std::set<std::string> copies = loaded_copies(needle);
        if (copies.size() > 1) {
            std::string why = std::string("two copies of ") + needle + " are mapped in this process:";
            for (const std::string &c : copies) why += " " + c;
            return why;
        }
    }
    return std::string();
}

const RcclApi *rccl_api()
{
    static RcclApi api{};
    static bool ok = false;
    static std::once_flag once;
    std::call_once(once, [] {
        // by soname: a copy already in the process wins (e.g. the one PyTorch-ROCm ships with its HIP runtime)
        const char *override_path = getenv("CCP_GS_RCCL_LIB");
        // RTLD_NOLOAD first: a collective library the process already carries is THE one to use — it belongs to
        // the HIP runtime the process runs on.  Only a process without one gets the soname lookup.
        //
        // RTLD_LOCAL, never RTLD_GLOBAL.  RCCL pulls in librocm_smi64.so, which exports C++ globals
        // (amd::smi::Device::devInfoTypesStrings, a std::map) that libamd_smi.so — loaded by PyTorch-ROCm when it is
        // imported — exports as well.  With librocm_smi64 in the GLOBAL lookup scope, libamd_smi's references bind to
        // librocm_smi64's instance: both libraries' static initialisers construct the one map, both exit handlers
        // destroy it, and the process dies at exit with "double free or corruption (!prev)" inside
        // std::map<amd::smi::DevInfoTypes, char const*>::~map() (round 2's gpurun_out/exit_v4.txt; stack and library
        // list taken with rocgdb: tools/exit_probe.py, profiles/r03_exit_probe.txt).  Only dlsym() on this handle is
        // ever needed, so nothing has to be global.
        void *h = nullptr;
        if (!override_path) {
            h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL);
            if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL);
        }
        if (!h) h = dlopen(override_path ? override_path : "librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h && !override_path) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) {
            const char *why = dlerror();
            g_rccl_refusal = std::string("cannot load RCCL: ") + (why ? why : "?");
            if (getenv("CCP_GS_DEBUG")) fprintf(stderr, "[ccp_gs] %s\n", g_rccl_refusal.c_str());
            return;
        }
        g_rccl_refusal = two_copies_refusal();
        if (!g_rccl_refusal.empty()) {
            if (getenv("CCP_GS_DEBUG")) fprintf(stderr, "[ccp_gs] RCCL refused: %s\n", g_rccl_refusal.c_str());
            return;
        }
        bool all = true;
        auto sym = [&](const char *name) -> void * {
            void *p = dlsym(h, name);
            if (!p) all = false;
            return p;
        };
        api.GetVersion = reinterpret_cast<decltype(api.GetVersion)>(sym("ncclGetVersion"));
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.CommAbort = reinterpret_cast<decltype(api.CommAbort)>(sym("ncclCommAbort"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
        api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
        api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
        api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
        api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
        api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
        ok = all;
    });
    return ok ? &api : nullptr;
}

}  // namespace ccp

using namespace ccp;

extern "C" {

static_assert(sizeof(ncclUniqueId) == CCP_COMM_ID_BYTES, "ccp_gs.h promises the size of ncclUniqueId");

// Can this process take part in a communicator on `device`?  Not collective: a host program calls it on every
// rank and lets the ranks agree BEFORE the collective ccp_comm_create, so that a rank without RCCL (or without
// the device) does not leave the others waiting inside ncclCommInitRank.
int ccp_comm_probe(int32_t device)
try {
    const RcclApi *api = rccl_api();
    if (!api) {
        if (getenv("CCP_GS_DEBUG")) fprintf(stderr, "[ccp_gs] ccp_comm_probe: %s\n", g_rccl_refusal.c_str());
        return CCP_ERR_RCCL;
    }
    const std::string later = two_copies_refusal();             // (a second copy mapped since RCCL was bound)
    if (!later.empty()) {
        if (getenv("CCP_GS_DEBUG")) fprintf(stderr, "[ccp_gs] ccp_comm_probe: %s\n", later.c_str());
        return CCP_ERR_RCCL;
    }
    int v = 0;
    if (api->GetVersion(&v) != ncclSuccess) return CCP_ERR_RCCL;
    return select_device(device);
} CCP_ABI_CATCH

int ccp_comm_unique_id(uint8_t *id_out)
try {
    if (!id_out) return CCP_ERR_BAD_ARG;
    const RcclApi *api = rccl_api();
    if (!api) return CCP_ERR_RCCL;
    ncclUniqueId id;
    CCP_RCCL(api->GetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof(id));
    return CCP_OK;
} CCP_ABI_CATCH

int ccp_comm_create(const uint8_t *id_bytes, int32_t rank, int32_t world, int32_t device, ccp_comm **out)
try {
    if (!out) return CCP_ERR_BAD_ARG;
    *out = nullptr;
    if (!id_bytes || world < 1 || rank < 0 || rank >= world) return CCP_ERR_BAD_ARG;
    CCP_TRY(select_device(device));
    const RcclApi *api = rccl_api();
    if (!api) return CCP_ERR_RCCL;
    if (!two_copies_refusal().empty()) return CCP_ERR_RCCL;     // as ccp_comm_probe (which the ranks agree on first)
    ccp_comm *c = new (std::nothrow) ccp_comm();
    if (!c) return CCP_ERR_ALLOC;
    c->rank = rank;
    c->world = world;
    c->device = device;
    (void)api->GetVersion(&c->version);
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, sizeof(id));
    const ncclResult_t r = api->CommInitRank(&c->comm, world, id, rank);      // collective over all ranks
    if (r != ncclSuccess) {
        delete c;
        return rccl_fail(r, "ncclCommInitRank", __FILE__, __LINE__);
    }
    const int st = c->scratch.alloc((size_t)std::max(64, 8 * world));
    if (st != CCP_OK) {
        (void)api->CommDestroy(c->comm);
        delete c;
        return st;
    }
    *out = c;
    return CCP_OK;
} CCP_ABI_CATCH

int ccp_comm_destroy(ccp_comm *c)
try {
    if (!c) return CCP_OK;
    (void)hipSetDevice(c->device);
    const RcclApi *api = rccl_api();
    int st = CCP_OK;
    if (api && c->comm) {
        (void)hipDeviceSynchronize();                       // nothing of ours may still be queued on it
        if (api->CommDestroy(c->comm) != ncclSuccess) st = CCP_ERR_RCCL;
    }
    delete c;
    return st;
} CCP_ABI_CATCH

int ccp_comm_info(ccp_comm *c, int32_t *rank, int32_t *world, int32_t *device, int32_t *rccl_version)
try {
    if (!c) return CCP_ERR_BAD_ARG;
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (device) *device = c->device;
    if (rccl_version) *rccl_version = c->version;
    return CCP_OK;
} CCP_ABI_CATCH

// In-place sum over all ranks of `count` host doubles (ncclAllReduce on the communicator's device;
// staged through its scratch buffer).  For callers that keep their own statistics.
int ccp_comm_all_reduce_sum(ccp_comm *c, double *values, int32_t count)
try {
    if (!c || !values || count < 0 || (size_t)count > c->scratch.n) return CCP_ERR_BAD_ARG;
    if (count == 0) return CCP_OK;
    if (hipSetDevice(c->device) != hipSuccess) return CCP_ERR_NO_DEVICE;
    const RcclApi *api = rccl_api();
    if (!api) return CCP_ERR_RCCL;
    CCP_HIP(hipMemcpyAsync(c->scratch.p, values, sizeof(double) * count, hipMemcpyHostToDevice, nullptr));
    CCP_RCCL(api->AllReduce(c->scratch.p, c->scratch.p, (size_t)count, ncclDouble, ncclSum, c->comm, nullptr));
    CCP_HIP(hipMemcpyAsync(values, c->scratch.p, sizeof(double) * count, hipMemcpyDeviceToHost, nullptr));
    CCP_HIP(hipStreamSynchronize(nullptr));
    return CCP_OK;
} CCP_ABI_CATCH

// In-place maximum over all ranks (the bench's max-over-ranks wall time).
int ccp_comm_all_reduce_max(ccp_comm *c, double *values, int32_t count)
try {
    if (!c || !values || count < 0 || (size_t)count > c->scratch.n) return CCP_ERR_BAD_ARG;
    if (count == 0) return CCP_OK;
    if (hipSetDevice(c->device) != hipSuccess) return CCP_ERR_NO_DEVICE;
    const RcclApi *api = rccl_api();
    if (!api) return CCP_ERR_RCCL;
    CCP_HIP(hipMemcpyAsync(c->scratch.p, values, sizeof(double) * count, hipMemcpyHostToDevice, nullptr));
    CCP_RCCL(api->AllReduce(c->scratch.p, c->scratch.p, (size_t)count, ncclDouble, ncclMax, c->comm, nullptr));
    CCP_HIP(hipMemcpyAsync(values, c->scratch.p, sizeof(double) * count, hipMemcpyDeviceToHost, nullptr));
    CCP_HIP(hipStreamSynchronize(nullptr));
    return CCP_OK;
} CCP_ABI_CATCH

}  // extern "C"
