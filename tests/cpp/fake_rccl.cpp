// fake_rccl.cpp — a TEST-ONLY stand-in for librccl.so.1 (never shipped, never linked by the product).
//
// libccp_gs.so binds RCCL at run time by name (csrc/ccp_comm.hip: the twelve nccl* entry points below) and
// CCP_GS_RCCL_LIB names the library to bind.  A test box has ONE GPU, so the library's own multi-rank path —
// ccp_grid_attach_comm's all-gather, issue_exchange()'s grouped ncclSend/ncclRecv pairs, the all-reduced
// stop rule of ccp_grid_gauss_seidel_rowblocked — could never run with a neighbour there.  This transport lets
// it: the ranks of a communicator are host THREADS of one process that share the card;
//   * ncclSend / ncclRecv are matched FIFO per (source, destination) pair and executed as ONE device-to-device
//     copy on the RECEIVER's stream, which first waits for an event the sender recorded on ITS stream when it
//     posted the message; the sender's stream then waits for the copy (its buffer may be reused after that) —
//     the stream semantics of the real calls;
//   * a group posts all its sends, then performs all its receives, then completes its sends, so the
//     send-up / recv-up / send-down / recv-down pattern of a halo exchange cannot deadlock;
//   * ncclAllReduce / ncclAllGather are host-staged between two barriers (tiny messages: partition check,
//     stop rule, norms), reduced in rank order on every rank alike.
// Every wait is bounded (FAKE_RCCL_TIMEOUT_S, default 60 s): a rank that never arrives is an error
// (ncclInternalError), not a hang.  fake_rccl_stats() lets a test prove the messages went through here.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace {

constexpr int kFakeVersion = 99901;            // what ncclGetVersion reports: tests recognise the transport by it

std::atomic<long> g_sends{0}, g_recvs{0}, g_bytes{0}, g_allreduces{0}, g_allgathers{0};

int timeout_s()
{
    static const int t = getenv("FAKE_RCCL_TIMEOUT_S") ? std::max(1, atoi(getenv("FAKE_RCCL_TIMEOUT_S"))) : 60;
    return t;
}

size_t type_bytes(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}

struct Msg {
    const void *src = nullptr;
    size_t bytes = 0;
    hipEvent_t ready = nullptr;       // recorded on the sender's stream when the message was posted
    hipEvent_t done = nullptr;        // recorded on the receiver's stream behind the copy
    bool copied = false;
    bool failed = false;
};

struct World {
    int n = 0;
    std::mutex m;
    std::condition_variable cv;
    int joined = 0, left = 0;
    std::vector<char> rank_taken;
    int bar_count = 0;
    long bar_gen = 0;
    std::map<std::pair<int, int>, std::deque<std::shared_ptr<Msg>>> box;     // (source, destination) -> posted, not yet received
    std::vector<std::vector<char>> slots;                                    // one staging slot per rank (collectives)
};

struct FakeComm {
    std::shared_ptr<World> w;
    int rank = 0;
    std::string key;
    std::vector<hipEvent_t> garbage;      // events of finished messages, destroyed with the communicator
};

std::mutex g_registry_mutex;
std::map<std::string, std::shared_ptr<World>> g_registry;

struct Op {
    bool send;
    const void *src;
    void *dst;
    size_t bytes;
    int peer;
    FakeComm *comm;
    hipStream_t stream;
    std::shared_ptr<Msg> msg;
};
thread_local int t_group_depth = 0;
thread_local std::vector<Op> t_ops;

template <class Pred>
bool wait_for(World &w, std::unique_lock<std::mutex> &lk, Pred p)
{
    return w.cv.wait_for(lk, std::chrono::seconds(timeout_s()), p);
}

// all ranks of the world; false on timeout
bool barrier(World &w)
{
    std::unique_lock<std::mutex> lk(w.m);
    const long gen = w.bar_gen;
    if (++w.bar_count == w.n) {
        w.bar_count = 0;
        ++w.bar_gen;
        w.cv.notify_all();
        return true;
    }
    return wait_for(w, lk, [&] { return w.bar_gen != gen; });
}

ncclResult_t run_ops(std::vector<Op> &ops)
{
    ncclResult_t result = ncclSuccess;
    // 1. post every send (never blocks)
    for (Op &o : ops) {
        if (!o.send) continue;
        auto m = std::make_shared<Msg>();
        m->src = o.src;
        m->bytes = o.bytes;
        if (hipEventCreateWithFlags(&m->ready, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&m->done, hipEventDisableTiming) != hipSuccess ||
            hipEventRecord(m->ready, o.stream) != hipSuccess)
            return ncclUnhandledCudaError;
        o.msg = m;
        World &w = *o.comm->w;
        {
            std::lock_guard<std::mutex> lk(w.m);
            w.box[{o.comm->rank, o.peer}].push_back(m);
        }
        w.cv.notify_all();
        g_sends++;
    }
    // 2. every receive: wait for the matching send to be posted, copy on MY stream behind the sender's event
    for (Op &o : ops) {
        if (o.send) continue;
        World &w = *o.comm->w;
        std::shared_ptr<Msg> m;
        {
            std::unique_lock<std::mutex> lk(w.m);
            auto &q = w.box[{o.peer, o.comm->rank}];
            if (!wait_for(w, lk, [&] { return !q.empty(); })) {
                fprintf(stderr, "[fake_rccl] rank %d: no send from rank %d arrived within %d s\n", o.comm->rank, o.peer, timeout_s());
                result = ncclInternalError;
                continue;
            }
            m = q.front();
            q.pop_front();
        }
        bool ok = m->bytes == o.bytes;
        if (!ok) fprintf(stderr, "[fake_rccl] rank %d <- %d: message of %zu bytes meets a receive of %zu\n", o.comm->rank, o.peer, m->bytes, o.bytes);
        ok = ok && hipStreamWaitEvent(o.stream, m->ready, 0) == hipSuccess;
        ok = ok && (o.bytes == 0 || hipMemcpyAsync(o.dst, m->src, o.bytes, hipMemcpyDeviceToDevice, o.stream) == hipSuccess);
        ok = ok && hipEventRecord(m->done, o.stream) == hipSuccess;
        {
            std::lock_guard<std::mutex> lk(w.m);
            m->copied = true;
            m->failed = !ok;
        }
        w.cv.notify_all();
        if (!ok) result = ncclInvalidArgument;
        g_recvs++;
        g_bytes += (long)o.bytes;
    }
    // 3. complete every send: my stream may touch the buffer again once the receiver's copy is done
    for (Op &o : ops) {
        if (!o.send || !o.msg) continue;
        World &w = *o.comm->w;
        bool arrived;
        {
            std::unique_lock<std::mutex> lk(w.m);
            arrived = wait_for(w, lk, [&] { return o.msg->copied; });
        }
        if (!arrived) {
            fprintf(stderr, "[fake_rccl] rank %d: rank %d never received within %d s\n", o.comm->rank, o.peer, timeout_s());
            result = ncclInternalError;
            continue;
        }
        if (o.msg->failed) result = ncclInvalidArgument;
        else if (hipStreamWaitEvent(o.stream, o.msg->done, 0) != hipSuccess) result = ncclUnhandledCudaError;
        o.comm->garbage.push_back(o.msg->ready);
        o.comm->garbage.push_back(o.msg->done);
    }
    return result;
}

ncclResult_t p2p(bool send, const void *src, void *dst, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
    FakeComm *c = reinterpret_cast<FakeComm *>(comm);
    const size_t tb = type_bytes(type);
    if (!c || tb == 0 || peer < 0 || peer >= c->w->n || peer == c->rank) return ncclInvalidArgument;
    Op o{send, src, dst, count * tb, peer, c, stream, nullptr};
    if (t_group_depth > 0) {
        t_ops.push_back(o);
        return ncclSuccess;
    }
    std::vector<Op> one{o};
    return run_ops(one);
}

template <typename T>
void reduce_into(std::vector<char> &out, const std::vector<std::vector<char>> &slots, size_t count, ncclRedOp_t op)
{
    T *o = reinterpret_cast<T *>(out.data());
    for (size_t i = 0; i < count; ++i) {
        T acc = reinterpret_cast<const T *>(slots[0].data())[i];
        for (size_t r = 1; r < slots.size(); ++r) {
            const T v = reinterpret_cast<const T *>(slots[r].data())[i];
            if (op == ncclSum) acc = acc + v;
            else if (op == ncclMax) acc = v > acc ? v : acc;
            else if (op == ncclMin) acc = v < acc ? v : acc;
            else if (op == ncclProd) acc = acc * v;
        }
        o[i] = acc;
    }
}

}  // namespace

extern "C" {

// test hook (not an nccl symbol)
void fake_rccl_stats(long *sends, long *recvs, long *bytes, long *allreduces, long *allgathers)
{
    if (sends) *sends = g_sends;
    if (recvs) *recvs = g_recvs;
    if (bytes) *bytes = g_bytes;
    if (allreduces) *allreduces = g_allreduces;
    if (allgathers) *allgathers = g_allgathers;
}

ncclResult_t ncclGetVersion(int *version)
{
    if (!version) return ncclInvalidArgument;
    *version = kFakeVersion;
    return ncclSuccess;
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    static std::atomic<unsigned long> next{1};
    if (!id) return ncclInvalidArgument;
    std::memset(id, 0, sizeof(*id));
    std::snprintf(id->internal, sizeof(id->internal), "fake-rccl-%d-%lu", (int)getpid(), next++);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    const std::string key(id.internal, sizeof(id.internal));
    std::shared_ptr<World> w;
    {
        std::lock_guard<std::mutex> lk(g_registry_mutex);
        auto &slot = g_registry[key];
        if (!slot) {
            slot = std::make_shared<World>();
            slot->n = nranks;
            slot->rank_taken.assign((size_t)nranks, 0);
            slot->slots.resize((size_t)nranks);
        }
        w = slot;
    }
    {
        std::unique_lock<std::mutex> lk(w->m);
        if (w->n != nranks || w->rank_taken[(size_t)rank]) return ncclInvalidArgument;
        w->rank_taken[(size_t)rank] = 1;
        ++w->joined;
        w->cv.notify_all();
        if (!wait_for(*w, lk, [&] { return w->joined == w->n; })) {        // collective, as the real call
            fprintf(stderr, "[fake_rccl] rank %d: only %d of %d ranks joined within %d s\n", rank, w->joined, w->n, timeout_s());
            return ncclInternalError;
        }
    }
    FakeComm *c = new FakeComm();
    c->w = w;
    c->rank = rank;
    c->key = key;
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    FakeComm *c = reinterpret_cast<FakeComm *>(comm);
    if (!c) return ncclSuccess;
    (void)hipDeviceSynchronize();
    for (hipEvent_t e : c->garbage) (void)hipEventDestroy(e);
    bool last;
    {
        std::lock_guard<std::mutex> lk(c->w->m);
        last = ++c->w->left == c->w->n;
    }
    if (last) {
        std::lock_guard<std::mutex> lk(g_registry_mutex);
        g_registry.erase(c->key);
    }
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclCommAbort(ncclComm_t comm) { return ncclCommDestroy(comm); }

const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error (fake transport)";
    case ncclUnhandledCudaError: return "unhandled HIP error (fake transport)";
    case ncclInternalError: return "a rank did not arrive in time (fake transport)";
    case ncclInvalidArgument: return "invalid argument / mismatched message (fake transport)";
    default: return "error (fake transport)";
    }
}

ncclResult_t ncclGroupStart()
{
    ++t_group_depth;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
    if (t_group_depth <= 0) return ncclInvalidUsage;
    if (--t_group_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(t_ops);
    return run_ops(ops);
}

ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    return p2p(true, sendbuff, nullptr, count, datatype, peer, comm, stream);
}

ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    return p2p(false, nullptr, recvbuff, count, datatype, peer, comm, stream);
}

ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream)
{
    FakeComm *c = reinterpret_cast<FakeComm *>(comm);
    const size_t tb = type_bytes(datatype);
    if (!c || tb == 0 || t_group_depth > 0) return ncclInvalidArgument;
    if (datatype != ncclFloat64 && datatype != ncclFloat32 && datatype != ncclInt32 && datatype != ncclInt64) return ncclInvalidArgument;
    World &w = *c->w;
    const size_t bytes = count * tb;
    std::vector<char> mine(bytes);
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (bytes && hipMemcpy(mine.data(), sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    {
        std::lock_guard<std::mutex> lk(w.m);
        w.slots[(size_t)c->rank] = mine;
    }
    if (!barrier(w)) return ncclInternalError;
    bool same = true;
    for (const auto &s : w.slots) same = same && s.size() == bytes;
    std::vector<char> out(bytes);
    if (same) {
        if (datatype == ncclFloat64) reduce_into<double>(out, w.slots, count, op);
        else if (datatype == ncclFloat32) reduce_into<float>(out, w.slots, count, op);
        else if (datatype == ncclInt32) reduce_into<int>(out, w.slots, count, op);
        else reduce_into<long long>(out, w.slots, count, op);
    }
    if (!barrier(w)) return ncclInternalError;                       // every rank has read the slots
    if (!same) return ncclInvalidArgument;
    if (bytes && hipMemcpy(recvbuff, out.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    g_allreduces++;
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream)
{
    FakeComm *c = reinterpret_cast<FakeComm *>(comm);
    const size_t tb = type_bytes(datatype);
    if (!c || tb == 0 || t_group_depth > 0) return ncclInvalidArgument;
    World &w = *c->w;
    const size_t bytes = sendcount * tb;
    std::vector<char> mine(bytes);
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (bytes && hipMemcpy(mine.data(), sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    {
        std::lock_guard<std::mutex> lk(w.m);
        w.slots[(size_t)c->rank] = mine;
    }
    if (!barrier(w)) return ncclInternalError;
    std::vector<char> out(bytes * (size_t)w.n);
    bool same = true;
    for (int r = 0; r < w.n; ++r) {
        same = same && w.slots[(size_t)r].size() == bytes;
        if (same && bytes) std::memcpy(out.data() + bytes * (size_t)r, w.slots[(size_t)r].data(), bytes);
    }
    if (!barrier(w)) return ncclInternalError;
    if (!same) return ncclInvalidArgument;
    if (bytes && hipMemcpy(recvbuff, out.data(), out.size(), hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    g_allgathers++;
    return ncclSuccess;
}

}  // extern "C"
