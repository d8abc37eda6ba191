// loopback_nccl.hip — TEST INFRASTRUCTURE, not part of the product: a stand-in for librccl that runs an N-rank
// communicator between THREADS OF ONE PROCESS whose contexts share one GPU, so that the library's own multi-GPU code
// (pd_comm_init / pd_comm_init_all, pd_sliced_sum_start / _finish: grouped send/recv, message chunking, the all-reduce
// of tile sums and exception counts, the all-gather of exception blocks, the gather of the partials to the root) can be
// executed for N = 2, 3, 8 on a 1-GPU box.  The library dlopens whatever PANDEPTH_RCCL_LIB names before it looks for
// librccl.so.1; this file exports exactly the eleven entry points it resolves, with RCCL's signatures (rccl.h), and moves
// the bytes with hipMemcpyAsync / a sum kernel on the callers' own streams, ordered by events the way RCCL's calls order
// themselves on the stream they are given.
//
// Semantics kept: a group of sends/receives completes as a unit; a receive from rank p matches p's sends to this rank in
// the order they were issued; collectives are matched in call order per communicator; all calls only ENQUEUE on the
// caller's stream (the host blocks inside a call only until its peers have made the matching calls, as RCCL may).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {

__global__ void k_sum_i32(const int32_t *const *src, int n_src, int32_t *dst, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        int32_t a = 0;
        for (int k = 0; k < n_src; ++k) a += src[k][i];
        dst[i] = a;
    }
}

size_t dtype_bytes(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    case ncclFloat16: case ncclBfloat16: return 2;
    default: return 0;
    }
}

struct SendRec {
    const void *src; size_t bytes; hipEvent_t ready;      // `ready`: recorded on the sender's stream when the group was issued
    hipEvent_t copied = nullptr; bool matched = false;      // filled in by the receiver
};

struct World {
    int n = 0;
    std::mutex mu; std::condition_variable cv;
    int joined = 0;
    std::map<std::pair<int, int>, std::deque<SendRec *>> box;   // (from, to) -> sends not yet received
    // collectives: one at a time per world (matched in call order)
    uint64_t gen = 0; int waiting = 0;
    std::vector<const void *> c_send; std::vector<void *> c_recv; std::vector<hipEvent_t> c_ready, c_finished;
    std::vector<void *> stage; std::vector<size_t> stage_cap;
    const int32_t **d_ptrs = nullptr;                           // device array of n staged pointers
};

struct Comm { World *w; int rank; };

std::mutex g_mu;
std::map<std::string, World *> g_worlds;                        // unique id -> world being assembled
uint64_t g_next_id = 1;

struct Op { int kind; const void *sbuf; void *rbuf; size_t bytes; int peer; Comm *c; hipStream_t st; };   // kind 0 send, 1 recv
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;

hipEvent_t new_event() { hipEvent_t e = nullptr; (void)hipEventCreateWithFlags(&e, hipEventDisableTiming); return e; }

ncclResult_t run_group(std::vector<Op> &ops)
{
    // 1. post every send
    std::vector<SendRec *> mine;
    for (Op &o : ops) {
        if (o.kind != 0) continue;
        SendRec *r = new SendRec{o.sbuf, o.bytes, new_event()};
        if (hipEventRecord(r->ready, o.st) != hipSuccess) return ncclUnhandledCudaError;
        { std::lock_guard<std::mutex> lk(o.c->w->mu); o.c->w->box[{o.c->rank, o.peer}].push_back(r); }
        o.c->w->cv.notify_all();
        mine.push_back(r);
        o.rbuf = r;                                             // (remember the record for step 3)
    }
    // 2. every receive: wait for the matching send to be posted, copy on the receiver's stream behind the sender's event
    for (Op &o : ops) {
        if (o.kind != 1) continue;
        World *w = o.c->w;
        SendRec *r = nullptr;
        {
            std::unique_lock<std::mutex> lk(w->mu);
            auto &q = w->box[{o.peer, o.c->rank}];
            w->cv.wait(lk, [&] { return !q.empty(); });
            r = q.front(); q.pop_front();
        }
        if (r->bytes != o.bytes) return ncclInvalidArgument;    // RCCL would hang or corrupt: a test must see it
        if (hipStreamWaitEvent(o.st, r->ready, 0) != hipSuccess) return ncclUnhandledCudaError;
        if (o.bytes && hipMemcpyAsync(o.rbuf, r->src, o.bytes, hipMemcpyDeviceToDevice, o.st) != hipSuccess) return ncclUnhandledCudaError;
        hipEvent_t done = new_event();
        if (hipEventRecord(done, o.st) != hipSuccess) return ncclUnhandledCudaError;
        { std::lock_guard<std::mutex> lk(w->mu); r->copied = done; r->matched = true; }
        w->cv.notify_all();
    }
    // 3. this rank's sends are complete (for its stream) once the receivers' copies are: nothing enqueued later may overwrite them
    for (Op &o : ops) {
        if (o.kind != 0) continue;
        SendRec *r = (SendRec *)o.rbuf;
        World *w = o.c->w;
        { std::unique_lock<std::mutex> lk(w->mu); w->cv.wait(lk, [&] { return r->matched; }); }
        if (hipStreamWaitEvent(o.st, r->copied, 0) != hipSuccess) return ncclUnhandledCudaError;
        // (events are left to the process: a test harness, and destroying an event another stream still waits on is not worth the risk)
        delete r;
    }
    return ncclSuccess;
}

void barrier(World *w)
{
    std::unique_lock<std::mutex> lk(w->mu);
    const uint64_t g = w->gen;
    if (++w->waiting == w->n) { w->waiting = 0; ++w->gen; w->cv.notify_all(); }
    else w->cv.wait(lk, [&] { return w->gen != g; });
}

// all ranks call this with the same sequence of collectives; `body` runs on every rank once all have arrived
template <class F> ncclResult_t collective(Comm *c, const void *sbuf, void *rbuf, size_t stage_bytes, hipStream_t st, F body)
{
    World *w = c->w;
    const int r = c->rank;
    // stage this rank's contribution (in-place calls must not race with the peers' reads)
    {
        std::lock_guard<std::mutex> lk(w->mu);
        if (w->stage_cap[r] < stage_bytes) {
            if (w->stage[r]) (void)hipFree(w->stage[r]);
            if (hipMalloc(&w->stage[r], stage_bytes + 256) != hipSuccess) return ncclSystemError;
            w->stage_cap[r] = stage_bytes;
        }
    }
    if (stage_bytes && hipMemcpyAsync(w->stage[r], sbuf, stage_bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return ncclUnhandledCudaError;
    hipEvent_t ready = new_event();
    if (hipEventRecord(ready, st) != hipSuccess) return ncclUnhandledCudaError;
    { std::lock_guard<std::mutex> lk(w->mu); w->c_ready[r] = ready; w->c_recv[r] = rbuf; }
    barrier(w);                                                 // every rank has staged and posted
    for (int k = 0; k < w->n; ++k) if (hipStreamWaitEvent(st, w->c_ready[k], 0) != hipSuccess) return ncclUnhandledCudaError;
    const ncclResult_t rc = body(w, r);
    hipEvent_t fin = new_event();
    if (hipEventRecord(fin, st) != hipSuccess) return ncclUnhandledCudaError;
    { std::lock_guard<std::mutex> lk(w->mu); w->c_finished[r] = fin; }
    barrier(w);                                                 // every rank has enqueued its reads of the staging buffers
    // the staging buffers are reused by the next collective: every rank's stream waits until every rank has consumed them
    for (int k = 0; k < w->n; ++k) if (hipStreamWaitEvent(st, w->c_finished[k], 0) != hipSuccess) return ncclUnhandledCudaError;
    barrier(w);                                                 // ... and nobody posts the next collective's events before all have read these
    return rc;
}

World *make_world(int n)
{
    World *w = new World; w->n = n;
    w->c_send.assign(n, nullptr); w->c_recv.assign(n, nullptr); w->c_ready.assign(n, nullptr); w->c_finished.assign(n, nullptr);
    w->stage.assign(n, nullptr); w->stage_cap.assign(n, 0);
    return w;
}

} // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof *id);
    std::lock_guard<std::mutex> lk(g_mu);
    const uint64_t k = g_next_id++;
    memcpy(id->internal, "pdloop", 6); memcpy(id->internal + 8, &k, 8);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    const std::string key(id.internal, sizeof id.internal);
    World *w;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_worlds.find(key);
        if (it == g_worlds.end()) it = g_worlds.emplace(key, make_world(nranks)).first;
        w = it->second;
    }
    if (w->n != nranks) return ncclInvalidArgument;
    {
        std::unique_lock<std::mutex> lk(w->mu);
        ++w->joined; w->cv.notify_all();
        w->cv.wait(lk, [&] { return w->joined >= w->n; });      // like RCCL: returns when every rank has joined
    }
    *comm = (ncclComm_t) new Comm{w, rank};
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t *comm, int ndev, const int *)
{
    if (!comm || ndev < 1) return ncclInvalidArgument;
    World *w = make_world(ndev);
    w->joined = ndev;
    for (int i = 0; i < ndev; ++i) comm[i] = (ncclComm_t) new Comm{w, i};
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) { delete (Comm *)comm; return ncclSuccess; }   // (worlds live as long as the test process)

ncclResult_t ncclGroupStart() { ++t_depth; return ncclSuccess; }

ncclResult_t ncclGroupEnd()
{
    if (t_depth <= 0) return ncclInvalidUsage;
    if (--t_depth) return ncclSuccess;
    std::vector<Op> ops; ops.swap(t_ops);
    return run_group(ops);
}

ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t st)
{
    Comm *c = (Comm *)comm;
    if (!c || peer < 0 || peer >= c->w->n || !dtype_bytes(dt)) return ncclInvalidArgument;
    t_ops.push_back(Op{0, sendbuff, nullptr, count * dtype_bytes(dt), peer, c, st});
    if (t_depth) return ncclSuccess;
    std::vector<Op> ops; ops.swap(t_ops);
    return run_group(ops);
}

ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t st)
{
    Comm *c = (Comm *)comm;
    if (!c || peer < 0 || peer >= c->w->n || !dtype_bytes(dt)) return ncclInvalidArgument;
    t_ops.push_back(Op{1, nullptr, recvbuff, count * dtype_bytes(dt), peer, c, st});
    if (t_depth) return ncclSuccess;
    std::vector<Op> ops; ops.swap(t_ops);
    return run_group(ops);
}

ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm, hipStream_t st)
{
    Comm *c = (Comm *)comm;
    if (!c || dt != ncclInt32 || op != ncclSum) return ncclInvalidArgument;       // what the library uses
    return collective(c, sendbuff, recvbuff, count * 4, st, [&](World *w, int r) -> ncclResult_t {
        // a per-rank table of the staged buffers, then one sum kernel into this rank's destination
        const int32_t **tab = nullptr;
        if (hipMalloc((void **)&tab, sizeof(void *) * (size_t)w->n) != hipSuccess) return ncclSystemError;
        std::vector<const int32_t *> h((size_t)w->n);
        for (int k = 0; k < w->n; ++k) h[(size_t)k] = (const int32_t *)w->stage[k];
        if (hipMemcpyAsync(tab, h.data(), sizeof(void *) * (size_t)w->n, hipMemcpyHostToDevice, st) != hipSuccess) return ncclUnhandledCudaError;
        if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;  // (h is a local; a test transport may wait)
        if (count) hipLaunchKernelGGL(k_sum_i32, dim3(256), dim3(256), 0, st, tab, w->n, (int32_t *)recvbuff, count);
        if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
        (void)hipFree(tab);
        (void)r;
        return hipGetLastError() == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
    });
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t dt, ncclComm_t comm, hipStream_t st)
{
    Comm *c = (Comm *)comm;
    const size_t eb = dtype_bytes(dt);
    if (!c || !eb) return ncclInvalidArgument;
    const size_t bytes = sendcount * eb;
    return collective(c, sendbuff, recvbuff, bytes, st, [&](World *w, int) -> ncclResult_t {
        for (int k = 0; k < w->n; ++k)
            if (bytes && hipMemcpyAsync((char *)recvbuff + (size_t)k * bytes, w->stage[k], bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return ncclUnhandledCudaError;
        return ncclSuccess;
    });
}

const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "loopback transport: a HIP call failed";
    case ncclSystemError: return "loopback transport: allocation failed";
    case ncclInvalidArgument: return "loopback transport: invalid argument (or a send/recv pair of different sizes)";
    case ncclInvalidUsage: return "loopback transport: invalid usage";
    default: return "loopback transport: error";
    }
}

} // extern "C"
