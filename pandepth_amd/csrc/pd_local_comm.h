// pd_local_comm.h — the IN-PROCESS transport of the sliced sum (include/pandepth_amd.h: pd_comm_init_local).
//
// The `pandepth` executable is ONE process that owns every GPU of the node (`#.list` mode: one context, one rank thread per GPU;
// reference: the sequential accumulate loop PD:2704-3014).  Between threads of one process nothing has to be bootstrapped: a rank
// PULLS what it needs out of its peers' buffers with hipMemcpyPeerAsync (xGMI peer copies on its own stream, ordered by events
// the peers recorded on theirs) and sums with the engine's own add kernel.  No library is loaded — librccl's load alone costs a
// short-lived process 1.1 s warm and 5 s cold and, while it registers its code objects, holds the runtime lock every kernel launch
// of the decode needs (profiles/r05_comm_init.txt) — so a list run pays nothing for having a communicator.
//
// The entry points have the signatures and the stream semantics of the RCCL calls the collective code makes (grouped send / recv,
// all-reduce of int32 sums, all-gather of bytes; everything only ENQUEUES on the caller's stream, the host blocks only until the
// peers have made the matching call), so pd_sliced_sum_start / _finish / pd_sliced_window_sum / pd_sliced_interval_sum run unchanged
// over either transport; RCCL stays the transport between PROCESSES (pd_comm_init: bench.py --gpus N, any launcher with one
// process per GPU).  Contexts may share a GPU (tests: N ranks on one device).
//
// A rank that fails marks the world broken: every peer blocked in a matching call returns an error instead of waiting for ever.
#ifndef PD_LOCAL_COMM_H_
#define PD_LOCAL_COMM_H_
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <vector>
#include "pd_kernels.h"

namespace pdlocal {

struct World;
struct Comm {
    World *w = nullptr;
    int rank = 0, device = 0;
    hipEvent_t ready = nullptr, pulled = nullptr;     // collectives: my contribution is final | my reads of everybody's are queued
    std::vector<hipEvent_t> sent, taken;              // per peer: my sends to p are final | my receive from p is queued
    void *stage = nullptr; size_t stage_cap = 0;      // all-reduce: the peers' contributions, pulled over the links
};
struct SendRec { const void *src; size_t bytes; int device; hipEvent_t ready; hipEvent_t done; bool matched; };
struct World {
    int n = 0, alive = 0;
    std::mutex mu; std::condition_variable cv;
    uint64_t gen = 0; int waiting = 0; bool broken = false;
    std::vector<Comm *> member;
    std::map<std::pair<int, int>, std::deque<SendRec *>> box;    // (from, to) -> sends not yet received
    std::vector<const void *> pub;                               // per rank: the buffer published for the collective under way
};
struct Op { int kind; const void *sbuf; void *rbuf; size_t bytes; int peer; Comm *c; hipStream_t st; SendRec *rec; };   // kind 0 send, 1 receive

// dst[i] += src[i] for any word count and any 4-byte alignment (the engine's own add kernel works on whole int4s of aligned arrays)
__global__ __launch_bounds__(256) void k_local_add_words(int *__restrict__ dst, const int *__restrict__ src, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if ((((uintptr_t)dst | (uintptr_t)src) & 15) == 0) {
        const size_t n4 = n / 4;
        int4 *d4 = reinterpret_cast<int4 *>(dst); const int4 *s4 = reinterpret_cast<const int4 *>(src);
        for (size_t k = i; k < n4; k += stride) { int4 a = d4[k]; const int4 b = s4[k]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; d4[k] = a; }
        for (size_t k = n4 * 4 + i; k < n; k += stride) dst[k] += src[k];
        return;
    }
    for (; i < n; i += stride) dst[i] += src[i];
}
inline void add_words(hipStream_t st, int *dst, const int *src, size_t n)
{
    if (!n) return;
    size_t g = (n / 4 + 255) / 256 + 1; if (g > 8192) g = 8192;
    hipLaunchKernelGGL(k_local_add_words, dim3((unsigned)g), dim3(256), 0, st, dst, src, n);
}

inline void break_world(World *w) { { std::lock_guard<std::mutex> lk(w->mu); w->broken = true; } w->cv.notify_all(); }
#define PDL_HIP(c, call) do { if ((call) != hipSuccess) { (void)hipGetLastError(); break_world((c)->w); return ncclUnhandledCudaError; } } while (0)

// every rank thread arrives; false when some rank has failed meanwhile
inline bool barrier(World *w)
{
    std::unique_lock<std::mutex> lk(w->mu);
    if (w->broken) return false;
    const uint64_t g = w->gen;
    if (++w->waiting == w->n) { w->waiting = 0; ++w->gen; w->cv.notify_all(); }
    else w->cv.wait(lk, [&] { return w->gen != g || w->broken; });
    return !w->broken;
}

inline size_t dtype_bytes(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclInt32: case ncclUint32: return 4;
    case ncclInt64: case ncclUint64: return 8;
    default: return 0;
    }
}

inline thread_local int t_depth = 0;
inline thread_local std::vector<Op> t_ops;

// A group of sends and receives completes as a unit.  Sends are posted first (so that two ranks sending to each other cannot wait
// for one another), then every receive pulls its bytes behind the sender's event, then the sender's stream is ordered behind the
// receivers' copies: nothing it enqueues later can overwrite a buffer a peer is still reading.
inline ncclResult_t run_group(std::vector<Op> &ops)
{
    for (Op &o : ops) {
        if (o.kind != 0) continue;
        Comm *c = o.c;
        PDL_HIP(c, hipEventRecord(c->sent[(size_t)o.peer], o.st));
        o.rec = new SendRec{o.sbuf, o.bytes, c->device, c->sent[(size_t)o.peer], nullptr, false};
        { std::lock_guard<std::mutex> lk(c->w->mu); c->w->box[{c->rank, o.peer}].push_back(o.rec); }
        c->w->cv.notify_all();
    }
    ncclResult_t rc = ncclSuccess;
    for (Op &o : ops) {
        if (o.kind != 1 || rc != ncclSuccess) continue;
        Comm *c = o.c;
        World *w = c->w;
        SendRec *r = nullptr;
        {
            std::unique_lock<std::mutex> lk(w->mu);
            auto &q = w->box[{o.peer, c->rank}];
            w->cv.wait(lk, [&] { return !q.empty() || w->broken; });
            if (q.empty()) { rc = ncclSystemError; break; }
            r = q.front(); q.pop_front();
        }
        hipError_t e = hipSuccess;
        if (r->bytes != o.bytes) rc = ncclInvalidArgument;           // (RCCL would hang or corrupt: say so instead)
        else {
            e = hipStreamWaitEvent(o.st, r->ready, 0);
            if (e == hipSuccess && o.bytes) e = hipMemcpyPeerAsync(o.rbuf, c->device, r->src, r->device, o.bytes, o.st);
            if (e == hipSuccess) e = hipEventRecord(c->taken[(size_t)o.peer], o.st);
            if (e != hipSuccess) { (void)hipGetLastError(); rc = ncclUnhandledCudaError; }
        }
        { std::lock_guard<std::mutex> lk(w->mu); r->done = rc == ncclSuccess ? c->taken[(size_t)o.peer] : nullptr; r->matched = true; if (rc != ncclSuccess) w->broken = true; }
        w->cv.notify_all();
    }
    for (Op &o : ops) {
        if (o.kind != 0) continue;
        SendRec *r = o.rec;
        World *w = o.c->w;
        bool matched;
        { std::unique_lock<std::mutex> lk(w->mu); w->cv.wait(lk, [&] { return r->matched || w->broken; }); matched = r->matched; }
        if (!matched) { if (rc == ncclSuccess) rc = ncclSystemError; continue; }      // (the record stays in the box of a broken world)
        if (r->done && rc == ncclSuccess && hipStreamWaitEvent(o.st, r->done, 0) != hipSuccess) { (void)hipGetLastError(); rc = ncclUnhandledCudaError; }
        delete r;
    }
    if (rc != ncclSuccess && !ops.empty()) break_world(ops[0].c->w);
    return rc;
}

inline ncclResult_t GroupStart() { ++t_depth; return ncclSuccess; }
inline ncclResult_t GroupEnd()
{
    if (t_depth <= 0) return ncclInvalidUsage;
    if (--t_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(t_ops);
    return run_group(ops);
}
inline ncclResult_t p2p(int kind, const void *sbuf, void *rbuf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t st)
{
    Comm *c = (Comm *)comm;
    const size_t eb = dtype_bytes(dt);
    if (!c || !eb || peer < 0 || peer >= c->w->n || peer == c->rank) return ncclInvalidArgument;
    t_ops.push_back(Op{kind, sbuf, rbuf, count * eb, peer, c, st, nullptr});
    if (t_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(t_ops);
    return run_group(ops);
}
inline ncclResult_t Send(const void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t st) { return p2p(0, buf, nullptr, count, dt, peer, comm, st); }
inline ncclResult_t Recv(void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t st) { return p2p(1, nullptr, buf, count, dt, peer, comm, st); }

// publish my buffer, and once everybody has: my stream waits for every peer's contribution to be final
inline ncclResult_t publish(Comm *c, const void *buf, hipStream_t st)
{
    World *w = c->w;
    PDL_HIP(c, hipEventRecord(c->ready, st));
    { std::lock_guard<std::mutex> lk(w->mu); w->pub[(size_t)c->rank] = buf; }
    if (!barrier(w)) return ncclSystemError;
    return ncclSuccess;
}
// my reads are queued; once everybody's are, my stream may overwrite what I published
inline ncclResult_t retire(Comm *c, hipStream_t st)
{
    World *w = c->w;
    PDL_HIP(c, hipEventRecord(c->pulled, st));
    if (!barrier(w)) return ncclSystemError;
    for (int p = 0; p < w->n; ++p) if (p != c->rank) PDL_HIP(c, hipStreamWaitEvent(st, w->member[(size_t)p]->pulled, 0));
    return ncclSuccess;
}

// int32 sums only (tile sums, exception counts, the one-writer window arrays): every rank pulls every peer's words and adds them up
inline ncclResult_t AllReduce(const void *sendbuf, void *recvbuf, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm, hipStream_t st)
{
    Comm *c = (Comm *)comm;
    if (!c || (dt != ncclInt32 && dt != ncclUint32) || op != ncclSum) return ncclInvalidArgument;
    World *w = c->w;
    const size_t bytes = count * 4, slot = (bytes + 255) & ~(size_t)255, peers = (size_t)w->n - 1;
    if (c->stage_cap < peers * slot) {
        PDL_HIP(c, hipStreamSynchronize(st));
        if (c->stage) (void)hipFree(c->stage);
        c->stage = nullptr; c->stage_cap = 0;
        if (peers * slot && hipMalloc(&c->stage, peers * slot + 256) != hipSuccess) { (void)hipGetLastError(); break_world(w); return ncclSystemError; }
        c->stage_cap = peers * slot;
    }
    ncclResult_t rc = publish(c, sendbuf, st);
    if (rc != ncclSuccess) return rc;
    size_t k = 0;
    for (int p = 0; p < w->n; ++p) {
        if (p == c->rank) continue;
        const Comm *pc = w->member[(size_t)p];
        PDL_HIP(c, hipStreamWaitEvent(st, pc->ready, 0));
        if (bytes) PDL_HIP(c, hipMemcpyPeerAsync((char *)c->stage + k * slot, c->device, w->pub[(size_t)p], pc->device, bytes, st));
        ++k;
    }
    rc = retire(c, st);
    if (rc != ncclSuccess) return rc;
    if (recvbuf != sendbuf && bytes) PDL_HIP(c, hipMemcpyAsync(recvbuf, sendbuf, bytes, hipMemcpyDeviceToDevice, st));
    for (k = 0; k < peers && count; ++k) add_words(st, (int *)recvbuf, (const int *)((const char *)c->stage + k * slot), count);
    PDL_HIP(c, hipGetLastError());
    return ncclSuccess;
}

inline ncclResult_t AllGather(const void *sendbuf, void *recvbuf, size_t sendcount, ncclDataType_t dt, ncclComm_t comm, hipStream_t st)
{
    Comm *c = (Comm *)comm;
    const size_t eb = dtype_bytes(dt);
    if (!c || !eb) return ncclInvalidArgument;
    World *w = c->w;
    const size_t bytes = sendcount * eb;
    ncclResult_t rc = publish(c, sendbuf, st);
    if (rc != ncclSuccess) return rc;
    for (int p = 0; p < w->n; ++p) {
        char *dst = (char *)recvbuf + (size_t)p * bytes;
        if (p == c->rank) { if (dst != sendbuf && bytes) PDL_HIP(c, hipMemcpyAsync(dst, sendbuf, bytes, hipMemcpyDeviceToDevice, st)); continue; }
        const Comm *pc = w->member[(size_t)p];
        PDL_HIP(c, hipStreamWaitEvent(st, pc->ready, 0));
        if (bytes) PDL_HIP(c, hipMemcpyPeerAsync(dst, c->device, w->pub[(size_t)p], pc->device, bytes, st));
    }
    return retire(c, st);
}

inline ncclResult_t CommDestroy(ncclComm_t comm)
{
    Comm *c = (Comm *)comm;
    if (!c) return ncclSuccess;
    World *w = c->w;
    (void)hipSetDevice(c->device);
    for (hipEvent_t e : {c->ready, c->pulled}) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->sent) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->taken) if (e) (void)hipEventDestroy(e);
    if (c->stage) (void)hipFree(c->stage);
    bool last;
    { std::lock_guard<std::mutex> lk(w->mu); w->member[(size_t)c->rank] = nullptr; last = --w->alive == 0; }
    delete c;
    if (last) {
        for (auto &kv : w->box) for (SendRec *r : kv.second) delete r;
        delete w;
    }
    return ncclSuccess;
}

// One rank per entry of `devs` (entries may repeat: contexts sharing a GPU).  Peer access is switched on between every pair of
// distinct devices; a pair without it is an error (the caller falls back to RCCL, or adds the contexts into one GPU).
inline ncclResult_t CommInitAll(ncclComm_t *comms, int n, const int *devs)
{
    if (!comms || n < 1 || !devs) return ncclInvalidArgument;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            if (devs[i] == devs[j]) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devs[i], devs[j]) != hipSuccess || !can) { (void)hipGetLastError(); return ncclSystemError; }
        }
    World *w = new World;
    w->n = w->alive = n;
    w->member.assign((size_t)n, nullptr);
    w->pub.assign((size_t)n, nullptr);
    bool ok = true;
    for (int i = 0; i < n; ++i) {
        Comm *c = new Comm;
        c->w = w; c->rank = i; c->device = devs[i];
        w->member[(size_t)i] = c;
        comms[i] = (ncclComm_t)c;
        ok = ok && hipSetDevice(devs[i]) == hipSuccess;
        for (int j = 0; j < n && ok; ++j) {
            if (devs[j] == devs[i]) continue;
            const hipError_t e = hipDeviceEnablePeerAccess(devs[j], 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) ok = false;
            (void)hipGetLastError();
        }
        ok = ok && hipEventCreateWithFlags(&c->ready, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&c->pulled, hipEventDisableTiming) == hipSuccess;
        c->sent.assign((size_t)n, nullptr); c->taken.assign((size_t)n, nullptr);
        for (int j = 0; j < n && ok; ++j)
            ok = hipEventCreateWithFlags(&c->sent[(size_t)j], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&c->taken[(size_t)j], hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) {
        (void)hipGetLastError();
        for (int i = 0; i < n; ++i) { CommDestroy(comms[i]); comms[i] = nullptr; }
        return ncclUnhandledCudaError;
    }
    return ncclSuccess;
}

inline const char *GetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "a HIP call of the in-process transport failed";
    case ncclSystemError: return "in-process transport: no peer access between two of the GPUs, no memory, or another rank failed";
    case ncclInvalidArgument: return "in-process transport: invalid argument (mismatched message sizes, data type or peer)";
    case ncclInvalidUsage: return "in-process transport: invalid usage";
    default: return "in-process transport: error";
    }
}
#undef PDL_HIP

} // namespace pdlocal
#endif
