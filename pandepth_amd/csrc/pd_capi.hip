// pd_capi.hip — implementation of include/pandepth_amd.h on top of pd_kernels.hip.
//
// One context = one GPU, one compute stream, one copy stream, one int32 allocation holding all
// contig difference arrays followed by the tile sums.  Host batches go through a small pool of
// pinned staging slots (async H2D on the copy stream, scatter on the compute stream), so
// reader threads overlap decode, PCIe and the scatter kernel.  No CPU fallback exists here:
// without a gfx950 device pd_create fails.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <iostream>
#include <fcntl.h>
#include <unistd.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <mutex>
#include <thread>
#include <sys/mman.h>
#include <string>
#include <vector>
#include <map>
#include "pd_kernels.h"
#include "pd_local_comm.h"
#include "../../include/pandepth_amd_dev.h"
#include "pd_bamwalk.h"
#include <condition_variable>
#include <algorithm>
#include <atomic>
#include <deque>
#include <chrono>

using namespace pdk;

// ---- guarded device allocations (PANDEPTH_GUARD=1; pd_guard_check in include/pandepth_amd_dev.h) ---------------------------
// Every device buffer this file allocates goes through the two functions below.  Normally they ARE hipMalloc / hipFree.  With
// PANDEPTH_GUARD set, a buffer of n bytes is allocated as [256 B canary | n bytes | 256 B canary] (the second canary starting at
// byte n exactly, not at a rounded size), the canaries are filled with a pattern, and they are compared — after a device
// synchronize — whenever the buffer is freed and whenever pd_guard_check runs (pd_reset, pd_destroy, pd_comm_destroy call it):
// a kernel that writes in front of or behind its buffer is named by the line that allocated the buffer, instead of landing in
// the allocator's padding unseen.
namespace pdguard {
constexpr size_t G = 256;
constexpr unsigned char PAT = 0xC5;
struct Rec { size_t bytes; int line; };
std::mutex mu;
std::map<void *, Rec> live;
std::atomic<uint64_t> n_bad{0};
std::string last_msg;
bool on() { static const bool v = [] { const char *e = getenv("PANDEPTH_GUARD"); return e && *e && strcmp(e, "0") != 0; }(); return v; }

// caller holds mu; the device is idle
uint64_t check_one(void *user, const Rec &r)
{
    unsigned char h[2 * G];
    uint8_t *raw = (uint8_t *)user - G;
    if (hipMemcpy(h, raw, G, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(h + G, (uint8_t *)user + r.bytes, G, hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    int front = 0, back = 0, first_back = -1, last_back = -1, first_front = -1;
    for (size_t i = 0; i < G; ++i) if (h[i] != PAT) { ++front; if (first_front < 0) first_front = (int)i; }
    for (size_t i = 0; i < G; ++i) if (h[G + i] != PAT) { ++back; if (first_back < 0) first_back = (int)i; last_back = (int)i; }
    if (!front && !back) return 0;
    char m[320];
    snprintf(m, sizeof m, "[guard] device buffer of %zu bytes allocated at pd_capi.hip:%d was written out of bounds: %d byte(s) in front (first at -%d), "
             "%d byte(s) behind (offsets +%d .. +%d past the end)", r.bytes, r.line, front, first_front < 0 ? 0 : (int)G - first_front, back, first_back, last_back);
    fprintf(stderr, "%s\n", m);
    last_msg = m;
    // repair the canaries so that one overrun is reported once
    (void)hipMemset(raw, PAT, G); (void)hipMemset((uint8_t *)user + r.bytes, PAT, G);
    return 1;
}

hipError_t gmalloc(void **out, size_t bytes, int line)
{
    if (!on()) return hipMalloc(out, bytes);
    uint8_t *raw = nullptr;
    const hipError_t e = hipMalloc((void **)&raw, bytes + 2 * G);
    if (e != hipSuccess) return e;
    (void)hipMemset(raw, PAT, G);
    (void)hipMemset(raw + G + bytes, PAT, G);
    (void)hipDeviceSynchronize();
    *out = raw + G;
    std::lock_guard<std::mutex> g(mu);
    live[raw + G] = Rec{bytes, line};
    return hipSuccess;
}

hipError_t gfree(void *user)
{
    if (!on() || !user) return hipFree(user);
    std::lock_guard<std::mutex> g(mu);
    auto it = live.find(user);
    if (it == live.end()) return hipFree(user);          // not one of ours (cannot happen; stay safe)
    (void)hipDeviceSynchronize();
    n_bad += check_one(user, it->second);
    live.erase(it);
    return hipFree((uint8_t *)user - G);
}

// a sub-buffer of a larger allocation (pd_create packs the context's small buffers into one): the caller has left G bytes in front of
// and behind it; they become canaries, checked like everybody else's until drop()
void adopt(void *user, size_t bytes, int line)
{
    if (!on()) return;
    (void)hipMemset((uint8_t *)user - G, PAT, G);
    (void)hipMemset((uint8_t *)user + bytes, PAT, G);
    (void)hipDeviceSynchronize();
    std::lock_guard<std::mutex> g(mu);
    live[user] = Rec{bytes, line};
}
void drop(void *user)
{
    if (!on() || !user) return;
    std::lock_guard<std::mutex> g(mu);
    auto it = live.find(user);
    if (it == live.end()) return;
    (void)hipDeviceSynchronize();
    n_bad += check_one(user, it->second);
    live.erase(it);
}

uint64_t check_all()
{
    if (!on()) return 0;
    std::lock_guard<std::mutex> g(mu);
    int dev = 0; (void)hipGetDevice(&dev);
    (void)hipDeviceSynchronize();
    for (auto &kv : live) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, kv.first) == hipSuccess && at.device != dev) { (void)hipSetDevice(at.device); (void)hipDeviceSynchronize(); }
        n_bad += check_one(kv.first, kv.second);
    }
    (void)hipSetDevice(dev);
    return n_bad.load();
}
} // namespace pdguard

template <class T> static inline hipError_t pd_dmalloc(T **out, size_t bytes, int line) { return pdguard::gmalloc((void **)out, bytes, line); }
#define hipMalloc(p, n) pd_dmalloc((p), (n), __LINE__)
#define hipFree(p) pdguard::gfree((void *)(p))

extern "C" int pd_guard_check(char *msg, size_t cap)
{
    const uint64_t n = pdguard::check_all();
    if (msg && cap) { std::lock_guard<std::mutex> g(pdguard::mu); snprintf(msg, cap, "%s", pdguard::last_msg.c_str()); }
    return n > 0x7fffffff ? 0x7fffffff : (int)n;
}

// proves that the guard sees what it is there to see: a guarded buffer is allocated, ONE byte is written just behind it (and, second
// round, just in front of it), and the check must report exactly that.  Returns 0 when both are found, 1 when the guard is off, -1 when the
// guard is on and misses a write.  The findings it provokes are not counted (pd_guard_check's count is unchanged).
extern "C" int pd_guard_selftest(void)
{
    if (!pdguard::on()) return 1;
    int found = 0;
    for (int side = 0; side < 2; ++side) {
        uint8_t *p = nullptr;
        if (hipMalloc(&p, 1000) != hipSuccess) return -1;
        (void)hipMemset(side ? p - 1 : p + 1000, 0, 1);
        (void)hipDeviceSynchronize();
        const uint64_t before = pdguard::n_bad.load();
        (void)hipFree(p);
        if (pdguard::n_bad.load() == before + 1) ++found;
        pdguard::n_bad.store(before);
    }
    { std::lock_guard<std::mutex> g(pdguard::mu); pdguard::last_msg.clear(); }
    return found == 2 ? 0 : -1;
}

namespace {

constexpr int N_STAGE = 1024;                       // upper bound; slots are created on demand
constexpr size_t STAGE_CAP = (size_t)1 << 18;        // runs per staging slot (3 MiB pinned + 3 MiB HBM)
constexpr size_t DEV_BATCH_MAX = 0xFFFFFF00ull;       // runs per sorted batch (32-bit run indices)
constexpr uint64_t OVF_MAX = (uint64_t)64 << 20;     // overflow-list entries (ends of runs longer than lmax) per tile pass
constexpr uint32_t LMAX_DEFAULT = 512;               // look-back bound for owner tiles (cells)
constexpr uint32_t SAMPLE_DEFAULT = 64;              // sparse index stride (runs)

std::string g_create_err;                   // why the last pd_create failed (contexts may be created on several threads at once:
std::mutex g_create_err_mu;                 // written and read under this lock, handed out as a copy of the calling thread's own)

struct Stage {
    pd_iv *host = nullptr, *dev = nullptr;
    hipEvent_t copied = nullptr, done = nullptr;
    int state = 0;                                   // 0 free, 1 held by caller, 2 in flight
    uint64_t seq = 0;
};

struct ProfRec { std::string name; hipEvent_t a, b; };

struct Pending { const pd_iv *iv; uint32_t n; uint32_t disorder; int slot; pd_runs *cr = nullptr; };   // slot: staging slot or -1; cr: a compact sample (iv NULL until expanded)

} // namespace

// a whole sample in the compact form (include/pandepth_amd.h: pd_runs_create; layout: C8Sample in pd_kernels.h)
struct pd_runs {
    pd_ctx *ctx = nullptr;
    Run8 *r8 = nullptr;                                          // [sorted stream: n_s runs, file order | ... | other runs by bucket at o_base]
    uint32_t n_s = 0, n_o = 0, o_base = 0, n = 0;                // n = n_s + n_o
    uint32_t *b1 = nullptr, *o1 = nullptr;                       // (n_tiles << bshift) + 1 bucket starts per stream (one allocation: b1 | o1)
    uint32_t bshift = 4;                                         // 16 buckets of 512 cells per tile
    uint32_t n_long = 0;                                         // runs longer than a bucket: the direct kernels cannot use the sample
    pd_iv *iv12 = nullptr;                                       // the expanded copy, made on first need
    bool own_r8 = true;                                          // r8 is this object's allocation (false: it lives in the decode session's arena)
    C8Sample view() const { return C8Sample{r8, b1, o1, o_base, bshift}; }
};

static inline size_t slice_flag_bytes(uint64_t n_tiles) { return (size_t)((n_tiles + 16 + 15) / 16 * 16); }

struct pd_ctx {
    int device = 0;
    hipStream_t stream = nullptr, copy_stream = nullptr;
    int32_t n_contigs = 0;
    std::vector<uint32_t> len;
    std::vector<uint64_t> off;                       // first cell of each slot
    uint64_t n_cells = 0, n_tiles = 0, n_words = 0;
    int *buf = nullptr;                              // [n_cells diff | n_tiles sums | pad]
    uint8_t *slab = nullptr;                         // ONE allocation behind the seventeen small buffers below (carry .. chk)
    int *sums = nullptr, *carry = nullptr, *bsum = nullptr;
    uint64_t *d_off = nullptr; uint32_t *d_len = nullptr; uint32_t *d_tile_contig = nullptr;
    uint32_t *ub_a[PD_MAXPEND] = {}, *cand_lo[PD_MAXPEND] = {};   // per pending batch, indexed by 4096-cell tile
    BatchDesc *desc = nullptr; CheckWords *chk = nullptr;         // desc: PD_MAXPEND entries
    uint8_t *hstate = nullptr; uint32_t n_half = 0;               // "written since reset" per 4096 cells
    uint8_t *slice_flags = nullptr;                               // pd_slice_sweep_i4: tiles that own exceptions
    bool accumulate_packed = true;                                // pd_accumulate_from: 4-bit transport
    bool direct_windows = false;                                  // pd_keep_deferred: a whole deferred sample stays deferred, the direct kernels may read it
    bool pristine = true;                                         // nothing materialised in the arrays since the last reset
    bool sums_stale = false;                                      // the tile sums hold what a direct export wrote while the sample is still deferred
    uint32_t *direct_words = nullptr;                             // [n_long, fail, heavy_count, pad | heavy tile list]
    bool dec_crc = true;                                          // the decoder checks every member's CRC-32 ("decode_crc")
    unsigned lz_group = 16;                                       // chunks per workgroup of the LDS parse ("lz_group", up to 16; 0: every chunk parses with its text in memory).  Round 5's default: sixteen
                                                                  // chunks of 8 + 2 KiB share a CU's LDS (158 KB: 32 KiB of history + their text), 16 waves per CU — 8.6 ms against 11.8 for the 60 MB call of
                                                                  // profiles/r04_lz_parse_ab.txt, and the text is fetched once instead of ~1 000 times; chunks of 16 KiB fit seven to a CU and lose
    unsigned dec_waves = 20;                                      // one-wave inflate workgroups per CU and launch ("inflate_waves")
    std::atomic<uint32_t> dec_oth_div{41};                        // inflated bytes per slot for a later run in a batch's arrays: 41 (a kept record's minimum size) until a batch
                                                                  // does not fit (long reads: a later run per 8 bytes of CIGAR), then 8 for the batches that follow
    int dec_sync_event = 1;                                       // "decode_sync_event": pd_decode_collect waits for the batch's last event (0: for its stream, as until round 6)
    int dec_h2d_fifo = 1;                                         // "decode_h2d_fifo": the batches' compressed bytes go up ONE after the other on a copy stream of their own (see pd_decode_queue)
    std::mutex dec_copy_mu; int dec_h2d_lanes = 1; uint64_t dec_copy_seq = 0; hipStream_t dec_copy_st2 = nullptr;   // "decode_h2d_lanes": 2 = the batches' copies alternate between the main stream and a second one (two on the link at a time)
    int dec_h2d_kernel = 0;                                       // "decode_h2d_kernel": a batch's compressed bytes fetched from the pinned buffer by a copy KERNEL on the batch's stream instead of the copy engine
    bool dec_fast = true;                                         // the record chain of a batch is confirmed on the device where the session allows it ("decode_fast")
    uint32_t dec_spoil = 0;                                       // test hook: every k-th segment's guess is spoilt after pass 1 ("decode_spoil")
    uint32_t dec_max_redo = 256;                                  // ... with at most this many segments walking again per batch ("decode_max_redo")
    std::atomic<uint64_t> dec_n_fast{0}, dec_n_slow{0}, dec_n_redo{0};   // batches finished without / with the host's chain check; segments the device walked again
    uint32_t direct_sample = 256;                                 // index stride of the direct path (runs)
    int direct_un = 0;                                           // 0 = the default form of the wide direct kernel (launch_direct_tiles)
    bool all_valid_host = false;
    std::vector<Pending> pend;
    // ---- device decode (pd_decode_*): a few batch slots, each with its own stream and buffers ----
    struct DecSlot {
        bool busy = false;
        bool warming = false;                                     // the session's warm-up thread is still making this slot's buffer and stream (dec_mu)
        hipStream_t st = nullptr;
        hipEvent_t ev[6] = {};
        hipEvent_t ev_done = nullptr;                             // recorded behind everything pd_decode_queue puts on the stream: what pd_decode_collect waits for
        uint8_t *h_blob = nullptr; size_t h_cap = 0;              // page-locked (pin_alloc)
        bool h_mapped = false;                                    // ... as huge pages of its own registered with the runtime (freed by pin_free)
        uint8_t *h_small = nullptr; size_t h_small_cap = 0;       // pinned: the batch's small tables on their way to and from the device
        void *d[10] = {}; size_t cap[10] = {};                    // blob, inflated, tables (members | segments | member counter), status, -, lanes, redo list,
                                                                  // ChainOut + per-segment keys (compact emission), and the runs of a batch whose chain the device
                                                                  // confirms itself: first runs (8 B), later runs (12 B) — copied to exact arrays when the batch is collected
        void *d_tok = nullptr; unsigned tok_wg = 0;               // wave scratch (match tokens) and the number of workgroups it was sized for
        // a batch between pd_decode_queue and pd_decode_collect (pd_decode_submit: the two back to back)
        struct Job {
            bool open = false, queued = false, c8 = false, fast = false, owes_count = false, timed = false;
            bool collecting = false;                               // some thread is inside dec_collect on this slot (set and tested under dec_mu): one ticket, one collect
            uint64_t order = 0; size_t n_bytes = 0; uint64_t inflated = 0; uint32_t n_seg = 0;
            std::vector<pd_bgzf_block> blocks; std::vector<pd_decode_unit> units; std::vector<pdb2::Seg> segs; std::vector<uint32_t> seg0;
            size_t o_blk = 0, o_seg = 0, o_next = 0, o_up = 0, o_bst = 0, o_co = 0, o_so = 0, o_ord = 0;
            pdb2::Cfg cfg{};
            uint8_t *d_tab = nullptr;                              // the batch's tables on the device: the slot's table buffer, or behind the members in the blob buffer (one copy)
            uint64_t cap_first = 0, cap_other = 0, t_mark = 0, t_q0 = 0, t_q1 = 0;      // (t_q0 / t_q1: PANDEPTH_DEVTRACE)
        } job;
        uint32_t gen = 0;
    };
    struct RunSeg { uint64_t order; pd_iv *first; uint64_t n_first; pd_iv *other; uint64_t n_other; pd_iv *far; uint64_t n_far; uint32_t max_span; uint32_t unsorted; uint64_t first_key, last_key; uint64_t n_long = 0; };
    static constexpr int N_DEC = 12;
    uint8_t *arena = nullptr; size_t arena_cap = 0; std::atomic<size_t> arena_used{0};   // the batches' run arrays (bump allocated)
    DecSlot dec[N_DEC];
    std::mutex dec_mu; std::condition_variable dec_cv;
    bool dec_open = false;
    bool dec_warm_on = false;                                     // "decode_warm"
    std::thread dec_warm;                                         // pd_decode_begin's helper: the first slots' page-locked buffers, streams and hardware queues, one after the other, beside the caller
    void *dec_warm_word = nullptr;
    uint32_t dec_near_span = 0xFFFFFFFFu;                         // "decode_near_span": split the later runs into two streams (off)
    pd_decode_cfg dec_cfg{}; uint8_t *d_contig_on = nullptr; uint32_t *d_span_off = nullptr; int32_t *d_spans = nullptr;
    std::vector<RunSeg> run_segs;
    pd_iv *run_first = nullptr, *run_other = nullptr, *run_far = nullptr;   // the concatenated sample (owned until the next reset)
    pd_runs *dec_runs = nullptr;                                  // ... or the whole of it as a compact sample (PD_DECODE_COMPACT)
    // A decode session that emits the compact form directly (PD_DECODE_COMPACT + pd_decode_cfg::n_batches): every batch's pass 2 writes its
    // first runs as 8-byte compact runs (and marks the buckets' first runs, keyed by (batch, index in the batch)); as soon as every
    // earlier batch has been counted, a batch's runs are copied — on a stream of their own, behind the decode — to their FINAL places in
    // the sample's sorted stream, and its later runs behind those of the batches before it.  No feeder ever waits for another one, and at
    // the end nothing is concatenated or converted: only the marks become indices and the later runs are sorted by bucket.
    struct C8Dec {
        bool on = false;
        uint8_t *base = nullptr; size_t bytes = 0;               // ONE allocation: [Run8 x (cap_s + cap_o) | pd_iv x cap_o]
        size_t cap_s = 0, cap_o = 0;
        uint32_t *b1 = nullptr; size_t nbw = 0;                  // bucket starts: b1 | o1, nbw words each
        unsigned long long *marks = nullptr;                     // per bucket: min (batch << 32 | index in the batch) of a run that begins there
        uint32_t bshift = 4;
        uint64_t n_s = 0, n_o = 0, turn = 0, n_batches = 0;
        struct Batch { bool counted = false; uint64_t nf = 0, no = 0; Run8 *seg_s = nullptr; pd_iv *seg_o = nullptr; hipEvent_t ev = nullptr; };
        std::vector<Batch> batch;                                // by order
        std::vector<uint32_t> base_s;                            // first place of every batch's first runs in the sorted stream
        hipStream_t compose = nullptr;
        std::string err;                                         // what went wrong while runs were being placed (reported by pd_decode_end)
        std::mutex mu;
        Run8 *r8() const { return (Run8 *)base; }
        pd_iv *oth() const { return (pd_iv *)(base + (cap_s + cap_o) * sizeof(Run8)); }
    } c8;
    uint64_t *ovf = nullptr; uint32_t ovf_cap = 0;    // ends of runs longer than lmax (grown on demand)
    std::vector<Stage> stage;                        // grows on demand, up to N_STAGE
    uint64_t seq = 0;
    void *scratch = nullptr; size_t scratch_bytes = 0;
    int state = 0;                                   // 0 accumulating (diff), 1 depth
    uint32_t lmax = LMAX_DEFAULT, sample = SAMPLE_DEFAULT;
    unsigned grid_tiles = 0;                         // 0 = sized per pass from the number of runs
    int stile = 8192; int n_cu = 256;
    // pd_deflate_parse's work buffers (device memory, grown on demand, kept until pd_destroy): two slots, each with its stream, so that
    // two calls overlap (one's copies under the other's kernels)
    struct LzWork {
        static constexpr int N = 16;
        void *p[N] = {}; size_t cap[N] = {};
        bool fit(int k, size_t bytes)
        {
            if (bytes <= cap[k]) return true;
            if (p[k]) { (void)hipFree(p[k]); p[k] = nullptr; cap[k] = 0; }
            const size_t want = bytes + bytes / 8 + 4096;
            if (hipMalloc(&p[k], want) != hipSuccess) return false;
            cap[k] = want;
            return true;
        }
        void release() { for (int k = 0; k < N; ++k) { if (p[k]) (void)hipFree(p[k]); p[k] = nullptr; cap[k] = 0; } if (st) { (void)hipStreamDestroy(st); st = nullptr; }
                         if (h_stage) { (void)hipHostFree(h_stage); h_stage = nullptr; } for (auto &e : ev_stage) if (e) { (void)hipEventDestroy(e); e = nullptr; } }
        hipStream_t st = nullptr;
        void *h_stage = nullptr; hipEvent_t ev_stage[2] = {nullptr, nullptr};     // page-locked staging of the symbols' way back
        std::mutex mu;
    } lz[4];                                                      // (a round's provider calls in flight at once: two until round 6, up to four)
    std::atomic<unsigned> lz_turn{0}; unsigned lz_slots = 2;         // "lz_slots": 2 or 4 of lz[] in use
    bool lz_mix = false;                                          // "lz_mix": see lz_run
    // the statistics of the last window call stay on the device (pd_text_append_window_rows formats the table's rows from them)
    unsigned char *wk = nullptr; size_t wk_bytes = 0; uint32_t wk_w = 0; uint64_t wk_nw = 0; bool wk_valid = false; std::vector<uint64_t> wk_woff;
    bool prof = false;
    std::vector<ProfRec> prof_pending;
    std::vector<hipEvent_t> ev_pool;
    std::map<std::string, std::pair<double, uint64_t>> prof_acc;
    std::mutex mu;
    std::string err;
};

namespace {

int fail(pd_ctx *c, int code, const std::string &msg)
{
    if (c) c->err = msg; else { std::lock_guard<std::mutex> lk(g_create_err_mu); g_create_err = msg; }
    return code;
}

// every entry point that needs the arrays in a given state (0 accumulating, 1 depth)
int need_state(pd_ctx *c, int want, const char *fn)
{
    if (c->state == want) return PD_OK;
    const char *why = c->state == 1 ? "depth already materialised (call pd_reset, or use the pd_reduce_* calls)"
                                    : "call pd_scan first";
    return fail(c, PD_ESTATE, std::string(fn) + ": " + why);
}

#define HIPOK(ctx, call)                                                                         \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(ctx, PD_EHIP, std::string(#call) + ": " + hipGetErrorString(e_));        \
    } while (0)

ContigTab tab_of(pd_ctx *c) { return ContigTab{c->d_off, c->d_len, c->n_contigs}; }

hipEvent_t get_event(pd_ctx *c)
{
    if (!c->ev_pool.empty()) { hipEvent_t e = c->ev_pool.back(); c->ev_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

struct ProfScope {
    pd_ctx *c; ProfRec r; bool on;
    ProfScope(pd_ctx *ctx, const char *name) : c(ctx), on(ctx->prof)
    {
        if (on) { r.name = name; r.a = get_event(c); r.b = get_event(c); (void)hipEventRecord(r.a, c->stream); }
    }
    ~ProfScope()
    {
        if (on) { (void)hipEventRecord(r.b, c->stream); c->prof_pending.push_back(r); }
    }
};

int prof_collect(pd_ctx *c)
{
    if (c->prof_pending.empty()) return PD_OK;
    HIPOK(c, hipStreamSynchronize(c->stream));
    for (auto &r : c->prof_pending) {
        float ms = 0.f;
        HIPOK(c, hipEventElapsedTime(&ms, r.a, r.b));
        auto &acc = c->prof_acc[r.name];
        acc.first += ms; acc.second += 1;
        c->ev_pool.push_back(r.a); c->ev_pool.push_back(r.b);
    }
    c->prof_pending.clear();
    return PD_OK;
}

int ensure_scratch(pd_ctx *c, size_t bytes)
{
    if (bytes <= c->scratch_bytes) return PD_OK;
    if (c->scratch) { HIPOK(c, hipStreamSynchronize(c->stream)); HIPOK(c, hipFree(c->scratch)); c->scratch = nullptr; c->scratch_bytes = 0; }
    size_t want = bytes + bytes / 4 + 4096;
    if (hipMalloc(&c->scratch, want) != hipSuccess) return fail(c, PD_ENOMEM, "scratch allocation failed");
    c->scratch_bytes = want;
    return PD_OK;
}

// Reset = forget every cell: mark all half-tiles "not written" (the owner-tile kernel stores into
// them without reading, the sweep reads them as zeros) and zero the tile sums.  No 12 GB fill.
int do_reset(pd_ctx *c)
{
    ProfScope ps(c, "reset");
    HIPOK(c, hipMemsetAsync(c->hstate, 0, c->n_half, c->stream));
    HIPOK(c, hipMemsetAsync(c->sums, 0, (c->n_words - c->n_cells) * 4, c->stream));
    HIPOK(c, hipMemsetAsync(c->chk, 0, sizeof(CheckWords), c->stream));
    HIPOK(c, hipMemsetAsync(c->desc, 0, sizeof(BatchDesc) * PD_MAXPEND, c->stream));
    c->all_valid_host = false;
    c->pristine = true;
    c->sums_stale = false;
    return PD_OK;
}

// every cell readable/atomically addable: zero-fill what has not been written since the reset
int ensure_all_valid(pd_ctx *c)
{
    if (c->all_valid_host) return PD_OK;
    ProfScope ps(c, "fill");
    launch_fill_invalid(c->stream, c->buf, c->hstate, c->n_half, c->chk, false, (unsigned)c->n_cu * 8);
    HIPOK(c, hipGetLastError());
    c->all_valid_host = true;
    return PD_OK;
}

// a compact pending sample that has to take a path that reads 12-byte runs: expanded once (the copy stays with the sample).
// Inside a bucket the runs are in no particular order: the batch is sorted up to one bucket's cells of disorder.
int expand_compact(pd_ctx *c, Pending &p)
{
    if (!p.cr || p.iv) return PD_OK;
    pd_runs *r = p.cr;
    if (!r->iv12) {
        if (hipMalloc(&r->iv12, (size_t)r->n * sizeof(pd_iv)) != hipSuccess) return fail(c, PD_ENOMEM, "compact sample: allocation of the expanded runs failed");
        ProfScope ps(c, "expand_runs");
        launch_c8_expand(c->stream, r->view(), c->d_tile_contig, c->d_off, (uint32_t)c->n_tiles, r->iv12);
        HIPOK(c, hipGetLastError());
    }
    p.iv = r->iv12;
    p.disorder = (uint32_t)PD_TILE >> r->bshift;
    return PD_OK;
}

// one owner-tile pass over all pending sorted batches
int flush_pending(pd_ctx *c)
{
    if (c->pend.empty()) return PD_OK;
    for (auto &p : c->pend) { const int re = expand_compact(c, p); if (re) return re; }
    if (c->sums_stale) {                      // a direct export wrote this (still deferred) sample's tile sums; the scatter adds to them
        HIPOK(c, hipMemsetAsync(c->sums, 0, (c->n_words - c->n_cells) * 4, c->stream));
        c->sums_stale = false;
    }
    c->pristine = false;
    uint64_t total = 0;
    for (auto &p : c->pend) total += p.n;
    if (total > OVF_MAX) total = OVF_MAX;     // more long runs than this in ONE pass is reported (err bit 4):
                                              // such data belongs on the PD_PUSH_DEFAULT path
    if (total > c->ovf_cap) {
        if (c->ovf) { HIPOK(c, hipStreamSynchronize(c->stream)); HIPOK(c, hipFree(c->ovf)); c->ovf = nullptr; c->ovf_cap = 0; }
        const uint64_t cap = total;
        if (hipMalloc(&c->ovf, (size_t)cap * 8) != hipSuccess) return fail(c, PD_ENOMEM, "overflow list allocation failed");
        c->ovf_cap = (uint32_t)cap;
    }
    const uint32_t n_stiles = (uint32_t)(c->n_cells / c->stile);
    PendSet ps{};
    ps.nb = (int)c->pend.size(); ps.lmax = c->lmax;
    for (int b = 0; b < ps.nb; ++b) {
        const Pending &p = c->pend[b];
        ps.b[b] = PendBatch{p.iv, c->ub_a[b], c->cand_lo[b], c->desc + b, p.n, 0};
        ProfScope sc(c, "scatter_index");
        launch_scatter_index(c->stream, p.iv, p.n, tab_of(c), c->lmax, p.disorder, c->sample, c->ub_a[b], c->cand_lo[b],
                             n_stiles, c->stile, c->desc + b);
    }
    // Grid of the tile pass: measured best is ~64 K workgroups for a whole-genome pass (each walks a
    // handful of tiles; the hardware overlaps their load/flush phases); small streaming batches
    // touch few tiles and get a proportionally smaller grid.
    unsigned grid = c->grid_tiles;
    if (!grid) {
        uint64_t all = 0;
        for (auto &p : c->pend) all += p.n;
        uint64_t g = all / 256;
        if (g < (uint64_t)c->n_cu * 4) g = (uint64_t)c->n_cu * 4;
        if (g > 65536) g = 65536;
        grid = (unsigned)g;
    }
    { ProfScope sc(c, "scatter_tiles");
      launch_scatter_tiles(c->stream, ps, tab_of(c), c->d_tile_contig, n_stiles, c->stile, c->buf, c->sums, c->hstate,
                           c->ovf, c->ovf_cap, c->chk, grid); }
    { ProfScope sc(c, "scatter_finish");
      if (!c->all_valid_host) launch_fill_invalid(c->stream, c->buf, c->hstate, c->n_half, c->chk, true, (unsigned)c->n_cu * 8);
      launch_scatter_finish(c->stream, ps, c->buf, c->sums, c->ovf, c->ovf_cap, c->chk); }
    HIPOK(c, hipGetLastError());
    for (auto &p : c->pend)
        if (p.slot >= 0) {
            Stage &st = c->stage[p.slot];
            HIPOK(c, hipEventRecord(st.done, c->stream));
            st.state = 2; st.seq = ++c->seq;
        }
    c->pend.clear();
    return PD_OK;
}

// scatter a device-resident batch on the compute stream; *deferred = true when a staged slot must
// stay in flight until flush_pending records its event
int scatter_device(pd_ctx *c, const pd_iv *d, size_t n, unsigned flags, int slot, bool *deferred)
{
    if (deferred) *deferred = false;
    if (n == 0) return PD_OK;
    if (flags & PD_PUSH_SORTED) {
        const uint64_t dis64 = (uint64_t)(flags >> 8) * 256u;
        const uint32_t disorder = dis64 > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)dis64;
        for (size_t o = 0; o < n; o += DEV_BATCH_MAX) {
            const uint32_t m = (uint32_t)(n - o < DEV_BATCH_MAX ? n - o : DEV_BATCH_MAX);
            const bool last = o + m >= n;
            c->pend.push_back(Pending{d + o, m, disorder, last ? slot : -1});
            if (c->pend.size() == PD_MAXPEND || !(last && (flags & PD_PUSH_MORE))) {
                int rc = flush_pending(c);
                if (rc) return rc;
            }
        }
        if (deferred) *deferred = !c->pend.empty();
    } else {
        int rc = flush_pending(c);
        if (rc) return rc;
        rc = ensure_all_valid(c);
        if (rc) return rc;
        c->pristine = false;
        ProfScope ps(c, "scatter_atomic");
        launch_scatter_atomic(c->stream, d, n, tab_of(c), c->buf, c->sums);
    }
    HIPOK(c, hipGetLastError());
    return PD_OK;
}

int check_words(pd_ctx *c)
{
    CheckWords h;
    HIPOK(c, hipMemcpyAsync(&h, c->chk, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIPOK(c, hipStreamSynchronize(c->stream));
    if (h.unsorted_batches || h.err) {
        char m[256];
        snprintf(m, sizeof m, "%llu batch(es) pushed with PD_PUSH_SORTED were not sorted by (tid,beg), held invalid "
                 "contig ids or too many runs longer than lmax (err bits 0x%x); depth arrays are not valid",
                 (unsigned long long)h.unsorted_batches, h.err);
        return fail(c, PD_EINVAL, m);
    }
    return PD_OK;
}

// caller holds c->mu
int stage_acquire(pd_ctx *c, int *slot)
{
    for (;;) {
        int oldest = -1;
        for (int i = 0; i < (int)c->stage.size(); ++i) {
            Stage &s = c->stage[i];
            if (s.state == 2 && hipEventQuery(s.done) == hipSuccess) s.state = 0;
            if (s.state == 0) { *slot = i; s.state = 1; return PD_OK; }
            if (s.state == 2 && (oldest < 0 || s.seq < c->stage[oldest].seq)) oldest = i;
        }
        // a new slot while the pool may still grow and fewer than 4 batches are queued on the GPU
        int inflight = 0;
        for (auto &s : c->stage) inflight += s.state == 2;
        if ((int)c->stage.size() < N_STAGE && (oldest < 0 || inflight < 4)) {
            Stage s;
            if (hipEventCreateWithFlags(&s.copied, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&s.done, hipEventDisableTiming) != hipSuccess)
                return fail(c, PD_EHIP, "cannot create staging events");
            s.state = 1;
            c->stage.push_back(s);
            *slot = (int)c->stage.size() - 1;
            return PD_OK;
        }
        if (oldest < 0) {
            if (!c->pend.empty()) { int rc = flush_pending(c); if (rc) return rc; continue; }
            return fail(c, PD_ESTATE, "all staging slots are held by callers");
        }
        HIPOK(c, hipEventSynchronize(c->stage[oldest].done));
        c->stage[oldest].state = 0;
    }
}

int stage_submit(pd_ctx *c, int slot, size_t n, unsigned flags)
{
    Stage &s = c->stage[slot];
    if (n == 0) { s.state = 0; return PD_OK; }
    if (!c->copy_stream) HIPOK(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    HIPOK(c, hipMemcpyAsync(s.dev, s.host, n * sizeof(pd_iv), hipMemcpyHostToDevice, c->copy_stream));
    HIPOK(c, hipEventRecord(s.copied, c->copy_stream));
    HIPOK(c, hipStreamWaitEvent(c->stream, s.copied, 0));
    bool deferred = false;
    int rc = scatter_device(c, s.dev, n, flags, slot, &deferred);
    if (rc) return rc;
    Stage &s2 = c->stage[slot];               // (the vector may not have grown, but stay safe)
    if (deferred) { s2.state = 3; return PD_OK; }          // event recorded by flush_pending
    if (s2.state != 2) { HIPOK(c, hipEventRecord(s2.done, c->stream)); s2.state = 2; s2.seq = ++c->seq; }
    return PD_OK;
}

} // namespace

extern "C" {

int pd_abi_version(void) { return PD_ABI_VERSION; }

const char *pd_strerror(const pd_ctx *ctx)
{
    if (ctx) return ctx->err.c_str();
    static thread_local std::string mine;
    { std::lock_guard<std::mutex> lk(g_create_err_mu); mine = g_create_err; }
    return mine.c_str();
}

// A page-locked host buffer for a decode slot.  hipHostMalloc of 34 MB takes 7-8 ms (the runtime allocates and maps it 4 KiB page by page): six of them were
// 46 ms in front of every run's first read.  Anonymous memory the kernel backs with 2 MiB pages (MADV_HUGEPAGE; where transparent huge pages are off it is
// ordinary memory), touched, then registered with the runtime (hipHostRegister) takes 1.5 ms and copies at the same 56 GB/s (tools/ubench/pin_cost.hip,
// profiles/r06_h2d_fill.txt).  Falls back to hipHostMalloc.
static uint8_t *pin_alloc(size_t bytes, bool *mapped)
{
    *mapped = false;
    if (!getenv("PANDEPTH_NO_HUGE_PIN")) {
        const size_t al = (size_t)2 << 20, len = (bytes + al - 1) / al * al;
        void *m = mmap(nullptr, len + al, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (m != MAP_FAILED) {
            uint8_t *q = (uint8_t *)(((uintptr_t)m + al - 1) / al * al);
            // (the unaligned head and tail go back at once: the region is exactly [q, q + len))
            if (q > (uint8_t *)m) munmap(m, (size_t)(q - (uint8_t *)m));
            if ((uint8_t *)m + len + al > q + len) munmap(q + len, (size_t)((uint8_t *)m + len + al - (q + len)));
            (void)madvise(q, len, MADV_HUGEPAGE);
            for (size_t o = 0; o < len; o += 4096) q[o] = 0;              // (fault it in before the runtime walks it)
            if (hipHostRegister(q, len, hipHostRegisterDefault) == hipSuccess) { *mapped = true; return q; }
            (void)hipGetLastError();
            munmap(q, len);
        }
    }
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return (uint8_t *)p;
}
static void pin_free(uint8_t *p, size_t bytes, bool mapped)
{
    if (!p) return;
    if (!mapped) { (void)hipHostFree(p); return; }
    const size_t al = (size_t)2 << 20, len = (bytes + al - 1) / al * al;
    (void)hipHostUnregister(p);
    munmap(p, len);
}

int pd_create(int device, int32_t n_contigs, const uint32_t *contig_len, pd_ctx **out)
{
    if (!out || n_contigs <= 0 || !contig_len) return fail(nullptr, PD_EINVAL, "pd_create: bad arguments");
    *out = nullptr;
    const bool tm_on = getenv("PANDEPTH_TIMING") != nullptr;
    const auto tm_t0 = std::chrono::steady_clock::now();
    auto tm_mark = [&](const char *what) { if (tm_on) fprintf(stderr, "[timing]   pd_create: %-28s at %.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - tm_t0).count()); };
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, PD_ENODEV, "pd_create: no HIP device visible (this engine has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(nullptr, PD_ENODEV, "pd_create: device index out of range");
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, device) != hipSuccess) return fail(nullptr, PD_ENODEV, "pd_create: cannot query device");
    if (!strstr(pr.gcnArchName, "gfx950"))
        return fail(nullptr, PD_ENODEV, std::string("pd_create: device is ") + pr.gcnArchName + ", kernels are built for gfx950 only");
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, PD_ENODEV, "pd_create: hipSetDevice failed");
    tm_mark("runtime up, device chosen");

    pd_ctx *c = new pd_ctx;
    c->device = device;
    c->n_contigs = n_contigs;
    c->len.assign(contig_len, contig_len + n_contigs);
    c->off.resize((size_t)n_contigs + 1);
    uint64_t o = 0;
    for (int32_t i = 0; i < n_contigs; ++i) {
        c->off[i] = o;
        o += ((uint64_t)contig_len[i] + 1 + PD_TILE - 1) / PD_TILE * PD_TILE;   // room for the -1 at cell len
    }
    c->off[n_contigs] = o;
    c->n_cells = o;
    c->n_tiles = o / PD_TILE;
    if (c->n_tiles >= 0xFFFFFFF0ull) { delete c; return fail(nullptr, PD_EINVAL, "pd_create: genome too large"); }
    c->n_words = c->n_cells + (c->n_tiles + 3) / 4 * 4;
    c->n_cu = pr.multiProcessorCount;

#define CREATE_OK(call)                                                                                  \
    do { hipError_t e_ = (call); if (e_ != hipSuccess) {                                                 \
        std::string m_ = std::string("pd_create: ") + #call + ": " + hipGetErrorString(e_);            \
        pd_destroy(c); return fail(nullptr, e_ == hipErrorOutOfMemory ? PD_ENOMEM : PD_EHIP, m_); } } while (0)

    CREATE_OK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    // (the copy stream of the host staging path is made by its first user: a queue costs ~15 ms, and the device-decode path
    // of the executable never stages runs from the host)
    tm_mark("streams");
    CREATE_OK(hipMalloc(&c->buf, c->n_words * 4));
    tm_mark("cell buffer");
    c->sums = c->buf + c->n_cells;
    c->n_half = (uint32_t)(c->n_cells / PD_HALF);
    {
        // the small buffers come out of ONE allocation (seventeen hipMalloc calls were 17 ms of every run's start-up), each on a 256-byte
        // boundary; with PANDEPTH_GUARD they keep a canary in front of and behind them inside the slab, like separate allocations would
        struct Want { void **pp; size_t bytes; int line; };
        std::vector<Want> wants;
#define WANT(field, bytes_) wants.push_back(Want{(void **)&(field), (size_t)(bytes_), __LINE__})
        WANT(c->carry, (c->n_tiles + 4) * 4);
        WANT(c->bsum, (c->n_tiles / 1024 + 4) * 4);
        WANT(c->d_off, ((size_t)n_contigs + 1) * 8);
        WANT(c->d_len, (size_t)n_contigs * 4);
        WANT(c->d_tile_contig, (c->n_tiles + 1) * 4);
        for (int b = 0; b < PD_MAXPEND; ++b) {
            WANT(c->ub_a[b], (c->n_tiles * 2 + 4) * 4);          // indexed by 4096-cell scatter tile
            WANT(c->cand_lo[b], (c->n_tiles * 2 + 4) * 4);
        }
        WANT(c->hstate, c->n_half + 16);
        WANT(c->slice_flags, slice_flag_bytes(c->n_tiles) + 2 * (16 + ((size_t)c->n_tiles + 4) * 4) + 16);   // flags | counter | list of flagged tiles | counter | list of the tiles the packed sweep leaves
        WANT(c->direct_words, 64 + (c->n_tiles + 4) * 4);        // [n_long, fail, heavy_count, diagnostics ... | heavy tile list at +16]
        WANT(c->desc, sizeof(BatchDesc) * PD_MAXPEND);
        WANT(c->chk, sizeof(CheckWords));
#undef WANT
        const size_t gap = pdguard::on() ? pdguard::G : 0;
        std::vector<size_t> at(wants.size());
        size_t total = 0;
        for (size_t k = 0; k < wants.size(); ++k) { total += gap; at[k] = total; total = (total + wants[k].bytes + gap + 255) / 256 * 256; }
        CREATE_OK(hipMalloc(&c->slab, total + 256));
        for (size_t k = 0; k < wants.size(); ++k) { *wants[k].pp = c->slab + at[k]; pdguard::adopt(*wants[k].pp, wants[k].bytes, wants[k].line); }
    }
    {
        std::vector<uint32_t> tc(c->n_tiles + 1, 0);
        for (int32_t i = 0; i < n_contigs; ++i)
            for (uint64_t t = c->off[i] / PD_TILE; t < c->off[i + 1] / PD_TILE; ++t) tc[t] = (uint32_t)i;
        // (on the context's stream, not the null stream: the process's null stream would be one more hardware queue to make — 9 ms — for three small copies)
        CREATE_OK(hipMemcpyAsync(c->d_tile_contig, tc.data(), (c->n_tiles + 1) * 4, hipMemcpyHostToDevice, c->stream));
        CREATE_OK(hipMemcpyAsync(c->d_off, c->off.data(), ((size_t)n_contigs + 1) * 8, hipMemcpyHostToDevice, c->stream));
        CREATE_OK(hipMemcpyAsync(c->d_len, c->len.data(), (size_t)n_contigs * 4, hipMemcpyHostToDevice, c->stream));
        CREATE_OK(hipStreamSynchronize(c->stream));
    }
#undef CREATE_OK
    tm_mark("other buffers + tables");
    int rc = do_reset(c);
    if (rc == PD_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = PD_EHIP;
    tm_mark("first reset (kernels loaded)");
    if (rc != PD_OK) { { std::lock_guard<std::mutex> lk(g_create_err_mu); g_create_err = c->err; } pd_destroy(c); return rc; }
    *out = c;
    return PD_OK;
}

static void runs_free(pd_runs *r);
static int runs_make(pd_ctx *c, const pd_iv *sorted, size_t n_sorted, const pd_iv *const *others, const size_t *n_others, int n_arr, pd_runs **out);

static inline uint64_t dec_now_us() { return (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int pd_destroy(pd_ctx *c)
{
    if (!c) return PD_OK;
    if (c->dec_warm.joinable()) c->dec_warm.join();
    const bool tm = getenv("PANDEPTH_TIMING") != nullptr;
    const uint64_t t0 = dec_now_us(); uint64_t t1 = t0, t2 = t0, t3 = t0;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    for (size_t i = 0; i < c->stage.size(); ++i) {
        if (c->stage[i].host) (void)hipHostFree(c->stage[i].host);
        if (c->stage[i].dev) (void)hipFree(c->stage[i].dev);
        if (c->stage[i].copied) (void)hipEventDestroy(c->stage[i].copied);
        if (c->stage[i].done) (void)hipEventDestroy(c->stage[i].done);
    }
    for (auto &r : c->prof_pending) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    for (void *p : {(void *)c->carry, (void *)c->bsum, (void *)c->d_off, (void *)c->d_len, (void *)c->d_tile_contig, (void *)c->ub_a[0], (void *)c->ub_a[1], (void *)c->ub_a[2],
                    (void *)c->ub_a[3], (void *)c->cand_lo[0], (void *)c->cand_lo[1], (void *)c->cand_lo[2], (void *)c->cand_lo[3], (void *)c->hstate, (void *)c->slice_flags,
                    (void *)c->direct_words, (void *)c->desc, (void *)c->chk}) pdguard::drop(p);            // (parts of the slab)
    void *ptrs[] = {c->buf, c->slab, c->ovf, c->scratch, c->wk};
    t1 = dec_now_us();
    for (void *p : ptrs) if (p) (void)hipFree(p);
    t2 = dec_now_us();
    for (auto &sl : c->dec) {
        if (sl.st) (void)hipStreamSynchronize(sl.st);
        pin_free(sl.h_blob, sl.h_cap, sl.h_mapped);
        if (sl.h_small) (void)hipHostFree(sl.h_small);
        for (void *p : sl.d) if (p) (void)hipFree(p);
        if (sl.d_tok) (void)hipFree(sl.d_tok);
        for (hipEvent_t e : sl.ev) if (e) (void)hipEventDestroy(e);
        if (sl.ev_done) (void)hipEventDestroy(sl.ev_done);
        if (sl.st) (void)hipStreamDestroy(sl.st);
    }
    for (auto &r : c->run_segs) {
        const auto ina = [&](const void *p) { return c->arena && (const uint8_t *)p >= c->arena && (const uint8_t *)p < c->arena + c->arena_cap; };
        if (r.first && !ina(r.first)) (void)hipFree(r.first);
        if (r.other && !ina(r.other)) (void)hipFree(r.other);
        if (r.far && !ina(r.far)) (void)hipFree(r.far);
    }
    for (void *p : {(void *)c->d_contig_on, (void *)c->d_span_off, (void *)c->d_spans, (void *)c->run_first, (void *)c->run_other, (void *)c->run_far, (void *)c->arena}) if (p) (void)hipFree(p);
    runs_free(c->dec_runs);
    for (void *p : {(void *)c->c8.base, (void *)c->c8.b1}) if (p) (void)hipFree(p);
    t3 = dec_now_us();
    for (auto &w : c->lz) { std::lock_guard<std::mutex> g(w.mu); w.release(); }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->dec_copy_st2) (void)hipStreamDestroy(c->dec_copy_st2);
    if (c->dec_warm_word) (void)hipFree(c->dec_warm_word);
    delete c;
    if (tm) fprintf(stderr, "[timing]   pd_destroy: sync + staging %.3f s, cells and tables %.3f s, decode slots and runs %.3f s, parse buffers and streams %.3f s\n",
                    (t1 - t0) * 1e-6, (t2 - t1) * 1e-6, (t3 - t2) * 1e-6, (dec_now_us() - t3) * 1e-6);
    return PD_OK;
}

int pd_reset(pd_ctx *c)
{
    if (!c) return PD_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    c->wk_valid = false;
    HIPOK(c, hipSetDevice(c->device));
    // deferred batches are forgotten with everything else (they used to be scattered first, a whole-genome pass for nothing);
    // their staging slots are free once the stream has passed this point
    for (auto &p : c->pend)
        if (p.slot >= 0) {
            Stage &st = c->stage[p.slot];
            HIPOK(c, hipEventRecord(st.done, c->stream));
            st.state = 2; st.seq = ++c->seq;
        }
    c->pend.clear();
    c->state = 0;
    if (pdguard::on()) (void)pdguard::check_all();
    int rc = do_reset(c);
    if (rc == PD_OK && (c->run_first || c->run_other || c->run_far || c->dec_runs)) {          // the decoded sample's runs go with it
        HIPOK(c, hipStreamSynchronize(c->stream));
        for (pd_iv **q : {&c->run_first, &c->run_other, &c->run_far}) if (*q) { (void)hipFree(*q); *q = nullptr; }
        runs_free(c->dec_runs); c->dec_runs = nullptr;
    }
    return rc;
}

int pd_keep_deferred(pd_ctx *c, int enable)
{
    if (!c) return PD_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    c->direct_windows = enable != 0;
    return PD_OK;
}

int pd_set_param(pd_ctx *c, const char *name, uint64_t value)
{
    if (!c || !name) return PD_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!strcmp(name, "lmax")) { if (value < 1 || value > 4096) return fail(c, PD_EINVAL, "lmax must be in [1, 4096]"); c->lmax = (uint32_t)value; return PD_OK; }
    if (!strcmp(name, "sample")) { if (value < 1 || value > 65536) return fail(c, PD_EINVAL, "sample must be in [1, 65536]"); c->sample = (uint32_t)value; return PD_OK; }
    if (!strcmp(name, "scatter_tile")) {
        if (value != 4096 && value != 8192) return fail(c, PD_EINVAL, "scatter_tile must be 4096 or 8192");
        c->stile = (int)value; return PD_OK;
    }
    if (!strcmp(name, "grid_tiles")) { if (value > (1u << 20)) return fail(c, PD_EINVAL, "grid_tiles out of range"); c->grid_tiles = (unsigned)value; return PD_OK; }
    if (!strcmp(name, "accumulate_packed")) { c->accumulate_packed = value != 0; return PD_OK; }
    if (!strcmp(name, "direct_un")) { c->direct_un = (int)value; return PD_OK; }
    if (!strcmp(name, "direct_sample")) { if (value < 1 || value > 65536) return fail(c, PD_EINVAL, "direct_sample must be in [1, 65536]"); c->direct_sample = (uint32_t)value; return PD_OK; }
    if (!strcmp(name, "decode_crc")) { c->dec_crc = value != 0; return PD_OK; }
    if (!strcmp(name, "sweep_i4_fast")) { pdk::set_sweep_i4_fast(value != 0); return PD_OK; }
    if (!strcmp(name, "lz_mix")) { c->lz_mix = value != 0; return PD_OK; }
    if (!strcmp(name, "lz_group")) { if (value > pdk::LZ_GROUP_MAX) return fail(c, PD_EINVAL, "lz_group must be in [0, 16]"); c->lz_group = (unsigned)value; return PD_OK; }
    if (!strcmp(name, "inflate_waves")) { if (value < 1 || value > 24) return fail(c, PD_EINVAL, "inflate_waves must be in [1, 24]"); c->dec_waves = (unsigned)value; return PD_OK; }
    if (!strcmp(name, "decode_spoil")) { c->dec_spoil = value > 0xFFFFFFFFull ? 0u : (uint32_t)value; return PD_OK; }
    if (!strcmp(name, "decode_fast")) { c->dec_fast = value != 0; return PD_OK; }
    if (!strcmp(name, "decode_h2d_kernel")) { c->dec_h2d_kernel = (int)value; return PD_OK; }
    if (!strcmp(name, "decode_h2d_fifo")) { c->dec_h2d_fifo = value != 0; return PD_OK; }
    if (!strcmp(name, "decode_warm")) { c->dec_warm_on = value != 0; return PD_OK; }
    if (!strcmp(name, "lz_slots")) { c->lz_slots = value >= 4 ? 4 : 2; return PD_OK; }
    if (!strcmp(name, "decode_h2d_lanes")) { c->dec_h2d_lanes = value > 1 ? 2 : 1; return PD_OK; }
    if (!strcmp(name, "decode_sync_event")) { c->dec_sync_event = value != 0; return PD_OK; }
    if (!strcmp(name, "decode_max_redo")) { c->dec_max_redo = value > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)value; return PD_OK; }
    if (!strcmp(name, "decode_near_span")) { c->dec_near_span = value > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)value; return PD_OK; }
    return fail(c, PD_EINVAL, std::string("unknown parameter ") + name);
}

static void runs_free(pd_runs *r)
{
    if (!r) return;
    if (r->own_r8 && r->r8) (void)hipFree(r->r8);
    for (void *q : {(void *)r->b1, (void *)r->iv12}) if (q) (void)hipFree(q);
    delete r;
}

static uint32_t runs_bshift(const pd_ctx *c)
{
    // buckets as wide as the look-back bound ("lmax", a power of two between 256 and 8192 cells; default 512)
    uint32_t cells = 256; while (cells < c->lmax && cells < (uint32_t)PD_TILE) cells <<= 1;
    uint32_t bs = 0; while (((uint32_t)PD_TILE >> bs) > cells) ++bs;
    return bs;
}

// The second half of making a compact sample, shared by pd_runs_create and pd_decode_end: the sorted stream is in r->r8[0 .. n_s) and the
// first run of every bucket has left its index in r->b1 (everything else 0xFFFFFFFF); `others` are the remaining runs as 12-byte arrays, any
// order.  Fills the bucket starts of the sorted stream (a suffix minimum over the marks), counts the other runs per bucket, places them
// behind o_base.  words (device, 2 x uint32, already holding the sorted stream's flags): [0] bad contig id, [1] runs longer than a bucket.
// Only enqueues on c->stream; `tmp` must hold 2 x (nb + 2) + (nb / 1024 + 4) words.
static void runs_finish(pd_ctx *c, pd_runs *r, const pd_iv *const *others, const size_t *n_others, int n_arr, uint32_t *tmp, uint32_t *words)
{
    hipStream_t st = c->stream;
    const uint32_t nb = (uint32_t)((uint64_t)c->n_tiles << r->bshift);
    const size_t nbw = (size_t)nb + 2;
    uint32_t *hist = tmp, *cursor = tmp + nbw, *bs = tmp + 2 * nbw;
    const ContigTab tab = tab_of(c);
    ProfScope ps(c, "compact_finish");
    launch_c8_fill_starts(st, r->b1, nb, r->n_s, bs);
    (void)hipMemsetAsync(hist, 0, 2 * nbw * 4, st);                       // the histogram and the buckets' cursors
    for (int k = 0; k < n_arr; ++k) launch_c8_hist(st, others[k], (uint32_t)n_others[k], tab, r->bshift, hist, words);
    launch_excl_scan_u32(st, hist, r->o1, nb + 1, bs);
    for (int k = 0; k < n_arr; ++k) launch_c8_place_other(st, others[k], (uint32_t)n_others[k], tab, r->bshift, r->o1, cursor, r->r8 + r->o_base);
}

// caller holds c->mu and has set the device.  `sorted` must be sorted by (tid, beg) — checked; the `others` may be in any order.
static int runs_make(pd_ctx *c, const pd_iv *sorted, size_t n_sorted, const pd_iv *const *others, const size_t *n_others, int n_arr, pd_runs **out)
{
    *out = nullptr;
    size_t n = n_sorted, n_o = 0;
    for (int k = 0; k < n_arr; ++k) n_o += n_others[k];
    n += n_o;
    if (n == 0 || n > DEV_BATCH_MAX) return fail(c, PD_EINVAL, "pd_runs_create: between 1 and 2^32 - 256 runs");
    if (n_arr > 2) return PD_EINVAL;
    pd_runs *r = new pd_runs;
    r->ctx = c; r->n = (uint32_t)n; r->n_s = (uint32_t)n_sorted; r->n_o = (uint32_t)n_o; r->o_base = (uint32_t)n_sorted;
    r->bshift = runs_bshift(c);
    const uint64_t nb64 = (uint64_t)c->n_tiles << r->bshift;
    if (nb64 > 0xFFFFFF00ull) { delete r; return fail(c, PD_EINVAL, "pd_runs_create: too many buckets for this genome"); }
    const uint32_t nb = (uint32_t)nb64;
    const size_t nbw = (size_t)nb + 2;
    uint32_t *tmp = nullptr, *words = nullptr;
    if (hipMalloc(&r->r8, n * sizeof(Run8) + 64) != hipSuccess || hipMalloc(&r->b1, 2 * nbw * 4) != hipSuccess ||
        hipMalloc(&tmp, (2 * nbw + nb / 1024 + 8) * 4) != hipSuccess || hipMalloc(&words, 16) != hipSuccess) {
        (void)hipGetLastError();
        for (void *q : {(void *)tmp, (void *)words}) if (q) (void)hipFree(q);
        runs_free(r);
        return fail(c, PD_ENOMEM, "pd_runs_create: allocation failed");
    }
    r->o1 = r->b1 + nbw;
    uint32_t h[2] = {0, 0};
    hipStream_t st = c->stream;
    hipError_t e = hipMemsetAsync(words, 0, 16, st);
    if (e == hipSuccess) e = hipMemsetAsync(r->b1, 0xFF, nbw * 4, st);
    if (e == hipSuccess) {
        // (this first pass is what the GPU decoder's emit kernel does as it writes a file's runs: 8-byte runs in file order, the buckets'
        // first runs marked)
        { ProfScope ps(c, "compact_runs"); launch_c8_from_sorted(st, sorted, (uint32_t)n_sorted, tab_of(c), r->bshift, r->r8, r->b1, words); }
        runs_finish(c, r, others, n_others, n_arr, tmp, words);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h, words, 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(tmp); (void)hipFree(words);
    if (e != hipSuccess) { runs_free(r); return fail(c, PD_EHIP, std::string("pd_runs_create: ") + hipGetErrorString(e)); }
    if (h[0]) { runs_free(r); return fail(c, PD_EINVAL, "pd_runs_create: the first batch is not sorted by (tid, beg), or a contig id is out of range"); }
    r->n_long = h[1];
    *out = r;
    return PD_OK;
}

int pd_runs_create(pd_ctx *c, const pd_iv *dev_sorted, size_t n_sorted, const pd_iv *dev_other, size_t n_other, pd_runs **out)
{
    if (!c || !out || (!dev_sorted && n_sorted) || (!dev_other && n_other)) return PD_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPOK(c, hipSetDevice(c->device));
    const pd_iv *o[1] = {dev_other}; const size_t no[1] = {n_other};
    return runs_make(c, dev_sorted, n_sorted, o, no, n_other ? 1 : 0, out);
}

int pd_runs_destroy(pd_runs *r)
{
    if (!r) return PD_OK;
    pd_ctx *c = r->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    (void)hipSetDevice(c->device);
    for (auto &p : c->pend) if (p.cr == r) return fail(c, PD_ESTATE, "pd_runs_destroy: the sample is still deferred on its context (pd_reset first)");
    (void)hipStreamSynchronize(c->stream);
    runs_free(r);
    return PD_OK;
}

int pd_push_runs(pd_ctx *c, const pd_runs *runs, unsigned flags)
{
    if (!c || !runs || runs->ctx != c || (flags & ~PD_PUSH_MORE)) return PD_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    if (int rs = need_state(c, 0, "pd_push_runs")) return rs;
    HIPOK(c, hipSetDevice(c->device));
    int rc = flush_pending(c);                 // what was deferred before it is scattered first
    if (rc) return rc;
    Pending p{nullptr, runs->n, 0u, -1};
    p.cr = const_cast<pd_runs *>(runs);
    c->pend.push_back(p);
    return (flags & PD_PUSH_MORE) ? PD_OK : flush_pending(c);
}

int pd_push_intervals_device(pd_ctx *c, const pd_iv *dev_iv, size_t n, unsigned flags)
{
    if (!c || (!dev_iv && n)) return PD_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    if (int rs = need_state(c, 0, "pd_push_intervals_device")) return rs;
    HIPOK(c, hipSetDevice(c->device));
    return scatter_device(c, dev_iv, n, flags, -1, nullptr);
}

int pd_stage_acquire(pd_ctx *c, pd_iv **host_buf, size_t *capacity)
{
    if (!c || !host_buf || !capacity) return PD_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPOK(c, hipSetDevice(c->device));
    int slot = -1;
    int rc = stage_acquire(c, &slot);
    if (rc) return rc;
    Stage &s = c->stage[slot];
    if (!s.host) {
        if (hipHostMalloc((void **)&s.host, STAGE_CAP * sizeof(pd_iv), hipHostMallocDefault) != hipSuccess ||
            hipMalloc((void **)&s.dev, STAGE_CAP * sizeof(pd_iv)) != hipSuccess) {
            s.state = 0;
            return fail(c, PD_ENOMEM, "staging slot allocation failed");
        }
    }
    *host_buf = s.host; *capacity = STAGE_CAP;
    return PD_OK;
}

int pd_stage_submit(pd_ctx *c, pd_iv *host_buf, size_t n, unsigned flags)
{
    if (!c || !host_buf) return PD_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    int slot = -1;
    for (int i = 0; i < (int)c->stage.size(); ++i) if (c->stage[i].host == host_buf && c->stage[i].state == 1) slot = i;
    if (slot < 0) return fail(c, PD_EINVAL, "pd_stage_submit: buffer was not handed out by pd_stage_acquire");
    if (n > STAGE_CAP) return fail(c, PD_EINVAL, "pd_stage_submit: more runs than the slot holds");
    if (c->state != 0) { c->stage[slot].state = 0; return fail(c, PD_ESTATE, "pd_stage_submit: depth already materialised (call pd_reset)"); }
    HIPOK(c, hipSetDevice(c->device));
    return stage_submit(c, slot, n, flags);
}

int pd_push_intervals(pd_ctx *c, const pd_iv *iv, size_t n, unsigned flags)
{
    if (!c || (!iv && n)) return PD_EINVAL;
    for (size_t o = 0; o < n; o += STAGE_CAP) {
        const size_t m = n - o < STAGE_CAP ? n - o : STAGE_CAP;
        pd_iv *hb = nullptr; size_t cap = 0;
        int rc = pd_stage_acquire(c, &hb, &cap);
        if (rc) return rc;
        memcpy(hb, iv + o, m * sizeof(pd_iv));
        rc = pd_stage_submit(c, hb, m, flags);
        if (rc) return rc;
    }
    return PD_OK;
}

int pd_scan(pd_ctx *c, unsigned wrap_bits)
{
    if (!c || wrap_bits > 32) return PD_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    if (int rs = need_state(c, 0, "pd_scan")) return rs;
    HIPOK(c, hipSetDevice(c->device));
    int rc = flush_pending(c);
    if (rc) return rc;
    rc = check_words(c);
    if (rc) return rc;
    const uint32_t mask = (wrap_bits == 0 || wrap_bits == 32) ? 0xFFFFFFFFu : ((1u << wrap_bits) - 1u);
    { ProfScope ps(c, "tile_carry"); launch_tile_carry(c->stream, c->sums, c->bsum, c->carry, (uint32_t)c->n_tiles); }
    { ProfScope ps(c, "scan"); launch_scan_write(c->stream, c->buf, c->carry, (uint32_t)c->n_tiles, mask, c->hstate); }
    HIPOK(c, hipGetLastError());
    HIPOK(c, hipMemsetAsync(c->hstate, 1, c->n_half, c->stream));     // the sweep wrote every cell
    launch_mark_all_valid(c->stream, c->chk);
    c->all_valid_host = true;
    c->state = 1;
    return PD_OK;
}

int pd_window_layout(const pd_ctx *c, uint32_t w, uint64_t *win_off)
{
    if (!c || !win_off || w == 0) return PD_EINVAL;
    uint64_t o = 0;
    for (int32_t i = 0; i < c->n_contigs; ++i) { win_off[i] = o; o += ((uint64_t)c->len[i] + w - 1) / w; }
    win_off[c->n_contigs] = o;
    return PD_OK;
}

static int win_keep_fit(pd_ctx *c, size_t bytes)
{
    c->wk_valid = false;
    if (bytes <= c->wk_bytes) return PD_OK;
    if (c->wk) { HIPOK(c, hipStreamSynchronize(c->stream)); HIPOK(c, hipFree(c->wk)); c->wk = nullptr; c->wk_bytes = 0; }
    const size_t want = bytes + bytes / 8 + 4096;
    if (hipMalloc(&c->wk, want) != hipSuccess) { (void)hipGetLastError(); return fail(c, PD_ENOMEM, "window statistics: device allocation failed"); }
    c->wk_bytes = want;
    return PD_OK;
}

static int windows_common(pd_ctx *c, uint32_t w, uint32_t min_dep, uint32_t mask, bool from_depth,
                          uint32_t *cover, uint64_t *sum)
{
    std::vector<uint64_t> wo((size_t)c->n_contigs + 1);
    pd_window_layout(c, w, wo.data());
    const uint64_t nw = wo[c->n_contigs];
    const size_t b_off = ((size_t)c->n_contigs + 1) * 8;
    const size_t b_sum = (size_t)nw * 8, b_cov = ((size_t)nw * 4 + 15) / 16 * 16;
    const size_t b_part = (size_t)c->n_tiles * sizeof(TilePart);      // wide windows: every tile's shares; narrow ones: the shares of the windows across tile boundaries
    const size_t b_offr = (b_off + 15) / 16 * 16;
    int rc = ensure_scratch(c, b_offr + b_part + 64);
    if (rc) return rc;
    rc = win_keep_fit(c, b_sum + b_cov + 64);
    if (rc) return rc;
    unsigned char *s = (unsigned char *)c->scratch;
    uint64_t *d_wo = (uint64_t *)s;
    unsigned long long *d_sum = (unsigned long long *)c->wk;
    uint32_t *d_cov = (uint32_t *)(c->wk + b_sum);
    TilePart *d_part = (TilePart *)(s + b_offr);
    HIPOK(c, hipMemcpyAsync(d_wo, wo.data(), b_off, hipMemcpyHostToDevice, c->stream));
    HIPOK(c, hipStreamSynchronize(c->stream));          // wo is a local
    if (w < PD_TILE) HIPOK(c, hipMemsetAsync(d_sum, 0, b_sum + b_cov, c->stream));   // windows nobody writes stay zero
    if (!from_depth) { ProfScope ps(c, "tile_carry"); launch_tile_carry(c->stream, c->sums, c->bsum, c->carry, (uint32_t)c->n_tiles); }
    {
        ProfScope ps(c, from_depth ? "reduce_windows" : "scan_reduce_windows");
        TileMap tm{c->d_tile_contig, c->d_off, c->d_len, d_wo};
        int e = launch_sweep_windows(c->stream, c->buf, c->carry, (uint32_t)c->n_tiles, mask, tm, w, min_dep,
                                     d_cov, d_sum, d_part, nw, c->n_contigs, from_depth, c->hstate);
        if (e) return fail(c, PD_EHIP, "window sweep: cannot reserve LDS");
    }
    HIPOK(c, hipGetLastError());
    HIPOK(c, hipMemcpyAsync(sum, d_sum, b_sum, hipMemcpyDeviceToHost, c->stream));
    HIPOK(c, hipMemcpyAsync(cover, d_cov, (size_t)nw * 4, hipMemcpyDeviceToHost, c->stream));
    HIPOK(c, hipStreamSynchronize(c->stream));
    c->wk_w = w; c->wk_nw = nw; c->wk_woff = wo; c->wk_valid = true;
    return PD_OK;
}

// The direct whole-sample path (k_direct_tiles): every run of the sample is still pending and the
// arrays hold nothing, so the windows are computed from the runs without ever materialising the
// difference arrays.  *done = false (and nothing consumed) when the device found a run longer than
// the look-back or a batch that was not sorted: the caller then takes the materialising path, which
// reports real errors.
static int direct_windows(pd_ctx *c, uint32_t w, uint32_t min_dep, uint32_t mask, uint32_t *cover, uint64_t *sum, bool *done)
{
    *done = false;
    std::vector<uint64_t> wo((size_t)c->n_contigs + 1);
    pd_window_layout(c, w, wo.data());
    const uint64_t nw = wo[c->n_contigs];
    const size_t b_off = ((size_t)c->n_contigs + 1) * 8;
    const size_t b_sum = (size_t)nw * 8, b_cov = ((size_t)nw * 4 + 15) / 16 * 16;
    const size_t b_part = (size_t)c->n_tiles * sizeof(TilePart);
    const size_t b_offr = (b_off + 15) / 16 * 16;
    int rc = ensure_scratch(c, b_offr + b_part + 64);
    if (rc) return rc;
    rc = win_keep_fit(c, b_sum + b_cov + 64);
    if (rc) return rc;
    unsigned char *s = (unsigned char *)c->scratch;
    uint64_t *d_wo = (uint64_t *)s;
    unsigned long long *d_sum = (unsigned long long *)c->wk;
    uint32_t *d_cov = (uint32_t *)(c->wk + b_sum);
    TilePart *d_part = (TilePart *)(s + b_offr);
    HIPOK(c, hipMemcpyAsync(d_wo, wo.data(), b_off, hipMemcpyHostToDevice, c->stream));
    HIPOK(c, hipStreamSynchronize(c->stream));          // wo is a local
    HIPOK(c, hipMemsetAsync(c->direct_words, 0, 64, c->stream));
    if (w < PD_TILE) HIPOK(c, hipMemsetAsync(d_sum, 0, b_sum + b_cov, c->stream));   // edge windows are accumulated
    const uint32_t n_stiles = (uint32_t)c->n_tiles;
    PendSet ps{};
    ps.nb = (int)c->pend.size(); ps.lmax = c->lmax;
    uint64_t all = 0;
    // a compact sample that is ALL there is serves wide windows through k_direct_c8; anywhere else it goes on as 12-byte runs
    const bool c8 = ps.nb == 1 && c->pend[0].cr && !c->pend[0].iv && w >= PD_TILE && !c->pend[0].cr->n_long;
    if (!c8) for (auto &p : c->pend) { const int re = expand_compact(c, p); if (re) return re; }
    for (int b = 0; b < ps.nb; ++b) {
        const Pending &p = c->pend[b];
        all += p.n;
        if (c8) break;
        ps.b[b] = PendBatch{p.iv, c->ub_a[b], c->cand_lo[b], c->desc + b, p.n, 0};
        ProfScope sc(c, "scatter_index");
        // a 4x sparser index than the arrays path's: measured neutral for the tile kernel, 0.43 -> 0.13 ms of index
        launch_scatter_index(c->stream, p.iv, p.n, tab_of(c), c->lmax, p.disorder, c->direct_sample, c->ub_a[b],
                             c->cand_lo[b], n_stiles, PD_TILE, c->desc + b);
    }
    unsigned grid = c->grid_tiles;
    if (!grid) {
        uint64_t g = all / 256;
        if (g < (uint64_t)c->n_cu * 4) g = (uint64_t)c->n_cu * 4;
        if (g > 65536) g = 65536;
        grid = (unsigned)g;
    }
    if (grid > c->n_tiles) grid = (unsigned)c->n_tiles;
    { ProfScope sc(c, "direct_tiles");
      if (c8) launch_direct_c8(c->stream, c->pend[0].cr->view(), tab_of(c), c->d_tile_contig, (uint32_t)c->n_tiles, mask, w, min_dep, d_part,
                               c->direct_words + 16, c->direct_words + 2, grid, c->direct_un);
      else launch_direct_tiles(c->stream, ps, tab_of(c), c->d_tile_contig, (uint32_t)c->n_tiles, mask, w, min_dep, d_part, d_wo,
                               d_cov, d_sum, c->direct_words, c->direct_words + 1, c->direct_words + 16, c->direct_words + 2, grid, c->direct_un); }
    if (w >= PD_TILE) {
        ProfScope sc(c, "gather_windows");
        TileMap tm{c->d_tile_contig, c->d_off, c->d_len, d_wo};
        launch_window_gather(c->stream, d_part, tm, c->n_contigs, w, nw, d_cov, d_sum);
    }
    HIPOK(c, hipGetLastError());
    uint32_t words[12] = {0};
    HIPOK(c, hipMemcpyAsync(words, c->direct_words, 48, hipMemcpyDeviceToHost, c->stream));
    HIPOK(c, hipMemcpyAsync(sum, d_sum, b_sum, hipMemcpyDeviceToHost, c->stream));
    HIPOK(c, hipMemcpyAsync(cover, d_cov, (size_t)nw * 4, hipMemcpyDeviceToHost, c->stream));
    HIPOK(c, hipStreamSynchronize(c->stream));
    if (words[1]) {                                      // not applicable: batches stay pending
        if (getenv("PANDEPTH_TIMING")) fprintf(stderr, "[timing]   direct window path declined: begins owned %llu of %llu runs, runs with cells %u vs ends owned %u (difference), index error bits 0x%x, "
                            "long runs %u: the arrays are materialised\n", (unsigned long long)words[3] | ((unsigned long long)words[4] << 32),
                    (unsigned long long)words[5] | ((unsigned long long)words[6] << 32), words[7], words[8], words[9], words[10]);
        return PD_OK;
    }
    // the call READ the sample: it stays deferred (like pd_export_i4's direct form), every other call still works on it
    *done = true;
    c->wk_w = w; c->wk_nw = nw; c->wk_woff = wo; c->wk_valid = true;
    return PD_OK;
}

int pd_scan_reduce_windows(pd_ctx *c, uint32_t w, uint32_t min_dep, unsigned wrap_bits, uint32_t *cover, uint64_t *sum)
{
    if (!c || !cover || !sum || w == 0 || wrap_bits > 32) return PD_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    if (int rs = need_state(c, 0, "pd_scan_reduce_windows")) return rs;
    HIPOK(c, hipSetDevice(c->device));
    if (c->direct_windows && c->pristine && !c->pend.empty() && w >= 64 && c->stile == PD_TILE) {
        const uint32_t m = (wrap_bits == 0 || wrap_bits == 32) ? 0xFFFFFFFFu : ((1u << wrap_bits) - 1u);
        bool done = false;
        int rd = direct_windows(c, w, min_dep, m, cover, sum, &done);
        if (rd || done) return rd;
    }
    int rc = flush_pending(c);
    if (rc) return rc;
    rc = check_words(c);
    if (rc) return rc;
    const uint32_t mask = (wrap_bits == 0 || wrap_bits == 32) ? 0xFFFFFFFFu : ((1u << wrap_bits) - 1u);
    return windows_common(c, w, min_dep, mask, false, cover, sum);
}

int pd_reduce_windows(pd_ctx *c, uint32_t w, uint32_t min_dep, uint32_t *cover, uint64_t *sum)
{
    if (!c || !cover || !sum || w == 0) return PD_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    if (int rs = need_state(c, 1, "pd_reduce_windows")) return rs;
    HIPOK(c, hipSetDevice(c->device));
    return windows_common(c, w, min_dep, 0xFFFFFFFFu, true, cover, sum);
}

int pd_reduce_intervals(pd_ctx *c, const pd_region *regs, size_t n, uint32_t min_dep, int32_t *cover, uint64_t *sum)
{
    if (!c || (n && (!regs || !cover || !sum))) return PD_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    if (int rs = need_state(c, 1, "pd_reduce_intervals")) return rs;
    if (n == 0) return PD_OK;
    if (n > 0xFFFFFFF0ull) return fail(c, PD_EINVAL, "too many regions");
    HIPOK(c, hipSetDevice(c->device));
    constexpr uint32_t PIECE = 16384;
    std::vector<Piece> pieces;
    pieces.reserve(n + n / 8);
    for (size_t i = 0; i < n; ++i) {
        const pd_region &r = regs[i];
        if (r.tid < 0 || r.tid >= c->n_contigs) return fail(c, PD_EINVAL, "pd_reduce_intervals: contig id out of range");
        // cells [first-1, second), clipped to the slot (the reference reads its padding; it is zero here)
        int64_t b = (int64_t)r.first - 1, e = r.second;
        const int64_t slot = (int64_t)(c->off[r.tid + 1] - c->off[r.tid]);
        if (b < 0) b = 0;
        if (e > slot) e = slot;
        for (int64_t p = b; p < e; p += PIECE) {
            Piece pc; pc.start = c->off[r.tid] + (uint64_t)p;
            pc.count = (uint32_t)((e - p) < PIECE ? (e - p) : PIECE); pc.region = (uint32_t)i;
            pieces.push_back(pc);
        }
    }
    const size_t b_p = pieces.size() * sizeof(Piece), b_sum = n * 8, b_cov = n * 4;
    int rc = ensure_scratch(c, b_p + b_sum + b_cov + 64);
    if (rc) return rc;
    unsigned char *s = (unsigned char *)c->scratch;
    unsigned long long *d_sum = (unsigned long long *)s;
    Piece *d_p = (Piece *)(s + b_sum);
    int *d_cov = (int *)(s + b_sum + b_p);
    if (!pieces.empty()) HIPOK(c, hipMemcpyAsync(d_p, pieces.data(), b_p, hipMemcpyHostToDevice, c->stream));
    HIPOK(c, hipMemsetAsync(d_sum, 0, b_sum, c->stream));
    HIPOK(c, hipMemsetAsync(d_cov, 0, b_cov, c->stream));
    HIPOK(c, hipStreamSynchronize(c->stream));          // pieces is a local
    {
        ProfScope ps(c, "reduce_intervals");
        launch_reduce_pieces(c->stream, c->buf, d_p, (uint32_t)pieces.size(), min_dep, d_cov, d_sum);
    }
    HIPOK(c, hipGetLastError());
    HIPOK(c, hipMemcpyAsync(sum, d_sum, b_sum, hipMemcpyDeviceToHost, c->stream));
    HIPOK(c, hipMemcpyAsync(cover, d_cov, b_cov, hipMemcpyDeviceToHost, c->stream));
    HIPOK(c, hipStreamSynchronize(c->stream));
    return PD_OK;
}

int pd_read_depth(pd_ctx *c, int32_t tid, uint32_t beg, size_t n, uint32_t *out)
{
    if (!c || (!out && n)) return PD_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    if (int rs = need_state(c, 1, "pd_read_depth")) return rs;
    if (tid < 0 || tid >= c->n_contigs) return fail(c, PD_EINVAL, "pd_read_depth: contig id out of range");
    if ((uint64_t)beg + n > c->off[tid + 1] - c->off[tid]) return fail(c, PD_EINVAL, "pd_read_depth: range past the contig slot");
    HIPOK(c, hipSetDevice(c->device));
    HIPOK(c, hipMemcpyAsync(out, c->buf + c->off[tid] + beg, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPOK(c, hipStreamSynchronize(c->stream));
    return PD_OK;
}

int pd_format_sites(pd_ctx *c, int32_t tid, uint32_t beg, size_t n, const char *name, size_t name_len, char *text, size_t cap, size_t *n_bytes)
{
    if (!c || !n_bytes || (!text && cap) || (!name && name_len)) return PD_EINVAL;
    *n_bytes = 0;
    std::lock_guard<std::mutex> lk(c->mu);
    if (int rs = need_state(c, 1, "pd_format_sites")) return rs;
    if (tid < 0 || tid >= c->n_contigs) return fail(c, PD_EINVAL, "pd_format_sites: contig id out of range");
    if ((uint64_t)beg + n > c->off[tid + 1] - c->off[tid]) return fail(c, PD_EINVAL, "pd_format_sites: range past the contig slot");
    if (name_len > 4096 || n > ((size_t)1 << 27)) return fail(c, PD_EINVAL, "pd_format_sites: at most 2^27 cells per call and 4096 bytes of name");
    if (n == 0) return PD_OK;
    HIPOK(c, hipSetDevice(c->device));
    const uint32_t nb = pdk::site_rows_blocks(n);
    const size_t b_name = (name_len + 15) / 16 * 16 + 16, b_cnt = ((size_t)nb * 4 + 15) / 16 * 16, b_off = ((size_t)nb + 1) * 8;
    const size_t worst = n * (name_len + 23);                  // name + 2 tabs + newline + 2 x 10 digits
    int rc = ensure_scratch(c, b_name + b_cnt + b_off + worst + 64);
    if (rc) return rc;
    unsigned char *s = (unsigned char *)c->scratch;
    char *d_name = (char *)s; uint32_t *d_cnt = (uint32_t *)(s + b_name); uint64_t *d_off = (uint64_t *)(s + b_name + b_cnt);
    char *d_text = (char *)(s + b_name + b_cnt + b_off);
    ProfScope ps(c, "format_sites");
    if (name_len) HIPOK(c, hipMemcpyAsync(d_name, name, name_len, hipMemcpyHostToDevice, c->stream));
    const uint32_t *depth = (const uint32_t *)(c->buf + c->off[tid] + beg);
    pdk::launch_site_rows(c->stream, depth, beg, n, (uint32_t)name_len, d_name, d_cnt, d_off, d_text, false);
    uint64_t total = 0;
    HIPOK(c, hipMemcpyAsync(&total, d_off + nb, 8, hipMemcpyDeviceToHost, c->stream));
    HIPOK(c, hipStreamSynchronize(c->stream));
    HIPOK(c, hipGetLastError());
    if (total > cap) return fail(c, PD_EINVAL, "pd_format_sites: the text buffer is too small");
    pdk::launch_site_rows(c->stream, depth, beg, n, (uint32_t)name_len, d_name, d_cnt, d_off, d_text, true);
    HIPOK(c, hipGetLastError());
    HIPOK(c, hipMemcpyAsync(text, d_text, (size_t)total, hipMemcpyDeviceToHost, c->stream));
    HIPOK(c, hipStreamSynchronize(c->stream));
    *n_bytes = (size_t)total;
    return PD_OK;
}

int pd_device_buffer(pd_ctx *c, void **dev_ptr, uint64_t *n_words, uint64_t *contig_off)
{
    if (!c) return PD_EINVAL;
    {   // whoever reads the raw buffer must see real zeros, not "not written since reset"
        std::lock_guard<std::mutex> lk(c->mu);
        HIPOK(c, hipSetDevice(c->device));
        int rc = flush_pending(c);
        if (rc) return rc;
        if (c->state == 0) { rc = ensure_all_valid(c); if (rc) return rc; }
    }
    if (dev_ptr) *dev_ptr = c->buf;
    if (n_words) *n_words = c->n_words;
    if (contig_off) for (int32_t i = 0; i < c->n_contigs; ++i) contig_off[i] = c->off[i];
    return PD_OK;
}

} // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// GPU-side BAM decode in asynchronous batches (include/pandepth_amd.h: pd_decode_*)
// ---------------------------------------------------------------------------------------------------------------
namespace {

enum { DS_BLOB, DS_INF, DS_BLK, DS_ST, DS_SEG /* (unused since the tables travel as one) */, DS_LANE, DS_ONLY, DS_SEGOUT, DS_R8, DS_OTH };

// PANDEPTH_TIMING=1: where the host side of the decode path spends its time (thread-microseconds, summed)
std::atomic<uint64_t> g_dec_us[8];
inline uint64_t dec_now() { return (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct DecTimer { int k; uint64_t t0; explicit DecTimer(int k_) : k(k_), t0(dec_now()) {} ~DecTimer() { g_dec_us[k] += dec_now() - t0; } };

int dec_fail(pd_ctx *c, int code, const std::string &msg) { std::lock_guard<std::mutex> lk(c->mu); return fail(c, code, msg); }

#define HIPDEC(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return dec_fail(c, PD_EHIP, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)

std::mutex g_alloc_mu;                   // pinned / device allocations of the decode slots, one at a time

int dec_ensure(pd_ctx *c, pd_ctx::DecSlot &sl, int k, size_t bytes)
{
    if (bytes <= sl.cap[k]) return PD_OK;
    std::lock_guard<std::mutex> al(g_alloc_mu);
    if (sl.d[k]) { HIPDEC(hipStreamSynchronize(sl.st)); HIPDEC(hipFree(sl.d[k])); sl.d[k] = nullptr; sl.cap[k] = 0; }
    const size_t want = bytes + bytes / 8 + 4096;
    if (hipMalloc(&sl.d[k], want) != hipSuccess) return dec_fail(c, PD_ENOMEM, "device-decode buffer allocation failed");
    sl.cap[k] = want;
    return PD_OK;
}

// the host side of a batch after pass 1: pdb2::check_chain (pd_bamwalk.h)
uint32_t dec_finish(std::vector<pdb2::Seg> &segs, std::vector<uint32_t> *redo) { return pdb2::check_chain(segs, redo); }

static_assert(sizeof(pdb2::R8) == sizeof(Run8), "the decoder's 8-byte run is the kernels' Run8");

// ---- compact decode sessions (pd_ctx::C8Dec) ----
void c8_drop(pd_ctx *c)
{
    pd_ctx::C8Dec &x = c->c8;
    if (x.compose) (void)hipStreamSynchronize(x.compose);
    if (x.base) { (void)hipFree(x.base); x.base = nullptr; }
    if (x.b1) { (void)hipFree(x.b1); x.b1 = nullptr; }
    if (x.marks) { (void)hipFree(x.marks); x.marks = nullptr; }
    const auto ina = [&](const void *p) { return c->arena && (const uint8_t *)p >= c->arena && (const uint8_t *)p < c->arena + c->arena_cap; };
    for (auto &b : x.batch) {
        if (b.seg_s && !ina(b.seg_s)) (void)hipFree(b.seg_s);
        if (b.seg_o && !ina(b.seg_o)) (void)hipFree(b.seg_o);
        if (b.ev) (void)hipEventDestroy(b.ev);
    }
    x.batch.clear(); x.base_s.clear();
    x.on = false; x.bytes = 0; x.cap_s = x.cap_o = 0; x.n_s = x.n_o = x.turn = x.n_batches = 0;
}

// room for n_s first runs and n_o later runs in the sample's final arrays (the caller holds c8.mu; copies placed earlier may still be
// running — the device is waited for before anything moves).  Returns PD_OK / PD_ENOMEM / PD_EHIP, no message.
int c8_reserve(pd_ctx *c, uint64_t n_s, uint64_t n_o, bool exact = false)
{
    pd_ctx::C8Dec &x = c->c8;
    if (n_s <= x.cap_s && n_o <= x.cap_o && x.base) return PD_OK;
    const size_t ns = std::max<size_t>((size_t)n_s + (exact ? 0 : (size_t)n_s / 2) + ((size_t)1 << 16), x.cap_s), no = std::max<size_t>((size_t)n_o + (exact ? 0 : (size_t)n_o / 2) + ((size_t)1 << 16), x.cap_o);
    const size_t bytes = (ns + no) * sizeof(Run8) + no * sizeof(pd_iv) + 256;
    uint8_t *nb = nullptr;
    if (x.base) (void)hipDeviceSynchronize();
    if (hipMalloc(&nb, bytes) != hipSuccess) { (void)hipGetLastError(); return PD_ENOMEM; }
    if (x.base) {
        hipError_t e = hipSuccess;
        if (x.n_s) e = hipMemcpy(nb, x.base, (size_t)x.n_s * sizeof(Run8), hipMemcpyDeviceToDevice);
        if (e == hipSuccess && x.n_o) e = hipMemcpy(nb + (ns + no) * sizeof(Run8), x.oth(), (size_t)x.n_o * sizeof(pd_iv), hipMemcpyDeviceToDevice);
        (void)hipFree(x.base);
        if (e != hipSuccess) { (void)hipFree(nb); x.base = nullptr; return PD_EHIP; }
    }
    x.base = nb; x.bytes = bytes; x.cap_s = ns; x.cap_o = no;
    return PD_OK;
}

// A batch has been counted (its runs are being written to its own segment, `ev` follows that kernel): it and every batch behind it whose
// predecessors are all counted now get their final places, and the copies there are queued on the compose stream.  Nobody waits.
void c8_counted(pd_ctx *c, uint64_t order, uint64_t nf, uint64_t no, Run8 *seg_s, pd_iv *seg_o, hipEvent_t ev)
{
    pd_ctx::C8Dec &x = c->c8;
    std::lock_guard<std::mutex> lk(x.mu);
    if (order >= x.batch.size() || x.batch[(size_t)order].counted) { if (x.err.empty()) x.err = "a batch number was submitted twice or lies outside the session"; return; }
    pd_ctx::C8Dec::Batch &me = x.batch[(size_t)order];
    me.counted = true; me.nf = nf; me.no = no; me.seg_s = seg_s; me.seg_o = seg_o; me.ev = ev;
    while (x.turn < x.n_batches && x.batch[(size_t)x.turn].counted) {
        pd_ctx::C8Dec::Batch &b = x.batch[(size_t)x.turn];
        x.base_s[(size_t)x.turn] = (uint32_t)x.n_s;
        if (b.nf + b.no) {
            hipError_t e = hipSuccess;
            // (when the sample outgrows its arrays — every growth waits for the device and moves what is there — they are made large enough for
            // the REST of the file at the rate seen so far, not half again: a long-read file has 1 600 later runs per first run where the first
            // estimate assumed one in four, and eight growths of gigabytes stalled every feeder — 8 thread-seconds on 128 batches)
            uint64_t want_s = x.n_s + b.nf, want_o = x.n_o + b.no;
            if ((want_s > x.cap_s || want_o > x.cap_o) && x.n_batches > x.turn + 1) {
                const double f = 1.05 * (double)x.n_batches / (double)(x.turn + 1);
                want_s = std::max<uint64_t>(want_s, (uint64_t)((double)want_s * f)); want_o = std::max<uint64_t>(want_o, (uint64_t)((double)want_o * f));
                if (c8_reserve(c, want_s, want_o, /*exact=*/true) != PD_OK) { want_s = x.n_s + b.nf; want_o = x.n_o + b.no; }      // (no room for the projection: what is needed now)
            }
            if (c8_reserve(c, x.n_s + b.nf, x.n_o + b.no) != PD_OK) e = hipErrorOutOfMemory;
            if (e == hipSuccess && b.ev) e = hipStreamWaitEvent(x.compose, b.ev, 0);
            if (e == hipSuccess && b.nf) launch_copy_words(x.compose, x.r8() + x.n_s, b.seg_s, b.nf * (sizeof(Run8) / 4));
            if (e == hipSuccess && b.no) launch_copy_words(x.compose, x.oth() + x.n_o, b.seg_o, b.no * (sizeof(pd_iv) / 4));
            if (e == hipSuccess) e = hipGetLastError();
            if (e != hipSuccess && x.err.empty()) x.err = std::string("placing a batch's runs: ") + hipGetErrorString(e);
            x.n_s += b.nf; x.n_o += b.no;
        }
        ++x.turn;
    }
}

} // namespace

extern "C" {

int pd_decode_begin(pd_ctx *c, const pd_decode_cfg *cfg)
{
    if (!c || !cfg) return PD_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    if (int rs = need_state(c, 0, "pd_decode_begin")) return rs;
    HIPOK(c, hipSetDevice(c->device));
    const uint64_t tb0 = dec_now_us(); uint64_t tb[6] = {tb0, tb0, tb0, tb0, tb0, tb0};
    struct BeginMarks { const uint64_t *t; ~BeginMarks() { if (getenv("PANDEPTH_TIMING") && t[5] - t[0] > 20000) fprintf(stderr, "[timing]   pd_decode_begin: tables %.3f s, marks + compose stream %.3f s, sample arrays %.3f s, arena %.3f s, buffers %.3f s\n", (t[1] - t[0]) / 1e6, (t[2] - t[1]) / 1e6, (t[3] - t[2]) / 1e6, (t[4] - t[3]) / 1e6, (t[5] - t[4]) / 1e6); } } begin_marks{tb};
    c->dec_cfg = *cfg;
    std::vector<uint8_t> on((size_t)c->n_contigs, 1);
    for (int32_t t = 0; t < c->n_contigs; ++t) on[(size_t)t] = cfg->contig_on ? (cfg->contig_on[t] != 0) : (c->len[(size_t)t] >= 2);
    if (!c->d_contig_on && hipMalloc(&c->d_contig_on, (size_t)c->n_contigs + 16) != hipSuccess) return fail(c, PD_ENOMEM, "pd_decode_begin: allocation failed");
    HIPOK(c, hipMemcpyAsync(c->d_contig_on, on.data(), on.size(), hipMemcpyHostToDevice, c->stream));
    if (c->d_span_off) { (void)hipFree(c->d_span_off); c->d_span_off = nullptr; }
    if (c->d_spans) { (void)hipFree(c->d_spans); c->d_spans = nullptr; }
    if (cfg->span_off && cfg->spans) {
        const size_t ns = cfg->span_off[c->n_contigs];
        if (hipMalloc(&c->d_span_off, ((size_t)c->n_contigs + 1) * 4) != hipSuccess || hipMalloc(&c->d_spans, ns * 8 + 16) != hipSuccess)
            return fail(c, PD_ENOMEM, "pd_decode_begin: allocation failed");
        HIPOK(c, hipMemcpyAsync(c->d_span_off, cfg->span_off, ((size_t)c->n_contigs + 1) * 4, hipMemcpyHostToDevice, c->stream));
        if (ns) HIPOK(c, hipMemcpyAsync(c->d_spans, cfg->spans, ns * 8, hipMemcpyHostToDevice, c->stream));
    }
    c->dec_cfg.contig_on = nullptr; c->dec_cfg.span_off = nullptr; c->dec_cfg.spans = nullptr;      // (the caller's arrays are not kept)
    // A sorted file read for whole-contig statistics (PD_DECODE_COMPACT), its batches numbered 0 .. n_batches - 1: the batches' runs go
    // straight to their final places in a compact sample (C8Dec).  Sized from the compressed bytes
    // (>= 32 B of BGZF per record of a real file; denser files make it grow): a first run per record, a later run for every fourth.
    HIPOK(c, hipStreamSynchronize(c->stream));                        // (the caller's arrays have been read)
    tb[1] = tb[2] = tb[3] = dec_now_us();
    {
        std::lock_guard<std::mutex> l8(c->c8.mu);
        pd_ctx::C8Dec &x = c->c8;
        if (x.on || !x.batch.empty()) { (void)hipDeviceSynchronize(); c8_drop(c); }     // (a session that was never ended)
        x.on = false; x.n_s = x.n_o = x.turn = 0; x.n_batches = 0; x.err.clear();
        const uint64_t nb64 = (uint64_t)c->n_tiles << runs_bshift(c);
        // (any genome size: a compact run keeps the low 32 bits of its flat begin and every consumer works relative to a tile; what is bounded is
        // the number of runs — 32-bit indices — so a file that promises more than that many records keeps 12-byte runs per batch)
        if ((cfg->flags & PD_DECODE_COMPACT) && cfg->n_batches && cfg->n_batches < (1ull << 31) && cfg->sorted && !cfg->spans && c->pend.empty() &&
            cfg->bytes_hint / 16 < DEV_BATCH_MAX && nb64 <= 0xFFFFFF00ull &&
            // (above 2^32 cells a compact session that outgrows its 32-bit run indices cannot fall back to 12-byte runs afterwards — pd_decode_end
            // would have to refuse a file already decoded — so there the caller must have said how large the file is; the executable always does)
            (c->n_cells < (1ull << 32) || cfg->bytes_hint != 0)) {
            x.bshift = runs_bshift(c);
            x.nbw = (size_t)nb64 + 2;
            if (x.b1) { (void)hipFree(x.b1); x.b1 = nullptr; }
            if (x.marks) { (void)hipFree(x.marks); x.marks = nullptr; }
            if (hipMalloc(&x.b1, 2 * x.nbw * 4) != hipSuccess || hipMalloc(&x.marks, x.nbw * 8) != hipSuccess) { (void)hipGetLastError(); return fail(c, PD_ENOMEM, "pd_decode_begin: allocation failed"); }
            HIPOK(c, hipMemsetAsync(x.marks, 0xFF, x.nbw * 8, c->stream));
            HIPOK(c, hipStreamSynchronize(c->stream));                // (the batches' kernels run on other streams)
            if (!x.compose) HIPOK(c, hipStreamCreateWithFlags(&x.compose, hipStreamNonBlocking));
            tb[2] = tb[3] = dec_now_us();
            // (>= 32 B of BGZF per record of a real short-read file: a first run per record, a later run for every fourth; c8_reserve adds
            // half again when the sample has to GROW, not to this first estimate — a 70 GB file would otherwise ask for 50 GB up front.)
            const uint64_t est = std::min<uint64_t>(cfg->bytes_hint ? cfg->bytes_hint / 32 + (1u << 20) : (uint64_t)8 << 20, DEV_BATCH_MAX);
            const int rsv = c8_reserve(c, est, est / 4, /*exact=*/true);
            tb[3] = dec_now_us();
            if (rsv == PD_OK) {
                x.n_batches = cfg->n_batches;
                x.batch.assign((size_t)cfg->n_batches, pd_ctx::C8Dec::Batch());
                x.base_s.assign((size_t)cfg->n_batches, 0u);
                x.on = true;
            } else {
                // not enough memory for the compact sample's arrays: the session goes on with 12-byte runs per batch, as sessions without
                // PD_DECODE_COMPACT do (pd_decode_end then takes the general paths)
                (void)hipGetLastError();
                if (x.b1) { (void)hipFree(x.b1); x.b1 = nullptr; }
                if (x.marks) { (void)hipFree(x.marks); x.marks = nullptr; }
            }
        }
    }
    // one arena for the batches' run arrays (a hipMalloc per batch waits for the other streams): about half the compressed
    // bytes is plenty for short reads (12 B per run against >= 30 B of BGZF per record); what does not fit is allocated singly
    // (a compact session's segments are 8-byte first runs + 12-byte later runs: a third less)
    // (round 6: a fifth in a compact session — 8 bytes per record and 12 per later run are 0.18 of a 53-bytes-per-record file — instead of a third: device
    // memory a process HOLDS is wiped when it leaves, and the next process's large allocations wait for that: 1.0-1.8 s now and then in this very call when
    // one run followed another within a second, tools/calls/r6_call27.sh)
    const size_t want = cfg->bytes_hint ? (size_t)(cfg->bytes_hint / (c->c8.on ? 5 : 2)) + ((size_t)16 << 20) : (size_t)256 << 20;
    if (c->arena_cap < want) {
        if (c->arena) { (void)hipFree(c->arena); c->arena = nullptr; c->arena_cap = 0; }
        if (hipMalloc(&c->arena, want) == hipSuccess) c->arena_cap = want; else (void)hipGetLastError();
    }
    c->arena_used = 0;
    tb[4] = tb[5] = dec_now_us();
    c->dec_n_fast = 0; c->dec_n_slow = 0; c->dec_n_redo = 0;
    for (auto &g : g_dec_us) g = 0;
    if (c->dec_warm.joinable()) c->dec_warm.join();
    if (cfg->batch_bytes && cfg->batches_in_flight) {
        const size_t need = std::max<size_t>((size_t)cfg->batch_bytes + 128, (size_t)8 << 20);
        const size_t want = need + std::max<size_t>((size_t)1 << 20, need / 32);            // (room for the batch's tables behind its bytes: pd_decode_acquire)
        if (!c->dec_warm_on) {
            // the first buffers page-locked here, from ONE thread (six readers pinning at once took 75-100 ms EACH, 5-8 ms alone)
            DecTimer ta(1);
            uint32_t k = 0;
            for (auto &sl : c->dec) {
                if (k++ >= cfg->batches_in_flight) break;
                if (sl.h_cap >= need) continue;
                if (sl.h_blob) { pin_free(sl.h_blob, sl.h_cap, sl.h_mapped); sl.h_blob = nullptr; sl.h_cap = 0; }
                if ((sl.h_blob = pin_alloc(want, &sl.h_mapped)) != nullptr) sl.h_cap = want;
            }
        } else {
        // "decode_warm" (measured and left off, tools/calls/r6_call26.sh): the first slots made ready by a helper thread, one after the other, while the caller
        // goes on — a slot's page-locked buffer, its stream and events, and by a first, empty launch the stream's hardware queue (the runtime makes it when
        // something is launched: 9 ms each, one after the other whoever asks); pd_decode_acquire hands a slot out when it is ready.  The first reader does
        // start after 20 ms — and its kernels wait until the LAST queue is made: every queue the process makes stops the ones it has (first batches collected
        // after 140-155 ms instead of 58-66 after a 50 ms pd_decode_begin).
        uint32_t n_warm = 0;
        {
            std::lock_guard<std::mutex> l2(c->dec_mu);
            for (auto &sl : c->dec) { if (n_warm >= cfg->batches_in_flight) break; sl.warming = true; ++n_warm; }
        }
        if (!c->dec_warm_word && hipMalloc(&c->dec_warm_word, 256) != hipSuccess) { (void)hipGetLastError(); c->dec_warm_word = nullptr; }
        c->dec_warm = std::thread([c, n_warm, need, want]() {
            (void)hipSetDevice(c->device);
            for (uint32_t k = 0; k < n_warm; ++k) {
                pd_ctx::DecSlot &sl = c->dec[k];
                {
                    DecTimer ta(1);
                    if (sl.h_cap < need) {
                        std::lock_guard<std::mutex> al(g_alloc_mu);
                        if (sl.h_blob) { pin_free(sl.h_blob, sl.h_cap, sl.h_mapped); sl.h_blob = nullptr; sl.h_cap = 0; }
                        if ((sl.h_blob = pin_alloc(want, &sl.h_mapped)) != nullptr) sl.h_cap = want;
                    }
                }
                if (!sl.st) {
                    bool ok = hipStreamCreateWithFlags(&sl.st, hipStreamNonBlocking) == hipSuccess;
                    for (auto &e : sl.ev) ok = ok && hipEventCreate(&e) == hipSuccess;
                    ok = ok && hipEventCreateWithFlags(&sl.ev_done, hipEventDisableTiming) == hipSuccess;
                    if (!ok) {                                              // (dec_queue makes what is missing and reports what cannot be made)
                        (void)hipGetLastError();
                        for (auto &e : sl.ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
                        if (sl.ev_done) { (void)hipEventDestroy(sl.ev_done); sl.ev_done = nullptr; }
                        if (sl.st) { (void)hipStreamDestroy(sl.st); sl.st = nullptr; }
                    } else if (c->dec_warm_word) {
                        (void)hipMemsetAsync(c->dec_warm_word, 0, 4, sl.st);      // the stream's first launch: its hardware queue is made now
                        (void)hipStreamSynchronize(sl.st);
                    }
                }
                { std::lock_guard<std::mutex> l2(c->dec_mu); sl.warming = false; }
                c->dec_cv.notify_all();
            }
        });
        }
    }
    tb[5] = dec_now_us();
    c->dec_open = true;
    return PD_OK;
}

int pd_decode_acquire(pd_ctx *c, size_t bytes, void **host_buf)
{
    if (!c || !host_buf) return PD_EINVAL;
    *host_buf = nullptr;
    std::unique_lock<std::mutex> lk(c->dec_mu);
    if (!c->dec_open) return dec_fail(c, PD_ESTATE, "pd_decode_acquire: call pd_decode_begin first");
    pd_ctx::DecSlot *sl = nullptr;
    { DecTimer tw(0); c->dec_cv.wait(lk, [&] { for (auto &x : c->dec) if (!x.busy && !x.warming) { sl = &x; return true; } return false; }); }
    sl->busy = true;
    lk.unlock();
    {
        DecTimer tsd(7);
        if (hipSetDevice(c->device) != hipSuccess) {
            { std::lock_guard<std::mutex> l2(c->dec_mu); sl->busy = false; }
            c->dec_cv.notify_one();
            return dec_fail(c, PD_EHIP, "hipSetDevice failed");
        }
    }
    DecTimer ta(1);
    if (bytes + 64 > sl->h_cap) {
        if (sl->h_blob) { pin_free(sl->h_blob, sl->h_cap, sl->h_mapped); sl->h_blob = nullptr; sl->h_cap = 0; }
        // (room behind the caller's bytes for the batch's small tables, which then travel with them in ONE copy: dec_queue)
        const size_t want = std::max<size_t>(bytes + 64, (size_t)8 << 20) + std::max<size_t>((size_t)1 << 20, bytes / 32);
        // one allocation at a time: six feeders pinning their first buffers at once took 75-100 ms EACH (4-5 ms alone)
        std::lock_guard<std::mutex> al(g_alloc_mu);
        if ((sl->h_blob = pin_alloc(want, &sl->h_mapped)) == nullptr) {
            { std::lock_guard<std::mutex> l2(c->dec_mu); sl->busy = false; }
            c->dec_cv.notify_one();
            return dec_fail(c, PD_ENOMEM, "pinned batch buffer allocation failed");
        }
        sl->h_cap = want;
    }
    *host_buf = sl->h_blob;
    return PD_OK;
}

} // extern "C"

namespace {

const bool g_dec_devtrace = getenv("PANDEPTH_DEVTRACE") != nullptr;  // (development: host-clock times at which a batch's stages were seen to end, a line per batch)
const bool g_dec_timing = getenv("PANDEPTH_TIMING") != nullptr || g_dec_devtrace;     // the per-batch device events are recorded only when somebody reads them

inline bool in_arena(const pd_ctx *c, const void *p) { return c->arena && (const uint8_t *)p >= c->arena && (const uint8_t *)p < c->arena + c->arena_cap; }

// room for a batch's runs: the arena first, an allocation of its own when that is full
bool dec_grab(pd_ctx *c, size_t bytes, void **out)
{
    bytes = (bytes + 255) & ~(size_t)255;
    const size_t at = c->arena_used.fetch_add(bytes);
    if (at + bytes <= c->arena_cap) { *out = c->arena + at; return true; }
    if (hipMalloc(out, bytes) == hipSuccess) return true;
    (void)hipGetLastError(); *out = nullptr;
    return false;
}

// a compact batch's segments and event belong to the call that made them until c8_counted has taken them
struct C8Segs {
    pd_ctx *c; Run8 *seg_s = nullptr; pd_iv *seg_o = nullptr; hipEvent_t ev = nullptr; bool kept = false;
    ~C8Segs()
    {
        if (kept) return;
        if (seg_s && !in_arena(c, seg_s)) (void)hipFree(seg_s);
        if (seg_o && !in_arena(c, seg_o)) (void)hipFree(seg_o);
        if (ev) (void)hipEventDestroy(ev);
    }
};

// every batch with an order below n_batches is counted exactly once, whatever way its calls end; a batch that had something queued
// and is counted empty through an error path leaves the session in error (its totals would silently disagree with the file)
struct C8Owes {
    pd_ctx *c; pd_ctx::DecSlot::Job *j; bool armed = true;
    ~C8Owes()
    {
        if (!armed || !j->owes_count) return;
        j->owes_count = false;
        if (j->queued) { std::lock_guard<std::mutex> lk(c->c8.mu); if (c->c8.err.empty()) c->c8.err = "a batch of the session failed on the device"; }
        c8_counted(c, j->order, 0, 0, nullptr, nullptr, nullptr);
    }
};

// ---- first half: everything the batch needs is put on the slot's stream; nothing is waited for -------------------------------------
int dec_queue(pd_ctx *c, pd_ctx::DecSlot &sl, const pd_decode_batch *bt)
{
    pd_ctx::DecSlot::Job &J = sl.job;                                 // (claimed by dec_slot_of: J.open is set)
    J.queued = false; J.fast = false; J.timed = false; J.t_q0 = dec_now_us(); J.order = bt->order; J.n_bytes = bt->n_bytes; J.inflated = bt->inflated_bytes; J.n_seg = 0;
    J.blocks.clear(); J.units.clear(); J.segs.clear(); J.seg0.clear();
    const bool c8 = J.c8 = c->c8.on;
    J.owes_count = c8 && bt->order < c->c8.n_batches;
    C8Owes owes{c, &J};
    if (c8 && bt->order >= c->c8.n_batches && bt->n_units) return dec_fail(c, PD_EINVAL, "pd_decode_submit: batch order outside [0, n_batches) of this session");
    if (!bt->n_units || !bt->n_blocks) return PD_OK;                  // (nothing to decode: the order is counted, empty)
    if (!bt->units || !bt->blocks) return dec_fail(c, PD_EINVAL, "pd_decode_submit: a batch with units needs its unit and member tables");
    if (bt->n_bytes + 64 > sl.h_cap) return dec_fail(c, PD_EINVAL, "pd_decode_submit: more bytes than were acquired");
    HIPDEC(hipSetDevice(c->device));
    uint64_t dq[8] = {}; int dqn = 0;                                  // (PANDEPTH_DEVTRACE: where the call's own time goes, first batches)
    auto dq_mark = [&]() { if (g_dec_devtrace && dqn < 8) dq[dqn++] = dec_now_us(); };
    dq_mark();
    if (!sl.st) {
        HIPDEC(hipStreamCreateWithFlags(&sl.st, hipStreamNonBlocking));
        for (auto &e : sl.ev) HIPDEC(hipEventCreate(&e));
        HIPDEC(hipEventCreateWithFlags(&sl.ev_done, hipEventDisableTiming));
    }
    J.units.assign(bt->units, bt->units + bt->n_units);
    J.blocks.assign(bt->blocks, bt->blocks + bt->n_blocks);
    // ---- segments of every unit (host) ----
    std::vector<pdb2::Seg> &segs = J.segs;
    J.seg0.assign(bt->n_units + 1, 0);
    bool guess = false;
    for (uint32_t u = 0; u < bt->n_units; ++u) {
        const pd_decode_unit &un = J.units[u];
        if (un.start > un.stop || un.start > un.avail || un.avail > bt->inflated_bytes || (uint64_t)un.first_block + un.n_blocks > bt->n_blocks)
            return dec_fail(c, PD_EINVAL, "pd_decode_submit: unit outside the inflated buffer");
        if (un.flags & PD_UNIT_GUESS) guess = true;
        J.seg0[u] = (uint32_t)segs.size();
        for (uint64_t b = un.start; b < un.stop; b += pdb2::SEG_BYTES) {
            pdb2::Seg sg; memset(&sg, 0, sizeof sg);
            sg.begin = b; sg.end = std::min<uint64_t>(b + pdb2::SEG_BYTES, un.stop); sg.avail = un.avail;
            sg.unit_first = b == un.start;
            sg.hint = (b == un.start && !(un.flags & PD_UNIT_GUESS)) ? un.start : pdb2::NONE;
            segs.push_back(sg);
        }
    }
    J.seg0[bt->n_units] = (uint32_t)segs.size();
    for (uint32_t b = 0; b < bt->n_blocks; ++b)
        if (J.blocks[b].in_off + J.blocks[b].in_len + 8 > bt->n_bytes + 8 || J.blocks[b].out_off + J.blocks[b].out_len > bt->inflated_bytes)
            return dec_fail(c, PD_EINVAL, "pd_decode_submit: block outside its buffer");
    const uint32_t n_seg = J.n_seg = (uint32_t)segs.size();
    if (!n_seg) return PD_OK;
    // The device confirms the record chain itself — no host round trip between the two passes — in sessions whose units all start at
    // known records (index cuts and index chunks: everything but no-index streams).  A kept read has at least one CIGAR operation, so its record is at least 41 bytes (4 + 32 fixed, a name of
    // one byte, one operation): inflated / 41 first runs is a bound, not an estimate.  Later runs are bounded only by the CIGAR bytes;
    // the same number of slots (several times what real reads need) is given and the chain kernel checks that they suffice.
    J.fast = c->dec_fast && !guess && (c8 || c->dec_near_span == 0xFFFFFFFFu);      // (every session whose units start at known records and whose later runs are one stream)
    J.cap_first = J.fast ? bt->inflated_bytes / 41 + 64 : 0;
    J.cap_other = J.fast ? bt->inflated_bytes / c->dec_oth_div.load() + 64 : 0;
    // (the inflate kernel's LDS lets 20 one-wave workgroups share a CU; "inflate_waves": fewer per launch, so that several batches' launches share the GPU)
    const unsigned n_wg = (unsigned)c->n_cu * c->dec_waves;
    int rc;
    dq_mark();
    J.t_mark = dec_now();
    auto lap = [&](int k) { const uint64_t n = dec_now(); g_dec_us[k] += n - J.t_mark; J.t_mark = n; };
    // The batch's small tables travel through a page-locked staging area of the slot — an "asynchronous" copy from or to pageable memory
    // is staged by the runtime on the calling thread, under a lock all streams share — and since round 5 as ONE copy each way: members,
    // segments and the (zeroed) member counter go up together into one device buffer laid out the same way; ChainOut + the segments'
    // keys (or, on the host's path, the member statuses and the segments) come back together.
    const auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    J.o_blk = 0; J.o_seg = J.o_blk + al((size_t)bt->n_blocks * sizeof(pd_bgzf_block)); J.o_next = J.o_seg + al((size_t)n_seg * sizeof(pdb2::Seg)); J.o_up = J.o_next + 256;
    J.o_bst = J.o_up; J.o_co = J.o_bst + al((size_t)bt->n_blocks * 4); J.o_so = J.o_co + sizeof(pdb2::ChainOut);
    J.o_ord = J.o_so + al((size_t)n_seg * sizeof(pdb2::SegOut));
    const size_t small_need = J.o_ord + 256;
    // the tables behind the members in the caller's (page-locked) buffer when it has the room: one host-to-device copy per batch instead of two
    const size_t tab_at = (bt->n_bytes + 64 + 255) & ~(size_t)255;
    const bool one_copy = c->dec_h2d_kernel == 0 && c->dec_h2d_fifo && tab_at + J.o_up <= sl.h_cap;
    if ((rc = dec_ensure(c, sl, DS_BLOB, one_copy ? tab_at + J.o_up : bt->n_bytes + 64)) || (rc = dec_ensure(c, sl, DS_INF, (size_t)bt->inflated_bytes + 256)) ||
        (rc = dec_ensure(c, sl, DS_BLK, J.o_up)) || (rc = dec_ensure(c, sl, DS_ST, (size_t)bt->n_blocks * 4 + 16)) ||
        (rc = dec_ensure(c, sl, DS_LANE, (size_t)n_seg * 64 * sizeof(pdb2::LaneOut))) ||
        (rc = dec_ensure(c, sl, DS_ONLY, (size_t)n_seg * 4 + 16)) || ((c8 || J.fast) && (rc = dec_ensure(c, sl, DS_SEGOUT, sizeof(pdb2::ChainOut) + (size_t)n_seg * sizeof(pdb2::SegOut)))) ||
        (J.fast && ((rc = dec_ensure(c, sl, DS_R8, (size_t)J.cap_first * (c8 ? sizeof(Run8) : sizeof(pd_iv)))) || (rc = dec_ensure(c, sl, DS_OTH, (size_t)J.cap_other * sizeof(pd_iv)))))) return rc;
    if (!sl.d_tok || sl.tok_wg < n_wg) {
        // (the scratch is indexed by workgroup: "inflate_waves" may have been raised since it was sized)
        std::lock_guard<std::mutex> al2(g_alloc_mu);
        if (sl.d_tok) { HIPDEC(hipStreamSynchronize(sl.st)); HIPDEC(hipFree(sl.d_tok)); sl.d_tok = nullptr; sl.tok_wg = 0; }
        if (hipMalloc(&sl.d_tok, bgzf_wave_scratch_bytes(n_wg)) != hipSuccess) { (void)hipGetLastError(); return dec_fail(c, PD_ENOMEM, "device-decode scratch allocation failed"); }
        sl.tok_wg = n_wg;
    }
    if (small_need > sl.h_small_cap) {
        std::lock_guard<std::mutex> al2(g_alloc_mu);
        if (sl.h_small) { HIPDEC(hipStreamSynchronize(sl.st)); (void)hipHostFree(sl.h_small); sl.h_small = nullptr; sl.h_small_cap = 0; }
        const size_t want = small_need + small_need / 4 + 4096;
        if (hipHostMalloc((void **)&sl.h_small, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return dec_fail(c, PD_ENOMEM, "pinned staging allocation failed"); }
        sl.h_small_cap = want;
    }
    uint8_t *const pin = sl.h_small;
    lap(2);                                                               // device buffers
    hipStream_t st = sl.st;
    dq_mark();
    uint8_t *d_blob = (uint8_t *)sl.d[DS_BLOB], *d_inf = (uint8_t *)sl.d[DS_INF], *d_tab = one_copy ? (uint8_t *)sl.d[DS_BLOB] + tab_at : (uint8_t *)sl.d[DS_BLK];
    J.d_tab = d_tab;
    pdb2::Seg *d_seg = (pdb2::Seg *)(d_tab + J.o_seg);
    pdb2::LaneOut *d_lane = (pdb2::LaneOut *)sl.d[DS_LANE];
    pdb2::Cfg &cfg = J.cfg;
    cfg = pdb2::Cfg{};
    cfg.buf = d_inf; cfg.avail = bt->inflated_bytes; cfg.n_ref = c->n_contigs; cfg.contig_len = c->d_len; cfg.contig_on = c->d_contig_on;
    cfg.flag_mask = c->dec_cfg.flag_mask; cfg.min_mapq = c->dec_cfg.min_mapq; cfg.span_off = c->d_span_off; cfg.spans = c->d_spans;
    cfg.near_span = c8 ? 0xFFFFFFFFu : c->dec_near_span;                   // (a compact session has one stream of later runs)
    cfg.c8 = pdb2::C8Out{};
    // ---- H2D, inflate, pass 1 ----
    // (from here on the device may be reading the caller's buffer and the slot's staging area: a call that fails half way waits for
    // what it has queued before the slot goes back)
    struct Settle { hipStream_t st; bool armed = true; ~Settle() { if (armed) (void)hipStreamSynchronize(st); } } settle{st};
    J.timed = g_dec_timing;
    memset((uint8_t *)bt->host_buf + bt->n_bytes, 0, 64);                  // (the decoder reads up to 8 bytes past a member's end)
    uint8_t *const up = one_copy ? (uint8_t *)bt->host_buf + tab_at : pin;    // where the tables are put together
    memcpy(up + J.o_blk, J.blocks.data(), (size_t)bt->n_blocks * sizeof(pd_bgzf_block));
    memcpy(up + J.o_seg, segs.data(), (size_t)n_seg * sizeof(pdb2::Seg));
    memset(up + J.o_next, 0, 256);
    // ("decode_h2d_kernel": the copy engine's transfer and the kernel behind it are ordered by a signal between two engines — 2.2 ms of idle queue per
    // batch in profiles/r05_decode_timeline.txt; a copy kernel reads the pinned bytes over the link itself and the inflate kernel follows it in the same queue)
    // (3: no copy at all — the inflate kernel reads the members straight out of the pinned buffer, which the slot holds until the batch is collected)
    if (c->dec_h2d_kernel == 0 && c->dec_h2d_fifo) {
        // ONE batch's bytes on the link at a time, in the order the batches were queued (round 6).  Copies issued on the batches' own streams share the
        // link: six readers that happen to queue together get their bytes together, six times later than the first of them could have had them, their
        // kernels then share the GPU and finish together, and the readers come back together — a convoy in which reading, copying and decoding take
        // turns instead of overlapping (tools/feeder_trace.py, profiles/r06_feeder_trace.txt: 1.25-1.35 ms per batch whatever the readers x buffers).
        // First come, first served, the first batch decodes while the second is on the link.  The copies ride on the context's main stream, which has
        // nothing else to do while a file is decoded (a stream of their own would be one more hardware queue to make: 10 ms).
        std::lock_guard<std::mutex> lk(c->dec_copy_mu);
        hipStream_t cs = c->stream;
        if (c->dec_h2d_lanes > 1 && (c->dec_copy_seq++ & 1)) {
            if (!c->dec_copy_st2) HIPDEC(hipStreamCreateWithFlags(&c->dec_copy_st2, hipStreamNonBlocking));
            cs = c->dec_copy_st2;
        }
        if (J.timed) HIPDEC(hipEventRecord(sl.ev[0], cs));
        if (one_copy) HIPDEC(hipMemcpyAsync(d_blob, bt->host_buf, tab_at + J.o_up, hipMemcpyHostToDevice, cs));
        else {
            HIPDEC(hipMemcpyAsync(d_tab, pin, J.o_up, hipMemcpyHostToDevice, cs));
            HIPDEC(hipMemcpyAsync(d_blob, bt->host_buf, bt->n_bytes + 64, hipMemcpyHostToDevice, cs));
        }
        if (J.timed) HIPDEC(hipEventRecord(sl.ev[1], cs));
        HIPDEC(hipEventRecord(sl.ev[5], cs));
        HIPDEC(hipStreamWaitEvent(st, sl.ev[5], 0));
    } else {
        if (J.timed) HIPDEC(hipEventRecord(sl.ev[0], st));
        if (c->dec_h2d_kernel == 3) d_blob = (uint8_t *)bt->host_buf;
        else if (c->dec_h2d_kernel) launch_copy_words(st, d_blob, bt->host_buf, (bt->n_bytes + 64 + 3) / 4);
        else HIPDEC(hipMemcpyAsync(d_blob, bt->host_buf, bt->n_bytes + 64, hipMemcpyHostToDevice, st));
        if (c->dec_h2d_kernel >= 2) launch_copy_words(st, d_tab, pin, (J.o_up + 3) / 4);
        else HIPDEC(hipMemcpyAsync(d_tab, pin, J.o_up, hipMemcpyHostToDevice, st));
        if (J.timed) HIPDEC(hipEventRecord(sl.ev[1], st));
    }
    dq_mark();
    launch_bgzf_inflate_wave(st, d_blob, (const pd_bgzf_block *)(d_tab + J.o_blk), bt->n_blocks, d_inf, (int *)sl.d[DS_ST], sl.d_tok, n_wg, c->dec_crc,
                             (uint32_t *)(d_tab + J.o_next), false);
    dq_mark();
    if (J.timed) HIPDEC(hipEventRecord(sl.ev[2], st));
    launch_walk_segments(st, cfg, d_seg, n_seg, d_lane, nullptr, 0);
    if (c->dec_spoil) launch_spoil_segments(st, cfg, d_seg, n_seg, d_lane, c->dec_spoil);     // (test hook)
    if (J.fast) {
        pd_ctx::C8Dec &x = c->c8;
        pdb2::ChainOut *d_co = (pdb2::ChainOut *)sl.d[DS_SEGOUT];
        launch_chain_segments(st, cfg, d_seg, n_seg, d_lane, (const int *)sl.d[DS_ST], bt->n_blocks, J.cap_first, J.cap_other, c->dec_max_redo, d_co);
        if (J.timed) HIPDEC(hipEventRecord(sl.ev[3], st));
        pdb2::Cfg c2 = cfg;
        if (c8) c2.c8 = pdb2::C8Out{(pdb2::R8 *)sl.d[DS_R8], x.marks, c->d_off, 13u - x.bshift, (pdb2::SegOut *)(d_co + 1), (uint32_t)bt->order};
        else { c2.c8 = pdb2::C8Out{}; c2.c8.seg_out = (pdb2::SegOut *)(d_co + 1); }      // (12-byte runs; the order keys ride along)
        launch_emit_segments(st, c2, d_seg, n_seg, d_lane, c8 ? nullptr : (pd_iv *)sl.d[DS_R8], (pd_iv *)sl.d[DS_OTH], nullptr, d_co);
        HIPDEC(hipMemcpyAsync(pin + J.o_co, d_co, sizeof(pdb2::ChainOut) + (size_t)n_seg * sizeof(pdb2::SegOut), hipMemcpyDeviceToHost, st));
        if (J.timed) HIPDEC(hipEventRecord(sl.ev[4], st));
    } else {
        HIPDEC(hipMemcpyAsync(pin + J.o_bst, sl.d[DS_ST], (size_t)bt->n_blocks * 4, hipMemcpyDeviceToHost, st));
        HIPDEC(hipMemcpyAsync(pin + J.o_seg, d_seg, (size_t)n_seg * sizeof(pdb2::Seg), hipMemcpyDeviceToHost, st));
        if (J.timed) HIPDEC(hipEventRecord(sl.ev[3], st));
    }
    // (what the collecting call waits for.  hipStreamSynchronize would put a marker of its own into the stream's HARDWARE queue at the time of the call — and
    // the process's streams share eight of those: the marker landed behind whatever another batch's stream had in the same queue, and a batch that had long
    // finished was "collected" 2 ms later, when the other batch was through: tools/calls/r6_call14.sh, profiles/r06_devtrace.txt)
    HIPDEC(hipEventRecord(sl.ev_done, st));
    HIPDEC(hipGetLastError());
    dq_mark();
    if (g_dec_devtrace && bt->order < 14)
        fprintf(stderr, "[devtrace] batch %llu pd_decode_queue: stream + events + segments %llu us, buffers %llu, copies issued %llu, inflate launched %llu, the rest launched %llu\n", (unsigned long long)bt->order,
                (unsigned long long)(dq[1] - dq[0]), (unsigned long long)(dq[2] - dq[1]), (unsigned long long)(dq[3] - dq[2]), (unsigned long long)(dq[4] - dq[3]), (unsigned long long)(dq[5] - dq[4]));
    J.queued = true; J.t_q1 = dec_now_us();
    owes.armed = false;                                                   // (the second half counts the order)
    settle.armed = false;
    return PD_OK;
}

// ---- second half: wait for the batch, finish it, report what pd_decode_submit reports ------------------------------------------------
int dec_collect(pd_ctx *c, pd_ctx::DecSlot &sl, int32_t *unit_status, pd_decode_result *res)
{
    pd_ctx::DecSlot::Job &J = sl.job;                                 // (J.open goes with the slot: dec_release)
    if (res) { memset(res, 0, sizeof *res); res->first_start = res->next_start = ~0ull; }
    const uint32_t n_units = (uint32_t)J.units.size(), n_blocks = (uint32_t)J.blocks.size(), n_seg = J.n_seg;
    if (unit_status) for (uint32_t u = 0; u < n_units; ++u) unit_status[u] = 0;
    C8Owes owes{c, &J};
    if (!J.queued) return PD_OK;
    if (!unit_status) return dec_fail(c, PD_EINVAL, "pd_decode_collect: unit_status is required for a batch with units");
    const bool c8 = J.c8;
    std::vector<pdb2::Seg> &segs = J.segs;
    const std::vector<uint32_t> &seg0 = J.seg0;
    HIPDEC(hipSetDevice(c->device));
    hipStream_t st = sl.st;
    uint8_t *const pin = sl.h_small;
    uint8_t *d_tab = J.d_tab ? J.d_tab : (uint8_t *)sl.d[DS_BLK];
    pdb2::Seg *d_seg = (pdb2::Seg *)(d_tab + J.o_seg);
    pdb2::LaneOut *d_lane = (pdb2::LaneOut *)sl.d[DS_LANE];
    pdb2::SegOut *d_so = c8 ? (pdb2::SegOut *)((uint8_t *)sl.d[DS_SEGOUT] + sizeof(pdb2::ChainOut)) : nullptr;
    pdb2::Cfg cfg = J.cfg;
    J.t_mark = dec_now();
    auto lap = [&](int k) { const uint64_t n = dec_now(); g_dec_us[k] += n - J.t_mark; J.t_mark = n; };
    if (g_dec_devtrace && J.timed) {
        uint64_t t[6] = {};
        for (int k = 0; k < 5; ++k) { if (k < 4 || J.fast) (void)hipEventSynchronize(sl.ev[k]); t[k] = dec_now_us(); }
        (void)hipEventSynchronize(sl.ev_done); t[5] = dec_now_us();
        fprintf(stderr, "[devtrace] batch %llu queue call %llu us; since its start: collect entered %llu, copy begun %llu, copied %llu, inflated %llu, walked %llu, emitted %llu, stream idle %llu\n",
                (unsigned long long)J.order, (unsigned long long)(J.t_q1 - J.t_q0), (unsigned long long)(J.t_mark - J.t_q0), (unsigned long long)(t[0] - J.t_q0), (unsigned long long)(t[1] - J.t_q0),
                (unsigned long long)(t[2] - J.t_q0), (unsigned long long)(t[3] - J.t_q0), (unsigned long long)(t[4] - J.t_q0), (unsigned long long)(t[5] - J.t_q0));
    }
    HIPDEC(c->dec_sync_event ? hipEventSynchronize(sl.ev_done) : hipStreamSynchronize(st));
    HIPDEC(hipGetLastError());
    lap(3);                                                               // waiting for the device
    auto times = [&](bool emitted) {
        if (!res || !J.timed) return;
        float ms = 0;
        if (hipEventElapsedTime(&ms, sl.ev[0], sl.ev[1]) == hipSuccess) res->ms_h2d = ms;
        if (hipEventElapsedTime(&ms, sl.ev[1], sl.ev[2]) == hipSuccess) res->ms_inflate = ms;
        if (hipEventElapsedTime(&ms, sl.ev[2], sl.ev[3]) == hipSuccess) res->ms_walk = ms;
        if (emitted && hipEventElapsedTime(&ms, sl.ev[3], sl.ev[4]) == hipSuccess) res->ms_emit = ms;
    };
    // the order of a compact batch's first runs across its segments (inside a lane and across the lanes of a segment the emission checked it)
    auto order_of = [&](const pdb2::SegOut *so, pd_ctx::RunSeg *rs) {
        uint64_t prev = 0, first = pdb2::NONE, n_long = 0; uint32_t bad = 0;
        for (uint32_t j = 0; j < n_seg; ++j) {
            bad |= so[j].unsorted; n_long += so[j].n_long;
            if (so[j].first_key == pdb2::NONE) continue;
            if (first == pdb2::NONE) first = so[j].first_key; else if (so[j].first_key < prev) bad = 1;
            prev = so[j].last_key;
        }
        rs->unsorted = bad ? 1u : 0u; rs->first_key = first; rs->last_key = prev; rs->n_long = n_long;
    };
    if (J.fast) {
        pdb2::ChainOut co;
        memcpy(&co, pin + J.o_co, sizeof co);
        c->dec_n_redo += co.n_redo;
        if (!co.slow) {
            // ---- the device has confirmed the chain and written the runs to the slot's arrays: exact arrays for them, copied behind the
            // emission on this stream (the slot's arrays are free again when its next batch gets there), and the batch is counted
            ++c->dec_n_fast;
            const uint64_t nf = co.n_first, no = co.n_other;
            pd_ctx::RunSeg rs{J.order, nullptr, nf, nullptr, no, nullptr, 0, co.max_span, 0u, 0ull, 0ull};
            if (!c8) {
                // 12-byte runs (every mode that needs the arrays): exact arrays from the arena, the batch listed for pd_decode_end
                struct Guard { pd_ctx *c; pd_ctx::RunSeg *r; bool keep = false;
                               ~Guard() { if (keep) return; for (pd_iv *q : {r->first, r->other}) if (q && !in_arena(c, q)) (void)hipFree(q); } } guard{c, &rs};
                if (nf + no) {
                    if ((nf && !dec_grab(c, (size_t)nf * sizeof(pd_iv), (void **)&rs.first)) || (no && !dec_grab(c, (size_t)no * sizeof(pd_iv), (void **)&rs.other)))
                        return dec_fail(c, PD_ENOMEM, "run array allocation failed");
                    if (nf) launch_copy_words(st, rs.first, sl.d[DS_R8], nf * (sizeof(pd_iv) / 4));
                    if (no) launch_copy_words(st, rs.other, sl.d[DS_OTH], no * (sizeof(pd_iv) / 4));
                    HIPDEC(hipGetLastError());                       // (pd_decode_end waits for the slots' streams before it reads these arrays)
                    order_of((const pdb2::SegOut *)(pin + J.o_so), &rs);
                    rs.n_long = 0;
                }
                if (res) { res->n_first = nf; res->n_other = no; res->n_reads = co.n_rec; res->unsorted = rs.unsorted; res->first_key = rs.first_key; res->last_key = rs.last_key;
                           res->first_start = co.first_start; res->next_start = co.next_start; }
                times(true);
                lap(5);
                if (nf + no) { std::lock_guard<std::mutex> lk(c->dec_mu); c->run_segs.push_back(rs); }
                guard.keep = true;
                return PD_OK;
            }
            C8Segs g{c};
            if (nf + no) {
                if ((nf && !dec_grab(c, (size_t)nf * sizeof(Run8), (void **)&g.seg_s)) || (no && !dec_grab(c, (size_t)no * sizeof(pd_iv), (void **)&g.seg_o)) ||
                    hipEventCreateWithFlags(&g.ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return dec_fail(c, PD_ENOMEM, "run segment allocation failed"); }
                if (nf) launch_copy_words(st, g.seg_s, sl.d[DS_R8], nf * (sizeof(Run8) / 4));
                if (no) launch_copy_words(st, g.seg_o, sl.d[DS_OTH], no * (sizeof(pd_iv) / 4));
                HIPDEC(hipGetLastError());
                HIPDEC(hipEventRecord(g.ev, st));
            }
            J.owes_count = false; g.kept = true;
            c8_counted(c, J.order, nf, no, g.seg_s, g.seg_o, g.ev);
            if (nf + no) order_of((const pdb2::SegOut *)(pin + J.o_so), &rs);
            if (res) { res->n_first = nf; res->n_other = no; res->n_reads = co.n_rec; res->unsorted = rs.unsorted; res->first_key = rs.first_key; res->last_key = rs.last_key;
                       res->first_start = co.first_start; res->next_start = co.next_start; }
            times(true);
            lap(5);
            if (nf + no) { std::lock_guard<std::mutex> lk(c->dec_mu); c->run_segs.push_back(rs); }
            return PD_OK;
        }
        // ---- out of the ordinary (ChainOut::slow says why; nothing was emitted): the member statuses and the segments as they stand come
        // to the host, which goes through the batch the way it always has
        ++c->dec_n_slow;
        // (more later runs than the batch's array holds — reads with thousands of CIGAR operations: the batches queued from now on get
        // a slot per 8 inflated bytes, which an alternation of matches and gaps cannot exceed)
        if ((co.slow & pdb2::CH_ROOM) && co.n_first <= J.cap_first) c->dec_oth_div.store(8);
        HIPDEC(hipMemcpyAsync(pin + J.o_bst, sl.d[DS_ST], (size_t)n_blocks * 4, hipMemcpyDeviceToHost, st));
        HIPDEC(hipMemcpyAsync(pin + J.o_seg, d_seg, (size_t)n_seg * sizeof(pdb2::Seg), hipMemcpyDeviceToHost, st));
        HIPDEC(hipStreamSynchronize(st));
    }
    if (!J.fast) ++c->dec_n_slow;
    std::vector<int> bst(n_blocks);
    memcpy(bst.data(), pin + J.o_bst, (size_t)n_blocks * 4);
    memcpy(segs.data(), pin + J.o_seg, (size_t)n_seg * sizeof(pdb2::Seg));
    // ---- the chain across segments; segments whose guess was wrong walk again from the corrected start ----
    std::vector<uint32_t> redo;
    for (int round = 0; dec_finish(segs, &redo) > 0; ++round) {
        if (round >= 24) { for (uint32_t j : redo) segs[j].flags |= pdb2::WF_BAD; break; }
        for (uint32_t j : redo) HIPDEC(hipMemcpyAsync(&d_seg[j].hint, &segs[j].hint, 8, hipMemcpyHostToDevice, st));
        HIPDEC(hipMemcpyAsync(sl.d[DS_ONLY], redo.data(), redo.size() * 4, hipMemcpyHostToDevice, st));
        launch_walk_segments(st, cfg, d_seg, n_seg, d_lane, (const uint32_t *)sl.d[DS_ONLY], (uint32_t)redo.size());
        HIPDEC(hipMemcpyAsync(pin + J.o_seg, d_seg, (size_t)n_seg * sizeof(pdb2::Seg), hipMemcpyDeviceToHost, st));
        HIPDEC(hipStreamSynchronize(st));
        memcpy(segs.data(), pin + J.o_seg, (size_t)n_seg * sizeof(pdb2::Seg));
    }
    // ---- unit outcomes; units handed back emit nothing ----
    uint64_t nf = 0, no = 0, nfar = 0, nrec = 0; uint32_t max_span = 0;
    for (uint32_t u = 0; u < n_units; ++u) {
        int stt = 0;
        const pd_decode_unit &un = J.units[u];
        for (uint32_t b = 0; b < un.n_blocks; ++b) { const int v = bst[un.first_block + b]; if (v < 0) stt = 2; else if (v > 0 && stt == 0) stt = 1; }
        for (uint32_t j = seg0[u]; j < seg0[u + 1]; ++j) {
            if (segs[j].flags & pdb2::WF_BAD) { if (stt != 2) stt = 3; }
            else if ((segs[j].flags & (pdb2::WF_MORE | pdb2::WF_HOST)) && stt == 0) stt = 1;
        }
        unit_status[u] = stt;
        for (uint32_t j = seg0[u]; j < seg0[u + 1]; ++j) {
            if (stt) { segs[j].n_first = segs[j].n_other = segs[j].n_far = 0; } else nrec += segs[j].n_rec;
            segs[j].base_first = nf; segs[j].base_other = no; segs[j].base_far = nfar;
            nf += segs[j].n_first; no += segs[j].n_other; nfar += segs[j].n_far;
            if (!stt && segs[j].max_span > max_span) max_span = segs[j].max_span;
        }
    }
    if (res) {
        res->n_first = nf; res->n_other = no + nfar; res->n_reads = nrec;
        uint64_t fs = ~0ull, E = 0;
        for (uint32_t j = seg0[0]; j < seg0[1]; ++j) { if (fs == ~0ull && segs[j].used_start != pdb2::NONE) fs = segs[j].used_start; if (segs[j].e_last > E) E = segs[j].e_last; }
        res->first_start = fs; res->next_start = E ? E : ~0ull;
    }
    // ---- pass 2: the runs ----
    pd_ctx::RunSeg rs{J.order, nullptr, nf, nullptr, no, nullptr, nfar, max_span, 0u, 0ull, 0ull};
    // run arrays taken outside the arena belong to this call until the batch is listed: every early return gives them back
    struct RunGuard {
        pd_ctx *c; pd_ctx::RunSeg *r; bool keep = false;
        ~RunGuard() { if (keep) return; for (pd_iv *q : {r->first, r->other, r->far}) if (q && !in_arena(c, q)) (void)hipFree(q); }
    } run_guard{c, &rs};
    lap(4);                                                               // host: chain check, unit outcomes
    bool have_so = false;
    if (c8) {
        // compact session: pass 2 writes the batch's first runs as 8-byte runs into a segment of its own and marks the buckets' first runs;
        // c8_counted then queues the copies to the runs' final places for every batch whose predecessors are all counted
        pd_ctx::C8Dec &x = c->c8;
        C8Segs g{c};
        if (nf + no) {
            if ((nf && !dec_grab(c, (size_t)nf * sizeof(Run8), (void **)&g.seg_s)) || (no && !dec_grab(c, (size_t)no * sizeof(pd_iv), (void **)&g.seg_o)) ||
                hipEventCreateWithFlags(&g.ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return dec_fail(c, PD_ENOMEM, "run segment allocation failed"); }
            memcpy(pin + J.o_seg, segs.data(), (size_t)n_seg * sizeof(pdb2::Seg));
            HIPDEC(hipMemcpyAsync(d_seg, pin + J.o_seg, (size_t)n_seg * sizeof(pdb2::Seg), hipMemcpyHostToDevice, st));
            cfg.c8 = pdb2::C8Out{(pdb2::R8 *)g.seg_s, x.marks, c->d_off, 13u - x.bshift, d_so, (uint32_t)J.order};
            launch_emit_segments(st, cfg, d_seg, n_seg, d_lane, nullptr, g.seg_o, nullptr, nullptr);
            HIPDEC(hipEventRecord(g.ev, st));
        }
        J.owes_count = false; g.kept = true;
        c8_counted(c, J.order, nf, no, g.seg_s, g.seg_o, g.ev);
        if (nf + no) {
            have_so = true;
            HIPDEC(hipMemcpyAsync(pin + J.o_so, d_so, (size_t)n_seg * sizeof(pdb2::SegOut), hipMemcpyDeviceToHost, st));
        }
        lap(5);
    } else if (nf + no + nfar) {
        if ((nf && !dec_grab(c, (size_t)nf * sizeof(pd_iv), (void **)&rs.first)) || (no && !dec_grab(c, (size_t)no * sizeof(pd_iv), (void **)&rs.other)) ||
            (nfar && !dec_grab(c, (size_t)nfar * sizeof(pd_iv), (void **)&rs.far))) return dec_fail(c, PD_ENOMEM, "run array allocation failed");
        memcpy(pin + J.o_seg, segs.data(), (size_t)n_seg * sizeof(pdb2::Seg));
        HIPDEC(hipMemcpyAsync(d_seg, pin + J.o_seg, (size_t)n_seg * sizeof(pdb2::Seg), hipMemcpyHostToDevice, st));
        lap(5);                                                           // run array allocation
        launch_emit_segments(st, cfg, d_seg, n_seg, d_lane, rs.first, rs.other, rs.far, nullptr);
    }
    // are the first runs in (tid, begin) order, as the header's SO:coordinate promises?  (DS_ONLY is free again: 6 words)
    uint32_t order_words[6] = {0, 0, 0, 0, 0, 0};
    if (nf && !c8) {
        HIPDEC(hipMemsetAsync(sl.d[DS_ONLY], 0, 24, st));
        launch_runs_sorted(st, rs.first, nf, (uint32_t *)sl.d[DS_ONLY]);
        HIPDEC(hipMemcpyAsync(pin + J.o_ord, sl.d[DS_ONLY], 24, hipMemcpyDeviceToHost, st));
    }
    if (J.timed) HIPDEC(hipEventRecord(sl.ev[4], st));
    HIPDEC(hipStreamSynchronize(st));
    HIPDEC(hipGetLastError());
    if (nf && !c8) {
        memcpy(order_words, pin + J.o_ord, 24);
        rs.unsorted = order_words[0];
        rs.first_key = (uint64_t)order_words[2] | ((uint64_t)order_words[3] << 32);
        rs.last_key = (uint64_t)order_words[4] | ((uint64_t)order_words[5] << 32);
    }
    if (have_so) order_of((const pdb2::SegOut *)(pin + J.o_so), &rs);
    if (res) { res->unsorted = rs.unsorted; res->first_key = rs.first_key; res->last_key = rs.last_key; }
    lap(6);                                                               // pass 2 (waiting)
    times(!J.fast);
    if (nf + no + nfar) { std::lock_guard<std::mutex> lk(c->dec_mu); c->run_segs.push_back(rs); }
    run_guard.keep = true;
    return PD_OK;
}

pd_ctx::DecSlot *dec_slot_of(pd_ctx *c, const void *host_buf)
{
    // (other feeders may be in pd_decode_acquire, re-allocating THEIR slots' pinned buffers: look the slot up under the lock)
    std::lock_guard<std::mutex> l0(c->dec_mu);
    for (auto &x : c->dec) if (x.busy && !x.job.open && x.h_blob == host_buf) { x.job.open = true; return &x; }      // (claimed: a batch is under way in this slot)
    return nullptr;
}
void dec_release(pd_ctx *c, pd_ctx::DecSlot *s) { { std::lock_guard<std::mutex> l(c->dec_mu); s->busy = false; s->job.open = false; s->job.collecting = false; } c->dec_cv.notify_all(); }

} // namespace

extern "C" {

int pd_decode_submit(pd_ctx *c, const pd_decode_batch *bt, int32_t *unit_status, pd_decode_result *res)
{
    if (!c || !bt || !bt->host_buf || !unit_status) return PD_EINVAL;
    pd_ctx::DecSlot *slp = dec_slot_of(c, bt->host_buf);
    if (!slp) return dec_fail(c, PD_EINVAL, "pd_decode_submit: buffer was not handed out by pd_decode_acquire");
    struct Release { pd_ctx *c; pd_ctx::DecSlot *s; ~Release() { dec_release(c, s); } } rel{c, slp};
    if (res) { memset(res, 0, sizeof *res); res->first_start = res->next_start = ~0ull; }
    for (uint32_t u = 0; u < bt->n_units; ++u) unit_status[u] = 0;
    // (a ticket of an earlier batch on this slot is stale from here on, and nobody else may collect or drain the batch about to be queued)
    { std::lock_guard<std::mutex> l0(c->dec_mu); ++slp->gen; slp->job.collecting = true; }
    const int rc = dec_queue(c, *slp, bt);
    if (rc) return rc;
    return dec_collect(c, *slp, unit_status, res);
}

int pd_decode_queue(pd_ctx *c, const pd_decode_batch *bt, uint64_t *ticket)
{
    if (!c || !bt || !bt->host_buf || !ticket) return PD_EINVAL;
    *ticket = 0;
    pd_ctx::DecSlot *slp = dec_slot_of(c, bt->host_buf);
    if (!slp) return dec_fail(c, PD_EINVAL, "pd_decode_queue: buffer was not handed out by pd_decode_acquire");
    const int rc = dec_queue(c, *slp, bt);
    if (rc) { dec_release(c, slp); return rc; }
    { std::lock_guard<std::mutex> l0(c->dec_mu); *ticket = ((uint64_t)++slp->gen << 8) | (uint64_t)(slp - c->dec + 1); }
    return PD_OK;
}

int pd_decode_collect(pd_ctx *c, uint64_t ticket, int32_t *unit_status, pd_decode_result *res)
{
    if (!c) return PD_EINVAL;
    const uint64_t k = ticket & 0xff;
    pd_ctx::DecSlot *slp = k >= 1 && k <= (uint64_t)pd_ctx::N_DEC ? &c->dec[k - 1] : nullptr;
    {
        std::lock_guard<std::mutex> l0(c->dec_mu);
        if (!slp || !slp->busy || !slp->job.open || slp->job.collecting || slp->gen != (uint32_t)(ticket >> 8)) slp = nullptr;
        else slp->job.collecting = true;
    }
    if (!slp) return dec_fail(c, PD_EINVAL, "pd_decode_collect: not the ticket of a queued batch (or the batch is being collected already)");
    struct Release { pd_ctx *c; pd_ctx::DecSlot *s; ~Release() { dec_release(c, s); } } rel{c, slp};
    return dec_collect(c, *slp, unit_status, res);
}

// batches that were queued and never collected: finished here (pd_decode_end: they count) or waited for and dropped (pd_decode_abort)
static void dec_drain(pd_ctx *c, bool finish)
{
    for (auto &sl : c->dec) {
        bool mine = false;
        // (a slot some thread is collecting, or is inside pd_decode_submit on, is left to that thread: the callers wait on dec_cv for it)
        { std::lock_guard<std::mutex> lk(c->dec_mu); mine = sl.busy && sl.job.open && !sl.job.collecting; if (mine) sl.job.collecting = true; }
        if (!mine) continue;
        if (finish) { std::vector<int32_t> st(sl.job.units.size() + 1, 0); (void)dec_collect(c, sl, st.data(), nullptr); }
        else { (void)hipSetDevice(c->device); if (sl.st) (void)hipStreamSynchronize(sl.st); C8Owes owes{c, &sl.job}; sl.job.queued = false; }
        dec_release(c, &sl);
    }
}

int pd_decode_end(pd_ctx *c)
{
    if (!c) return PD_EINVAL;
    if (c->dec_warm.joinable()) c->dec_warm.join();
    dec_drain(c, true);
    {   // every batch has returned; wait for stragglers that still hold a slot
        std::unique_lock<std::mutex> lk(c->dec_mu);
        c->dec_cv.wait(lk, [&] { for (auto &x : c->dec) if (x.busy) return false; return true; });
        c->dec_open = false;
    }
    std::lock_guard<std::mutex> lk(c->mu);
    if (int rs = need_state(c, 0, "pd_decode_end")) return rs;
    HIPOK(c, hipSetDevice(c->device));
    for (auto &sl : c->dec) if (sl.st) HIPOK(c, hipStreamSynchronize(sl.st));      // (the last copies of the batches' runs to their arrays)
    std::vector<pd_ctx::RunSeg> segs;
    { std::lock_guard<std::mutex> l2(c->dec_mu); segs.swap(c->run_segs); }
    std::sort(segs.begin(), segs.end(), [](const pd_ctx::RunSeg &a, const pd_ctx::RunSeg &b) { return a.order < b.order; });
    uint64_t nf = 0, no = 0, nfar = 0; uint32_t span = 0;
    for (auto &r : segs) { nf += r.n_first; no += r.n_other; nfar += r.n_far; if (r.max_span > span) span = r.max_span; }
    auto in_arena = [&](const void *p) { return c->arena && (const uint8_t *)p >= c->arena && (const uint8_t *)p < c->arena + c->arena_cap; };
    auto drop = [&]() { for (auto &r : segs) for (pd_iv *q : {r.first, r.other, r.far}) if (q && !in_arena(q)) (void)hipFree(q); };
    if (c->run_first || c->run_other || c->run_far || c->dec_runs) {
        // an earlier sample of this context (#.list: one file after another) may still be deferred on these arrays
        int rf = flush_pending(c);
        if (rf) { drop(); return rf; }
        HIPOK(c, hipStreamSynchronize(c->stream));
        for (pd_iv **q : {&c->run_first, &c->run_other, &c->run_far}) if (*q) { (void)hipFree(*q); *q = nullptr; }
        runs_free(c->dec_runs); c->dec_runs = nullptr;
    }
    if (c->c8.on) {
        // ---- a compact session: the runs are where they belong already (or on their way there, on the compose stream) ----
        pd_ctx::C8Dec &x = c->c8;
        std::lock_guard<std::mutex> l8(x.mu);
        x.on = false;
        if (x.turn != x.n_batches || !x.err.empty()) {
            const std::string why = x.err.empty() ? "not every batch of the compact session was submitted" : x.err;
            (void)hipDeviceSynchronize(); c8_drop(c);
            return fail(c, PD_ESTATE, "pd_decode_end: " + why);
        }
        bool ok_order = true; uint64_t prev = 0, n_long = 0; bool have = false;
        for (auto &r : segs) {
            n_long += r.n_long;
            if (!r.n_first) continue;
            if (r.unsorted || (have && r.first_key < prev)) ok_order = false;
            prev = r.last_key; have = true;
        }
        if (getenv("PANDEPTH_TIMING"))
            fprintf(stderr, "[timing]   decode entry points, thread-seconds: slot wait %.3f, pinned alloc %.3f, device buffers + queueing %.3f, waiting for the device %.3f, "
                            "host chain check %.3f, runs to their arrays %.3f, wait emit (host's path) %.3f, first HIP call of the feeder threads %.3f; %zu batches; runs (compact session): %llu first, %llu later (span %u); "
                            "chain confirmed on the device for %llu batches, by the host for %llu; segments the device walked again: %llu\n",
                    g_dec_us[0] / 1e6, g_dec_us[1] / 1e6, g_dec_us[2] / 1e6, g_dec_us[3] / 1e6, g_dec_us[4] / 1e6, g_dec_us[5] / 1e6, g_dec_us[6] / 1e6, g_dec_us[7] / 1e6, segs.size(),
                    (unsigned long long)x.n_s, (unsigned long long)x.n_o, span, (unsigned long long)c->dec_n_fast.load(), (unsigned long long)c->dec_n_slow.load(), (unsigned long long)c->dec_n_redo.load());
        if (x.n_s + x.n_o == 0) { (void)hipStreamSynchronize(x.compose); c8_drop(c); return PD_OK; }
        HIPOK(c, hipStreamSynchronize(x.compose));                 // every batch's runs have reached their places
        if (ok_order && x.n_s && x.n_s + x.n_o <= DEV_BATCH_MAX && !c->pend.empty()) {
            // the context holds other runs already (units the device handed back and the host decoded meanwhile, an earlier file of a list):
            // they go into the arrays now, and the compact sample is pushed behind them like any other deferred batch
            const int rf = flush_pending(c);
            if (rf) { (void)hipDeviceSynchronize(); c8_drop(c); return rf; }
        }
        if (ok_order && x.n_s && c->pend.empty() && x.n_s + x.n_o <= DEV_BATCH_MAX) {
            pd_runs *r = new pd_runs;
            r->ctx = c; r->r8 = x.r8(); r->own_r8 = true; r->n_s = (uint32_t)x.n_s; r->n_o = (uint32_t)x.n_o; r->n = r->n_s + r->n_o; r->o_base = r->n_s;      // (the later runs go right behind the sorted stream, as in pd_runs_create: what counts is how many runs there ARE, not how many were reserved)
            r->b1 = x.b1; r->o1 = x.b1 + x.nbw; r->bshift = x.bshift;
            const pd_iv *oth = x.oth(); const size_t no1 = (size_t)x.n_o;
            uint32_t *tmp = nullptr, *words = nullptr, *d_base = nullptr;
            const size_t nbw = x.nbw;
            unsigned long long *marks = x.marks;
            const std::vector<uint32_t> base_s = x.base_s;
            x.base = nullptr; x.b1 = nullptr; x.marks = nullptr; x.bytes = 0; x.cap_s = x.cap_o = 0;       // (they belong to the sample now; the marks go below)
            c8_drop(c);                                                                                        // the batches' segments and events
            if (hipMalloc(&tmp, (2 * nbw + nbw / 1024 + 8) * 4) != hipSuccess || hipMalloc(&words, 16) != hipSuccess || hipMalloc(&d_base, base_s.size() * 4 + 16) != hipSuccess) {
                (void)hipGetLastError(); for (void *q : {(void *)tmp, (void *)words, (void *)d_base, (void *)marks}) if (q) (void)hipFree(q); runs_free(r);
                return fail(c, PD_ENOMEM, "pd_decode_end: allocation failed");
            }
            uint32_t h[2] = {0, 0};
            hipError_t e = hipMemsetAsync(words, 0, 16, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(d_base, base_s.data(), base_s.size() * 4, hipMemcpyHostToDevice, c->stream);
            if (e == hipSuccess) {
                // the marks (batch, index in the batch) of the buckets' first runs become indices into the sorted stream
                { ProfScope ps(c, "compact_finish"); launch_c8_marks_to_index(c->stream, marks, (uint32_t)(nbw - 1), d_base, r->b1); }
                const pd_iv *o[1] = {oth}; const size_t non[1] = {no1};
                runs_finish(c, r, o, non, no1 ? 1 : 0, tmp, words);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipMemcpyAsync(h, words, 8, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            (void)hipFree(tmp); (void)hipFree(words); (void)hipFree(d_base); (void)hipFree(marks);
            if (e != hipSuccess) { runs_free(r); return fail(c, PD_EHIP, std::string("pd_decode_end: ") + hipGetErrorString(e)); }
            r->n_long = (uint32_t)std::min<uint64_t>(n_long + h[1], 0xFFFFFFFFull);
            c->dec_runs = r;
            Pending p{nullptr, r->n, 0u, -1};
            p.cr = r;
            c->pend.push_back(p);
            return PD_OK;
        }
        // not usable as a compact sample after all (the records are not in the order the header promised, more than 2^32 runs):
        // back to 12-byte arrays, which take the general paths below
        nf = x.n_s; no = x.n_o; nfar = 0;
        if (nf && c->n_cells >= (1ull << 32)) {
            // 32 bits of a flat begin name a cell only below 2^32 cells; above, it takes the sample's own bucket index to say which contig a
            // run lies in, and that index is exactly what an unordered stream does not have.  (The executable never gets here: it gives a
            // file whose records are not in the promised order to the host readers before anything is counted.)
            (void)hipDeviceSynchronize(); c8_drop(c);
            return fail(c, PD_ESTATE, "pd_decode_end: the records of this compact session are not in coordinate order (or are more than 2^32 - 256 runs) on a genome of 2^32 cells or more: "
                                      "decode it again without PD_DECODE_COMPACT");
        }
        if ((nf && hipMalloc(&c->run_first, (size_t)nf * sizeof(pd_iv)) != hipSuccess) || (no && hipMalloc(&c->run_other, (size_t)no * sizeof(pd_iv)) != hipSuccess)) {
            c8_drop(c); return fail(c, PD_ENOMEM, "pd_decode_end: run array allocation failed"); }
        if (nf) launch_r8_to_iv(c->stream, x.r8(), nf, tab_of(c), c->run_first);
        if (no) HIPOK(c, hipMemcpyAsync(c->run_other, x.oth(), (size_t)no * sizeof(pd_iv), hipMemcpyDeviceToDevice, c->stream));
        HIPOK(c, hipGetLastError());
        HIPOK(c, hipStreamSynchronize(c->stream));
        c8_drop(c);
        int rc = PD_OK;
        const bool near_ok = ok_order && span <= (1u << 14);
        if (nf) rc = scatter_device(c, c->run_first, (size_t)nf, ok_order ? (PD_PUSH_SORTED | PD_PUSH_MORE) : PD_PUSH_DEFAULT, -1, nullptr);
        if (rc == PD_OK && no) rc = scatter_device(c, c->run_other, (size_t)no, near_ok ? (PD_PUSH_SORTED | PD_PUSH_MORE | PD_PUSH_DISORDER(span + 1)) : PD_PUSH_DEFAULT, -1, nullptr);
        return rc;
    }
    if ((nf && hipMalloc(&c->run_first, (size_t)nf * sizeof(pd_iv)) != hipSuccess) || (no && hipMalloc(&c->run_other, (size_t)no * sizeof(pd_iv)) != hipSuccess) ||
        (nfar && hipMalloc(&c->run_far, (size_t)nfar * sizeof(pd_iv)) != hipSuccess)) { drop(); return fail(c, PD_ENOMEM, "pd_decode_end: run array allocation failed"); }
    uint64_t of = 0, oo = 0, ofar = 0;
    for (auto &r : segs) {
        if (r.n_first) HIPOK(c, hipMemcpyAsync(c->run_first + of, r.first, (size_t)r.n_first * sizeof(pd_iv), hipMemcpyDeviceToDevice, c->stream));
        if (r.n_other) HIPOK(c, hipMemcpyAsync(c->run_other + oo, r.other, (size_t)r.n_other * sizeof(pd_iv), hipMemcpyDeviceToDevice, c->stream));
        if (r.n_far) HIPOK(c, hipMemcpyAsync(c->run_far + ofar, r.far, (size_t)r.n_far * sizeof(pd_iv), hipMemcpyDeviceToDevice, c->stream));
        of += r.n_first; oo += r.n_other; ofar += r.n_far;
    }
    HIPOK(c, hipStreamSynchronize(c->stream));
    drop();
    // the sample, deferred, as up to three streams in file order: every read's first run (position sorted for a
    // coordinate-sorted file: exact tile bounds), its later runs that begin within NEAR_SPAN bases of its start (they trail
    // the sorted order by at most that), and the few that follow a long gap (N operations: they trail by up to `span`).
    // An unsorted file, or gaps of more than a few tiles, take the atomic path.
    bool sorted = c->dec_cfg.sorted != 0;
    if (sorted) {                                        // ... and only if the records really are in that order (the header may lie)
        uint64_t prev = 0; bool have = false;
        for (auto &r : segs) {
            if (!r.n_first) continue;
            if (r.unsorted || (have && r.first_key < prev)) { sorted = false; break; }
            prev = r.last_key; have = true;
        }
    }
    if (getenv("PANDEPTH_TIMING"))
        fprintf(stderr, "[timing]   decode entry points, thread-seconds: slot wait %.3f, pinned alloc %.3f, device buffers %.3f, wait H2D+inflate+walk %.3f, "
                        "host chain check %.3f, run arrays %.3f, wait emit %.3f, first HIP call of the feeder threads %.3f; %zu batches; runs: %llu first, %llu near, %llu far (span %u); "
                        "chain confirmed on the device for %llu batches, by the host for %llu; segments the device walked again: %llu\n", g_dec_us[0] / 1e6,
                g_dec_us[1] / 1e6, g_dec_us[2] / 1e6, g_dec_us[3] / 1e6, g_dec_us[4] / 1e6, g_dec_us[5] / 1e6, g_dec_us[6] / 1e6, g_dec_us[7] / 1e6, segs.size(),
                (unsigned long long)nf, (unsigned long long)no, (unsigned long long)nfar, span, (unsigned long long)c->dec_n_fast.load(), (unsigned long long)c->dec_n_slow.load(), (unsigned long long)c->dec_n_redo.load());
    int rc = PD_OK;
    // disorder of a stream = how far its runs may trail the sorted order: the near stream by near_span (when the split is on,
    // otherwise by the longest gap seen, like the far stream)
    const uint32_t near_dis = nfar ? (c->dec_near_span < span ? c->dec_near_span : span) : span;
    const bool near_sorted = sorted && near_dis <= (1u << 14), far_sorted = sorted && span <= (1u << 14);
    if (nf && sorted && (c->dec_cfg.flags & PD_DECODE_COMPACT) && c->pend.empty() && nf + no + nfar <= DEV_BATCH_MAX) {
        // the whole-contig modes: the sample stays as ONE compact sample (8 bytes per run, grouped by 512-cell bucket: the first
        // runs keep their order — checked again —, the later runs of multi-run reads are dropped into their buckets), the
        // 12-byte arrays go
        pd_runs *r = nullptr;
        const pd_iv *o[2] = {c->run_other, c->run_far}; const size_t non[2] = {(size_t)no, (size_t)nfar};
        if (runs_make(c, c->run_first, (size_t)nf, o, non, 2, &r) == PD_OK) {
            for (pd_iv **q : {&c->run_first, &c->run_other, &c->run_far}) if (*q) { (void)hipFree(*q); *q = nullptr; }
            c->dec_runs = r;
            Pending p{nullptr, r->n, 0u, -1};
            p.cr = r;
            c->pend.push_back(p);
            return PD_OK;
        }
    }
    if (nf) rc = scatter_device(c, c->run_first, (size_t)nf, sorted ? (PD_PUSH_SORTED | PD_PUSH_MORE) : PD_PUSH_DEFAULT, -1, nullptr);
    if (rc == PD_OK && no) rc = scatter_device(c, c->run_other, (size_t)no, near_sorted ? (PD_PUSH_SORTED | PD_PUSH_MORE | PD_PUSH_DISORDER(near_dis + 1)) : PD_PUSH_DEFAULT, -1, nullptr);
    if (rc == PD_OK && nfar) rc = scatter_device(c, c->run_far, (size_t)nfar, far_sorted ? (PD_PUSH_SORTED | PD_PUSH_MORE | PD_PUSH_DISORDER(span + 1)) : PD_PUSH_DEFAULT, -1, nullptr);
    return rc;
}

int pd_decode_abort(pd_ctx *c)
{
    if (!c) return PD_EINVAL;
    if (c->dec_warm.joinable()) c->dec_warm.join();
    dec_drain(c, false);
    std::unique_lock<std::mutex> lk(c->dec_mu);
    c->dec_cv.wait(lk, [&] { for (auto &x : c->dec) if (x.busy) return false; return true; });
    c->dec_open = false;
    (void)hipSetDevice(c->device);
    auto in_arena = [&](const void *p) { return c->arena && (const uint8_t *)p >= c->arena && (const uint8_t *)p < c->arena + c->arena_cap; };
    for (auto &r : c->run_segs) for (pd_iv *q : {r.first, r.other, r.far}) if (q && !in_arena(q)) (void)hipFree(q);
    c->run_segs.clear();
    { std::lock_guard<std::mutex> l8(c->c8.mu); if (c->c8.on || c->c8.base) { (void)hipDeviceSynchronize(); c8_drop(c); } }
    return PD_OK;
}

// The synchronous single-batch form (round 1's entry point, kept for its callers): one batch through the pipeline
// above, its runs scattered at once (first runs: owner tiles; the others: atomics).
int pd_push_bgzf_units(pd_ctx *c, const void *blob, size_t n_bytes, const pd_bgzf_block *blocks, uint32_t n_blocks,
                       const pd_bgzf_unit *units, uint32_t n_units, uint64_t inflated_bytes, uint32_t flag_mask,
                       int32_t min_mapq, int32_t *unit_status, uint64_t *n_records)
{
    if (!c || !blob || !blocks || !units || !unit_status) return PD_EINVAL;
    if (n_records) *n_records = 0;
    if (n_units == 0 || n_blocks == 0) return PD_OK;
    pd_decode_cfg cfg{}; cfg.flag_mask = flag_mask; cfg.min_mapq = min_mapq; cfg.sorted = 1;
    int rc = pd_decode_begin(c, &cfg);
    if (rc) return rc;
    // the session this call opens is closed on every path out of it (its batch is taken out of the list below, so the
    // abort drops nothing that was counted)
    struct Close { pd_ctx *c; ~Close() { (void)pd_decode_abort(c); } } close_session{c};
    void *hb = nullptr;
    if ((rc = pd_decode_acquire(c, n_bytes, &hb))) return rc;
    memcpy(hb, blob, n_bytes);
    std::vector<pd_decode_unit> du(n_units);
    for (uint32_t u = 0; u < n_units; ++u) du[u] = pd_decode_unit{units[u].start, units[u].stop, units[u].avail, units[u].first_block, units[u].n_blocks, 0, 0};
    static std::atomic<uint64_t> key{0};
    pd_decode_batch bt{}; bt.host_buf = hb; bt.n_bytes = n_bytes; bt.blocks = blocks; bt.n_blocks = n_blocks; bt.inflated_bytes = inflated_bytes;
    bt.units = du.data(); bt.n_units = n_units; bt.order = ((uint64_t)1 << 63) + key.fetch_add(1);
    pd_decode_result res;
    if ((rc = pd_decode_submit(c, &bt, unit_status, &res))) return rc;
    for (auto &sl : c->dec) if (sl.st) (void)hipStreamSynchronize(sl.st);           // (the batch's runs are scattered from another stream below)
    if (n_records) *n_records = res.n_reads;
    pd_ctx::RunSeg mine{0, nullptr, 0, nullptr, 0, nullptr, 0, 0};
    {
        std::lock_guard<std::mutex> lk(c->dec_mu);
        for (size_t i = 0; i < c->run_segs.size(); ++i)
            if (c->run_segs[i].order == bt.order) { mine = c->run_segs[i]; c->run_segs.erase(c->run_segs.begin() + (long)i); break; }
    }
    if (mine.n_first + mine.n_other + mine.n_far) {
        std::unique_lock<std::mutex> lk(c->mu);
        if (int rs = need_state(c, 0, "pd_push_bgzf_units")) return rs;
        HIPOK(c, hipSetDevice(c->device));
        if (mine.n_first) { rc = scatter_device(c, mine.first, (size_t)mine.n_first, PD_PUSH_SORTED, -1, nullptr); if (rc) return rc; }
        if (mine.n_other) { rc = scatter_device(c, mine.other, (size_t)mine.n_other, PD_PUSH_DEFAULT, -1, nullptr); if (rc) return rc; }
        if (mine.n_far) { rc = scatter_device(c, mine.far, (size_t)mine.n_far, PD_PUSH_DEFAULT, -1, nullptr); if (rc) return rc; }
        HIPOK(c, hipStreamSynchronize(c->stream));
        const auto ina = [&](const void *p) { return c->arena && (const uint8_t *)p >= c->arena && (const uint8_t *)p < c->arena + c->arena_cap; };
        for (pd_iv *q : {mine.first, mine.other, mine.far}) if (q && !ina(q)) (void)hipFree(q);
    }
    return PD_OK;
}

} // extern "C"

extern "C" {

int pd_device_count(int *n)
{
    if (!n) return PD_EINVAL;
    int k = 0;
    if (hipGetDeviceCount(&k) != hipSuccess) k = 0;
    *n = k;
    return k > 0 ? PD_OK : PD_ENODEV;
}

int pd_accumulate_from(pd_ctx *dst, pd_ctx *src)
{
    if (!dst || !src || dst == src) return PD_EINVAL;
    // lock both contexts in address order
    std::unique_lock<std::mutex> l1(dst < src ? dst->mu : src->mu), l2(dst < src ? src->mu : dst->mu);
    if (dst->n_words != src->n_words || dst->n_contigs != src->n_contigs || dst->len != src->len)
        return fail(dst, PD_EINVAL, "pd_accumulate_from: the contexts describe different contigs");
    if (dst->state != 0 || src->state != 0) return fail(dst, PD_ESTATE, "pd_accumulate_from: both contexts must be accumulating");
    HIPOK(src, hipSetDevice(src->device));
    int rc = flush_pending(src);
    if (rc) { dst->err = src->err; return rc; }
    rc = check_words(src);
    if (rc) { dst->err = src->err; return rc; }
    if (dst->device != src->device) {
        HIPOK(dst, hipSetDevice(dst->device));
        int can = 0;
        (void)hipDeviceCanAccessPeer(&can, dst->device, src->device);
        if (can) (void)hipDeviceEnablePeerAccess(src->device, 0);    // already enabled is fine
        (void)hipGetLastError();
        HIPOK(src, hipSetDevice(src->device));
    }
    // Packed transport (default): the source packs its cells to nibbles (d + 8) + an exception list
    // on its own GPU — 1.5 GB instead of 12 GB over the link for a 3 Gb genome; never-written
    // half-tiles need no fill.  More than EXC_CAP cells outside [-8, 7]: the int32 path below.
    constexpr uint32_t EXC_CAP = 1u << 20;
    const size_t img_bytes = dst->n_cells / 2, exc_bytes = (size_t)EXC_CAP * sizeof(pd_exc);
    bool packed = dst->accumulate_packed;
    uint32_t n_exc = 0;
    if (packed) {
        rc = ensure_scratch(src, 16 + exc_bytes + img_bytes);
        if (rc) { dst->err = src->err; return rc; }
        unsigned char *ss = (unsigned char *)src->scratch;
        HIPOK(src, hipMemsetAsync(ss, 0, 16, src->stream));
        { ProfScope ps(src, "export_i4");
          launch_export_i4(src->stream, src->buf, src->hstate, ss + 16 + exc_bytes, src->n_cells, (pd_exc *)(ss + 16), EXC_CAP,
                           (uint32_t *)ss); }
        HIPOK(src, hipMemcpyAsync(&n_exc, ss, 4, hipMemcpyDeviceToHost, src->stream));
        HIPOK(src, hipStreamSynchronize(src->stream));
        if (n_exc > EXC_CAP) packed = false;
    }
    if (!packed) {
        // int32 transport: materialise the source (zeros where nothing was written)
        rc = ensure_all_valid(src);
        if (rc) { dst->err = src->err; return rc; }
    }
    if (src->copy_stream) HIPOK(src, hipStreamSynchronize(src->copy_stream));
    HIPOK(src, hipStreamSynchronize(src->stream));
    HIPOK(dst, hipSetDevice(dst->device));
    rc = flush_pending(dst);
    if (rc) return rc;
    rc = ensure_all_valid(dst);
    if (rc) return rc;
    dst->pristine = false;
    const size_t CH = (size_t)64 << 20;                               // words per chunk (256 MiB)
    if (packed) {
        const unsigned char *ss = (const unsigned char *)src->scratch;
        const size_t CHB = CH * 4;                                    // image bytes per chunk = 512 Mi cells
        rc = ensure_scratch(dst, CHB + (size_t)n_exc * sizeof(pd_exc) + 64);
        if (rc) return rc;
        unsigned char *ds = (unsigned char *)dst->scratch;
        pd_exc *d_exc = (pd_exc *)(ds + CHB);
        if (n_exc) HIPOK(dst, hipMemcpyPeerAsync(d_exc, dst->device, ss + 16, src->device, (size_t)n_exc * sizeof(pd_exc), dst->stream));
        for (size_t o = 0; o < img_bytes; o += CHB) {
            const size_t n = img_bytes - o < CHB ? img_bytes - o : CHB;
            HIPOK(dst, hipMemcpyPeerAsync(ds, dst->device, ss + 16 + exc_bytes + o, src->device, n, dst->stream));
            ProfScope ps(dst, "accumulate_from");
            const bool last = o + CHB >= img_bytes;
            launch_add_i4(dst->stream, dst->buf + o * 2, ds, (uint32_t)(n / (PD_TILE / 2)), d_exc, last ? n_exc : 0, dst->n_cells, dst->buf);
        }
        // the int32 tile sums travel as they are (4 B per 8192 cells)
        const size_t n_sum_words = dst->n_words - dst->n_cells;
        HIPOK(dst, hipMemcpyPeerAsync(ds, dst->device, src->sums, src->device, n_sum_words * 4, dst->stream));
        launch_add_i32(dst->stream, dst->sums, (const int *)ds, n_sum_words);
    } else {
        rc = ensure_scratch(dst, CH * 4);
        if (rc) return rc;
        for (size_t o = 0; o < dst->n_words; o += CH) {
            const size_t n = dst->n_words - o < CH ? dst->n_words - o : CH;
            HIPOK(dst, hipMemcpyPeerAsync(dst->scratch, dst->device, src->buf + o, src->device, n * 4, dst->stream));
            ProfScope ps(dst, "accumulate_from");
            launch_add_i32(dst->stream, dst->buf + o, (const int *)dst->scratch, n);
        }
    }
    HIPOK(dst, hipGetLastError());
    HIPOK(dst, hipStreamSynchronize(dst->stream));
    return PD_OK;
}

int pd_device_layout(pd_ctx *c, uint64_t *n_cells, uint64_t *n_tile_sums)
{
    if (!c) return PD_EINVAL;
    if (n_cells) *n_cells = c->n_cells;
    if (n_tile_sums) *n_tile_sums = c->n_words - c->n_cells;
    return PD_OK;
}

int pd_export_i8(pd_ctx *c, int threshold, void *dev_i8, pd_exc *dev_exc, uint32_t exc_cap, uint32_t *dev_count)
{
    if (!c || !dev_i8 || !dev_count || threshold < 1 || threshold > 127 || (exc_cap && !dev_exc)) return PD_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    if (int rs = need_state(c, 0, "pd_export_i8")) return rs;
    HIPOK(c, hipSetDevice(c->device));
    int rc = flush_pending(c);
    if (rc) return rc;
    HIPOK(c, hipMemsetAsync(dev_count, 0, 4, c->stream));
    { ProfScope ps(c, "export_i8");
      launch_export_i8(c->stream, c->buf, c->hstate, dev_i8, c->n_cells, threshold, dev_exc, exc_cap, dev_count); }
    HIPOK(c, hipGetLastError());
    return PD_OK;
}

int pd_import_i8(pd_ctx *c, const void *dev_i8, int bias, const pd_exc *dev_exc, uint64_t n_exc)
{
    if (!c || !dev_i8 || bias < 0 || bias > 255 || (n_exc && !dev_exc)) return PD_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    if (int rs = need_state(c, 0, "pd_import_i8")) return rs;
    HIPOK(c, hipSetDevice(c->device));
    int rc = flush_pending(c);
    if (rc) return rc;
    c->pristine = false;
    { ProfScope ps(c, "import_i8");
      launch_import_i8(c->stream, dev_i8, c->buf, c->n_cells, bias, dev_exc, n_exc); }
    HIPOK(c, hipGetLastError());
    HIPOK(c, hipMemsetAsync(c->hstate, 1, c->n_half, c->stream));      // every cell was just written
    c->all_valid_host = true;
    launch_mark_all_valid(c->stream, c->chk);
    return PD_OK;
}

static_assert(sizeof(TilePart) == PD_TILE_PARTIAL_BYTES, "pd_gather_windows' partial layout");
static_assert(PD_TILE == PD_TILE_CELLS, "tile size in the public header");

// pd_export_i4 without the arrays ("direct_windows", sample entirely deferred, context pristine): the tile windows leave
// as the 4-bit image straight from LDS (k_direct_tiles<DirectExport>), the tile sums are written beside it.  *done = false
// (tile sums re-zeroed, batches still pending) when a tile was too heavy, a run too long or a batch not sorted.
static int direct_export(pd_ctx *c, void *dev_i4, pd_exc *dev_exc, uint32_t exc_cap, uint32_t *dev_count, bool *done)
{
    *done = false;
    HIPOK(c, hipMemsetAsync(dev_count, 0, 4, c->stream));
    HIPOK(c, hipMemsetAsync(c->direct_words, 0, 64, c->stream));
    PendSet ps{};
    ps.nb = (int)c->pend.size(); ps.lmax = c->lmax;
    uint64_t all = 0;
    const bool c8 = ps.nb == 1 && c->pend[0].cr && !c->pend[0].iv && !c->pend[0].cr->n_long;
    if (!c8) for (auto &p : c->pend) { const int re = expand_compact(c, p); if (re) return re; }
    for (int b = 0; b < ps.nb; ++b) {
        const Pending &p = c->pend[b];
        all += p.n;
        if (c8) break;
        ps.b[b] = PendBatch{p.iv, c->ub_a[b], c->cand_lo[b], c->desc + b, p.n, 0};
        ProfScope sc(c, "scatter_index");
        launch_scatter_index(c->stream, p.iv, p.n, tab_of(c), c->lmax, p.disorder, c->sample < 256 ? 256 : c->sample, c->ub_a[b],
                             c->cand_lo[b], (uint32_t)c->n_tiles, PD_TILE, c->desc + b);
    }
    unsigned grid = c->grid_tiles;
    if (!grid) {
        uint64_t g = all / 256;
        if (g < (uint64_t)c->n_cu * 4) g = (uint64_t)c->n_cu * 4;
        if (g > 65536) g = 65536;
        grid = (unsigned)g;
    }
    if (grid > c->n_tiles) grid = (unsigned)c->n_tiles;
    { ProfScope sc(c, "direct_export");
      if (c8) launch_direct_c8_export(c->stream, c->pend[0].cr->view(), tab_of(c), c->d_tile_contig, (uint32_t)c->n_tiles, dev_i4, dev_exc, exc_cap,
                                      dev_count, c->sums, c->direct_words + 16, c->direct_words + 2, grid);
      else launch_direct_export(c->stream, ps, tab_of(c), c->d_tile_contig, (uint32_t)c->n_tiles, dev_i4, dev_exc, exc_cap, dev_count,
                                c->sums, c->direct_words, c->direct_words + 1, c->direct_words + 16, c->direct_words + 2, grid); }
    HIPOK(c, hipGetLastError());
    uint32_t words[2] = {0, 0}, n_exc = 0;
    HIPOK(c, hipMemcpyAsync(words, c->direct_words, 8, hipMemcpyDeviceToHost, c->stream));
    HIPOK(c, hipMemcpyAsync(&n_exc, dev_count, 4, hipMemcpyDeviceToHost, c->stream));
    HIPOK(c, hipStreamSynchronize(c->stream));
    // more cells outside the 4-bit range than the caller's exception block holds (amplicon / very deep data): the image is
    // not usable, and the sample must stay whole for whatever the caller does instead — it is materialised below like any
    // other declined export, the arrays keep it
    if (n_exc > exc_cap) words[1] = 1;
    if (words[1]) {
        HIPOK(c, hipMemsetAsync(c->sums, 0, (c->n_words - c->n_cells) * 4, c->stream));      // the kernel wrote them
        c->sums_stale = false;
        char m[160];
        snprintf(m, sizeof m, "direct export declined (heavy tiles / long runs: %u; cells outside the 4-bit range: %u); the materialising path was used", words[0], n_exc);
        c->err = m;                                          // informational: the call still succeeds
        return PD_OK;
    }
    // The sample STAYS deferred: an export reads it.  (A caller that finds out later that the image is of no use — another
    // rank's sample did not fit its exception block, pd_sliced_sum_finish -> PD_ERANGE — still has every context's sample and
    // adds them up the general way.)  The tile sums now hold this sample's sums although the arrays hold nothing:
    // flush_pending zeroes them before it scatters, pd_reset forgets them with everything else.
    c->sums_stale = true;
    *done = true;
    return PD_OK;
}

int pd_export_i4(pd_ctx *c, void *dev_i4, pd_exc *dev_exc, uint32_t exc_cap, uint32_t *dev_count)
{
    if (!c || !dev_i4 || !dev_count || (exc_cap && !dev_exc)) return PD_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    if (int rs = need_state(c, 0, "pd_export_i4")) return rs;
    HIPOK(c, hipSetDevice(c->device));
    if (c->direct_windows && c->pristine && !c->pend.empty() && c->stile == PD_TILE) {
        bool done = false;
        const int rd = direct_export(c, dev_i4, dev_exc, exc_cap, dev_count, &done);
        if (rd || done) return rd;
    }
    int rc = flush_pending(c);
    if (rc) return rc;
    HIPOK(c, hipMemsetAsync(dev_count, 0, 4, c->stream));
    { ProfScope ps(c, "export_i4");
      launch_export_i4(c->stream, c->buf, c->hstate, dev_i4, c->n_cells, dev_exc, exc_cap, dev_count); }
    HIPOK(c, hipGetLastError());
    return PD_OK;
}

int pd_slice_sweep_i4(pd_ctx *c, const void *dev_parts, uint32_t n_parts, uint64_t part_stride, uint64_t tile_first,
                      uint64_t tile_count, const int32_t *dev_tile_sums, const pd_exc *dev_exc, uint64_t exc_stride,
                      const int32_t *dev_exc_counts, uint32_t w, uint32_t min_dep, unsigned wrap_bits, void *dev_partials)
{
    if (!c || !dev_parts || !n_parts || !dev_tile_sums || !dev_partials || wrap_bits > 32) return PD_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    if (w < PD_TILE) return fail(c, PD_EINVAL, "pd_slice_sweep_i4: windows narrower than a tile are not sliced");
    if (tile_first > c->n_tiles || tile_count > c->n_tiles - tile_first)
        return fail(c, PD_EINVAL, "pd_slice_sweep_i4: tile range outside the buffer");
    if (n_parts > 1 && part_stride < tile_count * (PD_TILE / 2)) return fail(c, PD_EINVAL, "pd_slice_sweep_i4: parts overlap");
    if (((uintptr_t)dev_parts | part_stride) & 15) return fail(c, PD_EINVAL, "pd_slice_sweep_i4: parts must be 16-byte aligned");
    HIPOK(c, hipSetDevice(c->device));
    const uint32_t mask = (wrap_bits == 0 || wrap_bits == 32) ? 0xFFFFFFFFu : ((1u << wrap_bits) - 1u);
    { ProfScope ps(c, "tile_carry"); launch_tile_carry(c->stream, dev_tile_sums, c->bsum, c->carry, (uint32_t)c->n_tiles); }
    { ProfScope ps(c, "slice_sweep");
      TileMap tm{c->d_tile_contig, c->d_off, c->d_len, nullptr};
      launch_sweep_i4(c->stream, dev_parts, n_parts, part_stride, (uint32_t)tile_first, (uint32_t)tile_count, dev_exc, exc_stride,
                      dev_exc_counts, c->slice_flags, slice_flag_bytes(c->n_tiles), c->carry, mask, tm, w, min_dep, (TilePart *)dev_partials, nullptr); }
    HIPOK(c, hipGetLastError());
    return PD_OK;
}

int pd_gather_windows(pd_ctx *c, const void *dev_partials, uint32_t w, uint32_t *cover, uint64_t *sum)
{
    if (!c || !dev_partials || !cover || !sum) return PD_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    if (w < PD_TILE) return fail(c, PD_EINVAL, "pd_gather_windows: windows narrower than a tile are not sliced");
    HIPOK(c, hipSetDevice(c->device));
    std::vector<uint64_t> wo((size_t)c->n_contigs + 1);
    pd_window_layout(c, w, wo.data());
    const uint64_t nw = wo[c->n_contigs];
    const size_t b_off = ((size_t)c->n_contigs + 1) * 8;
    const size_t b_sum = (size_t)nw * 8, b_cov = ((size_t)nw * 4 + 15) / 16 * 16;
    int rc = ensure_scratch(c, b_off + b_sum + b_cov + 64);
    if (rc) return rc;
    unsigned char *s = (unsigned char *)c->scratch;
    uint64_t *d_wo = (uint64_t *)s;
    unsigned long long *d_sum = (unsigned long long *)(s + b_off);
    uint32_t *d_cov = (uint32_t *)(s + b_off + b_sum);
    HIPOK(c, hipMemcpyAsync(d_wo, wo.data(), b_off, hipMemcpyHostToDevice, c->stream));
    HIPOK(c, hipStreamSynchronize(c->stream));          // wo is a local
    { ProfScope ps(c, "gather_windows");
      TileMap tm{c->d_tile_contig, c->d_off, c->d_len, d_wo};
      launch_window_gather(c->stream, (const TilePart *)dev_partials, tm, c->n_contigs, w, nw, d_cov, d_sum); }
    HIPOK(c, hipGetLastError());
    HIPOK(c, hipMemcpyAsync(sum, d_sum, b_sum, hipMemcpyDeviceToHost, c->stream));
    HIPOK(c, hipMemcpyAsync(cover, d_cov, (size_t)nw * 4, hipMemcpyDeviceToHost, c->stream));
    HIPOK(c, hipStreamSynchronize(c->stream));
    return PD_OK;
}

// zlib's level-6 LZ77 parse of a host text on the device (pd_deflate.hip / pd_lz77.h): positions sorted by (hash, position),
// one wave per chunk, the symbols gathered and copied back.  Buffers live for the call.
// A text stream that lives in HBM (include/pandepth_amd.h: pd_text_*): a ring of segments, each one append's bytes, contiguous.
struct pd_text {
    pd_ctx *ctx = nullptr;
    uint8_t *ring = nullptr; size_t cap = 0;
    struct Seg { uint64_t off; size_t phys, len; };
    std::deque<Seg> segs;
    uint64_t tail_off = 0; size_t phys_tail = 0;
    void *scratch = nullptr; size_t scratch_bytes = 0;         // name | block counts | block offsets of an append
    std::mutex mu;
};

// The parse of pd_deflate_parse / pd_text_parse: the text comes from the host (`text`) or from a device stream (`tx`, bytes
// [tx_off, tx_off + n_text)); crc_out (optional): CRC-32 of every chunk's first crc_span bytes.
static int lz_run(pd_ctx *c, const void *text, pd_text *tx, uint64_t tx_off, size_t n_text, const pd_lz_chunk *chunks, uint32_t n_chunks,
                  uint32_t *syms, size_t syms_cap, uint64_t *sym_off, uint32_t *crc_out, uint64_t crc_span)
{
    // The call works on its own stream and its own buffers: the context's lock is held only where the context is touched (its
    // error text, the profile), so that the per-site writer's producer (pd_format_sites on the context's stream) is not kept
    // waiting for the time a round's parse takes.  Two calls run at a time, each in its own slot (buffers + stream).
    const unsigned lz_slot = c->lz_turn.fetch_add(1) & (c->lz_slots >= 4 ? 3u : 1u);
    pd_ctx::LzWork &w = c->lz[lz_slot];
    // "lz_mix": of the two calls a stream keeps in flight, the second parses with its text in memory — a workgroup of the LDS parse fills a CU's
    // LDS, so two LDS parses run one after the other, while a parse from memory shares the CUs with either kind
    const unsigned lz_group = c->lz_mix && (lz_slot & 1u) ? 0u : c->lz_group;
    std::lock_guard<std::mutex> lz_lock(w.mu);
    auto fail = [&](pd_ctx *cc, int code, const std::string &msg) { std::lock_guard<std::mutex> lk(cc->mu); cc->err = msg; return code; };
    if (n_text < 3 || n_text > 0xFFFFFF00ull - 64) return fail(c, PD_EINVAL, "pd_deflate_parse: between 3 and 2^32 - 320 bytes of text");
    uint64_t stride = 0;
    for (uint32_t k = 0; k < n_chunks; ++k) {
        const pd_lz_chunk &ch = chunks[k];
        if (ch.origin > ch.start || ch.start > ch.end || ch.end > n_text || ch.start - ch.origin > 32768)
            return fail(c, PD_EINVAL, "pd_deflate_parse: a chunk lies outside the text or has more than 32768 bytes of history");
        stride = std::max<uint64_t>(stride, ch.end - ch.start);
    }
    sym_off[0] = 0;
    if (!n_chunks) return PD_OK;
    stride += 8;
    if (hipSetDevice(c->device) != hipSuccess) return fail(c, PD_EHIP, "pd_deflate_parse: hipSetDevice failed");
    if (!w.st && hipStreamCreateWithFlags(&w.st, hipStreamNonBlocking) != hipSuccess) { w.st = nullptr; return fail(c, PD_EHIP, "pd_deflate_parse: stream creation failed"); }
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    const bool prof = c->prof;
    if (prof) for (auto &e : ev) (void)hipEventCreate(&e);
    const bool dbg = getenv("PD_LZ_DEBUG") != nullptr;
    double tm[8] = {}; int ti = 0;
    auto tick = [&]() { if (dbg && ti < 8) tm[ti++] = dec_now() * 1e-6; };
    tick();
    const uint32_t np = (uint32_t)(n_text - 2);
    const uint32_t n_blocks = (np + 2047) / 2048;
    // work buffers: kept from call to call (a round of 200 MB of text needs 5 GB of them; allocating and freeing them costs more
    // than the kernels), grown on demand, released by pd_destroy of the context that made them
    const size_t nh = (size_t)256 * n_blocks + 16;
    const size_t want[pd_ctx::LzWork::N] = {n_text + 64, (size_t)np * 8 + 64, (size_t)np * 8 + 64, nh * 4, (nh / 1024 + 8) * 4, (size_t)np * 4 + 64, (size_t)n_text * 4 + 64,
                                    ((size_t)32768 + 8) * 4, (size_t)n_chunks * 24, ((size_t)n_chunks + 1) * 8, (size_t)n_chunks * stride * 4, (size_t)n_chunks * 4 + 16, 0,
                                    (size_t)n_chunks * 4 + 16, (size_t)n_chunks * sizeof(pdk::LzGroup) + 16, (size_t)n_chunks * 4 + 16};
    for (int k = 0; k < pd_ctx::LzWork::N; ++k)
        if (!w.fit(k, want[k])) { (void)hipGetLastError(); return fail(c, PD_ENOMEM, "pd_deflate_parse: device allocation failed"); }
    uint8_t *d_text = (uint8_t *)w.p[0]; uint64_t *ka = (uint64_t *)w.p[1], *kb = (uint64_t *)w.p[2];
    uint32_t *hist = (uint32_t *)w.p[3], *scan_tmp = (uint32_t *)w.p[4], *S = (uint32_t *)w.p[5], *R = (uint32_t *)w.p[6], *bucket = (uint32_t *)w.p[7];
    uint64_t *d_chunks = (uint64_t *)w.p[8], *d_off = (uint64_t *)w.p[9]; uint32_t *d_syms = (uint32_t *)w.p[10], *d_cnt = (uint32_t *)w.p[11], *d_crc = (uint32_t *)w.p[13];
    auto cleanup = [&]() { for (auto &x : ev) if (x) { (void)hipEventDestroy(x); x = nullptr; } };
    // Consecutive chunks whose text — the first one's history up to the last one's end — fits a CU's LDS parse as one workgroup
    // (pd_deflate.hip: k_lz_parse_lds).  A chunk qualifies when its history is zlib's whole window or begins with the text: its
    // candidates (nearer than 32 506 bytes) then lie inside what the group holds.  The others parse with the text in memory.
    std::vector<pdk::LzGroup> groups;
    std::vector<uint32_t> loose;
    uint32_t group_waves = 0; size_t lds_bytes = 0;
    {
        int lds_max = 0;
        if (lz_group == 0 || hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, c->device) != hipSuccess) { (void)hipGetLastError(); lds_max = 0; }
        if (lds_max > (int)pdk::LZ_LDS_MAX) lds_max = (int)pdk::LZ_LDS_MAX;
        if (lds_max > 0 && !pdk::lz_parse_lds_ready((size_t)lds_max)) lds_max = 0;
        auto fits = [&](const pd_lz_chunk &ch) { return ch.origin == 0 || ch.start - ch.origin == 32768; };
        uint32_t k = 0;
        while (k < n_chunks) {
            const uint64_t base = chunks[k].origin;
            uint32_t count = 0; uint64_t hi = 0, len = 0;
            for (uint32_t j = k; lds_max > 0 && j < n_chunks && count < lz_group && fits(chunks[j]) && chunks[j].origin >= base; ++j) {
                const uint64_t nhi = std::max<uint64_t>(hi, chunks[j].end);
                const uint64_t nlen = std::min<uint64_t>(nhi + pdk::LZ_LDS_SLACK, (uint64_t)n_text + 32) - base;
                if ((base & 15) + nlen + 16 > (uint64_t)lds_max) break;
                hi = nhi; len = nlen; ++count;
            }
            if (!count) { loose.push_back(k); ++k; continue; }
            groups.push_back(pdk::LzGroup{k, count, base, len});
            group_waves = std::max(group_waves, count);
            lds_bytes = std::max<size_t>(lds_bytes, (size_t)(((base & 15) + len + 15) & ~(uint64_t)15));
            k += count;
        }
    }
    pdk::LzGroup *d_groups = (pdk::LzGroup *)w.p[14]; uint32_t *d_loose = (uint32_t *)w.p[15];
    tick();
    hipStream_t st = w.st;
    std::vector<uint32_t> counts(n_chunks);
    hipError_t e = hipMemsetAsync(d_text + n_text, 0, 64, st);
    if (tx) {
        // the stretch, segment by segment (the segments stay where they are until the caller releases them).  The stream's lock is
        // released before anything is reported: fail() takes the context's lock, and the append paths take the two in the other order.
        uint64_t got = 0;
        {
            std::lock_guard<std::mutex> tlk(tx->mu);
            for (const auto &sg : tx->segs) {
                const uint64_t lo = std::max<uint64_t>(sg.off, tx_off), hi = std::min<uint64_t>(sg.off + sg.len, tx_off + n_text);
                if (lo >= hi) continue;
                if (e == hipSuccess) e = hipMemcpyAsync(d_text + (lo - tx_off), tx->ring + sg.phys + (lo - sg.off), (size_t)(hi - lo), hipMemcpyDeviceToDevice, st);
                got += hi - lo;
            }
        }
        if (got != n_text) { (void)hipStreamSynchronize(st); cleanup(); return fail(c, PD_EINVAL, "pd_text_parse: the stretch is not (or no longer) in the stream"); }
    } else if (e == hipSuccess) e = hipMemcpyAsync(d_text, text, n_text, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_chunks, chunks, (size_t)n_chunks * 24, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && !groups.empty()) e = hipMemcpyAsync(d_groups, groups.data(), groups.size() * sizeof(pdk::LzGroup), hipMemcpyHostToDevice, st);
    if (e == hipSuccess && !loose.empty()) e = hipMemcpyAsync(d_loose, loose.data(), loose.size() * 4, hipMemcpyHostToDevice, st);
    static_assert(sizeof(pd_lz_chunk) == 24, "pd_lz_chunk layout");
    if (dbg && e == hipSuccess) e = hipStreamSynchronize(st);
    tick();
    if (e == hipSuccess) {
        if (prof) (void)hipEventRecord(ev[0], st);
        launch_lz_sort(st, d_text, np, ka, kb, hist, scan_tmp, S, R, bucket);
        if (prof) (void)hipEventRecord(ev[1], st);
        if (dbg) { (void)hipStreamSynchronize(st); tick(); }
        launch_lz_parse(st, d_text, n_text, S, R, bucket, d_chunks, d_groups, (uint32_t)groups.size(), group_waves, lds_bytes, d_loose, (uint32_t)loose.size(), d_syms, stride, d_cnt);
        if (prof) (void)hipEventRecord(ev[2], st);
        if (crc_out) launch_lz_crc(st, d_text, d_chunks, n_chunks, crc_span, d_crc);
        e = hipGetLastError();
    }
    if (e == hipSuccess && crc_out) e = hipMemcpyAsync(crc_out, d_crc, (size_t)n_chunks * 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(counts.data(), d_cnt, (size_t)n_chunks * 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { cleanup(); return fail(c, PD_EHIP, std::string("pd_deflate_parse: ") + hipGetErrorString(e)); }
    tick();
    uint64_t total = 0;
    for (uint32_t k = 0; k < n_chunks; ++k) {
        if (counts[k] == 0xFFFFFFFFu) { cleanup(); return fail(c, PD_EHIP, "pd_deflate_parse: a chunk's symbols did not fit its buffer"); }
        total += counts[k]; sym_off[k + 1] = total;
    }
    if (total > syms_cap) { cleanup(); return fail(c, PD_ERANGE, "pd_deflate_parse: the symbol buffer is too small"); }
    if (total) {
        if (!w.fit(12, total * 4 + 64)) { (void)hipGetLastError(); cleanup(); return fail(c, PD_ENOMEM, "pd_deflate_parse: device allocation failed"); }
        uint32_t *d_out = (uint32_t *)w.p[12];
        e = hipMemcpyAsync(d_off, sym_off, ((size_t)n_chunks + 1) * 8, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) { launch_lz_gather(st, d_syms, stride, d_off, n_chunks, d_out); e = hipGetLastError(); }
        if (dbg && e == hipSuccess) { e = hipStreamSynchronize(st); tick(); }
        // the symbols come back through a page-locked buffer of the slot, in pieces, and are copied on from there: a large copy
        // straight into the caller's pageable memory makes the runtime lock and unlock those pages around it
        const size_t PIECE = (size_t)8 << 20;                        // bytes
        if (!w.h_stage && hipHostMalloc(&w.h_stage, 2 * PIECE, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); w.h_stage = nullptr; }
        if (w.h_stage && getenv("PD_LZ_DIRECT_COPY") == nullptr) {
            if (!w.ev_stage[0]) { (void)hipEventCreateWithFlags(&w.ev_stage[0], hipEventDisableTiming); (void)hipEventCreateWithFlags(&w.ev_stage[1], hipEventDisableTiming); }
            const size_t bytes = total * 4;
            const size_t n_pieces = (bytes + PIECE - 1) / PIECE;
            // piece k is on the wire while piece k - 1 is copied out of the other half of the buffer
            for (size_t k = 0; k <= n_pieces && e == hipSuccess; ++k) {
                if (k < n_pieces) {
                    const size_t off = k * PIECE, n = std::min(PIECE, bytes - off);
                    e = hipMemcpyAsync((uint8_t *)w.h_stage + (k & 1) * PIECE, (const uint8_t *)d_out + off, n, hipMemcpyDeviceToHost, st);
                    if (e == hipSuccess) e = hipEventRecord(w.ev_stage[k & 1], st);
                }
                if (k > 0 && e == hipSuccess) {
                    const size_t off = (k - 1) * PIECE, n = std::min(PIECE, bytes - off);
                    e = hipEventSynchronize(w.ev_stage[(k - 1) & 1]);
                    if (e == hipSuccess) memcpy((uint8_t *)syms + off, (const uint8_t *)w.h_stage + ((k - 1) & 1) * PIECE, n);
                }
            }
        } else {
            if (e == hipSuccess) e = hipMemcpyAsync(syms, d_out, total * 4, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
        }
    }
    tick();
    if (dbg && ti >= 7)
        fprintf(stderr, "[lz] %zu groups of up to %u chunks with their text in LDS (%zu bytes a workgroup), %zu chunks with the text in memory\n", groups.size(), group_waves, lds_bytes, loose.size());
    if (dbg && ti >= 7)
        fprintf(stderr, "[lz] %.1f MB, %u chunks, %.1f M symbols: buffers %.4f, text to the device %.4f, sort %.4f, parse %.4f, gather %.4f, symbols back %.4f s\n", n_text / 1e6,
                n_chunks, total / 1e6, tm[1] - tm[0], tm[2] - tm[1], tm[3] - tm[2], tm[4] - tm[3], tm[5] - tm[4], tm[6] - tm[5]);
    if (prof) {
        float a = 0, b = 0;
        if (e == hipSuccess && hipEventElapsedTime(&a, ev[0], ev[1]) == hipSuccess && hipEventElapsedTime(&b, ev[1], ev[2]) == hipSuccess) {
            std::lock_guard<std::mutex> lk(c->mu);
            auto &s1 = c->prof_acc["lz_sort"]; s1.first += a; s1.second += 1;
            auto &s2 = c->prof_acc["lz_parse"]; s2.first += b; s2.second += 1;
        }
    }
    cleanup();
    if (e != hipSuccess) return fail(c, PD_EHIP, std::string("pd_deflate_parse: ") + hipGetErrorString(e));
    return PD_OK;
}

int pd_deflate_parse(pd_ctx *c, const void *text, size_t n_text, const pd_lz_chunk *chunks, uint32_t n_chunks,
                     uint32_t *syms, size_t syms_cap, uint64_t *sym_off)
{
    if (!c || !text || !chunks || !syms || !sym_off) return PD_EINVAL;
    return lz_run(c, text, nullptr, 0, n_text, chunks, n_chunks, syms, syms_cap, sym_off, nullptr, 0);
}

// (the callers hold the context's lock)
static int text_scratch(pd_text *t, size_t bytes)
{
    pd_ctx *c = t->ctx;
    if (t->scratch_bytes >= bytes) return PD_OK;
    if (t->scratch) { HIPOK(c, hipStreamSynchronize(c->stream)); HIPOK(c, hipFree(t->scratch)); t->scratch = nullptr; t->scratch_bytes = 0; }
    const size_t want = bytes * 2;
    if (hipMalloc(&t->scratch, want) != hipSuccess) { (void)hipGetLastError(); return fail(c, PD_ENOMEM, "pd_text: device allocation failed"); }
    t->scratch_bytes = want;
    return PD_OK;
}
// a place for `total` bytes in the ring: behind the last segment, or at the ring's start once that end is free again
static int text_place(pd_text *t, uint64_t total, size_t *phys, const char *who)
{
    pd_ctx *c = t->ctx;
    std::lock_guard<std::mutex> tlk(t->mu);
    if (total > t->cap) return fail(c, PD_EINVAL, std::string(who) + ": more bytes than the stream's capacity");
    if (t->segs.empty()) { t->phys_tail = 0; *phys = 0; return PD_OK; }
    const size_t head = t->segs.front().phys;
    if (t->phys_tail >= head) {                                // live bytes in [head, tail)
        if (t->phys_tail + total <= t->cap) { *phys = t->phys_tail; return PD_OK; }
        if (total < head) { *phys = 0; return PD_OK; }
    } else if (t->phys_tail + total < head) { *phys = t->phys_tail; return PD_OK; }   // wrapped: live bytes in [head, ...) and [0, tail)
    return fail(c, PD_ERANGE, std::string(who) + ": the stream is full (release what has been consumed)");
}
static void text_commit(pd_text *t, size_t phys, uint64_t total)
{
    std::lock_guard<std::mutex> tlk(t->mu);
    t->segs.push_back(pd_text::Seg{t->tail_off, phys, (size_t)total});
    t->tail_off += total; t->phys_tail = phys + (size_t)total;
}

// ---- a text stream in HBM: the per-site rows are formatted, parsed and check-summed where the cells are ----
int pd_text_open(pd_ctx *c, size_t capacity, pd_text **out)
{
    if (!c || !out || capacity < ((size_t)1 << 20)) return PD_EINVAL;
    *out = nullptr;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPOK(c, hipSetDevice(c->device));
    pd_text *t = new pd_text;
    t->ctx = c; t->cap = capacity;
    if (hipMalloc(&t->ring, capacity + 64) != hipSuccess) { (void)hipGetLastError(); delete t; return fail(c, PD_ENOMEM, "pd_text_open: device allocation failed"); }
    *out = t;
    return PD_OK;
}

int pd_text_close(pd_text *t)
{
    if (!t) return PD_OK;
    pd_ctx *c = t->ctx;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        (void)hipSetDevice(c->device);
        for (auto &w : c->lz) { std::lock_guard<std::mutex> g(w.mu); if (w.st) (void)hipStreamSynchronize(w.st); }
        (void)hipStreamSynchronize(c->stream);
        if (t->ring) (void)hipFree(t->ring);
        if (t->scratch) (void)hipFree(t->scratch);
    }
    delete t;
    return PD_OK;
}

int pd_text_append_sites(pd_text *t, int32_t tid, uint32_t beg, size_t n, const char *name, size_t name_len, uint64_t *n_bytes)
{
    if (!t || !n_bytes || (!name && name_len)) return PD_EINVAL;
    *n_bytes = 0;
    pd_ctx *c = t->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    if (int rs = need_state(c, 1, "pd_text_append_sites")) return rs;
    if (tid < 0 || tid >= c->n_contigs) return fail(c, PD_EINVAL, "pd_text_append_sites: contig id out of range");
    if ((uint64_t)beg + n > c->off[tid + 1] - c->off[tid]) return fail(c, PD_EINVAL, "pd_text_append_sites: range past the contig slot");
    if (name_len > 4096 || n > ((size_t)1 << 27)) return fail(c, PD_EINVAL, "pd_text_append_sites: at most 2^27 cells per call and 4096 bytes of name");
    if (n == 0) return PD_OK;
    HIPOK(c, hipSetDevice(c->device));
    const uint32_t nb = pdk::site_rows_blocks(n);
    const size_t b_name = (name_len + 15) / 16 * 16 + 16, b_cnt = ((size_t)nb * 4 + 15) / 16 * 16, b_off = ((size_t)nb + 1) * 8;
    if (int rs = text_scratch(t, b_name + b_cnt + b_off)) return rs;
    unsigned char *s = (unsigned char *)t->scratch;
    char *d_name = (char *)s; uint32_t *d_cnt = (uint32_t *)(s + b_name); uint64_t *d_off = (uint64_t *)(s + b_name + b_cnt);
    ProfScope ps(c, "format_sites");
    if (name_len) HIPOK(c, hipMemcpyAsync(d_name, name, name_len, hipMemcpyHostToDevice, c->stream));
    const uint32_t *depth = (const uint32_t *)(c->buf + c->off[tid] + beg);
    // pass 1: the rows' lengths and where each workgroup's rows begin
    pdk::launch_site_rows(c->stream, depth, beg, n, (uint32_t)name_len, d_name, d_cnt, d_off, nullptr, false);
    uint64_t total = 0;
    HIPOK(c, hipMemcpyAsync(&total, d_off + nb, 8, hipMemcpyDeviceToHost, c->stream));
    HIPOK(c, hipStreamSynchronize(c->stream));
    HIPOK(c, hipGetLastError());
    size_t phys = 0;
    if (int rp = text_place(t, total, &phys, "pd_text_append_sites")) return rp;
    // pass 2: the bytes
    pdk::launch_site_rows(c->stream, depth, beg, n, (uint32_t)name_len, d_name, d_cnt, d_off, (char *)t->ring + phys, true);
    HIPOK(c, hipGetLastError());
    HIPOK(c, hipStreamSynchronize(c->stream));
    text_commit(t, phys, total);
    *n_bytes = total;
    return PD_OK;
}

int pd_text_append_window_rows(pd_text *t, int32_t tid, uint32_t w, uint64_t row_first, size_t n_rows, const char *name, size_t name_len, uint64_t *n_bytes)
{
    if (!t || !n_bytes || (!name && name_len) || !w) return PD_EINVAL;
    *n_bytes = 0;
    pd_ctx *c = t->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->wk_valid || c->wk_w != w) return fail(c, PD_ESTATE, "pd_text_append_window_rows: the last window call (pd_scan_reduce_windows / pd_reduce_windows) was not for this width");
    if (tid < 0 || tid >= c->n_contigs) return fail(c, PD_EINVAL, "pd_text_append_window_rows: contig id out of range");
    const uint64_t rows_here = c->wk_woff[(size_t)tid + 1] - c->wk_woff[(size_t)tid];
    if (row_first > rows_here || n_rows > rows_here - row_first) return fail(c, PD_EINVAL, "pd_text_append_window_rows: rows beyond the contig's windows");
    if (name_len > 4096 || n_rows > ((size_t)1 << 27)) return fail(c, PD_EINVAL, "pd_text_append_window_rows: at most 2^27 rows per call and 4096 bytes of name");
    if (n_rows == 0) return PD_OK;
    HIPOK(c, hipSetDevice(c->device));
    const uint32_t nb = pdk::window_rows_blocks(n_rows);
    const size_t b_name = (name_len + 15) / 16 * 16 + 16, b_cnt = ((size_t)nb * 4 + 15) / 16 * 16, b_off = ((size_t)nb + 1) * 8;
    if (int rs = text_scratch(t, b_name + b_cnt + b_off)) return rs;
    unsigned char *s = (unsigned char *)t->scratch;
    char *d_name = (char *)s; uint32_t *d_cnt = (uint32_t *)(s + b_name); uint64_t *d_off = (uint64_t *)(s + b_name + b_cnt);
    ProfScope ps(c, "format_window_rows");
    if (name_len) HIPOK(c, hipMemcpyAsync(d_name, name, name_len, hipMemcpyHostToDevice, c->stream));
    const uint64_t g0 = c->wk_woff[(size_t)tid] + row_first;
    const unsigned long long *d_sum = (const unsigned long long *)c->wk + g0;
    const uint32_t *d_cov = (const uint32_t *)(c->wk + (size_t)c->wk_nw * 8) + g0;
    pdk::launch_window_rows(c->stream, d_cov, d_sum, row_first, n_rows, w, c->len[(size_t)tid], (uint32_t)name_len, d_name, d_cnt, d_off, nullptr, false);
    uint64_t total = 0;
    HIPOK(c, hipMemcpyAsync(&total, d_off + nb, 8, hipMemcpyDeviceToHost, c->stream));
    HIPOK(c, hipStreamSynchronize(c->stream));
    HIPOK(c, hipGetLastError());
    size_t phys = 0;
    if (int rp = text_place(t, total, &phys, "pd_text_append_window_rows")) return rp;
    pdk::launch_window_rows(c->stream, d_cov, d_sum, row_first, n_rows, w, c->len[(size_t)tid], (uint32_t)name_len, d_name, d_cnt, d_off, (char *)t->ring + phys, true);
    HIPOK(c, hipGetLastError());
    HIPOK(c, hipStreamSynchronize(c->stream));
    text_commit(t, phys, total);
    *n_bytes = total;
    return PD_OK;
}

int pd_text_append_bytes(pd_text *t, const void *bytes, size_t n)
{
    if (!t || (!bytes && n)) return PD_EINVAL;
    if (!n) return PD_OK;
    pd_ctx *c = t->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPOK(c, hipSetDevice(c->device));
    size_t phys = 0;
    if (int rp = text_place(t, n, &phys, "pd_text_append_bytes")) return rp;
    HIPOK(c, hipMemcpyAsync(t->ring + phys, bytes, n, hipMemcpyHostToDevice, c->stream));
    HIPOK(c, hipStreamSynchronize(c->stream));
    text_commit(t, phys, n);
    return PD_OK;
}

int pd_text_parse(pd_text *t, uint64_t off, size_t n_text, const pd_lz_chunk *chunks, uint32_t n_chunks, uint32_t *syms, size_t syms_cap,
                  uint64_t *sym_off, uint32_t *crc, uint64_t crc_span)
{
    if (!t || !chunks || !syms || !sym_off) return PD_EINVAL;
    return lz_run(t->ctx, nullptr, t, off, n_text, chunks, n_chunks, syms, syms_cap, sym_off, crc, crc_span);
}

int pd_text_read(pd_text *t, uint64_t off, size_t n, void *out)
{
    if (!t || (!out && n)) return PD_EINVAL;
    pd_ctx *c = t->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPOK(c, hipSetDevice(c->device));
    std::lock_guard<std::mutex> tlk(t->mu);
    uint64_t got = 0;
    for (const auto &sg : t->segs) {
        const uint64_t lo = std::max<uint64_t>(sg.off, off), hi = std::min<uint64_t>(sg.off + sg.len, off + n);
        if (lo >= hi) continue;
        HIPOK(c, hipMemcpyAsync((uint8_t *)out + (lo - off), t->ring + sg.phys + (lo - sg.off), (size_t)(hi - lo), hipMemcpyDeviceToHost, c->stream));
        got += hi - lo;
    }
    HIPOK(c, hipStreamSynchronize(c->stream));
    if (got != n) return fail(c, PD_EINVAL, "pd_text_read: the stretch is not (or no longer) in the stream");
    return PD_OK;
}

int pd_text_release(pd_text *t, uint64_t off)
{
    if (!t) return PD_EINVAL;
    std::lock_guard<std::mutex> tlk(t->mu);
    while (!t->segs.empty() && t->segs.front().off + t->segs.front().len <= off) t->segs.pop_front();
    return PD_OK;
}

int pd_host_register(pd_ctx *c, void *ptr, size_t bytes)
{
    if (!c || !ptr || !bytes) return PD_EINVAL;
    if (hipSetDevice(c->device) != hipSuccess || hipHostRegister(ptr, bytes, hipHostRegisterDefault) != hipSuccess) {
        (void)hipGetLastError();
        std::lock_guard<std::mutex> lk(c->mu);
        return fail(c, PD_EHIP, "pd_host_register: hipHostRegister failed");
    }
    return PD_OK;
}

int pd_host_unregister(pd_ctx *c, void *ptr)
{
    if (!c || !ptr) return PD_EINVAL;
    if (hipSetDevice(c->device) != hipSuccess || hipHostUnregister(ptr) != hipSuccess) {
        (void)hipGetLastError();
        std::lock_guard<std::mutex> lk(c->mu);
        return fail(c, PD_EHIP, "pd_host_unregister: hipHostUnregister failed");
    }
    return PD_OK;
}

void *pd_stream(pd_ctx *c) { return c ? (void *)c->stream : nullptr; }

int pd_synchronize(pd_ctx *c)
{
    if (!c) return PD_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPOK(c, hipSetDevice(c->device));
    // a whole deferred sample in an otherwise empty context stays deferred when the direct window path may still take it
    const bool keep_deferred = c->direct_windows && c->pristine && !c->pend.empty() && c->state == 0;
    int rc = keep_deferred ? PD_OK : flush_pending(c);
    if (rc) return rc;
    if (c->copy_stream) HIPOK(c, hipStreamSynchronize(c->copy_stream));
    HIPOK(c, hipStreamSynchronize(c->stream));
    return PD_OK;
}

int pd_profile(pd_ctx *c, int enable)
{
    if (!c) return PD_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPOK(c, hipSetDevice(c->device));
    int rc = prof_collect(c);
    c->prof_acc.clear();
    c->prof = enable != 0;
    return rc;
}

int pd_profile_get(pd_ctx *c, const char *name, double *ms, uint64_t *launches)
{
    if (!c || !name) return PD_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPOK(c, hipSetDevice(c->device));
    int rc = prof_collect(c);
    if (rc) return rc;
    auto it = c->prof_acc.find(name);
    if (ms) *ms = it == c->prof_acc.end() ? 0.0 : it->second.first;
    if (launches) *launches = it == c->prof_acc.end() ? 0 : it->second.second;
    return PD_OK;
}

} // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// Multi-sample sum over several GPUs with RCCL called from here (include/pandepth_amd.h: pd_comm_*, pd_sliced_window_sum):
// the C++ form of pandepth_amd/multi.py's SlicedSum, for the CLI's `#.list` mode and for any host that is not Python.
// ---------------------------------------------------------------------------------------------------------------
namespace { struct Rccl; }
struct pd_comm {
    pd_ctx *ctx = nullptr;
    const Rccl *tp = nullptr;                     // the transport's entry points: librccl's (between processes) or the in-process peer-copy one
    ncclComm_t nccl = nullptr;
    int rank = 0, world = 1;
    uint64_t n_tiles = 0, slice_tiles = 0, slice_bytes = 0, tile_first = 0, tile_count = 0, n_sums = 0;
    // two slots of exchange buffers: sample k+1 is scattered and packed while sample k's image is on the links
    struct Slot {
        uint8_t *send = nullptr, *recv = nullptr;
        int32_t *meta = nullptr;                  // tile sums | exception counts per rank
        pd_exc *exc = nullptr, *exc_all = nullptr;
        uint32_t *count = nullptr;
        hipEvent_t packed = nullptr, landed = nullptr;
        bool busy = false;
    } slot[2];
    uint8_t *part_mine = nullptr, *part_all = nullptr;
    int *slice_depth = nullptr;                   // the summed depth of this rank's slice (narrow windows, intervals), made on demand
    hipStream_t links = nullptr;                  // every RCCL call is issued on this stream, ordered against the context's by events
    hipEvent_t swept = nullptr, gathered = nullptr;
    std::string err;
};

namespace {
// RCCL is loaded on first use (dlopen): librccl.so is half a gigabyte of code objects that a single-GPU run never needs,
// and the executable's start-up time is part of its end-to-end figure.
struct Rccl {
    void *h = nullptr;
    decltype(&::ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&::ncclCommInitRank) CommInitRank = nullptr;
    decltype(&::ncclCommInitAll) CommInitAll = nullptr;
    decltype(&::ncclCommDestroy) CommDestroy = nullptr;
    decltype(&::ncclGroupStart) GroupStart = nullptr;
    decltype(&::ncclGroupEnd) GroupEnd = nullptr;
    decltype(&::ncclSend) Send = nullptr;
    decltype(&::ncclRecv) Recv = nullptr;
    decltype(&::ncclAllReduce) AllReduce = nullptr;
    decltype(&::ncclAllGather) AllGather = nullptr;
    decltype(&::ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false, alt = false;
};
Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // PANDEPTH_RCCL_LIB names another library with RCCL's entry points (tests: a loopback transport between contexts that
        // share one GPU, tests/harness/loopback_nccl.hip — the only way to run N > 1 ranks of this code on a 1-GPU box)
        if (const char *alt = getenv("PANDEPTH_RCCL_LIB")) { if (alt[0]) r.h = dlopen(alt, RTLD_NOW | RTLD_LOCAL); if (r.h) r.alt = true; }
        if (!r.h) for (const char *p : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) if ((r.h = dlopen(p, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!r.h) return;
#define PD_SYM(name) r.name = (decltype(r.name))dlsym(r.h, "nccl" #name)
        PD_SYM(GetUniqueId); PD_SYM(CommInitRank); PD_SYM(CommInitAll); PD_SYM(CommDestroy); PD_SYM(GroupStart); PD_SYM(GroupEnd);
        PD_SYM(Send); PD_SYM(Recv); PD_SYM(AllReduce); PD_SYM(AllGather); PD_SYM(GetErrorString);
#undef PD_SYM
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommInitAll && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv && r.AllReduce &&
               r.AllGather && r.GetErrorString;
    });
    return r;
}
// (RCCL 2.27 prints a version banner on stdout when the first communicator is made, whatever NCCL_DEBUG says.  The library leaves
// descriptor 1 alone — until round 6 it pointed it at /dev/null meanwhile, which in a multi-threaded process can eat somebody else's
// line; a caller whose stdout is a contract keeps its own lines on a descriptor of its own: host/pipeline.cpp, OwnStdout.)
//
// The in-process transport (pd_local_comm.h) behind the same table: pd_comm_init_local.
Rccl &local_tp()
{
    static Rccl r = [] {
        Rccl t;
        t.CommInitAll = pdlocal::CommInitAll; t.CommDestroy = pdlocal::CommDestroy; t.GroupStart = pdlocal::GroupStart; t.GroupEnd = pdlocal::GroupEnd;
        t.Send = pdlocal::Send; t.Recv = pdlocal::Recv; t.AllReduce = pdlocal::AllReduce; t.AllGather = pdlocal::AllGather; t.GetErrorString = pdlocal::GetErrorString;
        t.ok = true; t.alt = true;                 // (alt: ranks may share a device)
        return t;
    }();
    return r;
}
// Communicators made ahead of their contexts (pd_comm_preinit): ncclCommInitAll wants device numbers only, and a short-lived process
// wants librccl's load and bootstrap over BEFORE its contexts load code objects and its readers launch kernels (see pd_comm_preinit).
struct ParkedComms { std::mutex mu; std::vector<int> devs; std::vector<ncclComm_t> nc; } g_parked;
constexpr uint32_t COMM_EXC_BLOCK = 1u << 18;       // exceptions (cells outside the 4-bit range) per rank
constexpr size_t COMM_MSG_BYTES = (size_t)1 << 28;  // RCCL 2.26 delivers only the first half of a send/recv above 1 GiB: stay far below

int comm_fail(pd_comm *m, int code, const std::string &msg) { m->err = msg; if (m->ctx) { std::lock_guard<std::mutex> lk(m->ctx->mu); m->ctx->err = msg; } return code; }
#define NCCLOK(m, call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) return comm_fail(m, PD_EHIP, std::string(#call) + ": " + (m)->tp->GetErrorString(r_)); } while (0)
#define HIPCM(m, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return comm_fail(m, PD_EHIP, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)

// a slot's exchange buffers, on first use (the calling rank's thread; its device is current)
int slot_setup(pd_comm *m, pd_comm::Slot &s)
{
    if (s.send) return PD_OK;
    pd_ctx *c = m->ctx;
    const size_t W = (size_t)m->world, img = (size_t)(c->n_cells / 2), all = W * (size_t)m->slice_bytes + 256;
    // one rank: nothing is received — the "received" slices are the image itself
    if (hipMalloc(&s.send, all) != hipSuccess || (W > 1 && hipMalloc(&s.recv, all) != hipSuccess) ||
        hipMalloc(&s.meta, (m->n_sums + W + 16) * 4) != hipSuccess || hipMalloc(&s.exc, (size_t)COMM_EXC_BLOCK * sizeof(pd_exc)) != hipSuccess ||
        hipMalloc(&s.exc_all, W * COMM_EXC_BLOCK * sizeof(pd_exc)) != hipSuccess || hipMalloc(&s.count, 64) != hipSuccess) {
        (void)hipGetLastError();
        return comm_fail(m, PD_ENOMEM, "pd_comm: buffer allocation failed");
    }
    if (W == 1) s.recv = s.send;
    HIPCM(m, hipEventCreateWithFlags(&s.packed, hipEventDisableTiming));
    HIPCM(m, hipEventCreateWithFlags(&s.landed, hipEventDisableTiming));
    // (the export writes every byte of the image every time: only the padding behind it — the last slice's tail — must be, and stay, zero)
    if (all > img) HIPCM(m, hipMemsetAsync(s.send + img, 0, all - img, c->stream));
    HIPCM(m, hipMemsetAsync(s.exc, 0, (size_t)COMM_EXC_BLOCK * sizeof(pd_exc), c->stream));
    return PD_OK;
}

int comm_setup(pd_comm *m)
{
    pd_ctx *c = m->ctx;
    m->n_tiles = c->n_tiles; m->n_sums = c->n_words - c->n_cells;
    m->slice_tiles = std::max<uint64_t>(1, (m->n_tiles + (uint64_t)m->world - 1) / (uint64_t)m->world);
    m->slice_bytes = m->slice_tiles * (PD_TILE / 2);
    m->tile_first = std::min<uint64_t>((uint64_t)m->rank * m->slice_tiles, m->n_tiles);
    m->tile_count = std::min<uint64_t>(m->slice_tiles, m->n_tiles - m->tile_first);
    const size_t W = (size_t)m->world;
    HIPCM(m, hipSetDevice(c->device));
    HIPCM(m, hipStreamCreateWithFlags(&m->links, hipStreamNonBlocking));
    HIPCM(m, hipEventCreateWithFlags(&m->swept, hipEventDisableTiming));
    HIPCM(m, hipEventCreateWithFlags(&m->gathered, hipEventDisableTiming));
    // (the slots' exchange buffers — twice the 4-bit image per slot, 3 GB for a 3 Gb genome — are made by the first pd_sliced_sum_start that
    // uses the slot: a device allocation of that size costs tenths of a second, and the executable only ever uses slot 0)
    if (hipMalloc(&m->part_mine, m->slice_tiles * PD_TILE_PARTIAL_BYTES + 64) != hipSuccess ||
        hipMalloc(&m->part_all, W * m->slice_tiles * PD_TILE_PARTIAL_BYTES + 64) != hipSuccess)
        return comm_fail(m, PD_ENOMEM, "pd_comm: buffer allocation failed");
    HIPCM(m, hipMemsetAsync(m->part_mine, 0, m->slice_tiles * PD_TILE_PARTIAL_BYTES + 64, c->stream));
    HIPCM(m, hipStreamSynchronize(c->stream));
    return PD_OK;
}
} // namespace

extern "C" {

int pd_comm_unique_id(void *id128)
{
    if (!id128) return PD_EINVAL;
    if (!rccl().ok) return PD_ENODEV;
    ncclUniqueId id;
    if (rccl().GetUniqueId(&id) != ncclSuccess) return PD_EHIP;
    static_assert(sizeof(id) == PD_UNIQUE_ID_BYTES, "unique id size");
    memcpy(id128, &id, sizeof id);
    return PD_OK;
}

int pd_comm_init(pd_ctx *ctx, const void *id128, int rank, int n_ranks, pd_comm **out)
{
    if (!ctx || !id128 || !out || rank < 0 || rank >= n_ranks) return PD_EINVAL;
    *out = nullptr;
    if (!rccl().ok) { std::lock_guard<std::mutex> lk(ctx->mu); ctx->err = "pd_comm_init: librccl.so.1 cannot be loaded"; return PD_ENODEV; }
    pd_comm *m = new pd_comm; m->ctx = ctx; m->tp = &rccl(); m->rank = rank; m->world = n_ranks;
    ncclUniqueId id; memcpy(&id, id128, sizeof id);
    const bool made = hipSetDevice(ctx->device) == hipSuccess && rccl().CommInitRank(&m->nccl, n_ranks, id, rank) == ncclSuccess;
    if (!made) {
        { std::lock_guard<std::mutex> lk(ctx->mu); ctx->err = "pd_comm_init: ncclCommInitRank failed"; }
        delete m; return PD_EHIP;
    }
    const int rc = comm_setup(m);
    if (rc) { pd_comm_destroy(m); return rc; }
    *out = m;
    return PD_OK;
}

// librccl loaded and an n-rank communicator bootstrapped over `devices` NOW, before any context exists; the next pd_comm_init_all
// over contexts on exactly these devices adopts it.  For a short-lived process: the load registers half a gigabyte of code objects
// under the runtime lock every kernel launch needs, so behind a running decode it costs the decode (profiles/r05_comm_init.txt:
// 0.68 -> 2.19 s); ahead of pd_create it overlaps header and index reads and slows nothing down.  devices == NULL: load only.
int pd_comm_preinit(const int *devices, int n)
{
    if (!rccl().ok) return PD_ENODEV;
    if (!devices || n < 1) return PD_OK;
    std::vector<int> devs(devices, devices + n);
    if (!rccl().alt)
        for (int i = 0; i < n; ++i) for (int j = 0; j < i; ++j) if (devs[(size_t)i] == devs[(size_t)j]) return PD_EINVAL;
    std::vector<ncclComm_t> nc((size_t)n, nullptr);
    if (rccl().CommInitAll(nc.data(), n, devs.data()) != ncclSuccess) return PD_EHIP;
    std::lock_guard<std::mutex> lk(g_parked.mu);
    for (ncclComm_t old : g_parked.nc) if (old) (void)rccl().CommDestroy(old);         // (made, never adopted)
    g_parked.devs = devs; g_parked.nc = nc;
    return PD_OK;
}

static int comm_init_over(Rccl &tp, const char *what, pd_ctx **ctxs, int n, pd_comm **comms)
{
    if (!ctxs || !comms || n < 1) return PD_EINVAL;
    std::vector<int> devs((size_t)n);
    for (int i = 0; i < n; ++i) { if (!ctxs[i]) return PD_EINVAL; devs[(size_t)i] = ctxs[i]->device; comms[i] = nullptr; }
    if (!tp.ok) { std::lock_guard<std::mutex> lk(ctxs[0]->mu); ctxs[0]->err = std::string(what) + ": librccl.so.1 cannot be loaded"; return PD_ENODEV; }
    if (!tp.alt)
        for (int i = 0; i < n; ++i) for (int j = 0; j < i; ++j) if (devs[(size_t)i] == devs[(size_t)j]) {
            std::lock_guard<std::mutex> lk(ctxs[0]->mu); ctxs[0]->err = std::string(what) + ": two contexts share a GPU (RCCL wants one rank per device)"; return PD_EINVAL; }
    std::vector<ncclComm_t> nc((size_t)n, nullptr);
    bool made = false;
    // (NOT `&tp == &rccl()`: rccl() LOADS librccl — round 6's first in-process communicators spent 1.1 s doing exactly that, and slowed the decode
    // beside them the way round 5's side thread had: profiles/r06_comm_transports.txt)
    const bool is_local = &tp == &local_tp();
    if (!is_local) {               // a communicator made ahead of the contexts (pd_comm_preinit) over the same devices
        std::lock_guard<std::mutex> lk(g_parked.mu);
        if (g_parked.devs == devs && !g_parked.nc.empty()) { nc = g_parked.nc; g_parked.nc.clear(); g_parked.devs.clear(); made = true; }
    }
    if (!made) made = tp.CommInitAll(nc.data(), n, devs.data()) == ncclSuccess;
    if (!made) {
        std::lock_guard<std::mutex> lk(ctxs[0]->mu);
        ctxs[0]->err = !is_local ? std::string("ncclCommInitAll failed") : std::string(what) + ": no peer access between the contexts' GPUs";
        return PD_EHIP;
    }
    // every rank's buffers side by side (one thread per rank: allocations on eight devices in a row were most of a list run's start-up)
    std::vector<int> rcs((size_t)n, PD_OK);
    for (int i = 0; i < n; ++i) {
        pd_comm *m = new pd_comm; m->ctx = ctxs[i]; m->tp = &tp; m->rank = i; m->world = n; m->nccl = nc[(size_t)i];
        comms[i] = m;
    }
    if (n == 1) rcs[0] = comm_setup(comms[0]);
    else {
        std::vector<std::thread> th;
        for (int i = 0; i < n; ++i) th.emplace_back([&, i] { rcs[(size_t)i] = comm_setup(comms[i]); });
        for (auto &t : th) t.join();
    }
    int rc = PD_OK;
    for (int i = 0; i < n; ++i) if (rcs[(size_t)i] != PD_OK && rc == PD_OK) { rc = rcs[(size_t)i]; std::lock_guard<std::mutex> lk(ctxs[0]->mu); ctxs[0]->err = comms[i]->err; }
    if (rc) for (int i = 0; i < n; ++i) { pd_comm_destroy(comms[i]); comms[i] = nullptr; }
    return rc;
}

int pd_comm_init_all(pd_ctx **ctxs, int n, pd_comm **comms) { return comm_init_over(rccl(), "pd_comm_init_all", ctxs, n, comms); }

int pd_comm_init_local(pd_ctx **ctxs, int n, pd_comm **comms) { return comm_init_over(local_tp(), "pd_comm_init_local", ctxs, n, comms); }

int pd_comm_destroy(pd_comm *m)
{
    if (!m) return PD_OK;
    if (m->ctx) (void)hipSetDevice(m->ctx->device);
    if (m->links) (void)hipStreamSynchronize(m->links);
    if (m->ctx && m->ctx->stream) (void)hipStreamSynchronize(m->ctx->stream);
    if (m->nccl && m->tp) (void)m->tp->CommDestroy(m->nccl);
    for (pd_comm::Slot &s : m->slot) {
        for (void *p : {(void *)s.send, (void *)(s.recv == s.send ? nullptr : s.recv), (void *)s.meta, (void *)s.exc, (void *)s.exc_all, (void *)s.count}) if (p) (void)hipFree(p);
        if (s.packed) (void)hipEventDestroy(s.packed);
        if (s.landed) (void)hipEventDestroy(s.landed);
    }
    if (m->part_mine) (void)hipFree(m->part_mine);
    if (m->part_all) (void)hipFree(m->part_all);
    if (m->slice_depth) (void)hipFree(m->slice_depth);
    if (m->swept) (void)hipEventDestroy(m->swept);
    if (m->gathered) (void)hipEventDestroy(m->gathered);
    if (m->links) (void)hipStreamDestroy(m->links);
    delete m;
    return PD_OK;
}

const char *pd_comm_strerror(const pd_comm *m) { return m ? m->err.c_str() : ""; }

// The slot's exchange buffers now instead of at its first pd_sliced_sum_start: not collective, any thread (the executable calls it on a
// side thread while the rank's file is being decoded: 3 GB of device allocations per rank for a 3 Gb genome are tenths of a second).
int pd_comm_prepare(pd_comm *m, int slot)
{
    if (!m || slot < 0 || slot > 1) return PD_EINVAL;
    HIPCM(m, hipSetDevice(m->ctx->device));
    return slot_setup(m, m->slot[slot]);
}

// Collective, first half: packs the context's sample (pd_export_i4) into `slot` and puts it on the links.  Only enqueues: the
// context may be reset and refilled right away, and the other slot may be started before this one is finished.
int pd_sliced_sum_start(pd_comm *m, int slot)
{
    if (!m || slot < 0 || slot > 1) return PD_EINVAL;
    pd_comm::Slot &s = m->slot[slot];
    if (s.busy) return comm_fail(m, PD_EINVAL, "pd_sliced_sum_start: the slot has not been finished");
    pd_ctx *c = m->ctx;
    const size_t W = (size_t)m->world, sb = (size_t)m->slice_bytes;
    HIPCM(m, hipSetDevice(c->device));
    hipStream_t st = c->stream, ln = m->links;
    // 1. this rank's 4-bit image (straight from the tile windows in LDS when the sample is still deferred), its own slice in place
    int rc = slot_setup(m, s);
    if (rc) return rc;
    rc = pd_export_i4(c, s.send, s.exc, COMM_EXC_BLOCK, s.count);
    if (rc) return comm_fail(m, rc, std::string("pd_export_i4: ") + c->err);
    HIPCM(m, hipMemcpyAsync(s.meta, c->sums, (size_t)m->n_sums * 4, hipMemcpyDeviceToDevice, st));
    HIPCM(m, hipMemsetAsync(s.meta + m->n_sums, 0, W * 4, st));
    HIPCM(m, hipMemcpyAsync(s.meta + m->n_sums + m->rank, s.count, 4, hipMemcpyDeviceToDevice, st));
    if (W > 1) HIPCM(m, hipMemcpyAsync(s.recv + (size_t)m->rank * sb, s.send + (size_t)m->rank * sb, sb, hipMemcpyDeviceToDevice, st));
    HIPCM(m, hipEventRecord(s.packed, st));
    HIPCM(m, hipStreamWaitEvent(ln, s.packed, 0));
    // 2. the all-to-all: every pair of GPUs moves 1/world of the image over its own xGMI link, all links at once
    for (size_t c0 = 0; c0 < sb && W > 1; c0 += COMM_MSG_BYTES) {
        const size_t n = std::min(COMM_MSG_BYTES, sb - c0);
        NCCLOK(m, m->tp->GroupStart());
        for (int p = 0; p < m->world; ++p) {
            if (p == m->rank) continue;
            NCCLOK(m, m->tp->Send(s.send + (size_t)p * sb + c0, n, ncclUint8, p, m->nccl, ln));
            NCCLOK(m, m->tp->Recv(s.recv + (size_t)p * sb + c0, n, ncclUint8, p, m->nccl, ln));
        }
        NCCLOK(m, m->tp->GroupEnd());
    }
    // 3. tile sums (+ exception counts) summed over the ranks; everybody's exception block to everybody
    NCCLOK(m, m->tp->AllReduce(s.meta, s.meta, (size_t)m->n_sums + W, ncclInt32, ncclSum, m->nccl, ln));
    NCCLOK(m, m->tp->AllGather(s.exc, s.exc_all, (size_t)COMM_EXC_BLOCK * sizeof(pd_exc), ncclUint8, m->nccl, ln));
    HIPCM(m, hipEventRecord(s.landed, ln));
    s.busy = true;
    return PD_OK;
}

// Collective, second half: on `root`, cover / sum receive what pd_scan_reduce_windows would give on a context holding the sum
// of all ranks' samples started in `slot` (windows of w >= 8192 cells).  Blocks until this rank's part is complete.
int pd_sliced_sum_finish(pd_comm *m, int slot, uint32_t w, uint32_t min_dep, unsigned wrap_bits, int root, uint32_t *cover, uint64_t *sum)
{
    if (!m || slot < 0 || slot > 1 || root < 0 || root >= m->world || w < PD_TILE || wrap_bits > 32) return PD_EINVAL;
    if (m->rank == root && (!cover || !sum)) return PD_EINVAL;
    pd_comm::Slot &s = m->slot[slot];
    if (!s.busy) return comm_fail(m, PD_EINVAL, "pd_sliced_sum_finish: the slot has not been started");
    s.busy = false;
    pd_ctx *c = m->ctx;
    const size_t W = (size_t)m->world, sb = (size_t)m->slice_bytes;
    HIPCM(m, hipSetDevice(c->device));
    hipStream_t st = c->stream, ln = m->links;
    // 4. this rank's slice: sum of the images, prefix sum, wrap, per-tile partials of the windows
    HIPCM(m, hipStreamWaitEvent(st, s.landed, 0));
    int rc = pd_slice_sweep_i4(c, s.recv, (uint32_t)W, sb, m->tile_first, m->tile_count, s.meta, s.exc_all, COMM_EXC_BLOCK, s.meta + m->n_sums, w, min_dep,
                               wrap_bits, m->part_mine);
    if (rc) return comm_fail(m, rc, std::string("pd_slice_sweep_i4: ") + c->err);
    // 5. 24 bytes per tile to the root
    const size_t pb = (size_t)m->slice_tiles * PD_TILE_PARTIAL_BYTES;
    if (m->rank == root) HIPCM(m, hipMemcpyAsync(m->part_all + (size_t)root * pb, m->part_mine, pb, hipMemcpyDeviceToDevice, st));
    if (W > 1) {
        HIPCM(m, hipEventRecord(m->swept, st));
        HIPCM(m, hipStreamWaitEvent(ln, m->swept, 0));
        NCCLOK(m, m->tp->GroupStart());
        if (m->rank == root) { for (int p = 0; p < m->world; ++p) if (p != root) NCCLOK(m, m->tp->Recv(m->part_all + (size_t)p * pb, pb, ncclUint8, p, m->nccl, ln)); }
        else NCCLOK(m, m->tp->Send(m->part_mine, pb, ncclUint8, root, m->nccl, ln));
        NCCLOK(m, m->tp->GroupEnd());
        HIPCM(m, hipEventRecord(m->gathered, ln));
        HIPCM(m, hipStreamWaitEvent(st, m->gathered, 0));
    }
    std::vector<int32_t> counts(W);
    HIPCM(m, hipMemcpyAsync(counts.data(), s.meta + m->n_sums, W * 4, hipMemcpyDeviceToHost, st));
    HIPCM(m, hipStreamSynchronize(st));
    // (every rank holds the same all-reduced counts, so every rank leaves here with the same code; the samples are intact —
    // an export that overflows never consumes a deferred sample, it leaves it in the rank's difference arrays)
    for (int32_t k : counts) if (k < 0 || (uint32_t)k > COMM_EXC_BLOCK)
        return comm_fail(m, PD_ERANGE, "a sample has more cells outside the 4-bit range than the exception block holds: the sliced sum does not apply, "
                                       "every context still holds its sample (pd_accumulate_from adds them up)");
    if (m->rank == root) {
        rc = pd_gather_windows(c, m->part_all, w, cover, sum);
        if (rc) return comm_fail(m, rc, std::string("pd_gather_windows: ") + c->err);
    }
    return PD_OK;
}

// Collective: after pd_sliced_sum_start(slot) — this rank's slice of the SUMMED sample as int32 depth cells (prefix-summed, wrapped) in
// m->slice_depth: what pd_scan would leave in cells [tile_first, tile_first + tile_count) x 8192 of a context holding every rank's sample.
// The statistics that need the cells themselves (narrow windows, annotation intervals) then run on the rank that owns them.
static int sliced_depth(pd_comm *m, int slot, unsigned wrap_bits)
{
    pd_comm::Slot &s = m->slot[slot];
    if (!s.busy) return comm_fail(m, PD_EINVAL, "sliced statistics: the slot has not been started");
    s.busy = false;
    pd_ctx *c = m->ctx;
    const size_t W = (size_t)m->world, sb = (size_t)m->slice_bytes;
    HIPCM(m, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    HIPCM(m, hipStreamWaitEvent(st, s.landed, 0));
    std::vector<int32_t> counts(W);
    HIPCM(m, hipMemcpyAsync(counts.data(), s.meta + m->n_sums, W * 4, hipMemcpyDeviceToHost, st));
    HIPCM(m, hipStreamSynchronize(st));
    for (int32_t k : counts) if (k < 0 || (uint32_t)k > COMM_EXC_BLOCK)
        return comm_fail(m, PD_ERANGE, "a sample has more cells outside the 4-bit range than the exception block holds: the sliced sum does not apply, "
                                       "every context still holds its sample (pd_accumulate_from adds them up)");
    if (!m->slice_depth && hipMalloc(&m->slice_depth, (size_t)m->slice_tiles * PD_TILE * 4 + 64) != hipSuccess) {
        (void)hipGetLastError();
        return comm_fail(m, PD_ENOMEM, "sliced statistics: the slice's depth cells could not be allocated");
    }
    {
        std::lock_guard<std::mutex> lk(c->mu);          // (comm_fail takes this lock: nothing below fails under it)
        const uint32_t mask = (wrap_bits == 0 || wrap_bits == 32) ? 0xFFFFFFFFu : ((1u << wrap_bits) - 1u);
        { ProfScope ps(c, "tile_carry"); launch_tile_carry(st, s.meta, c->bsum, c->carry, (uint32_t)c->n_tiles); }
        { ProfScope ps(c, "slice_depth");
          TileMap tm{c->d_tile_contig, c->d_off, c->d_len, nullptr};
          launch_sweep_i4(st, s.recv, (uint32_t)W, sb, (uint32_t)m->tile_first, (uint32_t)m->tile_count, s.exc_all, COMM_EXC_BLOCK, s.meta + m->n_sums,
                          c->slice_flags, slice_flag_bytes(c->n_tiles), c->carry, mask, tm, PD_TILE, 0, nullptr, m->slice_depth); }
    }
    HIPCM(m, hipGetLastError());
    return PD_OK;
}

// windows of w < 8192 cells: every rank reduces the windows of its slice's cells; the window arrays are merged (each window has one
// writer: an all-reduce of the words adds zeros to it), the windows across tile edges are put together from the tiles' shares
static int sliced_narrow_windows(pd_comm *m, uint32_t w, uint32_t min_dep, unsigned wrap_bits, int root, uint32_t *cover, uint64_t *sum)
{
    int rc = pd_sliced_sum_start(m, 0);
    if (rc) return rc;
    rc = sliced_depth(m, 0, wrap_bits);
    if (rc) return rc;
    pd_ctx *c = m->ctx;
    hipStream_t st = c->stream, ln = m->links;
    // (the collective is the context's only user while it runs — one thread per rank; the context's lock is taken only where its
    // shared scratch and profile records are touched, never across a comm_fail, which takes it itself)
    std::vector<uint64_t> wo((size_t)c->n_contigs + 1);
    pd_window_layout(c, w, wo.data());
    const uint64_t nw = wo[c->n_contigs];
    const size_t b_off = ((size_t)c->n_contigs + 1) * 8, b_sum = (size_t)nw * 8, b_cov = ((size_t)nw * 4 + 15) / 16 * 16;
    std::string emsg;
    { std::lock_guard<std::mutex> lk(c->mu);
      rc = ensure_scratch(c, b_off + 64);
      if (!rc) rc = win_keep_fit(c, b_sum + b_cov + 64);
      if (rc) emsg = c->err; }
    if (rc) return comm_fail(m, rc, emsg);
    uint64_t *d_wo = (uint64_t *)c->scratch;
    unsigned long long *d_sum = (unsigned long long *)c->wk;
    uint32_t *d_cov = (uint32_t *)(c->wk + b_sum);
    HIPCM(m, hipMemcpyAsync(d_wo, wo.data(), b_off, hipMemcpyHostToDevice, st));
    HIPCM(m, hipStreamSynchronize(st));                 // wo is a local
    HIPCM(m, hipMemsetAsync(d_sum, 0, b_sum + b_cov, st));
    TileMap tm{c->d_tile_contig, c->d_off, c->d_len, d_wo};
    TilePart *parts = (TilePart *)m->part_all;           // indexed by tile of the genome: the ranks' slices are consecutive blocks of it
    int e_lds = 0;
    { std::lock_guard<std::mutex> lk(c->mu);
      ProfScope ps(c, "slice_windows");
      e_lds = launch_sweep_windows_slice(st, m->slice_depth, (uint32_t)m->tile_first, (uint32_t)m->tile_count, tm, w, min_dep, d_cov, d_sum, parts); }
    if (e_lds) return comm_fail(m, PD_EHIP, "window sweep: cannot reserve LDS");
    HIPCM(m, hipGetLastError());
    if (m->world > 1) {
        HIPCM(m, hipEventRecord(m->swept, st));
        HIPCM(m, hipStreamWaitEvent(ln, m->swept, 0));
        const size_t pb = (size_t)m->slice_tiles * PD_TILE_PARTIAL_BYTES;
        NCCLOK(m, m->tp->AllGather(m->part_all + (size_t)m->rank * pb, m->part_all, pb, ncclUint8, m->nccl, ln));
        NCCLOK(m, m->tp->AllReduce(d_sum, d_sum, (b_sum + b_cov) / 4, ncclInt32, ncclSum, m->nccl, ln));
        HIPCM(m, hipEventRecord(m->gathered, ln));
        HIPCM(m, hipStreamWaitEvent(st, m->gathered, 0));
    }
    launch_window_edges(st, parts, tm, (uint32_t)c->n_tiles, w, d_cov, d_sum);
    HIPCM(m, hipGetLastError());
    if (m->rank == root) {
        HIPCM(m, hipMemcpyAsync(sum, d_sum, b_sum, hipMemcpyDeviceToHost, st));
        HIPCM(m, hipMemcpyAsync(cover, d_cov, (size_t)nw * 4, hipMemcpyDeviceToHost, st));
    }
    HIPCM(m, hipStreamSynchronize(st));
    { std::lock_guard<std::mutex> lk(c->mu); c->wk_w = w; c->wk_nw = nw; c->wk_woff = wo; c->wk_valid = true; }   // (every rank holds the merged statistics)
    return PD_OK;
}

int pd_sliced_window_sum(pd_comm *m, uint32_t w, uint32_t min_dep, unsigned wrap_bits, int root, uint32_t *cover, uint64_t *sum)
{
    if (!m || root < 0 || root >= m->world || w == 0 || wrap_bits > 32) return PD_EINVAL;
    if (m->rank == root && (!cover || !sum)) return PD_EINVAL;
    if (w < PD_TILE) return sliced_narrow_windows(m, w, min_dep, wrap_bits, root, cover, sum);
    const int rc = pd_sliced_sum_start(m, 0);
    return rc ? rc : pd_sliced_sum_finish(m, 0, w, min_dep, wrap_bits, root, cover, sum);
}

// Collective: pd_reduce_intervals (PD:329-348 over the CDS / BED regions) on the sum of every rank's sample without any GPU holding the
// summed arrays: a rank reduces the stretches of the regions that lie in its slice, the per-region partial results of all ranks are
// gathered and added on `root`.
int pd_sliced_interval_sum(pd_comm *m, const pd_region *regs, size_t n, uint32_t min_dep, unsigned wrap_bits, int root, int32_t *cover, uint64_t *sum)
{
    if (!m || root < 0 || root >= m->world || wrap_bits > 32 || (n && !regs)) return PD_EINVAL;
    if (m->rank == root && n && (!cover || !sum)) return PD_EINVAL;
    if (n > 0xFFFFFFF0ull) return comm_fail(m, PD_EINVAL, "too many regions");
    int rc = pd_sliced_sum_start(m, 0);
    if (rc) return rc;
    rc = sliced_depth(m, 0, wrap_bits);
    if (rc) return rc;
    if (n == 0) return PD_OK;
    pd_ctx *c = m->ctx;
    hipStream_t st = c->stream, ln = m->links;
    constexpr uint32_t PIECE = 16384;
    const uint64_t lo_cell = m->tile_first * PD_TILE, hi_cell = (m->tile_first + m->tile_count) * PD_TILE;
    std::vector<Piece> pieces;
    for (size_t i = 0; i < n; ++i) {
        const pd_region &r = regs[i];
        if (r.tid < 0 || r.tid >= c->n_contigs) return comm_fail(m, PD_EINVAL, "pd_sliced_interval_sum: contig id out of range");
        int64_t b = (int64_t)r.first - 1, e = r.second;          // cells [first - 1, second), clipped to the slot
        const int64_t slot = (int64_t)(c->off[r.tid + 1] - c->off[r.tid]);
        if (b < 0) b = 0;
        if (e > slot) e = slot;
        if (b >= e) continue;
        uint64_t gb = c->off[r.tid] + (uint64_t)b, ge = c->off[r.tid] + (uint64_t)e;
        if (gb < lo_cell) gb = lo_cell;
        if (ge > hi_cell) ge = hi_cell;
        for (uint64_t p = gb; p < ge; p += PIECE) {
            Piece pc; pc.start = p - lo_cell; pc.count = (uint32_t)std::min<uint64_t>(PIECE, ge - p); pc.region = (uint32_t)i;
            pieces.push_back(pc);
        }
    }
    const size_t W = (size_t)m->world;
    const size_t b_sum = n * 8, b_cov = (n * 4 + 7) / 8 * 8, blk = b_sum + b_cov, b_p = (pieces.size() * sizeof(Piece) + 15) / 16 * 16;
    std::string emsg;
    { std::lock_guard<std::mutex> lk(c->mu); rc = ensure_scratch(c, b_p + blk * (W + 1) + 64); if (rc) emsg = c->err; }
    if (rc) return comm_fail(m, rc, emsg);
    unsigned char *sc = (unsigned char *)c->scratch;
    Piece *d_p = (Piece *)sc;
    unsigned char *mine = sc + b_p, *all = mine + blk;
    unsigned long long *d_sum = (unsigned long long *)mine;
    int *d_cov = (int *)(mine + b_sum);
    if (!pieces.empty()) HIPCM(m, hipMemcpyAsync(d_p, pieces.data(), pieces.size() * sizeof(Piece), hipMemcpyHostToDevice, st));
    HIPCM(m, hipMemsetAsync(mine, 0, blk, st));
    HIPCM(m, hipStreamSynchronize(st));                 // pieces is a local
    { std::lock_guard<std::mutex> lk(c->mu);
      ProfScope ps(c, "slice_intervals");
      launch_reduce_pieces(st, m->slice_depth, d_p, (uint32_t)pieces.size(), min_dep, d_cov, d_sum); }
    HIPCM(m, hipGetLastError());
    if (m->world > 1) {
        HIPCM(m, hipEventRecord(m->swept, st));
        HIPCM(m, hipStreamWaitEvent(ln, m->swept, 0));
        NCCLOK(m, m->tp->AllGather(mine, all, blk, ncclUint8, m->nccl, ln));
        HIPCM(m, hipEventRecord(m->gathered, ln));
        HIPCM(m, hipStreamWaitEvent(st, m->gathered, 0));
    } else HIPCM(m, hipMemcpyAsync(all, mine, blk, hipMemcpyDeviceToDevice, st));
    if (m->rank == root) {
        std::vector<unsigned char> host(blk * W);
        HIPCM(m, hipMemcpyAsync(host.data(), all, blk * W, hipMemcpyDeviceToHost, st));
        HIPCM(m, hipStreamSynchronize(st));
        for (size_t i = 0; i < n; ++i) { cover[i] = 0; sum[i] = 0; }
        for (size_t k = 0; k < W; ++k) {
            const uint64_t *hs = (const uint64_t *)(host.data() + k * blk);
            const int32_t *hc = (const int32_t *)(host.data() + k * blk + b_sum);
            for (size_t i = 0; i < n; ++i) { cover[i] += hc[i]; sum[i] += hs[i]; }
        }
    } else HIPCM(m, hipStreamSynchronize(st));
    return PD_OK;
}

} // extern "C"
