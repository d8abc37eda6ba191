// pipeline.cpp — one `pandepth` invocation: inputs -> reads the reference would count -> M/=/X
// runs -> depth engine (through pd_engine_api) -> CoveredSite/TotalDepth -> .stat.gz tables.
//
// What stays on the host is everything the reference does around its depth arrays:
//   * which cell type a mode uses (PD:4127 / PD:4413 / PD:4553 / PD:2687): uint32 for a single
//     indexed BAM without -a and without -w<150, the 18-bit SiteInfo cell everywhere else;
//   * which reads are fed to the increment loop: index fetch of the merged target spans widened
//     by one base (PD:419-434), the sorted no-index stream's span cursor (PD:4608-4646), or every
//     read (PD:4679-4711);
//   * the table text (PD:4879-5127) and the per-site file (PD:4264-4284).
// The increment loop itself, the statistics and the window sweep run on the engine.
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <functional>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <condition_variable>
#include <deque>
#include <thread>
#include "bam.h"
#include "engine_api.h"
#include "options.h"
#include "regions.h"
#include "paf.h"
#include "report.h"
#include "pgzip.h"

namespace pdh {

namespace {

// PANDEPTH_TIMING=1: phase wall times on stderr (diagnostics only)
struct PhaseTimer {
    bool on = getenv("PANDEPTH_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(), last = t0;
    void mark(const char *what)
    {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[timing] %-28s %8.3f s   (total %.3f s)\n", what,
                std::chrono::duration<double>(now - last).count(), std::chrono::duration<double>(now - t0).count());
        last = now;
    }
};

struct Engine {
    const pd_engine_api *api = nullptr;
    pd_ctx *ctx = nullptr;
    std::mutex err_mu;
    std::string err;
    std::atomic<bool> cancel{false};            // the run is being abandoned: a writer working behind the statistics stops where it is
    void fail(const std::string &m) { std::lock_guard<std::mutex> lk(err_mu); if (err.empty()) err = m; }
    bool ok() { std::lock_guard<std::mutex> lk(err_mu); return err.empty(); }
    std::string message() { std::lock_guard<std::mutex> lk(err_mu); return err; }
    bool ck(int rc, const char *what)
    {
        if (rc == 0) return true;
        const char *m = api->strerror(ctx);
        fail(std::string(what) + ": " + (m ? m : "engine error"));
        return false;
    }
};

// Per-thread producer of run batches.  Runs that keep (tid, beg) non-decreasing go to the sorted
// stream; the rest (later runs of multi-run reads, unsorted input) to a second stream whose
// measured disorder decides between the owner-tile path and the atomic path.  Batches are built
// in thread-local memory and handed to pd_push_intervals, which copies them into one of the
// engine's few pinned staging slots (the copy is ~1 % of the inflate cost and keeps the amount of
// pinned memory independent of the number of reader threads).
class RunSink {
public:
    static constexpr size_t CAP = (size_t)1 << 18;
    explicit RunSink(Engine *e) : e_(e) { s_.reserve(CAP); o_.reserve(CAP / 4); }
    ~RunSink() { flush(); }
    inline void emit(int32_t tid, int32_t beg, int32_t end)
    {
        const uint64_t key = ((uint64_t)(uint32_t)tid << 32) | (uint32_t)(beg < 0 ? 0 : beg);
        if (key >= s_last_) {
            s_.push_back(pd_iv{tid, beg, end});
            s_last_ = key;
            if (s_.size() >= CAP) flush_sorted();
        } else {
            if (key < o_max_) {
                const uint64_t d = (key >> 32) == (o_max_ >> 32) ? (o_max_ - key) : (uint64_t)1 << 40;
                if (d > o_dis_) o_dis_ = d;
            } else o_max_ = key;
            o_.push_back(pd_iv{tid, beg, end});
            if (o_.size() >= CAP / 4) flush_other();
        }
    }
    void flush() { flush_sorted(); flush_other(); }
private:
    // PANDEPTH_DECODE_ONLY=1 (diagnostics): decode and expand, but drop the batches
    const bool drop_ = tune("decode_only") != nullptr;
    void flush_sorted()
    {
        if (!drop_ && !s_.empty() && e_->ok()) e_->ck(e_->api->push_intervals(e_->ctx, s_.data(), s_.size(), PD_PUSH_SORTED), "pd_push_intervals");
        s_.clear(); s_last_ = 0;
    }
    void flush_other()
    {
        // beyond a few tiles of disorder the owner tiles would re-read too much: use the atomic kernel
        const unsigned f = o_dis_ <= (1u << 14) ? (PD_PUSH_SORTED | PD_PUSH_DISORDER((unsigned)o_dis_)) : PD_PUSH_DEFAULT;
        if (!drop_ && !o_.empty() && e_->ok()) e_->ck(e_->api->push_intervals(e_->ctx, o_.data(), o_.size(), f), "pd_push_intervals");
        o_.clear(); o_max_ = 0; o_dis_ = 0;
    }
    Engine *e_;
    std::vector<pd_iv> s_, o_;
    uint64_t s_last_ = 0, o_max_ = 0, o_dis_ = 0;
};

// PD:438-460: CIGAR walk with an int32 cursor; one run per M/=/X operation, D/N advance only.
inline void emit_runs(const AlnRec &r, RunSink *sink)
{
    int32_t cur = r.pos;
    for (uint32_t i = 0; i < r.n_cigar; ++i) {
        const uint32_t op = r.cigar[i] & 0xf;
        const int32_t len = (int32_t)(r.cigar[i] >> 4);
        const int32_t nxt = (int32_t)((uint32_t)cur + (uint32_t)len);      // the int cursor's wrap on corrupt lengths, spelled out
        if (op == 0 || op == 7 || op == 8) { sink->emit(r.tid, cur, nxt); cur = nxt; }
        else if (op == 2 || op == 3) cur = nxt;
    }
}

struct ReadFilter {
    uint32_t flag_mask; int min_mapq; int32_t n_contigs;
    inline bool pass(const AlnRec &r) const
    {
        return !(r.flag & flag_mask) && (int)r.mapq >= min_mapq && r.tid >= 0 && r.tid < n_contigs;
    }
};

// htslib multi-region fetch of "chr:max(s-1,1)-min(e+1,len)" for every merged span (PD:419-430):
// a read is returned when pos < region_end && endpos > region_begin0.
struct SpanIndex {
    std::vector<std::vector<std::pair<int32_t, int32_t>>> per_tid;   // (begin0, end) sorted
    std::vector<char> whole;                                         // spans cover the whole contig
    bool synthetic = true;                                           // whole-contig bins (modes 0/5/6)
    void build(const RegionModel &rm, const AlnHeader &h, bool synth)
    {
        synthetic = synth;
        per_tid.assign(h.names.size(), {}); whole.assign(h.names.size(), 0);
        for (auto &kv : rm.merged) {
            if (kv.first < 0 || (size_t)kv.first >= h.names.size()) continue;
            const int64_t len = h.lens[kv.first];
            auto &v = per_tid[kv.first];
            for (auto &se : kv.second) {
                int64_t b = (int64_t)se.first - 1; if (b < 1) b = 1;
                int64_t e = (int64_t)se.second + 1; if (e > len) e = len;
                v.emplace_back((int32_t)(b - 1), (int32_t)e);
            }
            if (synth) whole[kv.first] = 1;    // bins tile [1,len] (minus at most the last base): widened, they cover every read
        }
    }
    inline bool hit(const AlnRec &r) const
    {
        const auto &v = per_tid[r.tid];
        if (v.empty()) return false;
        if (whole[r.tid]) return true;
        // spans are disjoint and ordered, so region ends increase: first span whose end is past pos
        size_t lo = 0, hi = v.size();
        while (lo < hi) { const size_t m = (lo + hi) / 2; if (v[m].second > r.pos) hi = m; else lo = m + 1; }
        return lo < v.size() && r.endpos() > v[lo].first;
    }
};

bool index_exists(const std::string &p) { return file_exists(p + ".bai") || file_exists(p + ".crai") || file_exists(p + ".csi"); }

// ---- readers ---------------------------------------------------------------------------------
// records that START in [begin, stop) of an open BAM, through the host decoder
bool decode_range(AlnReader &rd, uint64_t begin, uint64_t stop, const ReadFilter &flt, const struct SpanIndex &spans,
                  RunSink *sink, uint64_t *n_rec, std::string *err);

struct DevRange;
int read_bam_device(const std::string &path, const Options &o, const AlnHeader &main_hdr, const SpanIndex &spans, const RegionModel &rm,
                    int kind, uint64_t first_voff, const BaiIndex *bai, bool sorted, Engine *eng);

bool read_indexed(const std::string &path, const Options &o, const AlnHeader &main_hdr, const SpanIndex &spans,
                  Engine *eng, const RegionModel *rm = nullptr)
{
    ReadFilter flt{o.flag_mask, o.min_mapq, (int32_t)main_hdr.names.size()};
    BaiIndex bai;
    std::string err;
    const auto t_idx = std::chrono::steady_clock::now();
    const bool have_bai = bai.load_for(path, &err);        // .bai or .csi
    if (getenv("PANDEPTH_TIMING")) fprintf(stderr, "[timing]   index of %s: %s in %.3f s\n", path.c_str(), have_bai ? "loaded" : "none", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_idx).count());
    AlnReader probe;
    if (!probe.open(path, &err)) { std::cerr << "Error: Failed to open the index file or BAM/CRAM file: " << path << std::endl; return true; }
    if (probe.is_cram()) {
        // CRAM with a .crai next to it: the reads the reference's index fetch would return (PD:419-434), found by walking
        // the file in order (containers are decoded whole; the .crai itself is not consulted)
        RunSink sink(eng);
        AlnRec r;
        uint64_t n = 0;
        int k;
        probe.set_threads(o.threads);                        // containers decoded ahead on helper threads
        if (!spans.synthetic)
            // GFF / BED targets: only containers whose stretch meets a (widened) merged span can hold selected reads
            probe.cram().set_container_filter([&spans](int32_t tid, int64_t b0, int64_t e) {
                if (tid < 0 || (size_t)tid >= spans.per_tid.size()) return false;
                for (auto &sp : spans.per_tid[(size_t)tid])
                    if ((int64_t)sp.first < e + 2 && (int64_t)sp.second > b0 - 2) return true;
                return false;
            });
        while ((k = probe.next(&r)) == 1) {
            ++n;
            if (flt.pass(r) && spans.hit(r)) emit_runs(r, &sink);
        }
        if (k < 0) { eng->fail(probe.error() + " (" + path + ")"); return false; }
        if (getenv("PANDEPTH_TIMING"))
            fprintf(stderr, "[timing] cram (indexed selection): %llu records, %llu containers decoded, %llu stepped over\n", (unsigned long long)n,
                    (unsigned long long)probe.cram().containers_read(), (unsigned long long)probe.cram().containers_skipped());
        return true;
    }
    if (!probe.is_bam()) { std::cerr << "Error: Failed to open the index file or BAM/CRAM file: " << path << std::endl; return true; }
    // Work list: [begin, end) virtual-offset ranges, each starting at a record boundary.
    //  * whole-genome modes: the file cut at linear-index offsets into ranges of similar size;
    //  * GFF/BED targets: only the index chunks that can hold reads overlapping a (widened) merged
    //    span, like the reference's multi-region iterator (PD:698-730) — most of the file is
    //    never inflated.
    if (have_bai && rm) {
        const int r = read_bam_device(path, o, main_hdr, spans, *rm, 0, probe.tell(), &bai, probe.header().sorted_coordinate(), eng);
        if (r != 0) return r > 0;
    }
    std::vector<BaiIndex::Chunk> work;
    int threads = o.threads < 1 ? 1 : o.threads;
    if (have_bai && !spans.synthetic) {
        for (size_t t = 0; t < spans.per_tid.size(); ++t)
            for (auto &sp : spans.per_tid[t]) bai.query((int32_t)t, sp.first, sp.second, &work);
        BaiIndex::normalise(&work);
    } else {
        std::vector<uint64_t> cuts;
        if (have_bai && threads > 1) cuts = bai.split(probe.tell(), file_size(path), threads * 8);
        else { cuts.push_back(probe.tell()); cuts.push_back(UINT64_MAX); }
        for (size_t i = 0; i + 1 < cuts.size(); ++i) work.emplace_back(cuts[i], cuts[i + 1]);
    }
    // tasks = runs of consecutive ranges of similar total compressed size
    std::vector<std::pair<size_t, size_t>> tasks;        // [first, last) into work
    {
        uint64_t total = 0;
        auto csz = [&](const BaiIndex::Chunk &c) { return ((c.second == UINT64_MAX ? file_size(path) : (c.second >> 16)) - (c.first >> 16)) + 65536; };
        for (auto &c : work) total += csz(c);
        const uint64_t per = total / (uint64_t)(threads * 8) + 1;
        size_t first = 0; uint64_t acc = 0;
        for (size_t i = 0; i < work.size(); ++i) {
            acc += csz(work[i]);
            if (acc >= per) { tasks.emplace_back(first, i + 1); first = i + 1; acc = 0; }
        }
        if (first < work.size()) tasks.emplace_back(first, work.size());
    }
    const size_t n_tasks = tasks.size();
    if ((size_t)threads > n_tasks) threads = n_tasks ? (int)n_tasks : 1;
    std::atomic<size_t> next{0};
    std::atomic<uint64_t> n_rec{0}, busy_us{0};
    auto worker = [&]() {
        const auto w0 = std::chrono::steady_clock::now();
        uint64_t my_rec = 0;
        struct Tally { std::atomic<uint64_t> &n, &us; uint64_t &mine; std::chrono::steady_clock::time_point t0;
                       ~Tally() { n += mine; us += (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count(); } }
            tally{n_rec, busy_us, my_rec, w0};
        AlnReader rd;
        std::string e2;
        if (!rd.open(path, &e2)) { eng->fail(e2); return; }
        RunSink sink(eng);
        for (;;) {
            const size_t t = next.fetch_add(1);
            if (t >= n_tasks || !eng->ok()) break;
            for (size_t w = tasks[t].first; w < tasks[t].second; ++w) {
                std::string e3;
                if (!decode_range(rd, work[w].first, work[w].second, flt, spans, &sink, &my_rec, &e3)) { eng->fail(e3 + " (" + path + ")"); return; }
            }
        }
    };
    if (threads <= 1) worker();
    else {
        std::vector<std::thread> th;
        for (int i = 0; i < threads; ++i) th.emplace_back(worker);
        for (auto &t : th) t.join();
    }
    if (getenv("PANDEPTH_TIMING"))
        fprintf(stderr, "[timing] indexed read: %d threads, %zu ranges, %llu records, thread-seconds %.2f, inflate backend %s\n",
                threads, n_tasks, (unsigned long long)n_rec.load(), busy_us.load() / 1e6, Inflater::backend());
    return eng->ok();
}

bool decode_range(AlnReader &rd, uint64_t begin, uint64_t stop, const ReadFilter &flt, const SpanIndex &spans,
                  RunSink *sink, uint64_t *n_rec, std::string *err)
{
    if (!rd.seek(begin)) { *err = "seek failed"; return false; }
    AlnRec r;
    for (;;) {
        if (rd.tell() >= stop) break;
        const int k = rd.next(&r);
        if (k == 0) break;
        if (k < 0) { *err = rd.error(); return false; }
        ++*n_rec;
        if (!flt.pass(r)) continue;
        if (!spans.hit(r)) continue;
        emit_runs(r, sink);
    }
    return true;
}

// ---- GPU-side decode (the default for BAM input) -----------------------------------------------------------------
// The host only READS: compressed bytes go from the page cache straight into the engine's pinned batch buffers, BGZF
// member boundaries are found from the 18-byte headers, and pd_decode_submit does the rest on the device (inflate, one
// wave per member; record boundaries; filter; CIGAR walk; runs resident in HBM).  Three input shapes:
//   * indexed, whole-contig modes: the file is cut at index offsets (record boundaries) into batches of ~96 MB;
//   * indexed, GFF / BED targets: the index chunks that can hold reads overlapping a (widened) merged span (the
//     reference's multi-region iterator, PD:698-730) become the units of the batches; the region test runs on the device;
//   * no index (whole-contig modes): the file is cut at arbitrary offsets, every batch finds its first BGZF member by
//     the member signature and its first record on the device (PD_UNIT_GUESS); afterwards batch k's "next record" must
//     be batch k+1's "first record" (compared as virtual offsets), otherwise the input is decoded on the host instead.
// Units the device hands back (a record longer than the spare members, CIGARs in the CG tag) are decoded by the host
// reader.  Returns 1 done, 0 not applicable / declined (nothing was counted: the caller uses the host path), -1 error.
struct DevRange { uint64_t vbeg, vend; };                  // one unit: records starting in [vbeg, vend) (virtual offsets; vend UINT64_MAX = EOF)

int read_bam_device(const std::string &path, const Options &o, const AlnHeader &main_hdr, const SpanIndex &spans, const RegionModel &rm,
                    int kind, uint64_t first_voff, const BaiIndex *bai, bool sorted, Engine *eng)
{
    const pd_engine_api *api = eng->api;
    const auto t_enter = std::chrono::steady_clock::now();
    if (!api->decode_begin || !api->decode_acquire || !api->decode_submit || !api->decode_end || !api->decode_abort) return 0;
    if (const char *e = tune("device_decode")) if (e[0] == '0') return 0;
    if (kind != 0 && !spans.synthetic) return 0;            // the no-index span cursor (PD:4608-4646) stays on the host
    const uint64_t F = file_size(path);
    if (F < 28) return 0;
    const uint64_t SPARE = 5 * 65536;                        // bytes read past a unit's end so that its last record can finish
    // An index chunk ends where its last record ends (SAM spec 5.1.3: chunk_end is the virtual offset behind it), so a unit
    // made of chunks needs the member its end offset points into and nothing behind it; members are at most 64 KiB.  With
    // thousands of small targets the tail is most of what is read: 5 x 64 KiB behind each of 33 688 genes is 11 GB.
    // (An index whose chunk ends are not record ends: the last record runs past the unit's bytes, the unit goes to the host.)
    const bool chunk_units = kind == 0 && !spans.synthetic;
    auto spare_of = [&](uint64_t vend) -> uint64_t { return !chunk_units || vend == UINT64_MAX ? SPARE : (vend & 0xffff) ? 65536 : 0; };
    uint64_t batch_bytes = (uint64_t)32 << 20;
    if (const char *e = tune("dd_batch_mb")) batch_bytes = std::max<uint64_t>(1, strtoull(e, nullptr, 10)) << 20;
    // ---- the work list: batches of units ----
    std::vector<std::vector<DevRange>> batches;
    std::vector<BaiIndex::Chunk> orig;                       // region fetch: the index chunks as the host reader visits them
    const bool guess = kind != 0;
    if (!guess) {
        std::vector<BaiIndex::Chunk> work;
        if (!spans.synthetic) {
            for (size_t t = 0; t < spans.per_tid.size(); ++t)
                for (auto &sp : spans.per_tid[t]) bai->query((int32_t)t, sp.first, sp.second, &work);
            BaiIndex::normalise(&work);
            orig = work;
            // Chunks are disjoint in virtual offsets but neighbours share members (two exons of a gene lie in one or two
            // members), and a unit reads the whole member its end lies in: a chunk that begins within 64 KiB of the previous
            // one's end joins it.  The records in between meet no span (the index would have named them), and the span test drops
            // them like it does inside a chunk.  Fewer bytes are read and inflated, never more.
            const uint64_t unit_cap = std::max<uint64_t>(batch_bytes / 4, (uint64_t)1 << 20);
            size_t w = 0;
            for (size_t i = 0; i < work.size(); ++i) {
                if (w && work[i].second != UINT64_MAX && (work[i].first >> 16) <= (work[w - 1].second >> 16) + 65536 &&
                    (work[i].second >> 16) - (work[w - 1].first >> 16) <= unit_cap) work[w - 1].second = work[i].second;
                else work[w++] = work[i];
            }
            work.resize(w);
            if (getenv("PANDEPTH_TIMING")) fprintf(stderr, "[timing] region fetch: %zu index chunks in %zu units\n", orig.size(), work.size());
            // and a chunk larger than a batch (a target that is a whole chromosome) is cut at record starts the index names
            bool big = false;
            for (auto &c : work) if ((c.second == UINT64_MAX ? F : (c.second >> 16)) - (c.first >> 16) > batch_bytes) big = true;
            if (big) {
                const std::vector<uint64_t> starts = bai->record_starts();
                std::vector<BaiIndex::Chunk> cut;
                for (auto &c : work) {
                    uint64_t a = c.first;
                    const uint64_t stop = c.second;
                    auto it = std::upper_bound(starts.begin(), starts.end(), a);
                    while ((stop == UINT64_MAX ? F : (stop >> 16)) - (a >> 16) > batch_bytes) {
                        it = std::lower_bound(it, starts.end(), ((a >> 16) + batch_bytes / 2) << 16);
                        if (it == starts.end() || *it >= stop) break;
                        cut.emplace_back(a, *it);
                        a = *it;
                    }
                    cut.emplace_back(a, stop);
                }
                work.swap(cut);
            }
        } else {
            const std::vector<uint64_t> cuts = bai->split(first_voff, F, (int)std::min<uint64_t>(1u << 20, F / batch_bytes + 1));
            for (size_t i = 0; i + 1 < cuts.size(); ++i) work.emplace_back(cuts[i], cuts[i + 1]);
        }
        uint64_t acc = 0;
        for (auto &c : work) {
            const uint64_t sz = ((c.second == UINT64_MAX ? F : (c.second >> 16)) - (c.first >> 16)) + spare_of(c.second);
            if (batches.empty() || acc + sz > batch_bytes) { batches.emplace_back(); acc = 0; }
            batches.back().push_back(DevRange{c.first, c.second});
            acc += sz;
        }
    } else {
        for (uint64_t a = first_voff >> 16; a < F; a += batch_bytes) batches.push_back({DevRange{a == (first_voff >> 16) ? first_voff : (a << 16), UINT64_MAX}});
    }
    if (batches.empty()) return 1;
    // ---- configuration: who is counted ----
    std::vector<uint8_t> on(main_hdr.names.size(), 0);
    for (size_t t = 0; t < on.size(); ++t) on[t] = rm.has((int32_t)t) ? 1 : 0;
    std::vector<uint32_t> soff; std::vector<int32_t> sflat;
    pd_decode_cfg cfg{};
    cfg.flag_mask = o.flag_mask; cfg.min_mapq = o.min_mapq; cfg.contig_on = on.data(); cfg.sorted = sorted ? 1 : 0;
    // whole-contig statistics over windows of >= 8192 cells (mode 0's 10 Mb bins, -w >= 8192) read the sample once, in the
    // engine's compact form; every other mode needs the arrays and keeps 12-byte runs
    if (spans.synthetic && !o.site_out && (o.mode == 0 || (o.mode == 5 && o.win >= 8192))) cfg.flags |= PD_DECODE_COMPACT;
    { uint64_t b = 0; for (auto &v : batches) for (auto &r : v) b += ((r.vend == UINT64_MAX ? F : (r.vend >> 16)) - (r.vbeg >> 16)) + 65536; cfg.bytes_hint = std::min(b, 2 * F); }
    if (!spans.synthetic) {
        soff.assign(on.size() + 1, 0);
        for (size_t t = 0; t < on.size(); ++t) {
            soff[t] = (uint32_t)(sflat.size() / 2);
            if (t < spans.per_tid.size()) for (auto &sp : spans.per_tid[t]) { sflat.push_back(sp.first); sflat.push_back(sp.second); }
            on[t] = t < spans.per_tid.size() && !spans.per_tid[t].empty();
        }
        soff[on.size()] = (uint32_t)(sflat.size() / 2);
        if (sflat.empty()) sflat.push_back(0);
        cfg.span_off = soff.data(); cfg.spans = sflat.data();
    }
    int feeders = o.decode_readers > 0 ? o.decode_readers : std::min<int>(std::max(1, o.threads), 6);
    if (const char *e = tune("dd_threads")) feeders = std::max(1, atoi(e));
    if ((size_t)feeders > batches.size()) feeders = (int)batches.size();
    // A reader does not wait for the device: it queues its batch (decode_queue: copy, inflate, both record passes and the chain check between
    // them all go on the batch's stream) and reads the next one meanwhile; it holds `depth` buffers — the one it is filling and depth - 1
    // queued batches — and collects the oldest when it needs a buffer back.  (The engine has twelve batch slots.)
    // Measured (profiles/r05_decode_matrix.txt): with six readers the GPU is the bound already (some kernel runs 91 % of the phase) and a second
    // buffer per reader only puts twelve batches' working sets on the device at once — 0.75-0.81 s with one buffer each against 0.79-0.89 with
    // two on the 3e8-record file.  Fewer readers per context (a `#.list` run: four per GPU) get a second buffer each.
    int depth = api->decode_queue && api->decode_collect ? (feeders >= 6 ? 1 : 2) : 1;
    if (const char *e = tune("dd_depth")) depth = std::max(1, atoi(e));
    if (!api->decode_queue || !api->decode_collect) depth = 1;
    depth = std::max(1, std::min(depth, 12 / std::max(1, feeders)));
    {   // the largest batch the feeders will ask a buffer for
        uint64_t mx = 0;
        for (auto &v : batches) { uint64_t t = 0; for (auto &r : v) { const uint64_t a = r.vbeg >> 16, b = std::min(F, (r.vend == UINT64_MAX ? (guess ? std::min(F, a + batch_bytes) : F) : (r.vend >> 16)) + spare_of(r.vend)); t += b - a; } mx = std::max(mx, t); }
        // (one buffer per reader is pinned up front, from one thread; a reader pins its further buffers itself when it first asks for them — by
        // then the device is at work on the first batches, and page-locking costs 0.12 s per GB: twelve 32 MB buffers in a row were 46 ms before the first read)
        cfg.batch_bytes = mx + 64; cfg.batches_in_flight = (uint32_t)std::min<size_t>((size_t)tune_int("dd_pin_ahead", feeders), std::min<size_t>((size_t)feeders, batches.size()));   // (-X dd_pin_ahead=n: only n buffers pinned before the readers start)
    }
    cfg.n_batches = batches.size();
    if (const char *e = tune("lz_group")) if (api->set_param) (void)api->set_param(eng->ctx, "lz_group", (uint64_t)std::max(0, atoi(e)));             // (tuning: 0 = the parse reads its text from memory)
    for (const char *k : {"decode_fast", "decode_spoil", "decode_max_redo", "lz_mix", "decode_warm"})                                                                           // (tuning / test hooks)
        if (const char *e = tune(k)) if (api->set_param) (void)api->set_param(eng->ctx, k, (uint64_t)std::max(0, atoi(e)));
    if (const char *e = tune("inflate_waves")) if (api->set_param) (void)api->set_param(eng->ctx, "inflate_waves", (uint64_t)std::max(1, atoi(e)));   // (tuning)
    if (const char *e = tune("h2d_kernel")) if (api->set_param) (void)api->set_param(eng->ctx, "decode_h2d_kernel", (uint64_t)std::max(0, atoi(e)));      // (tuning)
    if (const char *e = tune("h2d_lanes")) if (api->set_param) (void)api->set_param(eng->ctx, "decode_h2d_lanes", (uint64_t)std::max(1, atoi(e)));        // (tuning)
    if (const char *e = tune("sync_event")) if (api->set_param) (void)api->set_param(eng->ctx, "decode_sync_event", (uint64_t)std::max(0, atoi(e)));      // (tuning: 0 = collect waits for the stream)
    if (const char *e = tune("h2d_fifo")) if (api->set_param) (void)api->set_param(eng->ctx, "decode_h2d_fifo", (uint64_t)std::max(0, atoi(e)));          // (tuning: 0 = every batch's copy on its own stream, as until round 6)
    const auto t_begin = std::chrono::steady_clock::now();
    if (!eng->ck(api->decode_begin(eng->ctx, &cfg), "pd_decode_begin")) return -1;
    if (getenv("PANDEPTH_TIMING")) fprintf(stderr, "[timing]   work list of %zu batches %.3f s, pd_decode_begin %.3f s\n", batches.size(), std::chrono::duration<double>(t_begin - t_enter).count(),
                                           std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count());

    const size_t n_batches = batches.size();
    std::atomic<size_t> next{0};
    std::atomic<uint64_t> n_dev{0}, n_host{0}, n_back{0}, us_read{0}, us_submit{0}, b_comp{0}, b_inf{0};
    std::atomic<int> declined{0};
    std::mutex why_mu; std::string decline_why;                  // (PANDEPTH_TIMING: what made the device pass give the file back)
    auto decline = [&](const std::string &why) { { std::lock_guard<std::mutex> lk(why_mu); if (decline_why.empty()) decline_why = why; } declined = 1; };
    std::vector<uint64_t> chain_first(n_batches, UINT64_MAX), chain_next(n_batches, UINT64_MAX);   // no-index: virtual offsets
    std::vector<uint64_t> key_first(n_batches, 0), key_last(n_batches, 0);                         // order of the first runs across batches
    std::vector<uint8_t> key_have(n_batches, 0);
    std::atomic<int> order_broken{0};
    std::vector<std::pair<uint64_t, uint64_t>> backlog;                                            // handed-back units: [begin, stop) virtual offsets for the host reader
    std::mutex back_mu;
    double ms_sum[4] = {0, 0, 0, 0}; std::mutex ms_mu;
    ReadFilter flt{o.flag_mask, o.min_mapq, (int32_t)main_hdr.names.size()};
    auto now_us = [] { return (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    // -X dd_trace=1 (with PANDEPTH_TIMING): a line per batch — who read it and when it was acquired, read, queued, waited for and collected (us since the
    // decode began), and the device's stage times — the raw material of tools/feeder_trace.py
    struct BatchTrace { int thread = -1; uint64_t acq = 0, rd0 = 0, rd1 = 0, queued = 0, col0 = 0, col1 = 0; float ms[4] = {0, 0, 0, 0}; };
    std::vector<BatchTrace> trace(tune("dd_trace") ? n_batches : 0);
    std::atomic<int> next_thread{0};
    std::mutex inflight_mu; std::condition_variable inflight_cv; int inflight_now = 0; const int inflight_cap = (int)tune_int("dd_inflight", 0);
    const uint64_t trace_t0 = now_us();
    auto feeder = [&]() {
        const int my_thread = next_thread.fetch_add(1);
        int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) { eng->fail("cannot open " + path); return; }
        // what a batch's answer is read against, kept from the moment the batch is queued until it is collected
        struct InFlight {
            size_t bi = 0; uint64_t ticket = 0, pos = 0, uo = 0;
            std::vector<pd_bgzf_block> blocks;
            std::vector<uint64_t> bfile;                     // file offset of every scanned member
            std::vector<pd_decode_unit> units;
            std::vector<int32_t> status;
        };
        std::deque<InFlight> fly;
        std::vector<InFlight> spare;                         // (recycled: their vectors keep their capacity)
        // the answer to one batch: counts, order keys, the record chain of a no-index stream, units handed back
        auto take = [&](InFlight &f, const pd_decode_result &res) {
            const size_t bi = f.bi;
            const std::vector<DevRange> &rs = batches[bi];
            const std::vector<pd_bgzf_block> &blocks = f.blocks; const std::vector<uint64_t> &bfile = f.bfile; const std::vector<int32_t> &status = f.status;
            n_dev += res.n_reads; b_comp += f.pos; b_inf += f.uo;
            if (res.n_first) { key_first[bi] = res.first_key; key_last[bi] = res.last_key; key_have[bi] = 1; if (res.unsorted) order_broken = 1; }
            { std::lock_guard<std::mutex> lk(ms_mu); ms_sum[0] += res.ms_h2d; ms_sum[1] += res.ms_inflate; ms_sum[2] += res.ms_walk; ms_sum[3] += res.ms_emit; }
            auto voff_of = [&](uint64_t u) -> uint64_t {      // inflated offset of the batch -> virtual file offset
                if (u == UINT64_MAX) return UINT64_MAX;
                size_t lo = 0, hi = blocks.size();
                while (hi - lo > 1) { const size_t m = (lo + hi) / 2; if (blocks[m].out_off <= u) lo = m; else hi = m; }
                // the end of a member is the start of the next one
                while (lo + 1 < blocks.size() && u >= blocks[lo].out_off + blocks[lo].out_len) ++lo;
                if (u >= blocks[lo].out_off + blocks[lo].out_len) return UINT64_MAX - 1;          // past everything this batch saw
                return (bfile[lo] << 16) | (u - blocks[lo].out_off);
            };
            if (guess) {
                if (status[0] != 0) { decline("batch " + std::to_string(bi) + ": unit status " + std::to_string(status[0])); return; }
                chain_first[bi] = bi == 0 ? first_voff : voff_of(res.first_start);
                chain_next[bi] = voff_of(res.next_start);
            } else {
                for (size_t k = 0; k < f.units.size(); ++k) {
                    if (status[k] == 0) continue;
                    // 1: the device leaves this unit to the host; 2: one of the members read for it does not inflate or fails its
                    // CRC-32 — possibly one of the spare members behind the unit that no record of it needs; 3: it could not follow
                    // the record chain.  In every case the host reader goes through exactly the bytes the unit needs and says what
                    // is wrong with them, if anything is
                    // It is only NOTED here: the pass can still be declined (a member scan that stops short in another batch, an
                    // SO:coordinate header that does not hold), and a declined pass must not have counted anything — the host reader
                    // then goes through the whole file.  The backlog is decoded after those checks, before pd_decode_end.
                    // (a unit made of several chunks goes back chunk by chunk: what lies between them is not the host reader's
                    // business, damaged or not)
                    ++n_back;
                    std::lock_guard<std::mutex> lk(back_mu);
                    if (orig.empty()) backlog.emplace_back(rs[k].vbeg, rs[k].vend);
                    else
                        for (auto it = std::upper_bound(orig.begin(), orig.end(), rs[k].vbeg, [](uint64_t v, const BaiIndex::Chunk &c) { return v < c.second; });
                             it != orig.end() && it->first < rs[k].vend; ++it)
                            backlog.emplace_back(std::max(it->first, rs[k].vbeg), std::min(it->second, rs[k].vend));
                }
            }
        };
        auto collect_oldest = [&]() {
            InFlight f = std::move(fly.front());
            fly.pop_front();
            const uint64_t t0 = now_us();
            pd_decode_result res;
            const bool ok = eng->ck(api->decode_collect(eng->ctx, f.ticket, f.status.data(), &res), "pd_decode_collect");
            us_submit += now_us() - t0;
            if (!trace.empty()) { BatchTrace &tr = trace[f.bi]; tr.col0 = t0 - trace_t0; tr.col1 = now_us() - trace_t0; tr.ms[0] = res.ms_h2d; tr.ms[1] = res.ms_inflate; tr.ms[2] = res.ms_walk; tr.ms[3] = res.ms_emit; }
            if (ok) take(f, res);
            spare.push_back(std::move(f));
        };
        for (;;) {
            while ((int)fly.size() >= depth) collect_oldest();   // (depth 1: nothing is ever queued)
            // The buffer FIRST, then the batch number: the engine hands the batches of a compact session their places in batch order, so
            // the lowest number any thread holds must always belong to a thread that also holds a buffer (pd_decode_cfg::n_batches).
            void *hb = nullptr;
            const uint64_t t_acq = now_us();
            if (!eng->ck(api->decode_acquire(eng->ctx, (size_t)cfg.batch_bytes, &hb), "pd_decode_acquire")) break;
            auto hand_back = [&](uint64_t order) { pd_decode_batch e{}; e.host_buf = hb; e.order = order; int32_t dummy = 0; api->decode_submit(eng->ctx, &e, &dummy, nullptr); };
            const size_t bi = next.fetch_add(1);
            if (bi >= n_batches) { hand_back(UINT64_MAX); break; }
            if (!eng->ok() || declined.load()) { hand_back(bi); continue; }      // (the run is being abandoned: every number is still passed on)
            const std::vector<DevRange> &rs = batches[bi];
            // bytes to read: every unit's members + spare
            std::vector<std::pair<uint64_t, uint64_t>> fr;   // file ranges
            uint64_t total = 0;
            for (auto &r : rs) {
                uint64_t a = r.vbeg >> 16, b = r.vend == UINT64_MAX ? (guess ? std::min(F, a + batch_bytes) : F) : (r.vend >> 16);
                b = std::min(F, b + spare_of(r.vend));
                if (guess && bi > 0) a = r.vbeg >> 16;       // arbitrary offset: the first member is found below
                fr.emplace_back(a, b); total += b - a;
            }
            uint8_t *buf = (uint8_t *)hb;
            const uint64_t t_a = now_us();
            InFlight f;
            if (!spare.empty()) { f = std::move(spare.back()); spare.pop_back(); }
            f.bi = bi;
            std::vector<pd_bgzf_block> &blocks = f.blocks; std::vector<uint64_t> &bfile = f.bfile; std::vector<pd_decode_unit> &units = f.units; std::vector<int32_t> &status = f.status;
            blocks.clear(); bfile.clear(); units.clear();
            uint64_t pos = 0, uo = 0; bool bad = false;
            for (size_t k = 0; k < rs.size() && !bad; ++k) {
                const uint64_t a = fr[k].first, b = fr[k].second;
                for (uint64_t got = 0; got < b - a;) {
                    const ssize_t n = pread(fd, buf + pos + got, (size_t)(b - a - got), (off_t)(a + got));
                    if (n <= 0) { eng->fail("read error on " + path); bad = true; break; }
                    got += (uint64_t)n;
                }
                if (bad) break;
                const uint8_t *w = buf + pos; const uint64_t wn = b - a;
                uint64_t p = 0;
                if (guess && bi > 0) {
                    // the first member at or after this arbitrary offset: the member signature, confirmed by the two that follow
                    bool found = false;
                    for (; p + 18 <= wn && p < 3 * 65536; ++p) {
                        if (w[p] != 0x1f || w[p + 1] != 0x8b || w[p + 2] != 8 || !(w[p + 3] & 4)) continue;
                        uint64_t q = p; int okc = 0;
                        for (; okc < 3 && q + 18 <= wn; ++okc) { uint32_t d0 = 0; const uint32_t bs = bgzf_block_size(w + q, wn - q, &d0); if (!bs) break; q += bs; }
                        if (okc == 3 || (okc > 0 && q + 18 > wn)) { found = true; break; }
                    }
                    if (!found) { decline("batch " + std::to_string(bi) + ": no BGZF member found at the guessed offset"); bad = true; break; }
                }
                const size_t b0 = blocks.size();
                const uint64_t own_end = guess ? std::min(F, (rs[k].vbeg >> 16) + batch_bytes) : 0;      // members starting before this belong to the batch
                long stop_blk = -1;
                for (; p + 18 <= wn;) {
                    uint32_t doff = 0;
                    const uint32_t bs = bgzf_block_size(w + p, wn - p, &doff);
                    if (bs == 0 || p + bs > wn) break;       // partial member at the end of the window
                    const uint32_t isize = w[p + bs - 4] | (w[p + bs - 3] << 8) | (w[p + bs - 2] << 16) | ((uint32_t)w[p + bs - 1] << 24);
                    if (guess && stop_blk < 0 && a + p >= own_end) stop_blk = (long)blocks.size();
                    bfile.push_back(a + p);
                    blocks.push_back(pd_bgzf_block{pos + p + doff, uo, bs - doff - 8, isize});
                    uo += isize; p += bs;
                }
                {   // the member scan must have covered what this unit owns: stopping earlier means a member header that is not one
                    // (or a file that ends inside a member) — the host reader goes through such a file and says what is wrong
                    const uint64_t need = guess ? own_end : (rs[k].vend == UINT64_MAX ? F : std::min(F, rs[k].vend >> 16));
                    if (a + p < need) { decline("batch " + std::to_string(bi) + ": the member scan stopped short"); bad = true; break; }
                }
                if (blocks.size() == b0) { if (guess) { continue; } eng->fail("index offsets of " + path + " do not match its BGZF blocks"); bad = true; break; }
                pd_decode_unit un{};
                un.first_block = (uint32_t)b0; un.n_blocks = (uint32_t)(blocks.size() - b0);
                un.avail = uo;
                if (guess) {
                    un.start = blocks[b0].out_off + (bi == 0 ? (rs[k].vbeg & 0xffff) : 0);
                    un.flags = bi == 0 ? 0 : PD_UNIT_GUESS;
                    un.stop = stop_blk >= 0 ? blocks[(size_t)stop_blk].out_off : uo;
                } else {
                    if (bfile[b0] != (rs[k].vbeg >> 16)) { eng->fail("index offsets of " + path + " do not match its BGZF blocks"); bad = true; break; }
                    un.start = blocks[b0].out_off + (rs[k].vbeg & 0xffff);
                    if (rs[k].vend == UINT64_MAX) un.stop = uo;
                    else {
                        const auto it = std::lower_bound(bfile.begin() + (long)b0, bfile.end(), rs[k].vend >> 16);
                        if (it == bfile.end() || *it != (rs[k].vend >> 16)) {
                            if ((rs[k].vend >> 16) >= F || (rs[k].vend & 0xffff) == 0) un.stop = uo;     // an offset at the very end of the file
                            else { eng->fail("index offsets of " + path + " do not match its BGZF blocks"); bad = true; break; }
                        } else un.stop = blocks[(size_t)(it - bfile.begin())].out_off + (rs[k].vend & 0xffff);
                    }
                }
                if (un.start > un.stop) un.start = un.stop;
                units.push_back(un);
                pos += wn;
            }
            const uint64_t t_b = now_us();
            us_read += t_b - t_a;
            if (bad || units.empty()) {
                // nothing to do for this batch: hand the buffer back with an empty submit (the other threads go on passing their numbers)
                hand_back(bi);
                continue;
            }
            status.assign(units.size(), 0);
            f.pos = pos; f.uo = uo;
            pd_decode_batch bt{}; bt.host_buf = hb; bt.n_bytes = (size_t)pos; bt.blocks = blocks.data(); bt.n_blocks = (uint32_t)blocks.size();
            bt.inflated_bytes = uo; bt.units = units.data(); bt.n_units = (uint32_t)units.size(); bt.order = bi;
            if (!trace.empty()) { BatchTrace &tr = trace[bi]; tr.thread = my_thread; tr.acq = t_acq - trace_t0; tr.rd0 = t_a - trace_t0; tr.rd1 = t_b - trace_t0; }
            if (depth > 1) {
                const bool ok = eng->ck(api->decode_queue(eng->ctx, &bt, &f.ticket), "pd_decode_queue");
                us_submit += now_us() - t_b;
                if (!trace.empty()) trace[bi].queued = now_us() - trace_t0;
                if (ok) fly.push_back(std::move(f)); else spare.push_back(std::move(f));      // (the loop's head passes the remaining numbers on)
                continue;
            }
            pd_decode_result res;
            // (-X dd_inflight=n: at most n batches on the device at a time, the other readers read meanwhile — an experiment in keeping the device's
            // batches in flight constant: tools/calls/r6_call24.sh)
            if (inflight_cap > 0) { std::unique_lock<std::mutex> lk(inflight_mu); inflight_cv.wait(lk, [&] { return inflight_now < inflight_cap; }); ++inflight_now; }
            const uint64_t t_sub = now_us();
            const bool ok = eng->ck(api->decode_submit(eng->ctx, &bt, status.data(), &res), "pd_decode_submit");
            if (inflight_cap > 0) { { std::lock_guard<std::mutex> lk(inflight_mu); --inflight_now; } inflight_cv.notify_one(); }
            us_submit += now_us() - t_b;
            if (!trace.empty()) { BatchTrace &tr = trace[bi]; tr.queued = tr.col0 = t_sub - trace_t0; tr.col1 = now_us() - trace_t0; tr.ms[0] = res.ms_h2d; tr.ms[1] = res.ms_inflate; tr.ms[2] = res.ms_walk; tr.ms[3] = res.ms_emit; }
            if (ok) take(f, res);
            spare.push_back(std::move(f));
        }
        while (!fly.empty()) collect_oldest();
        ::close(fd);
    };
    {
        std::vector<std::thread> th;
        for (int i = 1; i < feeders; ++i) th.emplace_back(feeder);
        feeder();
        for (auto &t : th) t.join();
    }
    if (!trace.empty() && getenv("PANDEPTH_TIMING"))
        for (size_t k = 0; k < trace.size(); ++k) {
            const BatchTrace &tr = trace[k];
            fprintf(stderr, "[trace] batch %zu thread %d acquire %llu read %llu %llu queued %llu collect %llu %llu device ms h2d %.3f inflate %.3f walk %.3f emit %.3f\n", k, tr.thread,
                    (unsigned long long)tr.acq, (unsigned long long)tr.rd0, (unsigned long long)tr.rd1, (unsigned long long)tr.queued, (unsigned long long)tr.col0, (unsigned long long)tr.col1,
                    tr.ms[0], tr.ms[1], tr.ms[2], tr.ms[3]);
        }
    if (guess && eng->ok() && !declined.load()) {
        // the record chain across the batches (virtual offsets; the end of a member equals the start of the next one)
        for (size_t k = 0; k + 1 < n_batches; ++k)
            if (chain_next[k] != chain_first[k + 1] || chain_next[k] >= UINT64_MAX - 1) {
                char b[160]; snprintf(b, sizeof b, "the record chain of batch %zu ends at %llx, batch %zu starts at %llx", k, (unsigned long long)chain_next[k], k + 1, (unsigned long long)chain_first[k + 1]);
                decline(b); break;
            }
    }
    if (sorted && eng->ok() && !declined.load()) {
        // the header said SO:coordinate; a file whose records are not in that order is read on the host, the reference's way
        // (its no-index cursor, PD:4604-4671, depends on the order the records come in)
        uint64_t prev = 0; bool have = false;
        for (size_t k = 0; k < n_batches && !order_broken.load(); ++k) {
            if (!key_have[k]) continue;
            if (have && key_first[k] < prev) order_broken = 1;
            prev = key_last[k]; have = true;
        }
        if (order_broken.load()) decline("the records are not in the order SO:coordinate promises");
    }
    if (getenv("PANDEPTH_TIMING"))
        fprintf(stderr, "[timing] device decode: %zu batches (%s), %d feeders holding %d buffers each, %llu records on the device, %llu units handed back (%llu records on the host)%s; "
                        "feeder thread-seconds: read+scan %.2f, submit %.2f; device ms summed over batches: H2D %.1f, inflate %.1f, walk %.1f, emit %.1f; "
                        "bytes: compressed %llu, inflated %llu\n",
                n_batches, guess ? "no index: guessed starts" : spans.synthetic ? "index cuts" : "index chunks of the targets", feeders, depth,
                (unsigned long long)n_dev.load(), (unsigned long long)n_back.load(), (unsigned long long)n_host.load(), declined.load() ? (" — DECLINED (" + decline_why + ")").c_str() : "",
                us_read.load() / 1e6, us_submit.load() / 1e6, ms_sum[0], ms_sum[1], ms_sum[2], ms_sum[3], (unsigned long long)b_comp.load(), (unsigned long long)b_inf.load());
    if (!eng->ok()) { api->decode_abort(eng->ctx); return -1; }
    if (declined.load()) { api->decode_abort(eng->ctx); return 0; }          // nothing of this input has been counted
    if (!backlog.empty()) {
        // the units the device handed back, now that the pass stands: the host reader goes through exactly the bytes each of
        // them needs and says what is wrong with them, if anything is
        std::sort(backlog.begin(), backlog.end());
        std::atomic<size_t> nb{0};
        auto host_worker = [&]() {
            AlnReader rd; std::string e2;
            if (!rd.open(path, &e2)) { eng->fail(e2); return; }
            RunSink sink(eng);
            for (;;) {
                const size_t i = nb.fetch_add(1);
                if (i >= backlog.size() || !eng->ok()) break;
                uint64_t nr = 0;
                if (!decode_range(rd, backlog[i].first, backlog[i].second, flt, spans, &sink, &nr, &e2)) { eng->fail(e2 + " (" + path + ")"); break; }
                n_host += nr;
            }
        };
        const int nt = (int)std::min<size_t>((size_t)std::max(1, o.threads), backlog.size());
        std::vector<std::thread> th;
        for (int i = 1; i < nt; ++i) th.emplace_back(host_worker);
        host_worker();
        for (auto &t : th) t.join();
        if (!eng->ok()) { api->decode_abort(eng->ctx); return -1; }
    }
    const uint64_t t_e = now_us();
    if (!eng->ck(api->decode_end(eng->ctx), "pd_decode_end")) return -1;
    if (getenv("PANDEPTH_TIMING")) fprintf(stderr, "[timing]   pd_decode_end (concatenate the batches' runs) %.3f s\n", (now_us() - t_e) / 1e6);
    return 1;
}

// No index, header says SO:coordinate (PD:4604-4671): one cursor per contig over its merged spans.
bool read_sorted_stream(AlnReader *rd, const Options &o, const AlnHeader &main_hdr, const RegionModel &rm, Engine *eng)
{
    const int32_t n = (int32_t)main_hdr.names.size();
    std::vector<char> done(n, 1);
    std::vector<const std::vector<std::pair<int32_t, int32_t>> *> sp(n, nullptr);
    std::vector<size_t> cur(n, 0);
    for (auto &kv : rm.merged) if (kv.first >= 0 && kv.first < n) { done[kv.first] = 0; sp[kv.first] = &kv.second; }
    RunSink sink(eng);
    AlnRec r;
    int k;
    while ((k = rd->next(&r)) > 0) {
        if (r.tid < 0 || r.tid >= n) continue;       // the reference indexes EndChr[tid] here (UB for tid = -1)
        if (done[r.tid]) continue;
        if ((int)r.mapq < o.min_mapq) continue;
        if (r.flag & o.flag_mask) continue;
        const auto &v = *sp[r.tid];
        if (r.endpos() < v[cur[r.tid]].first) continue;
        if (r.pos > v[cur[r.tid]].second) {
            size_t c = cur[r.tid] + 1;
            while (c < v.size() && !(r.pos <= v[c].second)) ++c;
            cur[r.tid] = c;
            if (c == v.size()) {
                done[r.tid] = 1;
                bool all = true;
                for (int32_t i = 0; i < n; ++i) if (!done[i]) { all = false; break; }
                if (all) break;                      // this read is NOT counted (PD:4641-4644)
            }
        }
        emit_runs(r, &sink);                         // counted even when the cursor just ran off the end
    }
    if (k < 0) eng->fail(rd->error());
    return eng->ok();
}

// No index, not coordinate sorted (PD:4677-4711): every read that passes the filter.
bool read_all(AlnReader *rd, const Options &o, const AlnHeader &main_hdr, const RegionModel &rm, Engine *eng)
{
    ReadFilter flt{o.flag_mask, o.min_mapq, (int32_t)main_hdr.names.size()};
    RunSink sink(eng);
    AlnRec r;
    int k;
    while ((k = rd->next(&r)) > 0) {
        if (!flt.pass(r)) continue;
        if (!rm.has(r.tid)) continue;                // depth on contigs without targets is never reported
        emit_runs(r, &sink);
    }
    if (k < 0) eng->fail(rd->error());
    return eng->ok();
}

// ---- output ------------------------------------------------------------------------------------
inline char *put_u32(char *p, uint32_t v)
{
    char t[12]; int n = 0;
    do { t[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *p++ = t[--n];
    return p;
}

// formats cells [b, b+n) of one contig as "name\tindex\tdepth\n" (PD:4278-4281: the index is 0-based)
void format_sites(const std::string &nm, uint32_t b, const uint32_t *d, size_t n, std::string *out)
{
    out->resize(n * (nm.size() + 24));
    char *p0 = &(*out)[0], *p = p0;
    for (size_t j = 0; j < n; ++j) {
        memcpy(p, nm.data(), nm.size()); p += nm.size();
        *p++ = '\t'; p = put_u32(p, b + (uint32_t)j); *p++ = '\t'; p = put_u32(p, d[j]); *p++ = '\n';
    }
    out->resize((size_t)(p - p0));
}

// Stage 1 of the byte-identical gzip streams (zlib's LZ77 parse) on the engine: pgz hands over the chunks of a round, the
// engine returns zlib's symbols (pd_deflate_parse).  -X device_deflate=0: zlib parses on the host threads as before.
pgz::ParseFn engine_parse(Engine *eng)
{
    if (!eng->api->deflate_parse) return nullptr;
    if (const char *e = tune("device_deflate")) if (e[0] == '0') return nullptr;
    return [eng](const uint8_t *text, size_t n, const uint64_t *chunks, size_t n_chunks, pgz::SymVec &syms, std::vector<uint64_t> &off) -> bool {
        static_assert(sizeof(pd_lz_chunk) == 3 * sizeof(uint64_t), "pd_lz_chunk is a (start, end, origin) triple");
        size_t bytes = 0;
        for (size_t k = 0; k < n_chunks; ++k) bytes += (size_t)(chunks[3 * k + 1] - chunks[3 * k]);
        off.assign(n_chunks + 1, 0);
        const auto t0 = std::chrono::steady_clock::now();
        // a symbol per 3 bytes is plenty for tables (they parse to a symbol per 7-8 bytes); text that needs more gets the full size
        int rc = PD_ERANGE;
        for (size_t cap : {bytes / 3 + 4096, bytes + 16}) {
            syms.resize(cap);
            rc = eng->api->deflate_parse(eng->ctx, text, n, reinterpret_cast<const pd_lz_chunk *>(chunks), (uint32_t)n_chunks, syms.data(), cap, off.data());
            if (rc != PD_ERANGE) break;
        }
        if (getenv("PANDEPTH_TIMING"))
            fprintf(stderr, "[timing]   pd_deflate_parse: %zu chunks of %.1f MB of text in %.3f s%s\n", n_chunks, n / 1e6,
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), rc ? " — FAILED, zlib parses these chunks" : "");
        return rc == 0;                                          // (a failure leaves the chunks to zlib on the host threads)
    };
}

// <prefix>.SiteDepth.gz (PD:4264-4284) with its text resident on the device (pd_text_*): the engine formats the rows, parses them
// the way zlib would (stage 1) and check-sums them where the cells are; the host receives symbols and CRCs, cuts blocks, builds the
// Huffman codes and writes the bits (pgz::Stream with a Remote source).  Only the stream's last chunk comes back as text (zlib
// parses it itself).  Returns 1 done, 0 declined (nothing usable written: the caller takes the host-text path), -1 error.
int write_site_depth_resident(const std::string &path, const AlnHeader &hdr, const RegionModel &rm, Engine *eng, int threads)
{
    const pd_engine_api *api = eng->api;
    if (!api->text_open || !api->text_close || !api->text_append_sites || !api->text_parse || !api->text_read || !api->text_release) return 0;
    if (const char *e = tune("device_deflate")) if (e[0] == '0') return 0;
    if (const char *e = tune("site_resident")) if (e[0] == '0') return 0;
    const bool timing = getenv("PANDEPTH_TIMING") != nullptr;
    const auto t_enter = std::chrono::steady_clock::now();
    pgz::Params prm = pgz::Params::for_device(nullptr);
    prm.calls = (unsigned)tune_int("lz_calls", 2);            // (a round's parse calls in flight: four measured the same as two — the parse kernels of a round take turns on the CUs' LDS either way, tools/calls/r6_call22.sh)
    if (api->set_param) (void)api->set_param(eng->ctx, "lz_slots", prm.calls >= 4 ? 4 : 2);
    const size_t ring = std::max<size_t>((size_t)1 << 30, 4 * prm.batch);
    pd_text *tx = nullptr;
    if (api->text_open(eng->ctx, ring, &tx) != 0 || !tx) return 0;
    struct Closer { const pd_engine_api *api; pd_text *t; ~Closer() { if (t && !getenv("PANDEPTH_KEEP_CONTEXT")) api->text_close(t); } } closer{api, tx};
    FILE *fp = fopen(path.c_str(), "wb");
    if (!fp) { std::cerr << "open OUT File error: " << path << std::endl; return -1; }
    bool io_ok = true;
    int rc = 1;
    double t_parse = 0, t_append = 0; size_t n_calls = 0;
    std::mutex tm_mu;
    pgz::Remote src;
    src.parse = [&](uint64_t off, size_t n, const uint64_t *chunks, size_t n_chunks, pgz::SymVec &syms, std::vector<uint64_t> &soff, uint32_t *crc, uint64_t crc_span) -> bool {
        size_t bytes = 0;
        for (size_t k = 0; k < n_chunks; ++k) bytes += (size_t)(chunks[3 * k + 1] - chunks[3 * k]);
        soff.assign(n_chunks + 1, 0);
        const auto t0 = std::chrono::steady_clock::now();
        int r = PD_ERANGE;
        for (size_t cap : {bytes / 3 + 4096, bytes + 16}) {
            if (syms.size() < cap) syms.resize(cap);
            r = api->text_parse(tx, off, n, reinterpret_cast<const pd_lz_chunk *>(chunks), (uint32_t)n_chunks, syms.data(), syms.size(), soff.data(), crc, crc_span);
            if (r != PD_ERANGE) break;
        }
        { std::lock_guard<std::mutex> lk(tm_mu); t_parse += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); ++n_calls; }
        // (a call that fails leaves its chunks to zlib: pgz::Stream fetches their text and parses them on the host threads)
        if (r != 0 && timing) fprintf(stderr, "[timing]   pd_text_parse failed (%d: %s): zlib parses these chunks\n", r, api->strerror(eng->ctx));
        return r == 0;
    };
    src.fetch = [&](uint64_t off, size_t n, uint8_t *dst) { return api->text_read(tx, off, n, dst) == 0; };
    src.release = [&](uint64_t off) { (void)api->text_release(tx, off); };
    {
        std::unique_ptr<pgz::Stream> st(new pgz::Stream(threads, [&](const uint8_t *b, size_t n) { io_ok = fwrite(b, 1, n, fp) == n && io_ok; return io_ok; }, prm, src));
        const size_t CH = (size_t)2 << 20;                 // cells per append: 30-60 MB of rows
        for (size_t t = 0; t < hdr.names.size() && rc == 1; ++t) {
            if (!rm.has((int32_t)t)) continue;
            const uint32_t len = hdr.lens[t];
            const std::string &nm = hdr.names[t];
            for (uint32_t b = 0; b < len && rc == 1; b += (uint32_t)CH) {
                if (eng->cancel.load()) { rc = -1; break; }
                const size_t n = std::min<size_t>(CH, len - b);
                uint64_t got = 0;
                const auto t0 = std::chrono::steady_clock::now();
                int r = api->text_append_sites(tx, (int32_t)t, b, n, nm.data(), nm.size(), &got);
                if (r == PD_ERANGE) {                       // the ring is full: what the round in flight holds is released when it is over
                    if (!st->wait_idle()) { rc = io_ok ? 0 : -1; break; }
                    r = api->text_append_sites(tx, (int32_t)t, b, n, nm.data(), nm.size(), &got);
                }
                t_append += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (r != 0) { rc = r == PD_ERANGE ? 0 : (eng->ck(r, "pd_text_append_sites"), -1); break; }
                if (!st->announce(got)) rc = io_ok ? 0 : -1;
            }
        }
        if (rc == 1 && !st->finish()) rc = io_ok ? 0 : -1;
        if (rc != 1) (void)st->wait_idle();
        if (timing)
            fprintf(stderr, "[timing]   per-site writer (text resident on the device): appends %.3f s; %zu parse calls, %.3f s in them (two at a time); %.3f s in all\n", t_append,
                    n_calls, t_parse, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_enter).count());
        if (getenv("PANDEPTH_KEEP_CONTEXT")) (void)st.release();   // (the process is about to end, main.cpp)
    }
    if (fclose(fp) != 0 && rc == 1) rc = -1;
    return rc;                                                   // (0 = declined, and nothing here has touched the run's error text: the host-text path starts the file over)
}

// The `-w` table (PD:4366-4389) with its text resident on the device: the engine formats the rows from the window statistics its
// last window call left in HBM (pd_text_append_window_rows), parses and check-sums them there; header and total lines go in as bytes.
// The finished gzip stream is written through `out` (GzWriter::raw).  Returns 1 done, 0 declined (nothing left in `out`), -1 error.
struct TableContig { int32_t tid; size_t n_rows; const std::string *name; };
int write_window_table_resident(GzWriter &out, Engine *eng, int threads, uint32_t w, const std::string &header, const std::vector<TableContig> &contigs,
                                const std::string &footer_line)
{
    const pd_engine_api *api = eng->api;
    if (!api->text_open || !api->text_close || !api->text_append_window_rows || !api->text_append_bytes || !api->text_parse || !api->text_read ||
        !api->text_release || !out.collecting())
        return 0;
    if (const char *e = tune("device_deflate")) if (e[0] == '0') return 0;
    if (const char *e = tune("table_resident")) if (e[0] == '0') return 0;
    size_t rows = 0;
    for (const auto &tc : contigs) rows += tc.n_rows;
    size_t min_rows = 100000;                                  // (small tables: the host formats them in milliseconds)
    if (const char *e = tune("table_resident_min")) min_rows = (size_t)strtoull(e, nullptr, 10);
    if (rows < min_rows) return 0;
    const auto t_enter = std::chrono::steady_clock::now();
    pgz::Params prm = pgz::Params::for_device(nullptr);
    pd_text *tx = nullptr;
    if (api->text_open(eng->ctx, std::max<size_t>((size_t)1 << 30, 4 * prm.batch), &tx) != 0 || !tx) return 0;
    struct Closer { const pd_engine_api *api; pd_text *t; ~Closer() { if (t && !getenv("PANDEPTH_KEEP_CONTEXT")) api->text_close(t); } } closer{api, tx};
    bool io_ok = true;
    int rc = 1;
    pgz::Remote src;
    src.parse = [&](uint64_t off, size_t n, const uint64_t *chunks, size_t n_chunks, pgz::SymVec &syms, std::vector<uint64_t> &soff, uint32_t *crc, uint64_t crc_span) -> bool {
        size_t bytes = 0;
        for (size_t k = 0; k < n_chunks; ++k) bytes += (size_t)(chunks[3 * k + 1] - chunks[3 * k]);
        soff.assign(n_chunks + 1, 0);
        int r = PD_ERANGE;
        for (size_t cap : {bytes / 3 + 4096, bytes + 16}) {
            if (syms.size() < cap) syms.resize(cap);
            r = api->text_parse(tx, off, n, reinterpret_cast<const pd_lz_chunk *>(chunks), (uint32_t)n_chunks, syms.data(), syms.size(), soff.data(), crc, crc_span);
            if (r != PD_ERANGE) break;
        }
        return r == 0;
    };
    src.fetch = [&](uint64_t off, size_t n, uint8_t *dst) { return api->text_read(tx, off, n, dst) == 0; };
    src.release = [&](uint64_t off) { (void)api->text_release(tx, off); };
    {
        std::unique_ptr<pgz::Stream> st(new pgz::Stream(threads, [&](const uint8_t *b, size_t n) { io_ok = out.raw(b, n) && io_ok; return io_ok; }, prm, src));
        auto bytes_in = [&](const std::string &s) {
            if (rc != 1 || s.empty()) return;
            int r = api->text_append_bytes(tx, s.data(), s.size());
            if (r == PD_ERANGE) { if (!st->wait_idle()) { rc = io_ok ? 0 : -1; return; } r = api->text_append_bytes(tx, s.data(), s.size()); }
            if (r != 0) { rc = 0; return; }
            if (!st->announce(s.size())) rc = io_ok ? 0 : -1;
        };
        bytes_in(header);
        const size_t STEP = (size_t)1 << 20;                    // rows per append: about 45 MB of text
        for (const auto &tc : contigs) {
            for (size_t k = 0; k < tc.n_rows && rc == 1; k += STEP) {
                const size_t n = std::min(STEP, tc.n_rows - k);
                uint64_t got = 0;
                int r = api->text_append_window_rows(tx, tc.tid, w, k, n, tc.name->data(), tc.name->size(), &got);
                if (r == PD_ERANGE) { if (!st->wait_idle()) { rc = io_ok ? 0 : -1; break; } r = api->text_append_window_rows(tx, tc.tid, w, k, n, tc.name->data(), tc.name->size(), &got); }
                if (r != 0) { rc = 0; break; }                  // (e.g. the statistics were summed over several GPUs and are not on this device)
                if (!st->announce(got)) rc = io_ok ? 0 : -1;
            }
            if (rc != 1) break;
        }
        bytes_in(footer_line);
        if (rc == 1 && !st->finish()) rc = io_ok ? 0 : -1;
        if (rc != 1) (void)st->wait_idle();
        if (getenv("PANDEPTH_TIMING"))
            fprintf(stderr, "[timing]   window table (text resident on the device): %zu rows, %s, %.3f s\n", rows, rc == 1 ? "written" : "declined",
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - t_enter).count());
        if (getenv("PANDEPTH_KEEP_CONTEXT")) (void)st.release();
    }
    if (rc != 1 && !out.raw_rewind()) return -1;
    return rc;
}

// <prefix>.SiteDepth.gz (PD:4264-4284), byte-identical to the reference's single zlib stream at any size, on all
// threads: 4 M-cell blocks are read back, formatted in parallel slices and fed, in order, to pgz::Stream
// (host/pgzip.h), which deflates them with zlib's own parse spread over the threads and bounded memory.
// Returns 1 done, 0 the parallel form declined (nothing usable written), -1 error.
int write_site_depth_identical(const std::string &path, const AlnHeader &hdr, const RegionModel &rm, Engine *eng, int threads, bool device_parse)
{
    FILE *fp = fopen(path.c_str(), "wb");
    if (!fp) { std::cerr << "open OUT File error: " << path << std::endl; return -1; }
    bool io_ok = true;
    int rc = 1;
    const auto t_enter = std::chrono::steady_clock::now();
    auto t_leave = t_enter;
    {
        const pgz::ParseFn dev_parse = device_parse ? engine_parse(eng) : nullptr;
        pgz::Params prm = dev_parse ? pgz::Params::for_device(dev_parse) : pgz::Params();
        if (dev_parse && eng->api->host_register && eng->api->host_unregister) {
            // the stream's two text buffers are handed to pd_deflate_parse round after round: page-locked, the copy runs at the link's rate
            prm.pin = [eng](void *p, size_t n) { return eng->api->host_register(eng->ctx, p, n) == 0; };
            prm.unpin = [eng](void *p) { (void)eng->api->host_unregister(eng->ctx, p); };
        }
        std::unique_ptr<pgz::Stream> st_own(new pgz::Stream(threads, [&](const uint8_t *b, size_t n) { io_ok = fwrite(b, 1, n, fp) == n && io_ok; return io_ok; }, prm));
        pgz::Stream &st = *st_own;
        // producer: read-back + formatting of the next blocks (a quarter of the threads) while the consumer deflates
        const size_t CH = (size_t)4 << 20;
        const int nt = std::max(1, threads / 4);
        std::mutex mu;
        std::condition_variable cv;
        std::deque<std::vector<std::string>> q;              // each entry: one block's slices, in order
        struct Block { std::unique_ptr<char[]> p; size_t n = 0, cap = 0; };
        std::deque<Block> qb; size_t qb_bytes = 0;           // ... or one block of text formatted by the engine
        bool done = false, stop = false, prod_ok = true;
        double t_format = 0, t_prod_wait = 0, t_cons_wait = 0, t_write = 0;      // PANDEPTH_TIMING: where the two threads spend the phase
        std::thread producer([&] {
            std::vector<uint32_t> d(CH);
            for (size_t t = 0; t < hdr.names.size(); ++t) {
                if (!rm.has((int32_t)t)) continue;
                const uint32_t len = hdr.lens[t];
                for (uint32_t b = 0; b < len; b += (uint32_t)CH) {
                    if (eng->cancel.load()) { prod_ok = false; goto out; }
                    const size_t n = std::min<size_t>(CH, len - b);
                    if (eng->api->format_sites) {
                        // the engine formats the rows where the cells are; one block of text comes back (into a buffer that is
                        // not zero-filled first: 112 MB per block; the queue may hold a round's worth of blocks so that the
                        // formatting runs ahead of the deflate rounds)
                        const std::string &nm = hdr.names[t];
                        Block blk;
                        blk.cap = n * (nm.size() + 23);
                        blk.p.reset(new char[blk.cap]);
                        size_t got = 0;
                        const auto tf0 = std::chrono::steady_clock::now();
                        if (!eng->ck(eng->api->format_sites(eng->ctx, (int32_t)t, b, n, nm.data(), nm.size(), blk.p.get(), blk.cap, &got), "pd_format_sites")) { prod_ok = false; goto out; }
                        blk.n = got;
                        const auto tf1 = std::chrono::steady_clock::now();
                        t_format += std::chrono::duration<double>(tf1 - tf0).count();
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return stop || qb_bytes < ((size_t)256 << 20); });
                        t_prod_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - tf1).count();
                        if (stop) goto out;
                        qb_bytes += blk.n;
                        qb.push_back(std::move(blk));
                        lk.unlock();
                        cv.notify_all();
                        continue;
                    }
                    if (!eng->ck(eng->api->read_depth(eng->ctx, (int32_t)t, b, n, d.data()), "pd_read_depth")) { prod_ok = false; goto out; }
                    {
                        std::vector<std::string> parts((size_t)nt);
                        const size_t per = (n + (size_t)nt - 1) / (size_t)nt;
                        std::vector<std::thread> th;
                        for (int k = 1; k < nt; ++k) {
                            const size_t lo = std::min(n, per * (size_t)k), hi = std::min(n, lo + per);
                            th.emplace_back([&, k, lo, hi] { format_sites(hdr.names[t], b + (uint32_t)lo, d.data() + lo, hi - lo, &parts[(size_t)k]); });
                        }
                        format_sites(hdr.names[t], b, d.data(), std::min(n, per), &parts[0]);
                        for (auto &x : th) x.join();
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return stop || q.size() < 2; });
                        if (stop) goto out;
                        q.push_back(std::move(parts));
                    }
                    cv.notify_all();
                }
            }
        out:
            { std::lock_guard<std::mutex> lk(mu); done = true; }
            cv.notify_all();
        });
        for (;;) {
            std::vector<std::string> parts;
            Block blk;
            const auto tw0 = std::chrono::steady_clock::now();
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return done || !q.empty() || !qb.empty(); });
                if (q.empty() && qb.empty()) break;
                if (!qb.empty()) { blk = std::move(qb.front()); qb.pop_front(); qb_bytes -= blk.n; }
                else { parts = std::move(q.front()); q.pop_front(); }
            }
            cv.notify_all();
            const auto tw1 = std::chrono::steady_clock::now();
            t_cons_wait += std::chrono::duration<double>(tw1 - tw0).count();
            if (rc == 1 && blk.n && !st.write(blk.p.get(), blk.n)) rc = io_ok ? 0 : -1;
            t_write += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw1).count();
            for (auto &p : parts)
                if (rc == 1 && !p.empty() && !st.write(p.data(), p.size())) rc = io_ok ? 0 : -1;
            if (rc != 1) { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); break; }
        }
        producer.join();
        if (!prod_ok) rc = -1;
        const auto tfin = std::chrono::steady_clock::now();
        if (rc == 1 && !st.finish()) rc = io_ok ? 0 : -1;
        if (getenv("PANDEPTH_TIMING"))
            fprintf(stderr, "[timing]   per-site writer: producer formatting %.3f s + waiting for room %.3f s; consumer waiting for text %.3f s, "
                            "in the stream (copy + deflate rounds) %.3f s, finishing %.3f s\n", t_format, t_prod_wait, t_cons_wait, t_write,
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - tfin).count());
        t_leave = std::chrono::steady_clock::now();
        // (the process is about to end, main.cpp: returning 0.7 GB of buffers page by page would only delay that)
        if (getenv("PANDEPTH_KEEP_CONTEXT")) (void)st_own.release();
    }
    const auto t_torn = std::chrono::steady_clock::now();
    if (fclose(fp) != 0 && rc == 1) rc = -1;
    if (getenv("PANDEPTH_TIMING"))
        fprintf(stderr, "[timing]   per-site writer: %.3f s in all; releasing the stream's buffers %.3f s, closing the file %.3f s\n",
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t_enter).count(), std::chrono::duration<double>(t_torn - t_leave).count(),
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t_torn).count());
    return rc;
}

// <prefix>.SiteDepth.gz: the byte-identical parallel stream above; if pgz declines (text it does not
// re-state, which depth tables never produced in testing) the file is started over — as ONE zlib stream
// below PANDEPTH_SITE_PARALLEL_MIN bytes of text (default 256 MiB), as concatenated gzip members produced by
// the reader threads above it (same decompressed bytes, different .gz bytes).  PANDEPTH_SITE_IDENTICAL=0
// skips the first attempt.
bool write_site_depth(const std::string &path, const AlnHeader &hdr, const RegionModel &rm, Engine *eng, int threads)
{
    const char *ident = tune("site_identical");
    if (threads > 1 && !(ident && ident[0] == '0')) {
        // the text on the device; else the text on the host with the engine's parse; else zlib's own parse on the host threads, whose
        // chunks are large (1 MiB + 64 KiB of overlap): a text whose parses do not meet inside the small chunks' overlap ends up there
        int r = write_site_depth_resident(path, hdr, rm, eng, threads);
        if (r == 0) r = write_site_depth_identical(path, hdr, rm, eng, threads, true);
        if (r == 0 && engine_parse(eng)) r = write_site_depth_identical(path, hdr, rm, eng, threads, false);
        if (r == 1) return true;
        if (r < 0) return false;
    }
    uint64_t estimate = 0;
    for (size_t t = 0; t < hdr.names.size(); ++t)
        if (rm.has((int32_t)t)) estimate += (uint64_t)hdr.lens[t] * (hdr.names[t].size() + 12);
    uint64_t par_min = (uint64_t)256 << 20;
    if (const char *e = tune("site_parallel_min")) par_min = strtoull(e, nullptr, 10);
    const size_t CH = (size_t)4 << 20;
    if (estimate >= par_min && threads > 1) {
        ParallelGzWriter out;
        if (!out.open(path, threads)) { std::cerr << "open OUT File error: " << path << std::endl; return false; }
        for (size_t t = 0; t < hdr.names.size(); ++t) {
            if (!rm.has((int32_t)t)) continue;
            const uint32_t len = hdr.lens[t];
            for (uint32_t b = 0; b < len; b += (uint32_t)CH) {
                const size_t n = std::min<size_t>(CH, len - b);
                auto d = std::make_shared<std::vector<uint32_t>>(n);
                if (!eng->ck(eng->api->read_depth(eng->ctx, (int32_t)t, b, n, d->data()), "pd_read_depth")) { out.close(); return false; }
                const std::string *nm = &hdr.names[t];
                out.submit([d, nm, b, n](std::string *txt) { format_sites(*nm, b, d->data(), n, txt); });
            }
        }
        return out.close();
    }
    GzWriter out;
    out.set_threads(threads);                            // one stream, the reference's bytes, deflated on all threads
    if (!out.open(path)) { std::cerr << "open OUT File error: " << path << std::endl; return false; }
    std::vector<uint32_t> d(CH);
    std::string txt;
    for (size_t t = 0; t < hdr.names.size(); ++t) {
        if (!rm.has((int32_t)t)) continue;
        const uint32_t len = hdr.lens[t];
        for (uint32_t b = 0; b < len; b += (uint32_t)CH) {
            const size_t n = std::min<size_t>(CH, len - b);
            if (!eng->ck(eng->api->read_depth(eng->ctx, (int32_t)t, b, n, d.data()), "pd_read_depth")) return false;
            format_sites(hdr.names[t], b, d.data(), n, &txt);
            out.write(txt);
        }
    }
    return out.close();
}

struct RowSums { uint64_t L = 0, C = 0, D = 0, G = 0; };

// Rows [0, n) of one contig's table, formatted by up to `threads` threads in contiguous slices (3e7 rows for -w 100 on a
// 3 Gb genome) and written in order.  `row(k, txt, sums)` appends row k and adds its three totals.
template <class F>
void write_rows(GzWriter &out, size_t n, int threads, RowSums *tot, F row)
{
    const size_t T = std::min<size_t>((size_t)std::max(1, threads), std::max<size_t>(1, n / 8192));
    std::vector<std::string> txt(T);
    std::vector<RowSums> sums(T);
    auto work = [&](size_t s) {
        const size_t lo = n * s / T, hi = n * (s + 1) / T;
        txt[s].reserve((hi - lo) * 56);
        for (size_t k = lo; k < hi; ++k) row(k, &txt[s], &sums[s]);
    };
    std::vector<std::thread> th;
    for (size_t s = 1; s < T; ++s) th.emplace_back(work, s);
    work(0);
    for (auto &x : th) x.join();
    for (size_t s = 0; s < T; ++s) { out.write(txt[s]); tot->L += sums[s].L; tot->C += sums[s].C; tot->D += sums[s].D; tot->G += sums[s].G; }
}

// gc_sum < 0: no GC column (PD:5122); otherwise PD:5005
std::string footer(uint64_t L, uint64_t C, uint64_t D, int64_t gc_sum = -1)
{
    return "##RegionLength: " + std::to_string(L) + "\tCoveredSite: " + std::to_string(C) +
           (gc_sum >= 0 ? "\tGC(%): " + fmt2((uint64_t)gc_sum * 100.0 / L) : std::string()) + "\tCoverage(%): " +
           fmt2(C * 100.0 / L) + "\tMeanDepth: " + fmt2(D * 1.0 / L) + "\n";
}

} // namespace

} // namespace pdh

using namespace pdh;

extern "C" int pandepth_main(int argc, char **argv, const pd_engine_api *api, int device)
{
    PhaseTimer tm;
    Options o;
    const int n_files = parse_options(argc, argv, &o);
    if (n_files == 0) return 0;
    const bool list_mode = n_files > 1;
    if (list_mode) std::cout << "INFO: Run multi-file data " << std::endl;

    std::string path = o.input, err;
    if (path.empty()) { std::cerr << "Error: Failed to open the BAM/CRAM file: " << path << std::endl; return 1; }
    const bool paf = is_paf_path(path);                  // PD:3466-3479 / PD:3420-3432: the first input's extension decides
    // One context per GPU for a `#.list` input (round robin over the files); see the transport notes below.
    int n_dev = 1, n_ctx = 1;
    if (list_mode && !paf && api->device_count && api->accumulate_from && api->device_count(&n_dev) == 0 && n_dev > 0) {
        n_ctx = n_dev;
        if (const char *e = tune("gpus")) n_ctx = atoi(e) > 0 ? atoi(e) : 1;     // may exceed n_dev (contexts then share GPUs)
        if (n_ctx > n_files) n_ctx = n_files;
    }
    const bool want_rccl = tune("transport") && !strncmp(tune("transport"), "rccl", 4);
    // -X comm=force: a communicator even for ONE context (single-GPU boxes exercise the collective path that way); comm=0: never one
    // (the contexts are added into the first GPU).  `rccl=force` / `rccl=0` are the names these had while RCCL was the only transport.
    const bool comm_forced = (tune("comm") && !strcmp(tune("comm"), "force")) || (tune("rccl") && !strcmp(tune("rccl"), "force"));
    const bool comm_off = (tune("comm") && tune("comm")[0] == '0') || (tune("rccl") && tune("rccl")[0] == '0');
    // RCCL prints a version banner on descriptor 1 when the first communicator is made, whatever NCCL_DEBUG says, and this program's
    // stdout is compared byte for byte with the reference's.  Until round 6 descriptor 1 pointed at /dev/null around every RCCL call,
    // which in a process with reader threads could eat a line of OURS.  Now: when RCCL may be used, our own lines (all of them go
    // through std::cout) are written to a private duplicate of the real stdout for the whole run and descriptor 1 belongs to the
    // libraries — pointed at /dev/null unless -X rccl_verbose asks to see them.  Nothing of ours can be lost, whoever prints when.
    struct OwnStdout {
        struct Buf : std::streambuf {       // (line-buffered, and locked: reader threads print warnings too)
            int fd = -1; std::string pend; std::mutex mu;
            void flush_locked() { size_t o = 0; while (o < pend.size()) { const ssize_t k = ::write(fd, pend.data() + o, pend.size() - o); if (k <= 0) break; o += (size_t)k; } pend.clear(); }
            int overflow(int c) override { std::lock_guard<std::mutex> lk(mu); if (c != EOF) { pend.push_back((char)c); if (c == '\n') flush_locked(); } return c == EOF ? 0 : c; }
            std::streamsize xsputn(const char *p, std::streamsize n) override { std::lock_guard<std::mutex> lk(mu); pend.append(p, (size_t)n); if (memchr(p, '\n', (size_t)n)) flush_locked(); return n; }
            int sync() override { std::lock_guard<std::mutex> lk(mu); flush_locked(); return 0; }
        } buf;
        std::streambuf *old = nullptr; int saved = -1;
        void engage(bool silence)
        {
            std::cout.flush(); fflush(stdout);
            buf.fd = dup(1);
            if (buf.fd < 0) return;
            old = std::cout.rdbuf(&buf);
            if (!silence) return;
            const int nul = ::open("/dev/null", O_WRONLY);
            if (nul < 0) return;
            saved = dup(1);
            if (saved >= 0) dup2(nul, 1);
            ::close(nul);
        }
        ~OwnStdout()
        {
            if (old) { std::cout.flush(); buf.sync(); std::cout.rdbuf(old); }
            if (saved >= 0) { fflush(stdout); dup2(saved, 1); ::close(saved); }
            if (buf.fd >= 0) ::close(buf.fd);
        }
    } own_stdout;
    const bool rccl_maybe = !paf && !o.site_out && api->comm_init_all && (n_ctx > 1 || comm_forced) && !comm_off;      // (comm=force: also for a single input)
    if (rccl_maybe) own_stdout.engage(!tune("rccl_verbose"));
    // -X transport=rccl: librccl's load (1.1 s warm, 5 s the first time on a box) and the communicator's bootstrap (0.6 s) start NOW, on
    // a thread beside the header / index / annotation reads, and are waited for BEFORE the contexts are made — while the library
    // registers its code objects it holds the runtime lock every kernel launch needs, so behind a running decode (round 5) the
    // decode crawled (0.68 -> 2.19 s on the 3e8-record list run).  The contexts' communicators adopt the one made here.
    struct CommAhead {
        std::thread th; bool started = false; int rc = 0; double secs = 0;
        void wait() { if (th.joinable()) th.join(); }
        ~CommAhead() { wait(); }
    } comm_ahead;
    if (rccl_maybe && want_rccl && api->comm_preinit && (n_ctx <= n_dev || getenv("PANDEPTH_RCCL_LIB"))) {
        comm_ahead.started = true;
        comm_ahead.th = std::thread([&, n_ctx, n_dev]() {
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<int> devs;
            for (int k = 0; k < n_ctx; ++k) devs.push_back((device + k) % n_dev);
            comm_ahead.rc = api->comm_preinit(devs.data(), n_ctx);
            comm_ahead.secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        });
    }
    AlnReader first;
    AlnHeader hdr;
    RefSeqs ref;                                         // -c -r: the GC(%) column (PD:3506-3538); host-side text work
    std::map<std::string, int32_t> paf_names;            // PAF: target name -> id (grows while records are read, PD:1559)
    bool regions_ok = true;
    RegionModel rm;
    if (paf) {
        std::cout << (list_mode ? "INFO: Run PAF format data " : "INFO: Run paf Format data ") << std::endl;
        if (o.gc && o.reference.empty()) { std::cerr << "Error: lack reference sequence (-r) for GC parse" << std::endl; return 0; }   // PD:909-913
        regions_ok = paf_targets(o, &hdr, &paf_names, &ref);
        tm.mark("options + targets");
        if (regions_ok) regions_ok = build_regions(&o, hdr, &rm, ref.loaded ? &ref : nullptr, o.threads, &paf_names);
    } else {
        if (!first.open(path, &err)) { std::cerr << "Error: Failed to open the BAM/CRAM file: " << path << std::endl; return 1; }
        hdr = first.header();
        if (hdr.names.empty()) { std::cerr << "Error: Failed to read the header for the BAM/CRAM file: " << path << std::endl; return 1; }
        if (first.is_cram() && !o.reference.empty()) {
            // PD:3486-3492 hands -r to htslib for CRAM input, and htslib then trusts the FASTA over the header: an @SQ
            // whose LN differs from the indexed sequence's length is rewritten to the FASTA's (cram_io.c
            // sanitise_SQ_lines).  Plain-text FASTA only (its faidx cannot index a gzip file, and nothing changes then).
            // Not detected: a plain FASTA that faidx refuses to index (ragged line lengths, blank lines inside a record).
            std::map<std::string, uint32_t> fa_len;
            std::string scratch;
            bool plain = false;
            { FILE *fp = fopen(o.reference.c_str(), "rb"); if (fp) { const int c0 = fgetc(fp), c1 = fgetc(fp); plain = !(c0 == 0x1f && c1 == 0x8b) && c0 != EOF; fclose(fp); } }
            if (plain && read_fasta_records(o.reference, &scratch, [&](const std::string &name, size_t off, size_t n) {
                    fa_len.insert({name, (uint32_t)n});          // the first record of a name is the indexed one
                    scratch.resize(off);
                }))
                for (size_t i = 0; i < hdr.names.size(); ++i) {
                    auto it = fa_len.find(hdr.names[i]);
                    if (it != fa_len.end()) hdr.lens[i] = it->second;
                }
        }
        if (o.gc) {
            // PD:3510-3532 (PD:2068-2090 for lists): -c needs -r, checked once the first input's header has been read
            if (o.reference.empty()) { std::cerr << "Error: lack reference sequence (-r) for GC parse" << std::endl; return 0; }
        }
        tm.mark("options + header");
        regions_ok = build_regions(&o, hdr, &rm, o.gc ? &ref : nullptr, o.threads);
    }
    if (!regions_ok) {
        // the reference's reader never returns from a NULL gzFile; an error is the usable answer
        std::cerr << "Error: Cannot open the reference sequence file: " << o.reference << std::endl;
        return 1;
    }
    const bool gc = ref.loaded;
    if (gc && o.mode != 6) ref.clear();                  // PD:4095-4097 (the bins and genes hold their counts by now)
    const bool synthetic = o.mode == 0 || o.mode == 5 || o.mode == 6;

    // output names (PD:4057-4090)
    std::string prefix = o.out.substr(0, o.out.size() - 3);
    {
        const size_t d = prefix.rfind('.');
        const std::string ext = d == std::string::npos ? std::string() : prefix.substr(d + 1);
        if (ext == "stat" || ext == "bed") prefix = prefix.substr(0, d);
    }
    std::string stat_path = prefix + ".gene.stat.gz";
    std::string header_line = "#Chr\tStart\tEnd\tGeneID\tLength\tCoveredSite\tTotalDepth\tCoverage(%)\tMeanDepth\n";
    if (o.mode == 3) { stat_path = prefix + ".bed.stat.gz"; header_line = "#Chr\tStart\tEnd\tRegionID\tLength\tCoveredSite\tTotalDepth\tCoverage(%)\tMeanDepth\n"; }
    else if (o.mode == 4) stat_path = prefix + ".bed.stat.gz";
    else if (o.mode == 5 || o.mode == 6) { stat_path = prefix + ".win.stat.gz"; header_line = "#Chr\tStart\tEnd\tLength\tCoveredSite\tTotalDepth\tCoverage(%)\tMeanDepth\n"; }
    else if (o.mode == 0) { stat_path = prefix + ".chr.stat.gz"; header_line = "#Chr\tLength\tCoveredSite\tTotalDepth\tCoverage(%)\tMeanDepth\n"; }
    if (gc) {                                            // PD:4095-4114: one more column, after TotalDepth
        const size_t at = header_line.find("\tCoverage(%)");
        header_line.insert(at, "\tGC(%)");
    }
    GzWriter OUT;
    OUT.set_threads(o.threads);                          // large tables: same bytes, LZ77 parse on all threads (host/pgzip.h)
    if (!OUT.open(stat_path)) { std::cerr << "open OUT File error: " << stat_path << std::endl; return 0; }

    tm.mark("region model");
    if (paf && hdr.names.empty()) {
        // an empty (or unreadable: the reference's gzstream reports nothing) first file: no targets, empty tables
        if (o.site_out) { GzWriter s; if (s.open(prefix + ".SiteDepth.gz")) s.close(); }
        OUT.write(header_line);
        std::cout << "INFO: Input data read done" << std::endl;
        OUT.write(footer(0, 0, 0, gc ? 0 : -1));
        OUT.close();
        return 0;
    }
    // One context per GPU.  A `#.list` input is sharded one file per GPU (round robin) when the engine
    // offers several devices; the contexts are summed into the first one before the statistics
    // (difference arrays are linear: PD:2704-3014 accumulates every file into one array).
    if (comm_ahead.started) {
        const auto t0 = std::chrono::steady_clock::now();
        comm_ahead.wait();
        if (tm.on) fprintf(stderr, "[timing] %-28s %8.3f s   (RCCL load + bootstrap on a thread since process entry, ahead of the contexts, rc %d; the contexts waited %.3f s for it)\n",
                           "comm ahead", comm_ahead.secs, comm_ahead.rc, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
    std::vector<std::unique_ptr<Engine>> engs;
    // (the executable leaves without tearing the engine down — the process is about to end; library users of pandepth_main keep the destroy)
    struct CtxGuard { std::vector<std::unique_ptr<Engine>> *v; ~CtxGuard() { if (getenv("PANDEPTH_KEEP_CONTEXT")) return; for (auto &e : *v) if (e->ctx) e->api->destroy(e->ctx); } } guard{&engs};
    for (int k = 0; k < n_ctx; ++k) { engs.emplace_back(new Engine); engs.back()->api = api; }
    {   // the contexts are made side by side (a context costs 0.04-0.09 s of runtime start-up, queues and buffers: eight in a row were 0.5 s
        // before the first byte was read); the first failure is the one reported
        std::vector<int> rcs((size_t)n_ctx, 0);
        auto make = [&](int k) { rcs[(size_t)k] = api->create((device + k) % n_dev, (int32_t)hdr.lens.size(), hdr.lens.data(), &engs[(size_t)k]->ctx); };
        if (n_ctx == 1) make(0);
        else {
            std::vector<std::thread> th;
            for (int k = 0; k < n_ctx; ++k) th.emplace_back(make, k);
            for (auto &t : th) t.join();
        }
        for (int k = 0; k < n_ctx; ++k)
            if (rcs[(size_t)k] != 0) {
                const char *m = api->strerror(nullptr);
                std::cerr << "Error: depth engine unavailable: " << (m ? m : "?") << std::endl;
                return 2;
            }
    }
    Engine &eng = *engs[0];
    // the table's gzip stream: zlib's LZ77 parse on the engine (large -w tables); the writer forgets the engine before it goes
    struct ParseGuard { GzWriter *w; ~ParseGuard() { w->set_parse(nullptr); } } parse_guard{&OUT};
    OUT.set_parse(engine_parse(&eng));
    // whole-contig statistics straight from the runs when a sample ends up resident and deferred (pd_scan_reduce_windows)
    if (api->keep_deferred) for (auto &e : engs) api->keep_deferred(e->ctx, 1);
    tm.mark("engine create");
    SpanIndex spans;
    spans.build(rm, hdr, synthetic);
    // Several contexts: their statistics are summed in slices by a collective (pd_sliced_*).  Transport (-X transport=peer|rccl):
    //   peer (default)  the in-process one — this executable IS one process with a rank thread per GPU, so a rank pulls its slices out of
    //                   its peers' buffers with xGMI peer copies; nothing to load or bootstrap (pd_comm_init_local), made in line;
    //   rccl            north_star's transport and the one between processes (bench.py --gpus N): librccl was loaded and the communicator
    //                   bootstrapped at process entry, AHEAD of the contexts (comm_ahead, above), and is adopted here.
    // Whichever is chosen falls back to the other when it cannot be made, and to adding the contexts into the first GPU after that.
    const bool comm_possible = !paf && (n_ctx > 1 || comm_forced) && !comm_off;
    auto make_comms = [&](std::vector<pd_comm *> *comms, std::string *how) -> int {
        std::vector<pd_ctx *> ctxs;
        for (auto &e : engs) ctxs.push_back(e->ctx);
        comms->assign((size_t)n_ctx, nullptr);
        int rc = -1;
        for (int attempt = 0; attempt < 2 && rc != 0; ++attempt) {
            const bool rccl = (attempt == 0) == want_rccl;
            if (rccl) {
                if (!api->comm_init_all || !(n_ctx <= n_dev || getenv("PANDEPTH_RCCL_LIB"))) continue;
                rc = api->comm_init_all(ctxs.data(), n_ctx, comms->data());
                *how = "RCCL";
            } else {
                if (!api->comm_init_local || (tune("transport") && !strcmp(tune("transport"), "rccl_only"))) continue;
                rc = api->comm_init_local(ctxs.data(), n_ctx, comms->data());
                *how = "in-process peer copies";
            }
            if (rc != 0 && tm.on) fprintf(stderr, "[timing] %s communicator unavailable (%s)\n", how->c_str(), api->strerror(eng.ctx));
        }
        return rc;
    };
    // The communicator and its exchange buffers (3 GB of device allocations per rank for a 3 Gb genome: tenths of a second) are made on a
    // side thread BESIDE the decode and picked up by the first collective (-X comm_early=0: in line, before the first collective).  With the
    // in-process transport nothing is loaded meanwhile — round 5's side thread loaded librccl there, which starved the decode's launches.
    struct CommEarly {
        std::thread th; std::vector<pd_comm *> comms; std::string how; int rc = -1; bool started = false, taken = false; double secs = 0;
        const pd_engine_api *api = nullptr;
        void wait() { if (th.joinable()) th.join(); }
        ~CommEarly() { wait(); if (started && !taken && rc == 0 && api && api->comm_destroy) for (pd_comm *m : comms) if (m) api->comm_destroy(m); }   // (made, never used: a run that failed meanwhile)
    } comm_early;
    comm_early.api = api;
    auto start_comm_early = [&]() {
        if (!comm_possible || o.site_out || comm_early.started || (!api->comm_init_all && !api->comm_init_local)) return;
        if (tune("comm_early") && tune("comm_early")[0] == '0') return;
        comm_early.started = true;
        comm_early.th = std::thread([&]() {
            const auto t0 = std::chrono::steady_clock::now();
            comm_early.rc = make_comms(&comm_early.comms, &comm_early.how);
            if (comm_early.rc == 0 && api->comm_prepare) {
                std::vector<std::thread> th;
                for (pd_comm *m : comm_early.comms) th.emplace_back([this_api = api, m] { (void)this_api->comm_prepare(m, 0); });     // (a failure shows again, with its message, at the first collective)
                for (auto &t : th) t.join();
            }
            comm_early.secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        });
    };

    bool wrap18 = list_mode;                     // PD:2687: the #.list path always uses SiteInfo cells
    if (paf) {
        // PD:1532-1616: every file of the list, line by line, into the same 18-bit arrays.  PAF lines come in query order,
        // so most runs take the sink's unordered stream (device atomics); text parsing, not the scatter, is the cost.
        wrap18 = true;
        struct SinkEmitter : RunEmitter {
            RunSink sink;
            explicit SinkEmitter(Engine *e) : sink(e) {}
            void emit(int32_t tid, int32_t beg, int32_t end) override { sink.emit(tid, beg, end); }
        };                                                 // (a sink flushes what it holds when it goes away)
        uint64_t n_rec = 0;
        for (const std::string &fp : o.inputs)
            read_paf(fp, o, paf_names, [&]() { return std::unique_ptr<RunEmitter>(new SinkEmitter(&eng)); }, o.threads, &n_rec);
        if (tm.on) fprintf(stderr, "[timing] paf: %llu records\n", (unsigned long long)n_rec);
    } else {
        // Classify the inputs in list order first: the reference prints its "No Index mode" warnings in
        // that order (it reads the files one after another), whatever order the GPUs finish in.
        struct Input { std::string path; int kind; };                  // 0 indexed, 1 sorted stream, 2 every read
        std::vector<Input> inputs;
        for (const std::string &fp : o.inputs) {
            if (index_exists(fp) && o.use_index) {
                if (o.site_out || o.mode == 6) wrap18 = true;            // PD:4127
                inputs.push_back({fp, 0});
                continue;
            }
            wrap18 = true;                                               // PD:4553
            bool sorted = false;
            if (!list_mode) sorted = first.header().sorted_coordinate();
            else {
                AlnReader probe;
                if (!probe.open(fp, &err)) { std::cerr << "Error: Failed to open the BAM/CRAM file: " << fp << std::endl; continue; }
                sorted = probe.header().sorted_coordinate();
            }
            if (sorted) std::cout << "Warning: PanDepth will run in No Index mode: " << fp << std::endl;
            else std::cout << "Warning: Can't find index file of input BAM/CRAM. PanDepth will run in No Index mode: " << fp << std::endl;
            inputs.push_back({fp, sorted ? 1 : 2});
        }
        Options o_part = o;
        if (n_ctx > 1) {
            // the CPU-heavy parts of a context (host readers, handed-back units) get their share of -t; the device decode's readers do not
            // shrink with it — a reader copies a batch out of the page cache (2 ms per 32 MB) and then waits for the device — so every
            // GPU keeps four of them, two buffers each, whatever -t / #GPUs comes to
            o_part.threads = std::max(1, o.threads / n_ctx);
            o_part.decode_readers = 4;
        }
        start_comm_early();
        auto run_inputs = [&](int k) {
            Engine *e = engs[k].get();
            for (size_t i = (size_t)k; i < inputs.size(); i += (size_t)n_ctx) {
                const Input &in = inputs[i];
                if (in.kind == 0) { if (!read_indexed(in.path, o_part, hdr, spans, e, &rm)) return; continue; }
                AlnReader rd;
                AlnReader *r = &rd;
                std::string e2;
                if (!list_mode) r = &first;                              // already positioned after the header
                else if (!rd.open(in.path, &e2)) { e->fail("cannot open " + in.path); return; }
                if (r->is_bam()) {
                    // whole-contig modes: the device decodes the stream (every read of a contig with targets is counted on
                    // all three of the reference's paths there); 0 = not applicable or declined, nothing counted yet
                    const int d = read_bam_device(in.path, o_part, hdr, spans, rm, in.kind, r->tell(), nullptr, in.kind == 1, e);
                    if (d > 0) continue;
                    if (d < 0) return;
                }
                r->set_threads(o_part.threads > 1 ? (o_part.threads > 32 ? 32 : o_part.threads) : 0);
                if (!(in.kind == 1 ? read_sorted_stream(r, o_part, hdr, rm, e) : read_all(r, o_part, hdr, rm, e))) return;
            }
        };
        if (n_ctx == 1) run_inputs(0);
        else {
            std::vector<std::thread> th;
            for (int k = 0; k < n_ctx; ++k) th.emplace_back(run_inputs, k);
            for (auto &t : th) t.join();
        }
    }
    for (int k = 0; k < n_ctx; ++k) {
        if (!engs[k]->ok() || !engs[k]->ck(api->synchronize(engs[k]->ctx), "pd_synchronize")) {
            std::cerr << "Error: " << engs[k]->err << std::endl;
            return 2;
        }
    }
    tm.mark("decode + scatter");
    const unsigned wrap_bits = wrap18 ? 18u : 0u;
    const uint32_t min_dep = (uint32_t)o.min_dep;
    // Several GPUs hold one partial sample each.  Wide-window statistics are summed in slices over RCCL (pd_sliced_window_sum:
    // every GPU receives 1/n of the others' 4-bit images, no GPU ever holds everybody's arrays); whatever needs the summed
    // cells themselves (per-site output, annotation intervals, narrow windows) adds the contexts into the first one.
    bool merged = n_ctx == 1 && !comm_forced;            // (comm=force: a 1-rank communicator, so that single-GPU boxes test this path)
    auto merge_contexts = [&]() -> bool {
        if (merged) return true;
        merged = true;
        for (int k = 1; k < n_ctx; ++k) {
            if (!eng.ck(api->accumulate_from(eng.ctx, engs[k]->ctx), "pd_accumulate_from")) return false;
            api->destroy(engs[k]->ctx); engs[k]->ctx = nullptr;
        }
        return true;
    };
    bool scanned = false;
    auto need_scan = [&]() -> bool {
        if (scanned) return true;
        if (!merge_contexts()) return false;
        scanned = true;
        return eng.ck(api->scan(eng.ctx, wrap_bits), "pd_scan");
    };
    std::function<void()> abandon_site_file = [] {};           // (set below: stops a per-site writer working behind the statistics)
    auto bail = [&]() { abandon_site_file(); std::cerr << "Error: " << eng.message() << std::endl; return 2; };
    // A collective over the contexts (one thread per rank): what several GPUs' statistics have in common.  `call(k, comm)` is the
    // rank's collective; returns 1 done, 0 not applicable (no communicator, or the samples do not fit the sliced sum's 4-bit images:
    // PD_ERANGE on every rank, nothing consumed — the contexts are then added into the first one), -1 error.
    auto sliced = [&](const std::function<int(int, pd_comm *)> &call, const char *what) -> int {
        if (merged || scanned || !comm_possible || (!api->comm_init_all && !api->comm_init_local)) return 0;
        std::vector<pd_comm *> comms;
        std::string how;
        if (comm_early.started && !comm_early.taken) {
            const auto t0 = std::chrono::steady_clock::now();
            comm_early.wait();
            comm_early.taken = true;
            if (tm.on) fprintf(stderr, "[timing] %-28s %8.3f s   (%s: communicator + exchange buffers on a side thread beside the decode; the first collective waited %.3f s for it)\n", "comm init", comm_early.secs,
                               comm_early.how.c_str(), std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
            if (comm_early.rc != 0) {
                if (tm.on) fprintf(stderr, "[timing] no communicator: the contexts are added into GPU %d instead\n", device);
                return 0;
            }
            comms = comm_early.comms; how = comm_early.how;
        } else {
            const auto t0 = std::chrono::steady_clock::now();
            const int irc = make_comms(&comms, &how);
            if (tm.on) fprintf(stderr, "[timing] %-28s %8.3f s   (%s, in line before the first collective%s)\n", "comm init", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(),
                               how.c_str(), comm_ahead.started ? "; librccl's load and bootstrap were done ahead of the contexts" : "");
            if (irc != 0) {
                if (tm.on) fprintf(stderr, "[timing] no communicator: the contexts are added into GPU %d instead\n", device);
                return 0;
            }
        }
        std::vector<int> rcs((size_t)n_ctx, 0);
        std::vector<std::thread> th;
        const auto t_coll = std::chrono::steady_clock::now();
        for (int k = 0; k < n_ctx; ++k) th.emplace_back([&, k]() { rcs[(size_t)k] = call(k, comms[(size_t)k]); });
        for (auto &t : th) t.join();
        if (tm.on) fprintf(stderr, "[timing] %-28s %8.3f s   (%s: export, exchange over the links, every rank's sweep of its slice, results to rank 0)\n", "collective",
                           std::chrono::duration<double>(std::chrono::steady_clock::now() - t_coll).count(), what);
        // PD_ERANGE (-6) on every rank: a sample with more cells outside the 4-bit image's range than the exception block
        // holds (amplicon, very deep RNA-seq).  Nothing was consumed; the contexts are added into the first one instead.
        bool ok = true, too_wide = true;
        for (int k = 0; k < n_ctx; ++k) if (rcs[(size_t)k] != PD_ERANGE) too_wide = false;
        for (int k = 0; k < n_ctx && !too_wide; ++k)
            if (rcs[(size_t)k] != 0 && ok) { ok = false; const char *m = api->comm_strerror ? api->comm_strerror(comms[(size_t)k]) : nullptr; eng.fail(std::string(what) + ": " + (m ? m : "?")); }
        if (api->comm_destroy) for (pd_comm *c : comms) api->comm_destroy(c);
        if (too_wide) {
            if (tm.on) fprintf(stderr, "[timing] the samples do not fit the sliced sum's 4-bit images: the contexts are added into GPU %d instead\n", device);
            return 0;
        }
        if (ok && tm.on) fprintf(stderr, "[timing] %s summed over %d GPUs in slices (%s)\n", what, n_ctx, how.c_str());
        return ok ? 1 : -1;
    };
    // cover / depth sum of every window of `width` cells (pd_window_layout order), whichever way the sample is held
    auto window_stats = [&](uint32_t width, uint32_t *cov, uint64_t *sum) -> bool {
        if (api->sliced_window_sum && (width >= PD_TILE_CELLS || (api->sliced_interval_sum && !o.site_out))) {   // (narrow windows: engines with the cell-level collectives)
            const int r = sliced([&](int k, pd_comm *cm) { return api->sliced_window_sum(cm, width, min_dep, wrap_bits, 0, k == 0 ? cov : nullptr, k == 0 ? sum : nullptr); },
                                 "window statistics");
            if (r) return r > 0;
        }
        if (!merge_contexts()) return false;
        const int rc = scanned ? api->reduce_windows(eng.ctx, width, min_dep, cov, sum) : api->scan_reduce_windows(eng.ctx, width, min_dep, wrap_bits, cov, sum);
        return eng.ck(rc, "window reduction");
    };
    // CoveredSite / TotalDepth of every region (PD:329-348), whichever way the sample is held
    auto interval_stats = [&](const std::vector<pd_region> &regs, int32_t *cov, uint64_t *sum) -> bool {
        if (api->sliced_interval_sum && !o.site_out) {
            const int r = sliced([&](int k, pd_comm *cm) { return api->sliced_interval_sum(cm, regs.data(), regs.size(), min_dep, wrap_bits, 0, k == 0 ? cov : nullptr, k == 0 ? sum : nullptr); },
                                 "interval statistics");
            if (r) return r > 0;
        }
        if (!need_scan()) return false;
        return eng.ck(api->reduce_intervals(eng.ctx, regs.data(), regs.size(), min_dep, cov, sum), "pd_reduce_intervals");
    };

    // The per-site file is written behind the statistics and the tables: both read the same depth cells, the engine serialises
    // its entry points, and the file's gzip stream keeps the host threads busy only part of the time.
    struct SiteJob {
        std::thread th; bool ok = true;
        void wait() { if (th.joinable()) th.join(); }
        ~SiteJob() { wait(); }
    } site_job;
    if (o.site_out) {
        if (!need_scan()) return bail();
        site_job.th = std::thread([&] { site_job.ok = write_site_depth(prefix + ".SiteDepth.gz", hdr, rm, &eng, o.threads); });
        if (tune("site_overlap") && tune("site_overlap")[0] == '0') { site_job.wait(); tm.mark("per-site file"); }
    }
    // a failed run does not wait for the whole per-site file: the writer is told to stop, and what it wrote is removed
    abandon_site_file = [&] {
        if (!site_job.th.joinable()) return;
        eng.cancel.store(true);
        site_job.wait();
        ::remove((prefix + ".SiteDepth.gz").c_str());
    };
    auto site_done = [&]() -> bool {
        const bool was_running = site_job.th.joinable();
        site_job.wait();
        if (was_running) tm.mark("per-site file (the rest of it)");
        return site_job.ok || eng.ok();
    };

    const size_t nctg = hdr.lens.size();
    uint64_t SL = 0, SC = 0, SD = 0, SG = 0;
    std::string txt;

    if (o.mode == 6) {
        // PD:4352-4394: windows straight off the cells; `for (j = 1; j < len; j += w)` drops a final
        // 1-base window
        const uint32_t w = (uint32_t)o.win;
        std::vector<uint64_t> woff(nctg + 1);
        api->window_layout(eng.ctx, w, woff.data());
        std::vector<uint32_t> cov(woff[nctg] ? woff[nctg] : 1);
        std::vector<uint64_t> sum(woff[nctg] ? woff[nctg] : 1);
        if (!window_stats(w, cov.data(), sum.data())) return bail();
        // Large tables without the GC column: the rows are formatted, parsed and check-summed on the device from the statistics the
        // window call left there; the host adds up the three totals of the last line and writes the finished stream.
        bool table_done = false;
        if (!gc) {
            std::vector<TableContig> tcs;
            for (size_t t = 0; t < nctg; ++t) {
                if (!rm.has((int32_t)t)) continue;
                const int64_t len = hdr.lens[t];
                tcs.push_back(TableContig{(int32_t)t, len > 1 ? (size_t)((len - 1 + (int64_t)w - 1) / (int64_t)w) : 0, &hdr.names[t]});
            }
            size_t rows = 0;
            for (const auto &tc : tcs) rows += tc.n_rows;
            size_t min_rows = 100000;
            if (const char *e = tune("table_resident_min")) min_rows = (size_t)strtoull(e, nullptr, 10);
            if (rows >= min_rows && api->text_append_window_rows && OUT.collecting()) {
                tm.mark("scan + window statistics");
                RowSums tot;
                {   // the three totals of the last line, over slices of a million rows on the threads
                    struct Item { size_t tc, lo, hi; };
                    std::vector<Item> items;
                    for (size_t x = 0; x < tcs.size(); ++x)
                        for (size_t lo = 0; lo < tcs[x].n_rows; lo += (size_t)1 << 20) items.push_back(Item{x, lo, std::min(tcs[x].n_rows, lo + ((size_t)1 << 20))});
                    const int nt = std::max(1, std::min(o.threads, 16));
                    std::vector<RowSums> part((size_t)nt);
                    std::atomic<size_t> next{0};
                    std::vector<std::thread> th;
                    for (int k = 0; k < nt; ++k)
                        th.emplace_back([&, k] {
                            RowSums r;
                            for (;;) {
                                const size_t it = next.fetch_add(1);
                                if (it >= items.size()) break;
                                const TableContig &tc = tcs[items[it].tc];
                                const int64_t len = hdr.lens[(size_t)tc.tid];
                                const uint64_t base = woff[(size_t)tc.tid];
                                for (size_t i = items[it].lo; i < items[it].hi; ++i) {
                                    const int64_t j = 1 + (int64_t)i * w;
                                    int64_t end = j - 1 + w; if (end > len) end = len;
                                    r.L += (uint64_t)(end - j + 1);
                                    r.C += (uint64_t)(int64_t)(int32_t)cov[base + i];
                                    r.D += (uint64_t)(int64_t)(int32_t)sum[base + i];
                                }
                            }
                            part[(size_t)k] = r;
                        });
                    for (auto &x : th) x.join();
                    for (const auto &r : part) { tot.L += r.L; tot.C += r.C; tot.D += r.D; }
                }
                const int r = write_window_table_resident(OUT, &eng, o.threads, w, header_line, tcs, footer(tot.L, tot.C, tot.D, -1));
                if (r < 0) return bail();
                table_done = r == 1;
            }
        }
        if (table_done) {
            tm.mark("totals + table (rows, parse and checksums on the device)");
            OUT.close();
            tm.mark("table close");
            if (!site_done()) return bail();
            std::cout << "INFO: Input data read done" << std::endl;
            return 0;
        }
        OUT.write(header_line);
        RowSums tot;
        for (size_t t = 0; t < nctg; ++t) {
            if (!rm.has((int32_t)t)) continue;
            const int64_t len = hdr.lens[t];
            const size_t n_rows = len > 1 ? (size_t)((len - 1 + (int64_t)w - 1) / (int64_t)w) : 0;    // j = 1 + k w < len
            const std::string &nm = hdr.names[t];
            const uint64_t base = woff[t];
            write_rows(OUT, n_rows, o.threads, &tot, [&](size_t k, std::string *row, RowSums *rs) {
                const int64_t j = 1 + (int64_t)k * w;
                int64_t end = j - 1 + w; if (end > len) end = len;
                const int64_t L = end - j + 1;
                const int32_t c = (int32_t)cov[base + k];
                const int32_t d = (int32_t)sum[base + k];             // `int GeneDepth` (PD:4364)
                *row += nm; *row += '\t'; append_i64(row, j); *row += '\t'; append_i64(row, end);
                *row += '\t'; append_i64(row, L); *row += '\t'; append_i64(row, c); *row += '\t';
                append_i64(row, d); *row += '\t';
                if (gc) {
                    // PD:4327-4332.  The reference has dropped its sequences by now (PD:4097) and counts whatever
                    // memory follows an empty string; the window's real G/C count is written here instead.
                    const int32_t g = (int32_t)ref.gc((int32_t)t, j, end);
                    append_fmt2(row, g * 100.0 / L); *row += '\t';
                    rs->G += (uint64_t)(int64_t)g;
                }
                append_fmt2(row, c * 100.0 / L); *row += '\t'; append_fmt2(row, d * 1.0 / L);
                *row += '\n';
                rs->C += (uint64_t)(int64_t)c; rs->L += (uint64_t)L; rs->D += (uint64_t)(int64_t)d;
            });
        }
        SL += tot.L; SC += tot.C; SD += tot.D;
        OUT.write(footer(SL, SC, SD, gc ? (int64_t)tot.G : -1));
        tm.mark("scan + statistics + table text");
        OUT.close();
        tm.mark("table gzip");
        if (!site_done()) return bail();
        std::cout << "INFO: Input data read done" << std::endl;
        return 0;
    }

    if (synthetic) {
        // modes 0 and 5: every bin is window (start-1)/width of its contig
        const uint32_t width = o.mode == 5 ? (uint32_t)o.win : 10000000u;
        std::vector<uint64_t> woff(nctg + 1);
        api->window_layout(eng.ctx, width, woff.data());
        std::vector<uint32_t> cov(woff[nctg] ? woff[nctg] : 1);
        std::vector<uint64_t> sum(woff[nctg] ? woff[nctg] : 1);
        if (!window_stats(width, cov.data(), sum.data())) return bail();
        for (auto &kv : rm.bins)
            for (Bin &b : kv.second) {
                const uint64_t k = (uint64_t)(b.start - 1) / width;
                b.cover = (int32_t)cov[woff[kv.first] + k];
                b.depth = sum[woff[kv.first] + k];
            }
    } else {
        std::vector<pd_region> regs;
        for (auto &kv : rm.genes)
            for (auto &g : kv.second)
                for (auto &c : g.second.cds) regs.push_back(pd_region{kv.first, c.first, c.second});
        std::vector<int32_t> cov(regs.size() ? regs.size() : 1);
        std::vector<uint64_t> sum(regs.size() ? regs.size() : 1);
        if (!interval_stats(regs, cov.data(), sum.data())) return bail();
        size_t i = 0;
        for (auto &kv : rm.genes)
            for (auto &g : kv.second)
                for (size_t c = 0; c < g.second.cds.size(); ++c, ++i) { g.second.cover += cov[i]; g.second.depth += sum[i]; }
        // The indexed uint32 path of the reference (ProDealChrBambai, PD:676-786) walks each contig's merged gene
        // spans in windows [MeMStart, MeMEnd] and gives a window's statistics to the genes with
        // GeneStart < MeMEnd && GeneEnd >= MeMStart (PD:299-303; <= when the window is a single position, PD:305-308).
        // A window ends at (last span end + 1) clipped to the contig length, so a gene that STARTS on the last base of
        // its contig is selected by no window — it keeps cover 0 / depth 0 — unless its window also starts there.
        if (!wrap18) {
            for (auto &kv : rm.genes) {
                const int64_t clen = (int64_t)hdr.lens[(size_t)kv.first];
                auto mit = rm.merged.find(kv.first);
                if (mit == rm.merged.end() || mit->second.empty()) continue;
                const auto &spans = mit->second;
                bool any_at_end = false;
                for (auto &g : kv.second) if ((int64_t)g.second.start >= clen) { any_at_end = true; break; }
                if (!any_at_end) continue;                              // the only genes this can concern
                std::vector<std::pair<int64_t, int64_t>> wins;          // the reference's windows on this contig
                int64_t ms = spans[0].first < 1 ? 1 : spans[0].first;
                int64_t me = std::min<int64_t>(ms + 10000000 - 1, clen);
                for (size_t k = 0; k < spans.size(); ++k) {
                    const int64_t end = std::min<int64_t>((int64_t)spans[k].second + 1, clen);
                    const bool last = k + 1 == spans.size();
                    if (end >= me || last) {
                        me = end;
                        wins.emplace_back(ms, me);
                        if (!last) {
                            ms = spans[k + 1].first;
                            if (ms - 150 > me) ms -= 150;
                        }
                        me = std::min<int64_t>(ms + 10000000, clen);
                    }
                }
                for (auto &g : kv.second) {
                    Gene &x = g.second;
                    if ((int64_t)x.start < clen) continue;
                    bool selected = false;
                    for (auto &w : wins) {
                        const bool out = w.second != w.first ? ((int64_t)x.start >= w.second || (int64_t)x.end < w.first)
                                                             : ((int64_t)x.start > w.second || (int64_t)x.end < w.first);
                        if (!out) { selected = true; break; }
                    }
                    if (!selected) { x.cover = 0; x.depth = 0; }
                }
            }
        }
    }
    std::cout << "INFO: Input data read done" << std::endl;
    tm.mark("scan + statistics");

    OUT.write(header_line);
    if (o.mode == 0) {
        for (auto &kv : rm.bins) {
            uint64_t L = 0, C = 0, D = 0, G = 0;
            for (const Bin &b : kv.second) { L += (uint64_t)(b.end - b.start + 1); C += (uint64_t)(int64_t)b.cover; D += b.depth; G += (uint64_t)(int64_t)b.gc; }
            SL += L; SC += C; SD += D; SG += G;
            OUT.write(hdr.names[kv.first] + "\t" + std::to_string(L) + "\t" + std::to_string(C) + "\t" + std::to_string(D) +
                      "\t" + (gc ? fmt2(G * 100.0 / L) + "\t" : std::string()) + fmt2(C * 100.0 / L) + "\t" + fmt2(D * 1.0 / L) + "\n");
        }
    } else if (o.mode == 5) {
        RowSums tot;
        for (auto &kv : rm.bins) {
            const std::string &chr = hdr.names[kv.first];
            const std::vector<Bin> &bins = kv.second;
            write_rows(OUT, bins.size(), o.threads, &tot, [&](size_t k, std::string *row, RowSums *rs) {
                const Bin &b = bins[k];
                const uint64_t L = (uint64_t)(b.end - b.start + 1);
                rs->C += (uint64_t)(int64_t)b.cover; rs->L += L; rs->D += b.depth; rs->G += (uint64_t)(int64_t)b.gc;
                *row += chr; *row += '\t'; append_i64(row, b.start); *row += '\t'; append_i64(row, b.end); *row += '\t';
                append_u64(row, L); *row += '\t'; append_i64(row, b.cover); *row += '\t'; append_u64(row, b.depth);
                if (gc) { *row += '\t'; append_fmt2(row, b.gc * 100.0 / L); }
                *row += '\t'; append_fmt2(row, b.cover * 100.0 / L); *row += '\t'; append_fmt2(row, b.depth * 1.0 / L); *row += '\n';
            });
        }
        SL += tot.L; SC += tot.C; SD += tot.D; SG += tot.G;
    } else {
        for (auto &kv : rm.genes) {
            // rows by start; equal starts keep the id order of the map (PD:5032-5041)
            std::map<int32_t, std::string> rows;
            const std::string &chr = hdr.names[kv.first];
            for (auto &g : kv.second) {
                const Gene &x = g.second;
                SC += (uint64_t)(int64_t)x.cover; SL += x.length; SD += x.depth; SG += (uint64_t)(int64_t)x.gc;
                std::string row = chr + "\t" + std::to_string(x.start) + "\t" + std::to_string(x.end) + "\t";
                if (o.mode != 5) { row += g.first; row += '\t'; }
                row += std::to_string(x.length) + "\t" + std::to_string(x.cover) + "\t" + std::to_string(x.depth) + "\t" +
                       (gc ? fmt2(x.gc * 100.0 / x.length) + "\t" : std::string()) + fmt2(x.cover * 100.0 / x.length) + "\t" + fmt2(x.depth * 1.0 / x.length);
                auto it = rows.find(x.start);
                if (it == rows.end()) rows.emplace(x.start, row);
                else { it->second += "\n"; it->second += row; }
            }
            txt.clear();
            for (auto &r : rows) { txt += r.second; txt += '\n'; if (txt.size() > (1u << 22)) { OUT.write(txt); txt.clear(); } }
            OUT.write(txt);
        }
    }
    OUT.write(footer(SL, SC, SD, gc ? (int64_t)SG : -1));
    tm.mark("table text");
    OUT.close();
    tm.mark("table gzip");
    if (!site_done()) return bail();
    return 0;
}
