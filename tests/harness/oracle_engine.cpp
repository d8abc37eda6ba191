// tests/harness/oracle_engine.cpp — TEST INFRASTRUCTURE.
// A pd_engine_api implemented on the CPU with the oracle's loops (oracle/pd_oracle.c: one
// increment per covered base, one compare+add per region base), used ONLY by the no-GPU tests to
// drive the product's HOST code (readers, read selection, region model, table writer) and
// compare its files byte-for-byte with the reference's golden outputs.  It is not linked into
// libpandepth_amd.so, libpandepth_host.a or the `pandepth` binary.
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <zlib.h>
#include <mutex>
#include <string>
#include <vector>
#include "../../pandepth_amd/host/engine_api.h"
#include "../../pandepth_amd/host/pgzip.h"
#include "../../pandepth_amd/csrc/pd_inflate_wave.h"       // the product's device decode cores, their 64 lanes emulated on the host
#include "../../pandepth_amd/csrc/pd_bamwalk.h"
#include <algorithm>

extern "C" {
void pdo_add_intervals(int64_t n, const int32_t *iv, uint32_t *depth, const int64_t *contig_off);
void pdo_wrap18(uint32_t *depth, int64_t n);
void pdo_stat_regions(const uint32_t *depth, const int64_t *contig_off, int64_t n_reg, const int32_t *reg,
                      uint32_t min_dep, int32_t *cover, uint64_t *sum);
}

struct pd_ctx {
    std::vector<uint32_t> len;
    std::vector<int64_t> off;
    std::vector<uint32_t> depth;
    bool scanned = false;
    // the last window call's statistics (pd_text_append_window_rows formats the table's rows from them)
    uint32_t wk_w = 0; std::vector<uint32_t> wk_cov; std::vector<uint64_t> wk_sum, wk_woff;
    std::mutex mu;
    std::string err;
    // decode_*: the same batch protocol as libpandepth_amd.so's, run with the product's cores (pd_inflate_wave.h,
    // pd_bamwalk.h) in host emulation, so that the CPU suite drives the feeder of host/pipeline.cpp too
    pd_decode_cfg dcfg{}; std::vector<uint8_t> on; std::vector<uint32_t> soff; std::vector<int32_t> spans;
    struct Buf { std::vector<uint8_t> b; bool busy = false;
                 // a batch between decode_queue and decode_collect: what decode_submit answered, kept until it is asked for
                 bool queued = false; uint64_t ticket = 0; int rc = 0; std::vector<int32_t> status; pd_decode_result res{}; };
    uint64_t tickets = 0; uint64_t n_chain_dev = 0, n_chain_host = 0, n_chain_redo = 0;
    std::vector<Buf *> bufs;
    struct Runs { uint64_t order; std::vector<pd_iv> first, other, far; };
    std::vector<Runs> runs;
    // a compact session (PD_DECODE_COMPACT + n_batches on a sorted file): pass 2 runs the product's COMPACT emission (8-byte runs with
    // flat begins, bucket marks, order keys) in host emulation; the stand-in turns the runs back into 12-byte ones for the oracle and
    // holds the emission to what the engine relies on.  orders_seen: every batch number exactly once (the feeder's side of the contract)
    bool compact = false; std::vector<uint64_t> flat_off; std::vector<uint8_t> orders_seen; std::string compact_err;
};


static int o_create(int, int32_t n, const uint32_t *len, pd_ctx **out)
{
    pd_ctx *c = new pd_ctx;
    c->len.assign(len, len + n);
    c->off.resize((size_t)n + 1);
    int64_t o = 0;
    for (int32_t i = 0; i < n; ++i) { c->off[i] = o; o += (int64_t)len[i] + 8; }
    c->off[n] = o;
    c->depth.assign((size_t)o, 0);
    *out = c;
    return 0;
}
static int o_destroy(pd_ctx *c) { delete c; return 0; }
static const char *o_strerror(const pd_ctx *c) { return c ? c->err.c_str() : "oracle engine"; }
static int o_push(pd_ctx *c, const pd_iv *b, size_t n, unsigned flags)
{
    std::lock_guard<std::mutex> lk(c->mu);
    int rc = 0;
    if (c->scanned) { c->err = "push after scan"; rc = -4; }
    // verify what the host promised about this batch
    if (!rc && (flags & PD_PUSH_SORTED)) {
        const int64_t D = (int64_t)(flags >> 8) * 256;
        int64_t mx_t = -1, mx_b = 0;
        for (size_t i = 0; i < n; ++i) {
            const int64_t t = b[i].tid, bb = b[i].beg < 0 ? 0 : b[i].beg;
            if (t < mx_t || (t == mx_t && bb < mx_b - D)) { c->err = "host broke its sortedness promise"; rc = -1; break; }
            if (t > mx_t) { mx_t = t; mx_b = bb; } else if (bb > mx_b) mx_b = bb;
        }
    }
    if (!rc) {
        std::vector<int32_t> iv;
        iv.reserve(n * 3);
        for (size_t i = 0; i < n; ++i) {
            const int32_t t = b[i].tid;
            if (t < 0 || (size_t)t >= c->len.size()) continue;
            int64_t x = b[i].beg, y = b[i].end;
            const int64_t L = c->len[t];
            if (x < 0) x = 0; if (x > L) x = L; if (y < 0) y = 0; if (y > L) y = L;
            if (x < y) { iv.push_back(t); iv.push_back((int32_t)x); iv.push_back((int32_t)y); }
        }
        pdo_add_intervals((int64_t)iv.size() / 3, iv.data(), c->depth.data(), c->off.data());
    }
    return rc;
}
static int o_scan(pd_ctx *c, unsigned wrap)
{
    if (c->scanned) { c->err = "already scanned"; return -4; }
    if (wrap == 18) pdo_wrap18(c->depth.data(), (int64_t)c->depth.size());
    c->scanned = true;
    return 0;
}
static int o_reduce_intervals(pd_ctx *c, const pd_region *r, size_t n, uint32_t md, int32_t *cov, uint64_t *sum)
{
    if (!c->scanned) { c->err = "scan first"; return -4; }
    // like pd_reduce_intervals: cells clipped to the contig (the product clips runs to [0, len], so the cells a region
    // reaches beyond the contig end hold zero; the oracle's arrays are contiguous, the next contig starts there)
    std::vector<pd_region> rr(r, r + n);
    for (auto &x : rr) {
        const int64_t len = (int64_t)c->len[(size_t)x.tid];
        if (x.second > len) x.second = (int32_t)len;
        if (x.first < 1) x.first = 1;
        if (x.first > x.second + 1) x.first = x.second + 1;
    }
    pdo_stat_regions(c->depth.data(), c->off.data(), (int64_t)n, (const int32_t *)rr.data(), md, cov, sum);
    return 0;
}
static int o_layout(const pd_ctx *c, uint32_t w, uint64_t *wo)
{
    uint64_t o = 0;
    for (size_t i = 0; i < c->len.size(); ++i) { wo[i] = o; o += ((uint64_t)c->len[i] + w - 1) / w; }
    wo[c->len.size()] = o;
    return 0;
}
static void windows(pd_ctx *c, uint32_t w, uint32_t md, uint32_t mask, uint32_t *cov, uint64_t *sum)
{
    uint64_t k = 0;
    c->wk_woff.assign(c->len.size() + 1, 0);
    for (size_t t = 0; t < c->len.size(); ++t) c->wk_woff[t + 1] = c->wk_woff[t] + ((uint64_t)c->len[t] + w - 1) / w;
    struct Keep { pd_ctx *c; uint32_t w; uint32_t *cov; uint64_t *sum; ~Keep() { c->wk_w = w; c->wk_cov.assign(cov, cov + c->wk_woff.back()); c->wk_sum.assign(sum, sum + c->wk_woff.back()); } } keep{c, w, cov, sum};
    for (size_t t = 0; t < c->len.size(); ++t)
        for (uint64_t s = 0; s < c->len[t]; s += w, ++k) {
            uint64_t e = s + w; if (e > c->len[t]) e = c->len[t];
            uint32_t cc = 0; uint64_t ss = 0;
            for (uint64_t p = s; p < e; ++p) { const uint32_t d = c->depth[(size_t)c->off[t] + p] & mask; if (d >= md) { ++cc; ss += d; } }
            cov[k] = cc; sum[k] = ss;
        }
}
static int o_scan_reduce_windows(pd_ctx *c, uint32_t w, uint32_t md, unsigned wrap, uint32_t *cov, uint64_t *sum)
{
    if (c->scanned) { c->err = "already scanned"; return -4; }
    windows(c, w, md, wrap == 18 ? 0x3FFFFu : 0xFFFFFFFFu, cov, sum);
    return 0;
}
static int o_reduce_windows(pd_ctx *c, uint32_t w, uint32_t md, uint32_t *cov, uint64_t *sum)
{
    if (!c->scanned) { c->err = "scan first"; return -4; }
    windows(c, w, md, 0xFFFFFFFFu, cov, sum);
    return 0;
}
static int o_read_depth(pd_ctx *c, int32_t t, uint32_t beg, size_t n, uint32_t *out)
{
    if (!c->scanned) { c->err = "scan first"; return -4; }
    memcpy(out, c->depth.data() + c->off[t] + beg, n * 4);
    return 0;
}
static int o_sync(pd_ctx *) { return 0; }
// several "GPUs" for the list-mode driver: as many as PANDEPTH_FAKE_GPUS says (default 1)
static int o_device_count(int *n) { const char *e = getenv("PANDEPTH_FAKE_GPUS"); *n = e ? atoi(e) : 1; return 0; }
static int o_accumulate_from(pd_ctx *dst, pd_ctx *src)
{
    if (dst->scanned || src->scanned || dst->depth.size() != src->depth.size()) { dst->err = "accumulate_from: bad state"; return -4; }
    // per-base counts are additive exactly like difference arrays
    for (size_t i = 0; i < dst->depth.size(); ++i) dst->depth[i] += src->depth[i];
    return 0;
}

// ---- decode_*: pd_decode_begin / acquire / submit / end / abort on the CPU -------------------------------------
static int o_decode_begin(pd_ctx *c, const pd_decode_cfg *cfg)
{
    std::lock_guard<std::mutex> lk(c->mu);
    c->dcfg = *cfg;
    const size_t n = c->len.size();
    c->on.assign(n, 1);
    for (size_t t = 0; t < n; ++t) c->on[t] = cfg->contig_on ? cfg->contig_on[t] != 0 : c->len[t] >= 2;
    c->soff.clear(); c->spans.clear();
    if (cfg->span_off && cfg->spans) { c->soff.assign(cfg->span_off, cfg->span_off + n + 1); c->spans.assign(cfg->spans, cfg->spans + 2 * (size_t)cfg->span_off[n]); c->spans.push_back(0); }
    c->runs.clear();
    c->compact = (cfg->flags & PD_DECODE_COMPACT) && cfg->n_batches && cfg->sorted && !cfg->spans;
    c->orders_seen.assign(c->compact ? (size_t)cfg->n_batches : 0, 0);
    c->compact_err.clear();
    c->flat_off.assign(n + 1, 0);                                 // the engine's flat cell space: every contig's slot rounded up to a tile
    for (size_t t = 0; t < n; ++t) c->flat_off[t + 1] = c->flat_off[t] + ((uint64_t)c->len[t] + 1 + 8191) / 8192 * 8192;
    return 0;
}
static int o_decode_acquire(pd_ctx *c, size_t bytes, void **out)
{
    std::lock_guard<std::mutex> lk(c->mu);
    pd_ctx::Buf *b = nullptr;
    for (auto *x : c->bufs) if (!x->busy) { b = x; break; }
    if (!b) { b = new pd_ctx::Buf; c->bufs.push_back(b); }
    b->busy = true; b->b.resize(bytes + 64);
    *out = b->b.data();
    return 0;
}
static int o_decode_batch(pd_ctx *c, const pd_decode_batch *bt, int32_t *status, pd_decode_result *res);
static int o_decode_submit(pd_ctx *c, const pd_decode_batch *bt, int32_t *status, pd_decode_result *res)
{
    pd_ctx::Buf *mine = nullptr;
    { std::lock_guard<std::mutex> lk(c->mu); for (auto *x : c->bufs) if (x->busy && !x->queued && x->b.data() == bt->host_buf) mine = x; }
    if (!mine) { c->err = "submit: unknown buffer"; return -1; }
    struct Rel { pd_ctx *c; pd_ctx::Buf *b; ~Rel() { std::lock_guard<std::mutex> lk(c->mu); b->busy = false; } } rel{c, mine};
    return o_decode_batch(c, bt, status, res);
}
// pd_decode_queue / pd_decode_collect: the batch is decoded at once (there is no device to wait for here) and its answer kept under a
// ticket; the buffer stays the engine's until the batch is collected, as the product's does
static int o_decode_queue(pd_ctx *c, const pd_decode_batch *bt, uint64_t *ticket)
{
    pd_ctx::Buf *mine = nullptr;
    { std::lock_guard<std::mutex> lk(c->mu); for (auto *x : c->bufs) if (x->busy && !x->queued && x->b.data() == bt->host_buf) mine = x; }
    if (!mine) { c->err = "queue: unknown buffer"; return -1; }
    mine->status.assign(bt->n_units + 1, 0);
    mine->rc = o_decode_batch(c, bt, mine->status.data(), &mine->res);
    if (mine->rc) { std::lock_guard<std::mutex> lk(c->mu); mine->busy = false; return mine->rc; }
    std::lock_guard<std::mutex> lk(c->mu);
    mine->queued = true; mine->ticket = *ticket = ++c->tickets;
    return 0;
}
static int o_decode_collect(pd_ctx *c, uint64_t ticket, int32_t *status, pd_decode_result *res)
{
    std::lock_guard<std::mutex> lk(c->mu);
    for (auto *x : c->bufs)
        if (x->busy && x->queued && x->ticket == ticket) {
            if (status) for (size_t u = 0; u + 1 < x->status.size(); ++u) status[u] = x->status[u];
            if (res) *res = x->res;
            x->queued = false; x->busy = false;
            return 0;
        }
    c->err = "collect: not the ticket of a queued batch";
    return -1;
}
static int o_decode_batch(pd_ctx *c, const pd_decode_batch *bt, int32_t *status, pd_decode_result *res)
{
    if (res) { memset(res, 0, sizeof *res); res->first_start = res->next_start = ~0ull; }
    if (c->compact && bt->order < c->orders_seen.size()) {
        std::lock_guard<std::mutex> lk(c->mu);
        if (c->orders_seen[(size_t)bt->order]++) c->compact_err = "a batch number was submitted twice";
    } else if (c->compact && bt->n_units) { std::lock_guard<std::mutex> lk(c->mu); c->compact_err = "a batch with units has a number outside the session"; }
    if (!bt->n_units || !bt->n_blocks) return 0;
    const uint8_t *blob = (const uint8_t *)bt->host_buf;
    std::vector<uint8_t> inf((size_t)bt->inflated_bytes + 256, 0);
    std::vector<int> bst(bt->n_blocks, 0);
    static thread_local pdw::Tables T;
    static thread_local std::vector<pdw::Token> tok(65536 / 3 + 64);
    for (uint32_t b = 0; b < bt->n_blocks; ++b) {
        const pd_bgzf_block &d = bt->blocks[b];
        bst[b] = pdw::inflate_member<pdw::HostWave>(blob + d.in_off, d.in_len, inf.data() + d.out_off, d.out_len, T, tok.data(), nullptr);
    }
    pdb2::Cfg cfg;
    cfg.buf = inf.data(); cfg.avail = bt->inflated_bytes; cfg.n_ref = (int32_t)c->len.size(); cfg.contig_len = c->len.data(); cfg.contig_on = c->on.data();
    cfg.flag_mask = c->dcfg.flag_mask; cfg.min_mapq = c->dcfg.min_mapq;
    cfg.span_off = c->soff.empty() ? nullptr : c->soff.data(); cfg.spans = c->soff.empty() ? nullptr : c->spans.data();
    cfg.near_span = getenv("PANDEPTH_TEST_NEAR_SPAN") && !c->compact ? (uint32_t)atoi(getenv("PANDEPTH_TEST_NEAR_SPAN")) : 0xFFFFFFFFu;
    cfg.c8 = pdb2::C8Out{};
    std::vector<pdb2::Seg> segs; std::vector<uint32_t> seg0(bt->n_units + 1, 0);
    for (uint32_t u = 0; u < bt->n_units; ++u) {
        const pd_decode_unit &un = bt->units[u];
        seg0[u] = (uint32_t)segs.size();
        for (uint64_t b = un.start; b < un.stop; b += pdb2::SEG_BYTES) {
            pdb2::Seg s; memset(&s, 0, sizeof s);
            s.begin = b; s.end = std::min<uint64_t>(b + pdb2::SEG_BYTES, un.stop); s.avail = un.avail; s.unit_first = b == un.start;
            s.hint = (b == un.start && !(un.flags & PD_UNIT_GUESS)) ? un.start : pdb2::NONE;
            segs.push_back(s);
        }
    }
    seg0[bt->n_units] = (uint32_t)segs.size();
    std::vector<pdb2::LaneOut> lanes(segs.size() * 64);
    for (size_t j = 0; j < segs.size(); ++j) pdb2::walk_segment<pdw::HostWave>(cfg, segs[j], &lanes[j * 64]);
    if (const char *e = getenv("PANDEPTH_TEST_SPOIL_GUESS")) {
        // (test hook: every k-th segment behind a unit's first walks again from a start that is no record — what a wrong guess of the
        // header search looks like to whoever confirms the chain, here far more often than real data ever shows one)
        const size_t k = (size_t)std::max(1, atoi(e));
        for (size_t j = 0; j < segs.size(); ++j)
            if (!segs[j].unit_first && j % k == 0) { const uint64_t h = segs[j].begin + 1 + (j % 7); pdb2::walk_segment<pdw::HostWave>(cfg, segs[j], &lanes[j * 64], &h); }
    }
    // the chain across segments, as the product confirms it: in a compact session whose units start at known records by the device form
    // (pdb2::chain_device, here with its lanes in a loop) — and, whenever that leaves the batch to the host, or in any other session, by
    // check_chain's rounds
    bool guess_unit = false;
    for (uint32_t u = 0; u < bt->n_units; ++u) if (bt->units[u].flags & PD_UNIT_GUESS) guess_unit = true;
    if (!guess_unit && (c->compact || cfg.near_span == 0xFFFFFFFFu) && !segs.empty() && !getenv("PANDEPTH_TEST_HOST_CHAIN")) {
        pdb2::ChainOut co;
        const uint32_t max_redo = getenv("PANDEPTH_TEST_MAX_REDO") ? (uint32_t)atoi(getenv("PANDEPTH_TEST_MAX_REDO")) : 256u;
        pdb2::chain_device<pdw::HostWave>(segs.data(), (uint32_t)segs.size(), bst.data(), bt->n_blocks, bt->inflated_bytes / 41 + 64, bt->inflated_bytes / 41 + 64, max_redo,
            [&](uint32_t j, uint64_t start) { return pdb2::walk_segment<pdw::HostWave>(cfg, segs[j], &lanes[(size_t)j * 64], &start); }, &co);
        std::lock_guard<std::mutex> lk(c->mu);
        if (co.slow) ++c->n_chain_host; else ++c->n_chain_dev;
        c->n_chain_redo += co.n_redo;
        if (!co.slow) {
            // what the device reports must be what the host's rounds would have arrived at: they now find nothing left to do
            uint64_t nf0 = 0, no0 = 0, nr0 = 0;
            for (auto &x : segs) { if (x.base_first != nf0 || x.base_other != no0) c->compact_err = "chain_device: a segment's place is not the sum of the counts before it"; nf0 += x.n_first; no0 += x.n_other; nr0 += x.n_rec; }
            if (co.n_first != nf0 || co.n_other != no0 || co.n_rec != nr0) c->compact_err = "chain_device: totals disagree with the segments";
            std::vector<uint32_t> r0;
            if (pdb2::check_chain(segs, &r0) != 0) c->compact_err = "chain_device: the host's check finds a segment that does not start where the chain ends";
            for (auto &x : segs) if (x.flags) c->compact_err = "chain_device: a flagged segment was not left to the host";
            uint64_t fs = ~0ull, E0 = 0;                          // unit 0's first record and chain end, as pd_decode_result reports them
            for (uint32_t j = seg0[0]; j < seg0[1]; ++j) { if (fs == ~0ull && segs[j].used_start != pdb2::NONE) fs = segs[j].used_start; if (segs[j].e_last > E0) E0 = segs[j].e_last; }
            if (co.first_start != fs || co.next_start != (E0 ? E0 : ~0ull)) c->compact_err = "chain_device: unit 0's first record / chain end disagree with the segments";
        }
    }
    std::vector<uint32_t> redo;
    for (int round = 0; pdb2::check_chain(segs, &redo) > 0; ++round) {
        if (round >= 24) { for (uint32_t j : redo) segs[j].flags |= pdb2::WF_BAD; break; }
        for (uint32_t j : redo) pdb2::walk_segment<pdw::HostWave>(cfg, segs[j], &lanes[(size_t)j * 64]);
    }
    uint64_t nf = 0, no = 0, nfar = 0, nrec = 0;
    for (uint32_t u = 0; u < bt->n_units; ++u) {
        int st = 0;
        const pd_decode_unit &un = bt->units[u];
        for (uint32_t b = 0; b < un.n_blocks; ++b) { const int v = bst[un.first_block + b]; if (v < 0) st = 2; else if (v > 0 && !st) st = 1; }
        for (uint32_t j = seg0[u]; j < seg0[u + 1]; ++j) { if (segs[j].flags & pdb2::WF_BAD) { if (st != 2) st = 3; } else if ((segs[j].flags & (pdb2::WF_MORE | pdb2::WF_HOST)) && !st) st = 1; }
        status[u] = st;

        for (uint32_t j = seg0[u]; j < seg0[u + 1]; ++j) {
            if (st) segs[j].n_first = segs[j].n_other = segs[j].n_far = 0; else nrec += segs[j].n_rec;
            segs[j].base_first = nf; segs[j].base_other = no; segs[j].base_far = nfar; nf += segs[j].n_first; no += segs[j].n_other; nfar += segs[j].n_far;
        }
    }
    if (res) {
        res->n_first = nf; res->n_other = no + nfar; res->n_reads = nrec;
        uint64_t fs = ~0ull, E = 0;
        for (uint32_t j = seg0[0]; j < seg0[1]; ++j) { if (fs == ~0ull && segs[j].used_start != pdb2::NONE) fs = segs[j].used_start; if (segs[j].e_last > E) E = segs[j].e_last; }
        res->first_start = fs; res->next_start = E ? E : ~0ull;
    }
    pd_ctx::Runs r; r.order = bt->order; r.first.resize(nf + 1); r.other.resize(no + 1); r.far.resize(nfar + 1);
    if (c->compact) {
        // the product's compact emission, then back to 12-byte runs; everything the engine relies on is checked on the way
        const uint32_t cshift = 9;
        std::vector<pdb2::R8> r8(nf + 1);
        std::vector<unsigned long long> b1((size_t)(c->flat_off.back() >> cshift) + 2, ~0ull);
        std::vector<pdb2::SegOut> so(segs.size());
        cfg.c8 = pdb2::C8Out{r8.data(), b1.data(), c->flat_off.data(), cshift, so.data(), (uint32_t)bt->order};
        for (size_t j = 0; j < segs.size(); ++j) {
            if (segs[j].n_first | segs[j].n_other | segs[j].n_far) pdb2::emit_segment<pdw::HostWave>(cfg, segs[j], &lanes[j * 64], nullptr, r.other.data(), r.far.data(), &so[j]);
            else so[j] = pdb2::SegOut{pdb2::NONE, 0, 0, 0};
        }
        std::string bad;
        uint64_t prev_flat = 0;
        for (uint64_t i = 0; i < nf; ++i) {
            // (the harness' genomes are far below 2^32 cells: the 32 bits are the flat begin)
            const uint64_t flat = r8[i].b;
            size_t t = std::upper_bound(c->flat_off.begin(), c->flat_off.end(), flat) - c->flat_off.begin() - 1;
            if (t >= c->len.size()) { bad = "a compact run lies outside the genome"; break; }
            const uint32_t beg = (uint32_t)(flat - c->flat_off[t]);
            if (beg > c->len[t] || r8[i].len > c->len[t] - beg) { bad = "a compact run is not clamped to its contig"; break; }
            r.first[i] = pd_iv{(int32_t)t, (int32_t)beg, (int32_t)(beg + r8[i].len)};
            const bool head = i == 0 || (prev_flat >> cshift) != (flat >> cshift);
            if (head && b1[flat >> cshift] != (((unsigned long long)(uint32_t)bt->order << 32) | (uint32_t)i) && flat >= prev_flat) { bad = "the first run of a bucket did not leave its (batch, index) mark"; break; }
            prev_flat = flat;
        }
        uint64_t first = pdb2::NONE, last = 0; uint32_t uns = 0;
        for (const auto &x : so) { uns |= x.unsorted; if (x.first_key == pdb2::NONE) continue; if (first == pdb2::NONE) first = x.first_key; else if (x.first_key < last) uns = 1; last = x.last_key; }
        if (bad.empty() && nf) {
            bool really = false;
            for (uint64_t i = 1; i < nf; ++i) if (r8[i].b < r8[i - 1].b) really = true;
            if (really != (uns != 0)) bad = "the emission's order flag disagrees with the runs";
            else if (!really && (first != r8[0].b || last != r8[nf - 1].b)) bad = "the emission's first / last keys disagree with the runs";
        }
        if (!bad.empty()) { std::lock_guard<std::mutex> lk(c->mu); c->compact_err = bad; }
        cfg.c8 = pdb2::C8Out{};
    } else {
        // 12-byte emission with the order keys riding along (what the product's device-confirmed batches report their order from)
        std::vector<pdb2::SegOut> so(segs.size(), pdb2::SegOut{pdb2::NONE, 0, 0, 0});
        cfg.c8 = pdb2::C8Out{}; cfg.c8.seg_out = so.data();
        for (size_t j = 0; j < segs.size(); ++j)
            if (segs[j].n_first | segs[j].n_other | segs[j].n_far) pdb2::emit_segment<pdw::HostWave>(cfg, segs[j], &lanes[j * 64], r.first.data(), r.other.data(), r.far.data(), &so[j]);
        cfg.c8 = pdb2::C8Out{};
        uint64_t first = pdb2::NONE, last = 0; uint32_t uns = 0;
        for (const auto &x : so) { uns |= x.unsorted; if (x.first_key == pdb2::NONE) continue; if (first == pdb2::NONE) first = x.first_key; else if (x.first_key < last) uns = 1; last = x.last_key; }
        auto key = [](const pd_iv &v) { return ((uint64_t)(uint32_t)v.tid << 32) | (uint32_t)v.beg; };
        bool really = false;
        for (uint64_t i = 1; i < nf; ++i) if (key(r.first[i]) < key(r.first[i - 1])) really = true;
        std::string bad;
        if (nf && really != (uns != 0)) bad = "the 12-byte emission's order flag disagrees with the runs";
        else if (nf && !really && (first != key(r.first[0]) || last != key(r.first[nf - 1]))) bad = "the 12-byte emission's first / last keys disagree with the runs";
        if (!bad.empty()) { std::lock_guard<std::mutex> lk(c->mu); c->err = bad; return -4; }
    }
    r.first.resize(nf); r.other.resize(no); r.far.resize(nfar);
    if (res && nf) {                                             // order of the first runs, as pd_decode_submit reports it
        auto key = [](const pd_iv &v) { return ((uint64_t)(uint32_t)v.tid << 32) | (uint32_t)v.beg; };
        res->first_key = key(r.first[0]); res->last_key = key(r.first[nf - 1]); res->unsorted = 0;
        for (uint64_t i = 1; i < nf; ++i) if (key(r.first[i]) < key(r.first[i - 1])) { res->unsorted = 1; break; }
    }
    std::lock_guard<std::mutex> lk(c->mu);
    c->runs.push_back(std::move(r));
    return 0;
}
static int o_decode_end(pd_ctx *c)
{
    {   // (the product collects and drops what the caller left queued; the stand-in holds the feeder to collecting every batch itself)
        std::lock_guard<std::mutex> lk(c->mu);
        for (auto *x : c->bufs) if (x->queued) { c->err = "decode_end: a queued batch was never collected"; return -4; }
        if (getenv("PANDEPTH_TEST_CHAIN_STATS")) fprintf(stderr, "[chain] batches confirmed by chain_device: %llu, left to check_chain: %llu, segments chain_device walked again: %llu\n", (unsigned long long)c->n_chain_dev, (unsigned long long)c->n_chain_host, (unsigned long long)c->n_chain_redo);
    }
    {
        std::lock_guard<std::mutex> lk(c->mu);
        if (c->compact) for (uint8_t seen : c->orders_seen) if (seen != 1 && c->compact_err.empty()) c->compact_err = "a batch number of the session was never submitted";
        if (!c->compact_err.empty()) { c->err = "decode session: " + c->compact_err; return -4; }      // (what the stand-in held the product's cores to, in any session)
    }
    std::vector<pd_ctx::Runs> rs;
    { std::lock_guard<std::mutex> lk(c->mu); rs.swap(c->runs); }
    std::sort(rs.begin(), rs.end(), [](const pd_ctx::Runs &a, const pd_ctx::Runs &b) { return a.order < b.order; });
    std::vector<pd_iv> first, other, far;
    for (auto &r : rs) { first.insert(first.end(), r.first.begin(), r.first.end()); other.insert(other.end(), r.other.begin(), r.other.end()); far.insert(far.end(), r.far.begin(), r.far.end()); }
    // the product pushes the first runs as a SORTED batch when the file is coordinate sorted: hold it to that promise
    int rc = first.empty() ? 0 : o_push(c, first.data(), first.size(), c->dcfg.sorted ? PD_PUSH_SORTED : PD_PUSH_DEFAULT);
    // ... and the near stream as sorted within NEAR_SPAN cells, as the product promises the engine
    if (!rc && !other.empty()) rc = o_push(c, other.data(), other.size(), (c->dcfg.sorted && getenv("PANDEPTH_TEST_NEAR_SPAN")) ? (PD_PUSH_SORTED | PD_PUSH_DISORDER((unsigned)atoi(getenv("PANDEPTH_TEST_NEAR_SPAN")))) : PD_PUSH_DEFAULT);
    if (!rc && !far.empty()) rc = o_push(c, far.data(), far.size(), PD_PUSH_DEFAULT);
    return rc;
}
static int o_decode_abort(pd_ctx *c) { std::lock_guard<std::mutex> lk(c->mu); c->runs.clear(); for (auto *x : c->bufs) { x->queued = false; x->busy = false; } return 0; }
static int o_set_param(pd_ctx *, const char *, uint64_t) { return 0; }
// pd_deflate_parse on the CPU: the product's parse (csrc/pd_lz77.h) with its 64 lanes in a loop
static int o_deflate_parse(pd_ctx *, const void *text, size_t n, const pd_lz_chunk *chunks, uint32_t n_chunks, uint32_t *syms, size_t cap, uint64_t *off)
{
    static const pgz::ParseFn fn = pgz::host_emulation_parse();
    pgz::SymVec sy; std::vector<uint64_t> of;
    if (!fn((const uint8_t *)text, n, reinterpret_cast<const uint64_t *>(chunks), n_chunks, sy, of)) return -5;
    if (sy.size() > cap) return -6;
    memcpy(syms, sy.data(), sy.size() * 4);
    memcpy(off, of.data(), of.size() * 8);
    return 0;
}

// pd_text_* on the CPU: the stream is a byte vector (with a small capacity, so that the ring's "full" answer is exercised too), rows
// formatted with printf, the parse by the product's core in host emulation, CRCs by zlib
struct pd_text { pd_ctx *c; std::vector<uint8_t> bytes; uint64_t base = 0; size_t cap; std::mutex mu; };
static int o_text_open(pd_ctx *c, size_t cap, pd_text **out)
{
    pd_text *t = new pd_text; t->c = c;
    t->cap = getenv("PANDEPTH_TEST_TEXT_CAP") ? (size_t)atoll(getenv("PANDEPTH_TEST_TEXT_CAP")) : cap;
    *out = t; return 0;
}
static int o_text_close(pd_text *t) { delete t; return 0; }
static int o_text_append_sites(pd_text *t, int32_t tid, uint32_t beg, size_t n, const char *name, size_t name_len, uint64_t *n_bytes)
{
    pd_ctx *c = t->c;
    if (!c->scanned) { c->err = "scan first"; return -4; }
    std::string rows, nm(name, name_len);
    char num[64];
    for (size_t j = 0; j < n; ++j) {
        rows += nm;
        snprintf(num, sizeof num, "\t%u\t%u\n", beg + (uint32_t)j, c->depth[(size_t)c->off[tid] + beg + j]);
        rows += num;
    }
    std::lock_guard<std::mutex> lk(t->mu);
    if (t->bytes.size() + rows.size() > t->cap) { c->err = "text stream full"; return -6; }
    t->bytes.insert(t->bytes.end(), rows.begin(), rows.end());
    *n_bytes = rows.size();
    return 0;
}
static int text_put(pd_text *t, const char *p, size_t n)
{
    std::lock_guard<std::mutex> lk(t->mu);
    if (t->bytes.size() + n > t->cap) { t->c->err = "text stream full"; return -6; }
    t->bytes.insert(t->bytes.end(), p, p + n);
    return 0;
}
static int o_text_append_window_rows(pd_text *t, int32_t tid, uint32_t w, uint64_t row_first, size_t n_rows, const char *name, size_t name_len, uint64_t *n_bytes)
{
    pd_ctx *c = t->c;
    if (c->wk_w != w || c->wk_woff.empty()) { c->err = "no window statistics of this width"; return -4; }
    std::string rows, nm(name, name_len);
    char buf[256];
    const int64_t len = c->len[(size_t)tid];
    for (size_t i = 0; i < n_rows; ++i) {
        const uint64_t k = row_first + i;
        const int64_t j = 1 + (int64_t)k * w;
        int64_t end = j - 1 + w; if (end > len) end = len;
        const int64_t L = end - j + 1;
        const int32_t cc = (int32_t)c->wk_cov[(size_t)(c->wk_woff[(size_t)tid] + k)], d = (int32_t)c->wk_sum[(size_t)(c->wk_woff[(size_t)tid] + k)];
        snprintf(buf, sizeof buf, "\t%lld\t%lld\t%lld\t%d\t%d\t%.2f\t%.2f\n", (long long)j, (long long)end, (long long)L, cc, d, cc * 100.0 / L, d * 1.0 / L);
        rows += nm; rows += buf;
    }
    const int rc = text_put(t, rows.data(), rows.size());
    if (rc) return rc;
    *n_bytes = rows.size();
    return 0;
}
static int o_text_append_bytes(pd_text *t, const void *p, size_t n) { return text_put(t, (const char *)p, n); }
static int o_text_parse(pd_text *t, uint64_t off, size_t n, const pd_lz_chunk *chunks, uint32_t n_chunks, uint32_t *syms, size_t cap, uint64_t *soff, uint32_t *crc, uint64_t crc_span)
{
    // PANDEPTH_TEST_TEXT_PARSE_FAIL=k: every k-th call fails (the host logic must then parse those chunks with zlib itself)
    static std::atomic<unsigned> calls{0};
    if (const char *e = getenv("PANDEPTH_TEST_TEXT_PARSE_FAIL")) { const unsigned k = (unsigned)atoi(e); if (k && (++calls % k) == 0) { t->c->err = "injected failure"; return -5; } }
    std::vector<uint8_t> text;
    {
        std::lock_guard<std::mutex> lk(t->mu);
        if (off < t->base || off + n > t->base + t->bytes.size()) { t->c->err = "stretch not in the stream"; return -1; }
        text.assign(t->bytes.begin() + (std::ptrdiff_t)(off - t->base), t->bytes.begin() + (std::ptrdiff_t)(off - t->base + n));
    }
    const int rc = o_deflate_parse(t->c, text.data(), n, chunks, n_chunks, syms, cap, soff);
    if (rc) return rc;
    if (crc) for (uint32_t k = 0; k < n_chunks; ++k) crc[k] = (uint32_t)crc32(crc32(0L, Z_NULL, 0), text.data() + chunks[k].start, (uInt)std::min<uint64_t>(crc_span, chunks[k].end - chunks[k].start));
    return 0;
}
static int o_text_read(pd_text *t, uint64_t off, size_t n, void *out)
{
    std::lock_guard<std::mutex> lk(t->mu);
    if (off < t->base || off + n > t->base + t->bytes.size()) { t->c->err = "stretch not in the stream"; return -1; }
    memcpy(out, t->bytes.data() + (off - t->base), n);
    return 0;
}
static int o_text_release(pd_text *t, uint64_t off)
{
    std::lock_guard<std::mutex> lk(t->mu);
    if (off > t->base) { const uint64_t k = std::min<uint64_t>(off - t->base, t->bytes.size()); t->bytes.erase(t->bytes.begin(), t->bytes.begin() + (std::ptrdiff_t)k); t->base += k; }
    return 0;
}

int main(int argc, char **argv)
{
    // (keep_deferred: nothing to do on this engine; deflate_parse: the product's parse core in host emulation, like the decoder's cores)
    const bool no_parse = getenv("PANDEPTH_TEST_NO_PARSE") != nullptr;      // (zlib alone writes the gzip streams)
    static const pd_engine_api api = {o_create, o_destroy, o_strerror, o_push, o_scan, o_reduce_intervals,
                                      o_layout, o_scan_reduce_windows, o_reduce_windows, o_read_depth, o_sync, nullptr, o_device_count, o_accumulate_from,
                                      o_decode_begin, o_decode_acquire, o_decode_submit, o_decode_end, o_decode_abort, o_set_param,
                                      nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, no_parse ? nullptr : o_deflate_parse, nullptr, nullptr,
                                      no_parse ? nullptr : o_text_open, o_text_close, o_text_append_sites, o_text_parse, o_text_read, o_text_release,
                                      no_parse ? nullptr : o_text_append_window_rows, o_text_append_bytes, nullptr,
                                      getenv("PANDEPTH_TEST_NO_QUEUE") ? nullptr : o_decode_queue, getenv("PANDEPTH_TEST_NO_QUEUE") ? nullptr : o_decode_collect};
    return pandepth_main(argc, argv, &api, 0);
}
