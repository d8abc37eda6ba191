// tests/harness/oracle_engine.cpp — TEST INFRASTRUCTURE.
// A pd_engine_api implemented on the CPU with the oracle's loops (oracle/pd_oracle.c: one
// increment per covered base, one compare+add per region base), used ONLY by the no-GPU tests to
// drive the product's HOST code (readers, read selection, region model, table writer) and
// compare its files byte-for-byte with the reference's golden outputs.  It is not linked into
// libpandepth_amd.so, libpandepth_host.a or the `pandepth` binary.
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <string>
#include <vector>
#include "../../pandepth_amd/host/engine_api.h"

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
    std::mutex mu;
    std::string err;
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
static void windows(const pd_ctx *c, uint32_t w, uint32_t md, uint32_t mask, uint32_t *cov, uint64_t *sum)
{
    uint64_t k = 0;
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

int main(int argc, char **argv)
{
    static const pd_engine_api api = {o_create, o_destroy, o_strerror, o_push, o_scan, o_reduce_intervals,
                                      o_layout, o_scan_reduce_windows, o_reduce_windows, o_read_depth, o_sync, nullptr, o_device_count, o_accumulate_from};
    return pandepth_main(argc, argv, &api, 0);
}
