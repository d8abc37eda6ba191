// cram.cpp — see cram.h.  Section numbers refer to the CRAM format specification, version 3.0.
#include "cram.h"
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include <map>
#include <memory>
#include "bam.h"

namespace pdh {

namespace {

// ---- byte cursor with the variable-length integers of §2.3 ------------------------------------------------------
struct Cur {
    const uint8_t *p, *e;
    bool ok = true;
    Cur(const uint8_t *b, size_t n) : p(b), e(b + n) {}
    int u8() { if (p >= e) { ok = false; return 0; } return *p++; }
    int32_t itf8()
    {
        const int b0 = u8();
        if (b0 < 0x80) return b0;
        if (b0 < 0xc0) return ((b0 & 0x3f) << 8) | u8();
        if (b0 < 0xe0) { const int b1 = u8(), b2 = u8(); return ((b0 & 0x1f) << 16) | (b1 << 8) | b2; }
        if (b0 < 0xf0) { const int b1 = u8(), b2 = u8(), b3 = u8(); return ((b0 & 0x0f) << 24) | (b1 << 16) | (b2 << 8) | b3; }
        const uint32_t b1 = (uint32_t)u8(), b2 = (uint32_t)u8(), b3 = (uint32_t)u8(), b4 = (uint32_t)u8();
        return (int32_t)((((uint32_t)b0 & 0x0f) << 28) | (b1 << 20) | (b2 << 12) | (b3 << 4) | (b4 & 0x0f));
    }
    int64_t ltf8()
    {
        const int b0 = u8();
        int extra = 0;
        while (extra < 8 && (b0 & (0x80 >> extra))) ++extra;
        uint64_t v = extra >= 7 ? 0 : (uint64_t)(b0 & (0xff >> (extra + 1)));
        for (int k = 0; k < extra; ++k) v = (v << 8) | (uint64_t)u8();
        return (int64_t)v;
    }
    uint32_t u32le() { uint32_t v = 0; for (int k = 0; k < 4; ++k) v |= (uint32_t)u8() << (8 * k); return v; }
    const uint8_t *take(size_t n) { if ((size_t)(e - p) < n) { ok = false; p = e; return nullptr; } const uint8_t *r = p; p += n; return r; }
    size_t left() const { return (size_t)(e - p); }
};

// ---- rANS 4x8 (§ "rANS codec"): four interleaved 32-bit states, 12-bit frequencies ------------------------------------
bool rans_read_freqs(Cur &c, uint16_t F[256], uint16_t C[256], uint8_t *lookup)
{
    memset(F, 0, 512); memset(C, 0, 512);
    int rle = 0, x = 0;
    int j = c.u8();
    do {
        int f = c.u8();
        if (f >= 128) f = ((f & 127) << 8) | c.u8();
        F[j] = (uint16_t)f; C[j] = (uint16_t)x;
        if (x + f > 4096 || !c.ok) return false;
        memset(lookup + x, j, (size_t)f);
        x += f;
        if (!rle && c.p < c.e && j + 1 == *c.p) { j = c.u8(); rle = c.u8(); }
        else if (rle) { --rle; ++j; }
        else j = c.u8();
    } while (j && j < 256 && c.ok);
    return c.ok && j < 256;
}

bool rans_decode(const uint8_t *in, size_t n_in, std::vector<uint8_t> *out)
{
    Cur c(in, n_in);
    const int order = c.u8();
    const uint32_t csize = c.u32le(), osize = c.u32le();
    (void)csize;
    if (!c.ok || osize > ((uint32_t)1 << 30)) return false;
    out->assign(osize, 0);
    if (osize == 0) return true;
    uint8_t *o = out->data();
    auto renorm = [&](uint32_t &R) { while (R < (1u << 23) && c.p < c.e) R = (R << 8) | *c.p++; };
    if (order == 0) {
        uint16_t F[256], C[256];
        std::vector<uint8_t> lookup(4096, 0);
        if (!rans_read_freqs(c, F, C, lookup.data())) return false;
        uint32_t R[4];
        for (int k = 0; k < 4; ++k) R[k] = c.u32le();
        if (!c.ok) return false;
        const size_t n4 = osize & ~(size_t)3;
        for (size_t i = 0; i < n4; i += 4)
            for (int k = 0; k < 4; ++k) {
                const uint32_t m = R[k] & 0xfff;
                const uint8_t s = lookup[m];
                o[i + k] = s;
                R[k] = F[s] * (R[k] >> 12) + m - C[s];
                renorm(R[k]);
            }
        for (size_t i = n4, k = 0; i < osize; ++i, ++k) {
            const uint32_t m = R[k] & 0xfff;
            const uint8_t s = lookup[m];
            o[i] = s;
            R[k] = F[s] * (R[k] >> 12) + m - C[s];
            renorm(R[k]);
        }
        return true;
    }
    if (order != 1) return false;
    // order 1: one frequency table per preceding symbol
    std::vector<uint16_t> F(256 * 256, 0), Cm(256 * 256, 0);
    std::vector<uint8_t> lookup((size_t)256 * 4096, 0);
    {
        int rle_i = 0;
        int i = c.u8();
        do {
            if (!rans_read_freqs(c, &F[(size_t)i * 256], &Cm[(size_t)i * 256], &lookup[(size_t)i * 4096])) return false;
            if (!rle_i && c.p < c.e && i + 1 == *c.p) { i = c.u8(); rle_i = c.u8(); }
            else if (rle_i) { --rle_i; ++i; }
            else i = c.u8();
        } while (i && i < 256 && c.ok);
        if (i >= 256) return false;
    }
    uint32_t R[4];
    for (int k = 0; k < 4; ++k) R[k] = c.u32le();
    if (!c.ok) return false;
    const size_t q = osize >> 2;
    size_t idx[4] = {0, q, 2 * q, 3 * q};
    int last[4] = {0, 0, 0, 0};
    for (; idx[0] < q; ) {
        for (int k = 0; k < 4; ++k) {
            const uint32_t m = R[k] & 0xfff;
            const size_t ctx = (size_t)last[k];
            const uint8_t s = lookup[ctx * 4096 + m];
            o[idx[k]++] = s;
            R[k] = F[ctx * 256 + s] * (R[k] >> 12) + m - Cm[ctx * 256 + s];
            renorm(R[k]);
            last[k] = s;
        }
    }
    for (; idx[3] < osize; ) {
        const uint32_t m = R[3] & 0xfff;
        const size_t ctx = (size_t)last[3];
        const uint8_t s = lookup[ctx * 4096 + m];
        o[idx[3]++] = s;
        R[3] = F[ctx * 256 + s] * (R[3] >> 12) + m - Cm[ctx * 256 + s];
        renorm(R[3]);
        last[3] = s;
    }
    return true;
}

// ---- rANS Nx16 (CRAM 3.1, block method 5): N = 4 or 32 interleaved states renormalised 16 bits at a time, 12- or
// 10-bit frequencies, with the byte-stream transforms around it (stripe, bit packing, run lengths, stored) ------------------
struct Cur7 : Cur {
    using Cur::Cur;
    uint32_t u7() { uint32_t v = 0; int c; int n = 0; do { c = u8(); v = (v << 7) | (uint32_t)(c & 0x7f); } while ((c & 0x80) && ok && ++n < 6); return v; }
};

bool nx16_alphabet(Cur7 &c, bool A[256])
{
    memset(A, 0, 256);
    int rle = 0, j = c.u8();
    do {
        A[j] = true;
        if (!rle && c.p < c.e && j + 1 == *c.p) { j = c.u8(); rle = c.u8(); }
        else if (rle) { --rle; ++j; }
        else j = c.u8();
    } while (j && j < 256 && c.ok);
    return c.ok;
}

// scales a row of frequencies whose sum is a smaller power of two up to 1 << bits and fills its lookup
bool nx16_finish_row(uint32_t *F, uint32_t *C, uint8_t *lookup, int bits)
{
    // every frequency comes straight from the file (a uint7 of up to 32 bits): bound each one and add them up in
    // 64 bits, so that a crafted row cannot wrap the sum back to a legal total and drive the memset below past the table
    const uint32_t want = 1u << bits;
    uint64_t tot64 = 0;
    for (int j = 0; j < 256; ++j) {
        if (F[j] > want) return false;
        tot64 += F[j];
        if (tot64 > want) return false;
    }
    const uint32_t tot = (uint32_t)tot64;
    if (tot == 0) return true;
    int shift = 0;
    while ((tot << shift) < want) ++shift;
    if ((tot << shift) != want) return false;
    uint32_t x = 0;
    for (int j = 0; j < 256; ++j) {
        F[j] <<= shift; C[j] = x;
        if (F[j] > want - x) return false;
        if (F[j]) memset(lookup + x, j, F[j]);
        x += F[j];
    }
    return true;
}

bool nx16_order0(Cur7 &c, uint8_t *o, size_t osize, int N)
{
    bool A[256];
    if (!nx16_alphabet(c, A)) return false;
    uint32_t F[256] = {0}, C[256] = {0};
    for (int j = 0; j < 256; ++j) if (A[j]) F[j] = c.u7();
    std::vector<uint8_t> lookup(4096, 0);
    if (!c.ok || !nx16_finish_row(F, C, lookup.data(), 12)) return false;
    uint32_t R[32];
    for (int k = 0; k < N; ++k) R[k] = c.u32le();
    if (!c.ok) return false;
    for (size_t i = 0; i < osize; ++i) {
        uint32_t &r = R[i % (size_t)N];
        const uint32_t m = r & 0xfff;
        const uint8_t s = lookup[m];
        o[i] = s;
        r = F[s] * (r >> 12) + m - C[s];
        if (r < (1u << 15) && c.e - c.p >= 2) { r = (r << 16) | (uint32_t)c.p[0] | ((uint32_t)c.p[1] << 8); c.p += 2; }
    }
    return true;
}

bool nx16_order1(Cur7 &c, uint8_t *o, size_t osize, int N)
{
    const int comp = c.u8();
    const int bits = comp >> 4;
    if (!c.ok || bits < 8 || bits > 12) return false;
    std::vector<uint8_t> table;
    Cur7 t(c.p, (size_t)(c.e - c.p));
    if (comp & 1) {                                       // the frequency table is itself order-0 compressed
        const uint32_t usz = c.u7(), csz = c.u7();
        const uint8_t *tp = c.take(csz);
        if (!c.ok) return false;
        table.assign(usz, 0);
        Cur7 tc(tp, csz);
        if (!nx16_order0(tc, table.data(), usz, 4)) return false;
        t = Cur7(table.data(), table.size());
    }
    bool A[256];
    if (!nx16_alphabet(t, A)) return false;
    const size_t W = (size_t)1 << bits;
    std::vector<uint32_t> F(256 * 256, 0), C(256 * 256, 0);
    std::vector<uint8_t> lookup(256 * W, 0);
    for (int i = 0; i < 256; ++i) {
        if (!A[i]) continue;
        int run = 0;
        uint32_t *Fi = &F[(size_t)i * 256];
        for (int j = 0; j < 256; ++j) {
            if (!A[j]) continue;
            if (run) { --run; continue; }
            Fi[j] = t.u7();
            if (Fi[j] == 0) run = t.u8();
        }
        if (!t.ok || !nx16_finish_row(Fi, &C[(size_t)i * 256], &lookup[(size_t)i * W], bits)) return false;
    }
    if (!(comp & 1)) c.p = t.p;                             // the table was read in place
    uint32_t R[32];
    for (int k = 0; k < N; ++k) R[k] = c.u32le();
    if (!c.ok) return false;
    const size_t q = osize / (size_t)N;
    size_t idx[32];
    int last[32];
    for (int k = 0; k < N; ++k) { idx[k] = (size_t)k * q; last[k] = 0; }
    const uint32_t mask = (uint32_t)W - 1;
    auto step = [&](int k) {
        uint32_t &r = R[k];
        const uint32_t m = r & mask;
        const size_t ctx = (size_t)last[k];
        const uint8_t s = lookup[ctx * W + m];
        o[idx[k]++] = s;
        r = F[ctx * 256 + s] * (r >> bits) + m - C[ctx * 256 + s];
        if (r < (1u << 15) && c.e - c.p >= 2) { r = (r << 16) | (uint32_t)c.p[0] | ((uint32_t)c.p[1] << 8); c.p += 2; }
        last[k] = s;
    };
    for (size_t i = 0; i < q; ++i) for (int k = 0; k < N; ++k) step(k);
    while (idx[N - 1] < osize) step(N - 1);
    return true;
}

bool nx16_decode(const uint8_t *in, size_t n_in, std::vector<uint8_t> *out, size_t known_size, bool have_size, int depth)
{
    if (depth > 3 || n_in == 0) { out->clear(); return n_in == 0 && known_size == 0; }
    Cur7 c(in, n_in);
    const int flags = c.u8();
    if (getenv("PANDEPTH_CRAM_DEBUG")) fprintf(stderr, "[cram]   nx16 flags 0x%02x in %zu depth %d\n", flags, n_in, depth);
    const bool order1 = flags & 0x01, x32 = flags & 0x04, stripe = flags & 0x08, nosz = flags & 0x10, cat = flags & 0x20, rle = flags & 0x40, pack = flags & 0x80;
    size_t osize = known_size;
    if (!nosz) { osize = c.u7(); if (!have_size && known_size && osize != known_size) return false; }     // (the block header's size, when the caller has one)
    else if (!have_size) return false;
    if (!c.ok || osize > ((size_t)1 << 30)) return false;
    if (stripe) {
        const int n = c.u8();
        if (n <= 0) return false;
        std::vector<uint32_t> clen((size_t)n);
        for (int k = 0; k < n; ++k) clen[(size_t)k] = c.u7();
        out->assign(osize, 0);
        for (int k = 0; k < n; ++k) {
            const size_t ulen = osize / (size_t)n + ((osize % (size_t)n) > (size_t)k ? 1 : 0);
            const uint8_t *sp = c.take(clen[(size_t)k]);
            if (!c.ok) return false;
            std::vector<uint8_t> sub;
            if (!nx16_decode(sp, clen[(size_t)k], &sub, ulen, true, depth + 1) || sub.size() != ulen) return false;
            for (size_t i = 0; i < ulen; ++i) (*out)[i * (size_t)n + (size_t)k] = sub[i];
        }
        return true;
    }
    // transforms announce themselves before the entropy-coded bytes, outermost first
    uint8_t pmap[256]; int psym = 0; size_t unpacked = 0;
    if (pack) {
        psym = c.u8();
        if (psym == 0) psym = 256;
        for (int k = 0; k < psym && k < 256; ++k) pmap[k] = (uint8_t)c.u8();
        unpacked = osize;
        osize = c.u7();
    }
    std::vector<uint8_t> rmeta; size_t unrle = 0;
    if (rle) {
        const uint32_t umeta = c.u7();
        const uint32_t rlen = c.u7();
        if (umeta & 1) { const uint8_t *mp = c.take(umeta / 2); if (!c.ok) return false; rmeta.assign(mp, mp + umeta / 2); }
        else {
            const uint32_t cmeta = c.u7();
            const uint8_t *mp = c.take(cmeta);
            if (!c.ok) return false;
            rmeta.assign(umeta / 2, 0);
            Cur7 mc(mp, cmeta);
            if (!nx16_order0(mc, rmeta.data(), rmeta.size(), x32 ? 32 : 4)) return false;      // the meta stream follows the block's interleave
        }
        unrle = osize;
        osize = rlen;
    }
    if (!c.ok || osize > ((size_t)1 << 30)) return false;
    std::vector<uint8_t> cur(osize, 0);
    if (cat) { const uint8_t *d = c.take(osize); if (!c.ok) return false; if (osize) memcpy(cur.data(), d, osize); }
    else if (osize) { if (!(order1 ? nx16_order1(c, cur.data(), osize, x32 ? 32 : 4) : nx16_order0(c, cur.data(), osize, x32 ? 32 : 4))) return false; }
    if (rle) {
        // meta: the symbols that carry run lengths, then the lengths (uint7) in order of appearance
        Cur7 m(rmeta.data(), rmeta.size());
        int ns = m.u8();
        if (ns == 0) ns = 256;
        bool has[256] = {false};
        for (int k = 0; k < ns; ++k) has[m.u8()] = true;
        std::vector<uint8_t> o2;
        o2.reserve(unrle);
        for (size_t i = 0; i < cur.size(); ++i) {
            const uint8_t b = cur[i];
            o2.push_back(b);
            if (has[b]) { const uint32_t run = m.u7(); if (o2.size() + run > unrle) return false; o2.insert(o2.end(), run, b); }
        }
        if (!m.ok || o2.size() != unrle) return false;
        cur.swap(o2);
    }
    if (pack) {
        std::vector<uint8_t> o2(unpacked, 0);
        if (psym <= 1) memset(o2.data(), pmap[0], unpacked);
        else if (psym <= 2) { for (size_t i = 0; i < unpacked; ++i) { if ((i >> 3) >= cur.size()) return false; o2[i] = pmap[(cur[i >> 3] >> (i & 7)) & 1]; } }
        else if (psym <= 4) { for (size_t i = 0; i < unpacked; ++i) { if ((i >> 2) >= cur.size()) return false; o2[i] = pmap[(cur[i >> 2] >> ((i & 3) * 2)) & 3]; } }
        else if (psym <= 16) { for (size_t i = 0; i < unpacked; ++i) { if ((i >> 1) >= cur.size()) return false; o2[i] = pmap[(cur[i >> 1] >> ((i & 1) * 4)) & 15]; } }
        else { if (cur.size() != unpacked) return false; o2 = cur; }
        cur.swap(o2);
    }
    out->swap(cur);
    return true;
}

// ---- adaptive arithmetic coder (CRAM 3.1, block method 6): a byte-wise range coder over adaptive frequency models, order
// 0 or 1, optionally with run lengths modelled beside the symbols, inside the same stripe / pack / stored framing ---------
struct RangeDec {
    const uint8_t *p, *e;
    uint32_t range = 0xffffffffu, code = 0;
    bool ok = true;
    RangeDec(const uint8_t *b, const uint8_t *end) : p(b), e(end)
    {
        if (e - p < 5) { ok = false; p = e; return; }
        for (int k = 0; k < 5; ++k) code = (code << 8) | *p++;      // the first byte is the encoder's carry cache
    }
    uint32_t freq(uint32_t tot) { range /= tot; if (!range) { ok = false; return 0; } return code / range; }
    void decode(uint32_t cum, uint32_t f)
    {
        code -= cum * range;
        range *= f;
        while (range < (1u << 24)) { if (p >= e) { ok = false; return; } code = (code << 8) | *p++; range <<= 8; }
    }
};

template <int NSYM> struct AdaptModel {
    static constexpr uint32_t MAXF = (1u << 16) - 17, STEP = 16;
    uint32_t tot;
    struct SF { uint16_t f, s; } v[NSYM + 2];            // v[0] is a sentinel that outranks everything, v[NSYM + 1] ends the list
    void init(int max_sym)
    {
        v[0].f = (uint16_t)MAXF; v[0].s = 0;
        for (int i = 0; i < NSYM; ++i) { v[i + 1].s = (uint16_t)i; v[i + 1].f = i < max_sym ? 1 : 0; }
        v[NSYM + 1].f = 0; v[NSYM + 1].s = 0;
        tot = (uint32_t)max_sym;
    }
    int get(RangeDec &rc)
    {
        const uint32_t fr = rc.freq(tot);
        if (!rc.ok || fr > MAXF) { rc.ok = false; return 0; }
        int k = 1;
        uint32_t acc = 0;
        while (k <= NSYM && (acc += v[k].f) <= fr) ++k;
        if (k > NSYM) { rc.ok = false; return 0; }
        acc -= v[k].f;
        rc.decode(acc, v[k].f);
        v[k].f = (uint16_t)(v[k].f + STEP); tot += STEP;
        if (tot > MAXF) { tot = 0; for (int j = 1; j <= NSYM && v[j].f; ++j) { v[j].f = (uint16_t)(v[j].f - (v[j].f >> 1)); tot += v[j].f; } }
        if (v[k].f > v[k - 1].f) { const SF t = v[k]; v[k] = v[k - 1]; v[k - 1] = t; return t.s; }      // kept roughly sorted
        return v[k].s;
    }
};

bool arith_decode(const uint8_t *in, size_t n_in, std::vector<uint8_t> *out, size_t known_size, bool have_size, int depth)
{
    if (depth > 3 || n_in == 0) { out->clear(); return n_in == 0 && known_size == 0; }
    Cur7 c(in, n_in);
    const int flags = c.u8();
    const bool order1 = flags & 0x01, ext = flags & 0x04, stripe = flags & 0x08, nosz = flags & 0x10, cat = flags & 0x20, rle = flags & 0x40, pack = flags & 0x80;
    size_t osize = known_size;
    if (!nosz) { osize = c.u7(); if (!have_size && known_size && osize != known_size) return false; }     // (the block header's size, when the caller has one)
    else if (!have_size) return false;
    if (!c.ok || osize > ((size_t)1 << 30) || ext || (flags & 0x02)) return false;         // EXT = bzip2 inside; order 2 does not exist
    if (stripe) {
        const int n = c.u8();
        if (n <= 0) return false;
        std::vector<uint32_t> clen((size_t)n);
        for (int k = 0; k < n; ++k) clen[(size_t)k] = c.u7();
        out->assign(osize, 0);
        for (int k = 0; k < n; ++k) {
            const size_t ulen = osize / (size_t)n + ((osize % (size_t)n) > (size_t)k ? 1 : 0);
            const uint8_t *sp = c.take(clen[(size_t)k]);
            if (!c.ok) return false;
            std::vector<uint8_t> sub;
            if (!arith_decode(sp, clen[(size_t)k], &sub, ulen, true, depth + 1) || sub.size() != ulen) return false;
            for (size_t i = 0; i < ulen; ++i) (*out)[i * (size_t)n + (size_t)k] = sub[i];
        }
        return true;
    }
    uint8_t pmap[256]; int psym = 0; size_t unpacked = 0;
    if (pack) {
        psym = c.u8();
        if (psym == 0) psym = 256;
        for (int k = 0; k < psym && k < 256; ++k) pmap[k] = (uint8_t)c.u8();
        unpacked = osize;
        osize = c.u7();
    }
    if (!c.ok || osize > ((size_t)1 << 30)) return false;
    std::vector<uint8_t> cur(osize, 0);
    if (cat) { const uint8_t *d = c.take(osize); if (!c.ok) return false; if (osize) memcpy(cur.data(), d, osize); }
    else if (osize) {
        int m = c.u8();
        if (m == 0) m = 256;
        RangeDec rc(c.p, c.e);
        typedef AdaptModel<256> BM;
        typedef AdaptModel<258> RM;
        std::vector<BM> bm(order1 ? 256 : 1);
        for (BM &x : bm) x.init(m);
        std::vector<RM> rm;
        if (rle) { rm.resize(258); for (RM &x : rm) x.init(4); }
        int last = 0;
        for (size_t i = 0; i < osize && rc.ok; ++i) {
            const int s = bm[order1 ? (size_t)last : 0].get(rc);
            cur[i] = (uint8_t)s;
            last = s;
            if (rle) {
                // the run that follows the symbol: parts of 0..3, a part of 3 says "more follows"; the context moves from the
                // symbol to 256, 257 and stays there
                uint32_t run = 0, part;
                int rctx = s;
                do {
                    part = (uint32_t)rm[(size_t)rctx].get(rc);
                    if (rctx == s) rctx = 256; else if (rctx < 257) ++rctx;
                    run += part;
                } while (part == 3 && rc.ok && run < ((uint32_t)1 << 30));
                for (uint32_t j = 0; j < run && i + 1 < osize; ++j) cur[++i] = (uint8_t)s;
            }
        }
        if (!rc.ok) return false;
    }
    if (pack) {
        std::vector<uint8_t> o2(unpacked, 0);
        if (psym <= 1) memset(o2.data(), pmap[0], unpacked);
        else if (psym <= 2) { for (size_t i = 0; i < unpacked; ++i) { if ((i >> 3) >= cur.size()) return false; o2[i] = pmap[(cur[i >> 3] >> (i & 7)) & 1]; } }
        else if (psym <= 4) { for (size_t i = 0; i < unpacked; ++i) { if ((i >> 2) >= cur.size()) return false; o2[i] = pmap[(cur[i >> 2] >> ((i & 3) * 2)) & 3]; } }
        else if (psym <= 16) { for (size_t i = 0; i < unpacked; ++i) { if ((i >> 1) >= cur.size()) return false; o2[i] = pmap[(cur[i >> 1] >> ((i & 1) * 4)) & 15]; } }
        else { if (cur.size() != unpacked) return false; o2 = cur; }
        cur.swap(o2);
    }
    out->swap(cur);
    return true;
}

// ---- blocks (§8.1) ------------------------------------------------------------------------------------------------
struct Block {
    int method = 0, type = 0;
    int32_t id = 0;
    const uint8_t *comp = nullptr;         // the compressed bytes inside the container body
    int32_t csize = 0, rsize = 0;
    bool ready = false;                    // data holds the uncompressed bytes
    std::vector<uint8_t> data;
};

bool inflate_block(Block *b, std::string *err);

// lazy = true: only the block header is parsed; inflate_block() runs when somebody needs the bytes (the external blocks
// of series the depth path never reads — qualities, names, bases, tags — are never decompressed)
// CRAM 2.1 differs from 3.0 in framing only: no CRC32 after container headers and blocks, 32-bit record counters
bool read_block(Cur &c, Block *b, std::string *err, bool lazy = false, bool v2 = false)
{
    b->method = c.u8(); b->type = c.u8(); b->id = c.itf8();
    b->csize = c.itf8(); b->rsize = c.itf8();
    if (!c.ok || b->csize < 0 || b->rsize < 0) { *err = "truncated CRAM block header"; return false; }
    b->comp = c.take((size_t)b->csize);
    if (!v2) c.take(4);                                  // CRC32
    if (!c.ok) { *err = "truncated CRAM block"; return false; }
    b->ready = false;
    return lazy ? true : inflate_block(b, err);
}

bool inflate_block(Block *b, std::string *err)
{
    if (b->ready) return true;
    b->ready = true;
    const uint8_t *d = b->comp;
    const int32_t csize = b->csize, rsize = b->rsize;
    if (getenv("PANDEPTH_CRAM_DEBUG")) fprintf(stderr, "[cram] block method %d%s type %d id %d %d -> %d bytes\n", b->method,
                                               b->method == 4 && csize > 0 ? (d[0] ? " (order 1)" : " (order 0)") : "", b->type, b->id, csize, rsize);
    switch (b->method) {
    case 0: b->data.assign(d, d + csize); return true;
    case 1: {
        b->data.assign((size_t)rsize, 0);
        z_stream z; memset(&z, 0, sizeof z);
        if (inflateInit2(&z, 15 + 32) != Z_OK) { *err = "zlib init failed"; return false; }
        z.next_in = const_cast<Bytef *>(d); z.avail_in = (uInt)csize;
        z.next_out = b->data.data(); z.avail_out = (uInt)rsize;
        const int rc = inflate(&z, Z_FINISH);
        inflateEnd(&z);
        if (rc != Z_STREAM_END && !(rc == Z_OK && z.avail_out == 0) && !(rc == Z_BUF_ERROR && z.avail_out == 0)) { *err = "corrupt gzip block in CRAM"; return false; }
        return true;
    }
    case 4:
        if (!rans_decode(d, (size_t)csize, &b->data) || b->data.size() != (size_t)rsize) { *err = "corrupt rANS block in CRAM"; return false; }
        return true;
    case 5:
        if (!nx16_decode(d, (size_t)csize, &b->data, (size_t)rsize, false, 0) || b->data.size() != (size_t)rsize) { *err = "corrupt rANS Nx16 block in CRAM"; return false; }
        return true;
    case 6:
        if (!arith_decode(d, (size_t)csize, &b->data, (size_t)rsize, false, 0) || b->data.size() != (size_t)rsize) { *err = "corrupt or unsupported (bzip2 inside) arithmetic-coded block in CRAM"; return false; }
        return true;
    default:
        *err = "CRAM block compression method " + std::to_string(b->method) + " is not supported (bzip2 / lzma; fqzcomp and the name tokeniser only hold qualities and names)";
        return false;
    }
}

// ---- encodings (§13) ------------------------------------------------------------------------------------------------
struct Enc {
    int codec = 0;                         // 0 NULL 1 EXTERNAL 3 HUFFMAN 4 BYTE_ARRAY_LEN 5 BYTE_ARRAY_STOP 6 BETA 7 SUBEXP 9 GAMMA
    int32_t ext = 0, offset = 0, bits = 0;
    uint8_t stop = 0;
    std::vector<int32_t> sym, len, code;   // HUFFMAN: sorted by (length, symbol), canonical codes
    std::unique_ptr<Enc> a, b;             // BYTE_ARRAY_LEN: lengths, values
};

bool parse_enc(Cur &c, Enc *e)
{
    e->codec = c.itf8();
    const int32_t plen = c.itf8();
    if (!c.ok || plen < 0) return false;
    const uint8_t *pp = c.take((size_t)plen);
    if (!c.ok) return false;
    Cur p(pp, (size_t)plen);
    switch (e->codec) {
    case 0: return true;
    case 1: e->ext = p.itf8(); return p.ok;
    case 3: {
        const int32_t n = p.itf8();
        std::vector<int32_t> s, l;
        for (int32_t k = 0; k < n; ++k) s.push_back(p.itf8());
        const int32_t m = p.itf8();
        for (int32_t k = 0; k < m; ++k) l.push_back(p.itf8());
        if (!p.ok || n != m || n <= 0) return false;
        for (int32_t x : l) if (x < 0 || x > 31) return false;
        std::vector<int> ord((size_t)n);
        for (int k = 0; k < n; ++k) ord[(size_t)k] = k;
        for (int i = 1; i < n; ++i)                       // insertion sort by (length, symbol): alphabets are tiny
            for (int j = i; j > 0; --j) {
                const int x = ord[(size_t)j - 1], y = ord[(size_t)j];
                if (l[(size_t)x] > l[(size_t)y] || (l[(size_t)x] == l[(size_t)y] && s[(size_t)x] > s[(size_t)y])) std::swap(ord[(size_t)j - 1], ord[(size_t)j]); else break;
            }
        int32_t code = 0, prev = l[(size_t)ord[0]];
        for (int k = 0; k < n; ++k) {
            const int32_t L = l[(size_t)ord[(size_t)k]];
            code <<= (L - prev); prev = L;
            e->sym.push_back(s[(size_t)ord[(size_t)k]]); e->len.push_back(L); e->code.push_back(code);
            ++code;
        }
        return true;
    }
    case 4:
        e->a.reset(new Enc); e->b.reset(new Enc);
        return parse_enc(p, e->a.get()) && parse_enc(p, e->b.get());
    case 5: e->stop = (uint8_t)p.u8(); e->ext = p.itf8(); return p.ok;
    case 6: e->offset = p.itf8(); e->bits = p.itf8(); return p.ok;
    case 7: e->offset = p.itf8(); e->bits = p.itf8(); return p.ok;      // bits = k
    case 9: e->offset = p.itf8(); return p.ok;
    default: return false;                                 // GOLOMB / GOLOMB_RICE: never written by htslib
    }
}

// ---- one slice's data: the core bit stream and the external byte streams -------------------------------------------------
struct SliceData {
    const uint8_t *core = nullptr; size_t core_n = 0, bitpos = 0;
    struct Ext { Block *blk; int64_t pos; const uint8_t *p, *e; };      // p/e valid once the block is inflated
    std::map<int32_t, Ext> ext;
    bool ok = true;
    std::string err;

    uint32_t bits(int n)
    {
        uint32_t v = 0;
        for (int k = 0; k < n; ++k) {
            const size_t byte = bitpos >> 3;
            if (byte >= core_n) { ok = false; return 0; }
            v = (v << 1) | ((core[byte] >> (7 - (bitpos & 7))) & 1u);
            ++bitpos;
        }
        return v;
    }
    // an external stream whose BYTES are needed: inflated on first use; earlier pointer-only skips are applied now
    Ext *find(int32_t id)
    {
        auto it = ext.find(id);
        if (it == ext.end()) { ok = false; return nullptr; }
        Ext &x = it->second;
        if (!x.p) {
            if (!inflate_block(x.blk, &err)) { ok = false; return nullptr; }
            x.p = x.blk->data.data(); x.e = x.p + x.blk->data.size();
            if (x.pos > (int64_t)x.blk->data.size()) { ok = false; x.p = x.e; } else x.p += x.pos;
        }
        return &x;
    }

    int32_t get_int(const Enc *e)
    {
        if (!e) { ok = false; return 0; }
        switch (e->codec) {
        case 1: { Ext *x = find(e->ext); if (!x) return 0; Cur c(x->p, (size_t)(x->e - x->p)); const int32_t v = c.itf8(); if (!c.ok) ok = false; x->p = c.p; return v; }
        case 3: {
            const size_t n = e->sym.size();
            if (n == 1 && e->len[0] == 0) return e->sym[0];
            int32_t code = 0, have = 0;
            for (size_t k = 0; k < n; ) {
                const int32_t need = e->len[k] - have;
                code = (int32_t)(((uint32_t)code << need) | bits(need)); have += need;
                for (; k < n && e->len[k] == have; ++k) if (e->code[k] == code) return e->sym[k];
                if (!ok) return 0;
            }
            ok = false; return 0;
        }
        case 6: if (e->bits < 0 || e->bits > 32) { ok = false; return 0; } return (int32_t)bits(e->bits) - e->offset;
        case 7: {
            int i = 0;
            while (ok && bits(1)) ++i;
            int32_t v;
            if (i == 0) v = (int32_t)bits(e->bits);
            else { const int b = i + e->bits - 1; if (b < 0 || b > 30) { ok = false; return 0; } v = (int32_t)((1u << b) | bits(b)); }
            return v - e->offset;
        }
        case 9: {
            int n = 0;
            while (ok && !bits(1)) ++n;
            if (n > 30) { ok = false; return 0; }
            return (int32_t)((1u << n) | bits(n)) - e->offset;
        }
        default: ok = false; return 0;
        }
    }
    int get_byte(const Enc *e)
    {
        if (!e) { ok = false; return 0; }
        if (e->codec == 1) { Ext *x = find(e->ext); if (!x) return 0; if (x->p >= x->e) { ok = false; return 0; } return *x->p++; }
        return get_int(e) & 0xff;
    }
    void skip_bytes(const Enc *e, int32_t n)                // n values of a byte series
    {
        if (n <= 0) return;
        if (!e) { ok = false; return; }
        if (e->codec == 1) {
            auto it = ext.find(e->ext);
            if (it == ext.end()) { ok = false; return; }
            Ext &x = it->second;
            if (x.p) { if ((int64_t)(x.e - x.p) < n) { ok = false; x.p = x.e; } else x.p += n; }
            else { x.pos += n; if (x.pos > (int64_t)x.blk->rsize) ok = false; }        // no bytes needed: the block stays compressed
            return;
        }
        for (int32_t k = 0; k < n && ok; ++k) (void)get_int(e);
    }
    int32_t skip_array(const Enc *e)                         // one byte array; returns its length
    {
        if (!e) { ok = false; return 0; }
        if (e->codec == 4) { const int32_t n = get_int(e->a.get()); if (n < 0) { ok = false; return 0; } skip_bytes(e->b.get(), n); return n; }
        if (e->codec == 5) {
            Ext *x = find(e->ext); if (!x) return 0;
            const uint8_t *q = (const uint8_t *)memchr(x->p, e->stop, (size_t)(x->e - x->p));
            if (!q) { ok = false; return 0; }
            const int32_t n = (int32_t)(q - x->p);
            x->p = q + 1;
            return n;
        }
        ok = false; return 0;
    }
};

inline uint16_t key2(char a, char b) { return (uint16_t)(((uint8_t)a << 8) | (uint8_t)b); }

struct CompHeader {
    bool rn = true, ap_delta = true;
    std::vector<std::vector<int32_t>> td;                  // tag dictionary: lines of tag keys
    std::map<uint16_t, Enc> ds;
    std::map<int32_t, Enc> tags;
    const Enc *get(char a, char b) const { auto it = ds.find(key2(a, b)); return it == ds.end() ? nullptr : &it->second; }
};

bool parse_comp_header(const std::vector<uint8_t> &d, CompHeader *h)
{
    Cur c(d.data(), d.size());
    {   // preservation map
        const int32_t size = c.itf8();
        if (!c.ok || size < 0) return false;
        const uint8_t *pp = c.take((size_t)size);
        if (!c.ok) return false;
        Cur p(pp, (size_t)size);
        const int32_t n = p.itf8();
        for (int32_t k = 0; k < n && p.ok; ++k) {
            const int a = p.u8(), b = p.u8();
            if (a == 'R' && b == 'N') h->rn = p.u8() != 0;
            else if (a == 'A' && b == 'P') h->ap_delta = p.u8() != 0;
            else if (a == 'R' && b == 'R') p.u8();
            else if (a == 'S' && b == 'M') p.take(5);
            else if (a == 'T' && b == 'D') {
                const int32_t len = p.itf8();
                const uint8_t *t = p.take(len < 0 ? 0 : (size_t)len);
                if (!p.ok) return false;
                std::vector<int32_t> line;
                for (int32_t i = 0; i < len; ) {
                    if (t[i] == 0) { h->td.push_back(line); line.clear(); ++i; continue; }
                    if (i + 3 > len) return false;
                    line.push_back((t[i] << 16) | (t[i + 1] << 8) | t[i + 2]);
                    i += 3;
                }
                if (!line.empty()) h->td.push_back(line);
            } else return false;
        }
        if (!p.ok) return false;
    }
    {   // data series encodings
        const int32_t size = c.itf8();
        if (!c.ok || size < 0) return false;
        const uint8_t *pp = c.take((size_t)size);
        if (!c.ok) return false;
        Cur p(pp, (size_t)size);
        const int32_t n = p.itf8();
        for (int32_t k = 0; k < n; ++k) {
            const int a = p.u8(), b = p.u8();
            Enc e;
            if (!p.ok || !parse_enc(p, &e)) return false;
            h->ds[key2((char)a, (char)b)] = std::move(e);
        }
    }
    {   // tag encodings
        const int32_t size = c.itf8();
        if (!c.ok || size < 0) return false;
        const uint8_t *pp = c.take((size_t)size);
        if (!c.ok) return false;
        Cur p(pp, (size_t)size);
        const int32_t n = p.itf8();
        for (int32_t k = 0; k < n; ++k) {
            const int32_t key = p.itf8();
            Enc e;
            if (!p.ok || !parse_enc(p, &e)) return false;
            h->tags[key] = std::move(e);
        }
    }
    return true;
}

// Which data series have to be walked at all.  Every external block is a byte stream of its own, so a series whose
// encoding touches neither the core bit stream nor a block that a NEEDED series reads can be left alone: its block is
// never decompressed (qualities, names, bases, tags, mate fields — most of the file).  Needed: what gives flag, contig,
// position, mapping quality and the feature list (BF CF RI RL AP FN FC FP DL RS HC PD MQ + the lengths of SC IN BB).
struct EncUse { bool core = false; std::vector<int32_t> ids; };

void enc_use(const Enc *e, EncUse *u)
{
    if (!e) return;
    switch (e->codec) {
    case 1: case 5: u->ids.push_back(e->ext); break;
    case 3: if (!(e->sym.size() == 1 && e->len[0] == 0)) u->core = true; break;
    case 4: enc_use(e->a.get(), u); enc_use(e->b.get(), u); break;
    case 6: case 7: case 9: u->core = true; break;
    default: break;
    }
}

struct Walk {
    std::map<uint16_t, bool> ds;              // optional series -> must be consumed
    std::map<int32_t, bool> tag;
    bool any_tag = false;
    bool operator()(char a, char b) const { auto it = ds.find(key2(a, b)); return it == ds.end() ? true : it->second; }
};

Walk plan_walk(const CompHeader &H)
{
    static const char *NEEDED[] = {"BF", "CF", "RI", "RL", "AP", "FN", "FC", "FP", "DL", "RS", "HC", "PD", "MQ", "SC", "IN", "BB"};
    static const char *OPTIONAL[] = {"RG", "RN", "MF", "NS", "NP", "TS", "NF", "TL", "QQ", "BS", "BA", "QS"};
    std::vector<int32_t> ids;
    auto touches = [&](const EncUse &u) { for (int32_t a : u.ids) for (int32_t b : ids) if (a == b) return true; return false; };
    for (const char *k : NEEDED) { EncUse u; enc_use(H.get(k[0], k[1]), &u); ids.insert(ids.end(), u.ids.begin(), u.ids.end()); }
    Walk w;
    for (const char *k : OPTIONAL) w.ds[key2(k[0], k[1])] = false;
    for (auto &kv : H.tags) w.tag[kv.first] = false;
    for (bool changed = true; changed; ) {
        changed = false;
        for (const char *k : OPTIONAL) {
            bool &keep = w.ds[key2(k[0], k[1])];
            if (keep) continue;
            EncUse u; enc_use(H.get(k[0], k[1]), &u);
            if (u.core || touches(u)) { keep = true; changed = true; ids.insert(ids.end(), u.ids.begin(), u.ids.end()); }
        }
        for (auto &kv : H.tags) {
            bool &keep = w.tag[kv.first];
            if (keep) continue;
            EncUse u; enc_use(&kv.second, &u);
            if (u.core || touches(u)) { keep = true; changed = true; ids.insert(ids.end(), u.ids.begin(), u.ids.end()); }
        }
    }
    for (auto &kv : w.tag) w.any_tag = w.any_tag || kv.second;
    if (w.any_tag) w.ds[key2('T', 'L')] = true;             // the tag line says which tags a record carries
    return w;
}

int fgetc_itf8(FILE *f, int32_t *v)
{
    uint8_t b[5];
    const int c0 = fgetc(f);
    if (c0 == EOF) return -1;
    b[0] = (uint8_t)c0;
    const int extra = b[0] < 0x80 ? 0 : b[0] < 0xc0 ? 1 : b[0] < 0xe0 ? 2 : b[0] < 0xf0 ? 3 : 4;
    if (extra && fread(b + 1, 1, (size_t)extra, f) != (size_t)extra) return -1;
    Cur c(b, (size_t)extra + 1);
    *v = c.itf8();
    return 0;
}

int fgetc_ltf8(FILE *f, int64_t *v)
{
    uint8_t b[9];
    const int c0 = fgetc(f);
    if (c0 == EOF) return -1;
    b[0] = (uint8_t)c0;
    int extra = 0;
    while (extra < 8 && (b[0] & (0x80 >> extra))) ++extra;
    if (extra && fread(b + 1, 1, (size_t)extra, f) != (size_t)extra) return -1;
    Cur c(b, (size_t)extra + 1);
    *v = c.ltf8();
    return 0;
}

struct ContainerHeader { int32_t length = 0, ref = 0, start = 0, span = 0, n_rec = 0, n_blocks = 0; std::vector<int32_t> landmarks; };

// 1 ok, 0 clean end of file, -1 truncated
int read_container_header(FILE *f, ContainerHeader *h, bool v2)
{
    uint8_t l[4];
    const size_t got = fread(l, 1, 4, f);
    if (got == 0) return 0;
    if (got != 4) return -1;
    h->length = (int32_t)((uint32_t)l[0] | ((uint32_t)l[1] << 8) | ((uint32_t)l[2] << 16) | ((uint32_t)l[3] << 24));
    int64_t t;
    int32_t n = 0;
    int32_t t32;
    if (fgetc_itf8(f, &h->ref) || fgetc_itf8(f, &h->start) || fgetc_itf8(f, &h->span) || fgetc_itf8(f, &h->n_rec) ||
        (v2 ? fgetc_itf8(f, &t32) : fgetc_ltf8(f, &t)) || fgetc_ltf8(f, &t) || fgetc_itf8(f, &h->n_blocks) || fgetc_itf8(f, &n)) return -1;
    h->landmarks.clear();
    for (int32_t k = 0; k < n; ++k) { int32_t v; if (fgetc_itf8(f, &v)) return -1; h->landmarks.push_back(v); }
    if (!v2 && fread(l, 1, 4, f) != 4) return -1;          // CRC32
    return h->length < 0 ? -1 : 1;
}

} // namespace

bool CramReader::is_cram(const std::string &path)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char m[4] = {0, 0, 0, 0};
    const size_t n = fread(m, 1, 4, f);
    fclose(f);
    return n == 4 && memcmp(m, "CRAM", 4) == 0;
}

void CramReader::close() { ahead_.clear(); if (f_) { fclose(f_); f_ = nullptr; } }

bool CramReader::open(const std::string &path, AlnHeader *hdr, std::string *err)
{
    close();
    err_.clear(); eof_ = false; ahead_.clear(); cur_batch_ = Batch(); cur_ = 0;
    auto bad = [&](const std::string &m) { err_ = m; if (err) *err = m; close(); return false; };
    f_ = fopen(path.c_str(), "rb");
    if (!f_) return bad("cannot open " + path);
    uint8_t def[26];
    if (fread(def, 1, 26, f_) != 26 || memcmp(def, "CRAM", 4) != 0) return bad("not a CRAM file: " + path);
    if (!((def[4] == 3 && def[5] <= 1) || (def[4] == 2 && def[5] == 1))) return bad("CRAM version " + std::to_string(def[4]) + "." + std::to_string(def[5]) + " is not supported (2.1, 3.0 and 3.1 only): " + path);
    // §6: the first container holds the SAM header
    ContainerHeader ch;
    v2_ = def[4] == 2;
    fsize_ = file_size(path);
    if (read_container_header(f_, &ch, v2_) != 1) return bad("truncated CRAM header container: " + path);
    // The header block is read from the file itself, not from `length` bytes of container body: writers of CRAM 2.1 files
    // state a container length a few bytes short of the (padded) block, and htslib reads it this way too.
    Block b;
    std::string e2;
    std::vector<uint8_t> raw;
    {
        uint8_t mt[2];
        int32_t id = 0, csize = 0, rsize = 0;
        const long at = ftell(f_);
        if (fread(mt, 1, 2, f_) != 2 || fgetc_itf8(f_, &id) || fgetc_itf8(f_, &csize) || fgetc_itf8(f_, &rsize) || csize < 0 || rsize < 0)
            return bad("truncated CRAM header block: " + path);
        if ((uint64_t)csize > file_size(path) || (uint64_t)rsize > ((uint64_t)1 << 31)) return bad("damaged CRAM header block: " + path);
        raw.resize((size_t)csize);
        if (csize && fread(raw.data(), 1, raw.size(), f_) != raw.size()) return bad("truncated CRAM header block: " + path);
        if (!v2_) { uint8_t crc[4]; if (fread(crc, 1, 4, f_) != 4) return bad("truncated CRAM header block: " + path); }
        b.method = mt[0]; b.type = mt[1]; b.id = id; b.comp = raw.data(); b.csize = csize; b.rsize = rsize;
        if (!inflate_block(&b, &e2)) return bad(e2);
        const long used = ftell(f_) - at;
        if ((long)ch.length > used && fseek(f_, (long)ch.length - used, SEEK_CUR) != 0) return bad("truncated CRAM header container: " + path);
    }
    if (b.type != 0 || b.data.size() < 4) return bad("CRAM header block missing: " + path);
    const uint32_t l_text = (uint32_t)b.data[0] | ((uint32_t)b.data[1] << 8) | ((uint32_t)b.data[2] << 16) | ((uint32_t)b.data[3] << 24);
    if ((size_t)l_text + 4 > b.data.size()) return bad("CRAM header text truncated: " + path);
    hdr->text.assign((const char *)b.data.data() + 4, l_text);
    const size_t z = hdr->text.find('\0');
    if (z != std::string::npos) hdr->text.resize(z);
    hdr->names.clear(); hdr->lens.clear();
    size_t p = 0;
    while (p < hdr->text.size()) {
        size_t nl = hdr->text.find('\n', p);
        if (nl == std::string::npos) nl = hdr->text.size();
        if (nl - p >= 3 && hdr->text.compare(p, 3, "@SQ") == 0) {
            std::string name; long long len = 0;
            size_t q = p;
            while (q < nl) {
                size_t t = hdr->text.find('\t', q);
                if (t == std::string::npos || t > nl) t = nl;
                if (t - q > 3 && hdr->text.compare(q, 3, "SN:") == 0) name = hdr->text.substr(q + 3, t - q - 3);
                else if (t - q > 3 && hdr->text.compare(q, 3, "LN:") == 0) len = atoll(hdr->text.substr(q + 3, t - q - 3).c_str());
                q = t + 1;
            }
            hdr->names.push_back(name); hdr->lens.push_back((uint32_t)len);
        }
        p = nl + 1;
    }
    return true;
}

namespace {

// one container body (everything after the container header) -> its records
bool decode_container_unchecked(const std::vector<uint8_t> &body, CramReader::Batch *out, bool v2);

// Sizes inside a container (block sizes, table sizes, record counts) are file-declared; whatever slips past the
// explicit bounds ends as std::bad_alloc / std::length_error.  On a helper thread that would reach future::get()
// uncaught and end the process, so it is turned into the reader's ordinary "corrupt CRAM" error here.
bool decode_container(const std::vector<uint8_t> &body, CramReader::Batch *out, bool v2)
{
    try { return decode_container_unchecked(body, out, v2); }
    catch (const std::exception &e) {
        out->recs.clear(); out->cigs.clear();
        if (out->err.empty()) out->err = std::string("corrupt CRAM container (") + e.what() + ")";
        return false;
    }
}

bool decode_container_unchecked(const std::vector<uint8_t> &body, CramReader::Batch *out, bool v2)
{
    typedef CramReader::Rec Rec;
    auto bad = [&](const std::string &m) { if (out->err.empty()) out->err = m; return false; };
    Cur c(body.data(), body.size());
    Block cb;
    std::string e2;
    if (!read_block(c, &cb, &e2, false, v2)) return bad(e2);
    if (cb.type != 1) return bad("CRAM compression header missing");
    CompHeader H;
    if (!parse_comp_header(cb.data, &H)) return bad("CRAM compression header uses an encoding this reader does not know");
    if (getenv("PANDEPTH_CRAM_DEBUG")) {
        fprintf(stderr, "[cram] container ref %d n_rec %d rn %d ap_delta %d td %zu\n", 0, 0, (int)H.rn, (int)H.ap_delta, H.td.size());
        for (auto &kv : H.ds) fprintf(stderr, "[cram]   %c%c codec %d ext %d%s\n", kv.first >> 8, kv.first & 0xff, kv.second.codec, kv.second.ext,
                                      kv.second.codec == 3 ? (kv.second.sym.size() == 1 ? " (const)" : " (huffman)") : "");
        for (auto &kv : H.tags) fprintf(stderr, "[cram]   tag %c%c%c codec %d ext %d\n", kv.first >> 16, (kv.first >> 8) & 0xff, kv.first & 0xff, kv.second.codec, kv.second.ext);
    }
    const Walk W = plan_walk(H);
    const bool wRG = W('R', 'G'), wRN = W('R', 'N'), wMF = W('M', 'F'), wNS = W('N', 'S'), wNP = W('N', 'P'), wTS = W('T', 'S'),
               wNF = W('N', 'F'), wTL = W('T', 'L'), wQQ = W('Q', 'Q'), wBS = W('B', 'S'), wBA = W('B', 'A'), wQS = W('Q', 'S');
    const Enc *BF = H.get('B', 'F'), *CF = H.get('C', 'F'), *RI = H.get('R', 'I'), *RL = H.get('R', 'L'), *AP = H.get('A', 'P'),
              *RG = H.get('R', 'G'), *RN = H.get('R', 'N'), *MF = H.get('M', 'F'), *NS = H.get('N', 'S'), *NP = H.get('N', 'P'),
              *TS = H.get('T', 'S'), *NF = H.get('N', 'F'), *TL = H.get('T', 'L'), *FN = H.get('F', 'N'), *FC = H.get('F', 'C'),
              *FP = H.get('F', 'P'), *DL = H.get('D', 'L'), *BB = H.get('B', 'B'), *QQ = H.get('Q', 'Q'), *BS = H.get('B', 'S'),
              *IN = H.get('I', 'N'), *RS = H.get('R', 'S'), *PD = H.get('P', 'D'), *HC = H.get('H', 'C'), *SC = H.get('S', 'C'),
              *MQ = H.get('M', 'Q'), *BA = H.get('B', 'A'), *QS = H.get('Q', 'S');
    while (c.left() > 0) {
        // §8.5 slice header, then its blocks
        Block sh;
        if (!read_block(c, &sh, &e2, false, v2)) return bad(e2);
        if (sh.type != 2) return bad("CRAM slice header expected");
        Cur s(sh.data.data(), sh.data.size());
        const int32_t ref = s.itf8(), start = s.itf8();
        s.itf8();                                       // span
        const int32_t n_rec = s.itf8();
        if (v2) s.itf8(); else s.ltf8();               // record counter
        const int32_t n_blocks = s.itf8();
        if (!s.ok || n_rec < 0 || n_blocks < 0) return bad("corrupt CRAM slice header");
        // counts come from the file: a block takes at least 6 bytes of the container, and a slice of more than 2^24
        // records is not something any writer produces (htslib: 10 000) — nothing below is sized from a larger number
        if ((size_t)n_blocks > c.left() / 6 + 1 || n_rec > (1 << 24)) return bad("corrupt CRAM slice header");
        std::vector<Block> blocks((size_t)n_blocks);
        SliceData sd;
        for (int32_t k = 0; k < n_blocks; ++k) {
            if (!read_block(c, &blocks[(size_t)k], &e2, true, v2)) return bad(e2);
            Block &b = blocks[(size_t)k];
            if (b.type == 5) { if (!inflate_block(&b, &e2)) return bad(e2); sd.core = b.data.data(); sd.core_n = b.data.size(); }
            else if (b.type == 4) sd.ext[b.id] = SliceData::Ext{&b, 0, nullptr, nullptr};
        }
        int32_t prev_ap = start;
        for (int32_t r = 0; r < n_rec; ++r) {
            // §10: the record
            const int32_t bf = sd.get_int(BF), cf = sd.get_int(CF);
            int32_t ri = ref;
            if (ref == -2) ri = sd.get_int(RI);
            const int32_t rl = sd.get_int(RL);
            int32_t ap = sd.get_int(AP);
            if (H.ap_delta) { prev_ap += ap; ap = prev_ap; }
            if (wRG) sd.get_int(RG);
            if (H.rn && wRN) sd.skip_array(RN);
            if (cf & 2) {
                if (wMF) sd.get_int(MF);
                if (!H.rn && wRN) sd.skip_array(RN);
                if (wNS) sd.get_int(NS);
                if (wNP) sd.get_int(NP);
                if (wTS) sd.get_int(TS);
            } else if ((cf & 4) && wNF) sd.get_int(NF);
            const int32_t tl = wTL ? sd.get_int(TL) : 0;
            if (!sd.ok) return bad("corrupt CRAM record");
            if (!W.any_tag) {}
            else if (tl >= 0 && (size_t)tl < H.td.size())
                for (int32_t key : H.td[(size_t)tl]) {
                    auto it = H.tags.find(key);
                    if (it == H.tags.end()) return bad("CRAM tag without an encoding");
                    if (W.tag.at(key)) sd.skip_array(&it->second);
                }
            else if (!H.td.empty() || tl != 0) return bad("corrupt CRAM tag line");
            Rec rec{ri, ap - 1, (uint16_t)bf, 0, (uint32_t)out->cigs.size(), 0};
            auto op = [&](uint32_t code, int32_t len) {
                if (len <= 0) return;
                if (out->cigs.size() > rec.cig_off && (out->cigs.back() & 0xf) == code) out->cigs.back() += (uint32_t)len << 4;
                else out->cigs.push_back(((uint32_t)len << 4) | code);
            };
            if (!(bf & 4)) {
                // §10.6: read features -> the CIGAR shape (M 0, I 1, D 2, N 3, S 4, H 5, P 6)
                const int32_t fn = sd.get_int(FN);
                int32_t prev = 0, seq_pos = 1;
                for (int32_t k = 0; k < fn && sd.ok; ++k) {
                    const int fc = sd.get_byte(FC);
                    const int32_t pos = prev + sd.get_int(FP);
                    prev = pos;
                    if (pos > seq_pos) { op(0, pos - seq_pos); seq_pos = pos; }
                    switch (fc) {
                    case 'S': { const int32_t n = sd.skip_array(SC); op(4, n); seq_pos += n; break; }
                    case 'X': if (wBS) sd.get_byte(BS); op(0, 1); ++seq_pos; break;
                    case 'D': op(2, sd.get_int(DL)); break;
                    case 'I': { const int32_t n = sd.skip_array(IN); op(1, n); seq_pos += n; break; }
                    case 'i': if (wBA) sd.get_byte(BA); op(1, 1); ++seq_pos; break;
                    case 'b': { const int32_t n = sd.skip_array(BB); op(0, n); seq_pos += n; break; }
                    case 'q': if (wQQ) sd.skip_array(QQ); break;
                    case 'B': if (wBA) sd.get_byte(BA); if (wQS) sd.get_byte(QS); op(0, 1); ++seq_pos; break;
                    case 'Q': if (wQS) sd.get_byte(QS); break;
                    case 'H': op(5, sd.get_int(HC)); break;
                    case 'P': op(6, sd.get_int(PD)); break;
                    case 'N': op(3, sd.get_int(RS)); break;
                    default: return bad("unknown CRAM read feature");
                    }
                }
                if (seq_pos <= rl) op(0, rl - seq_pos + 1);
                rec.mapq = (uint8_t)sd.get_int(MQ);
                if ((cf & 1) && wQS) sd.skip_bytes(QS, rl);
            } else {
                if (wBA) sd.skip_bytes(BA, rl);
                if ((cf & 1) && wQS) sd.skip_bytes(QS, rl);
            }
            if (!sd.ok) return bad(!sd.err.empty() ? sd.err : "corrupt CRAM record (record " + std::to_string(r) + " of its slice)");
            rec.n_cig = (uint32_t)out->cigs.size() - rec.cig_off;
            out->recs.push_back(rec);
        }
    }
    return true;
}

} // namespace

bool CramReader::read_body(std::vector<uint8_t> *body)
{
    for (;;) {
        ContainerHeader ch;
        const int rc = read_container_header(f_, &ch, v2_);
        if (rc == 0) { eof_ = true; return false; }
        if (rc < 0) { eof_ = true; return fail("truncated CRAM container header"); }
        if (keep_ && ch.n_rec > 0 && ch.ref != -2 &&
            (ch.ref < 0 || !keep_(ch.ref, (int64_t)ch.start - 1, (int64_t)ch.start - 1 + (ch.span > 0 ? ch.span : 1)))) {
            if (fseeko(f_, (off_t)ch.length, SEEK_CUR) != 0) { eof_ = true; return fail("seek failed in CRAM file"); }
            ++n_skipped_;
            continue;
        }
        if ((uint64_t)ch.length > fsize_) { eof_ = true; return fail("truncated CRAM container"); }      // longer than the file: do not size a buffer from it
        body->resize((size_t)ch.length);
        if (ch.length && fread(body->data(), 1, body->size(), f_) != body->size()) { eof_ = true; return fail("truncated CRAM container"); }
        if (ch.n_rec == 0) continue;                       // the end-of-file container (or an empty one)
        ++n_read_;
        return true;
    }
}

int CramReader::next(AlnRec *r)
{
    if (!f_) return -1;
    while (cur_ >= cur_batch_.recs.size()) {
        // keep up to threads_ containers in flight (one decoder thread each), hand them back in file order
        while (!eof_ && ahead_.size() < (size_t)threads_) {
            std::vector<uint8_t> body;
            if (!read_body(&body)) break;
            if (threads_ == 1) {
                std::promise<Batch> p;
                Batch b;
                decode_container(body, &b, v2_);
                p.set_value(std::move(b));
                ahead_.push_back(p.get_future());
            } else
                ahead_.push_back(std::async(std::launch::async, [](std::vector<uint8_t> bytes, bool v2) { Batch b; decode_container(bytes, &b, v2); return b; }, std::move(body), v2_));
        }
        if (ahead_.empty()) return err_.empty() ? 0 : -1;
        cur_batch_ = ahead_.front().get();
        ahead_.pop_front();
        cur_ = 0;
        if (!cur_batch_.err.empty()) { fail(cur_batch_.err); return -1; }
    }
    const Rec &x = cur_batch_.recs[cur_++];
    r->tid = x.tid; r->pos = x.pos; r->flag = x.flag; r->mapq = x.mapq;
    r->n_cigar = x.n_cig; r->cigar = cur_batch_.cigs.data() + x.cig_off;
    return 1;
}

} // namespace pdh
