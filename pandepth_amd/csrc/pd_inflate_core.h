// pd_inflate_core.h — raw DEFLATE (RFC 1951) decoder for one BGZF block, written once and compiled
// both for gfx950 (one lane decodes one block; pd_bgzf.hip) and for the host (g++, tests/: the same
// source is checked against zlib on every block of the test BAMs before it ever runs on a GPU).
//
// Shape: bit reader with a 64-bit buffer refilled 32 bits at a time; per-block Huffman tables in a
// caller-provided scratch area (LDS or global): a direct table for codes of <= 9 (literal/length)
// and <= 6 (distance) bits — the common case, one lookup per symbol — and canonical first-code /
// count arrays (RFC 1951 §3.2.2) for the rare longer codes.  All loops are bounded by the input
// and output sizes, so corrupt data ends with an error code, never with a hang.
#ifndef PD_INFLATE_CORE_H_
#define PD_INFLATE_CORE_H_
#include <stdint.h>

#if defined(__HIPCC__)
#define PDI_FN __host__ __device__ __forceinline__
#define PDI_FN_NOINLINE __host__ __device__
#else
#define PDI_FN inline
#define PDI_FN_NOINLINE inline
#endif

namespace pdi {

enum { LL_FAST_BITS = 8, D_FAST_BITS = 5, MAX_BITS = 15 };

template <int NSYM>
struct HuffT {                // canonical code, symbols ordered by (length, value)
    uint16_t count[MAX_BITS + 1];
    uint16_t symbol[NSYM];
};
typedef HuffT<288> HuffLL;
typedef HuffT<32> HuffD;      // 30 distance codes / the 19 code-length codes

struct Fast {                 // hot: one lookup per symbol.  576 bytes per lane -> a wave's 64 copies fit LDS 4x per CU
    uint16_t ll[1 << LL_FAST_BITS];        // (symbol << 4) | length, 0 = code longer than LL_FAST_BITS
    uint16_t d[1 << D_FAST_BITS];
};
struct Slow {                 // cold: canonical arrays for long codes + header scratch (global memory on the GPU)
    HuffLL ll;
    HuffD d;
    uint8_t cl[320];                       // code lengths of the block being set up
    uint8_t small[32];                     // the code-length code's lengths
};
struct Tables { Fast fast; Slow slow; };   // host convenience

struct Bits {
    const uint8_t *in;
    uint32_t pos, end;        // next unread byte, input length
    uint64_t buf;
    uint32_t cnt;
};

PDI_FN void bits_refill(Bits &b)
{
    // keep at least 32 valid bits while input lasts; past the end zeros are shifted in and the
    // caller's position check reports the overrun
    while (b.cnt <= 32) {
        uint32_t w;
        if (b.pos + 4 <= b.end) {
            __builtin_memcpy(&w, b.in + b.pos, 4);           // one (possibly unaligned) 32-bit load, little-endian hosts/GPUs
            b.buf |= (uint64_t)w << b.cnt; b.cnt += 32; b.pos += 4;
        } else if (b.pos < b.end) {
            b.buf |= (uint64_t)b.in[b.pos] << b.cnt; b.cnt += 8; b.pos += 1;
        } else {
            b.cnt += 32; b.pos += 4;                       // virtual zero bytes; detected by overrun()
        }
    }
}
PDI_FN uint32_t bits_peek(const Bits &b, int n) { return (uint32_t)(b.buf & ((1ull << n) - 1)); }
PDI_FN void bits_drop(Bits &b, int n) { b.buf >>= n; b.cnt -= (uint32_t)n; }
PDI_FN uint32_t bits_get(Bits &b, int n) { bits_refill(b); const uint32_t v = bits_peek(b, n); bits_drop(b, n); return v; }
// true when more bits were consumed than the input holds
PDI_FN bool bits_overrun(const Bits &b) { return (uint64_t)b.pos * 8 - b.cnt > (uint64_t)b.end * 8; }

PDI_FN uint32_t rev_bits(uint32_t code, int len)
{
    uint32_t r = 0;
    for (int i = 0; i < len; ++i) { r = (r << 1) | (code & 1); code >>= 1; }
    return r;
}

// builds the canonical arrays and the fast table from `n` code lengths; returns 0, or -1 for an
// over-subscribed / incomplete code (a single distance code is allowed, as zlib does)
template <class H>
PDI_FN_NOINLINE int build(H &h, uint16_t *fast, int fast_bits, const uint8_t *len, int n)
{
    for (int i = 0; i <= MAX_BITS; ++i) h.count[i] = 0;
    for (int i = 0; i < n; ++i) h.count[len[i]]++;
    for (int i = 0; i < (1 << fast_bits); ++i) fast[i] = 0;
    if (h.count[0] == n) return 0;                       // no codes at all (legal for distances)
    int left = 1;
    for (int l = 1; l <= MAX_BITS; ++l) {
        left <<= 1; left -= h.count[l];
        if (left < 0) return -1;
    }
    uint16_t offs[MAX_BITS + 2];
    offs[1] = 0;
    for (int l = 1; l < MAX_BITS; ++l) offs[l + 1] = (uint16_t)(offs[l] + h.count[l]);
    for (int i = 0; i < n; ++i) if (len[i]) h.symbol[offs[len[i]]++] = (uint16_t)i;
    // fast table: walk the codes in canonical order
    uint32_t code = 0; int idx = 0;
    for (int l = 1; l <= fast_bits; ++l) {
        for (int k = 0; k < h.count[l]; ++k, ++idx, ++code) {
            const uint32_t r = rev_bits(code, l);
            const uint16_t e = (uint16_t)((h.symbol[idx] << 4) | l);
            for (uint32_t j = r; j < (1u << fast_bits); j += (1u << l)) fast[j] = e;
        }
        code <<= 1;
    }
    return 0;        // an incomplete code is tolerated here: using one of its missing codes fails in decode()
}

// one symbol; -1 on an invalid code
template <class H>
PDI_FN int decode(Bits &b, const H &h, const uint16_t *fast, int fast_bits)
{
    bits_refill(b);
    const uint16_t e = fast[bits_peek(b, fast_bits)];
    if (e) { bits_drop(b, e & 15); return e >> 4; }
    // long code: canonical walk, one bit at a time (RFC 1951 §3.2.2)
    int code = 0, first = 0, index = 0;
    uint64_t v = b.buf;
    for (int l = 1; l <= MAX_BITS; ++l) {
        code |= (int)(v & 1); v >>= 1;
        const int c = h.count[l];
        if (code - c < first) { bits_drop(b, l); return h.symbol[index + (code - first)]; }
        index += c; first += c; first <<= 1; code <<= 1;
    }
    return -1;
}

// Inflates one raw DEFLATE stream of exactly out_len bytes.  Returns 0 on success, a negative
// code otherwise (-1 bad block type, -2 bad stored block, -3 bad code lengths, -4 invalid code,
// -5 output overrun / wrong size, -6 bad distance, -7 input overrun).
PDI_FN_NOINLINE int inflate_block(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_len, Fast &tf, Slow &t)
{
    static const uint16_t LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115,
                                       131, 163, 195, 227, 258};
    static const uint8_t LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537,
                                       2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    static const uint8_t CLORD[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    Bits b; b.in = in; b.pos = 0; b.end = in_len; b.buf = 0; b.cnt = 0;
    uint32_t o = 0;
    for (int guard = 0; guard < 70000; ++guard) {                       // a block emits >= 0 bytes; bound the block count
        const uint32_t last = bits_get(b, 1);
        const uint32_t type = bits_get(b, 2);
        if (type == 0) {
            bits_drop(b, (int)(b.cnt & 7));                             // to the byte boundary
            bits_refill(b);
            const uint32_t len = bits_get(b, 16), nlen = bits_get(b, 16);
            if ((len ^ 0xffff) != nlen) return -2;
            if (o + len > out_len) return -5;
            for (uint32_t i = 0; i < len; ++i) out[o++] = (uint8_t)bits_get(b, 8);
        } else if (type == 1 || type == 2) {
            if (type == 1) {
                for (int i = 0; i < 144; ++i) t.cl[i] = 8;
                for (int i = 144; i < 256; ++i) t.cl[i] = 9;
                for (int i = 256; i < 280; ++i) t.cl[i] = 7;
                for (int i = 280; i < 288; ++i) t.cl[i] = 8;
                if (build(t.ll, tf.ll, LL_FAST_BITS, t.cl, 288) < 0) return -3;
                for (int i = 0; i < 30; ++i) t.cl[i] = 5;
                if (build(t.d, tf.d, D_FAST_BITS, t.cl, 30) < 0) return -3;
            } else {
                const int nlen = (int)bits_get(b, 5) + 257, ndist = (int)bits_get(b, 5) + 1, ncode = (int)bits_get(b, 4) + 4;
                if (nlen > 286 || ndist > 30) return -3;
                for (int i = 0; i < 19; ++i) t.small[i] = 0;
                for (int i = 0; i < ncode; ++i) t.small[CLORD[i]] = (uint8_t)bits_get(b, 3);
                // the code-length code (<= 7 bits) borrows the distance tables until they are built
                if (build(t.d, tf.d, D_FAST_BITS, t.small, 19) < 0) return -3;
                int idx = 0;
                uint8_t *cl = t.cl;
                while (idx < nlen + ndist) {
                    const int sym = decode(b, t.d, tf.d, D_FAST_BITS);
                    if (sym < 0) return -4;
                    if (sym < 16) cl[idx++] = (uint8_t)sym;
                    else {
                        int rep; uint8_t val = 0;
                        if (sym == 16) { if (idx == 0) return -3; val = cl[idx - 1]; rep = 3 + (int)bits_get(b, 2); }
                        else if (sym == 17) rep = 3 + (int)bits_get(b, 3);
                        else rep = 11 + (int)bits_get(b, 7);
                        if (idx + rep > nlen + ndist) return -3;
                        while (rep--) cl[idx++] = val;
                    }
                }
                if (cl[256] == 0) return -3;                             // no end-of-block code
                if (build(t.ll, tf.ll, LL_FAST_BITS, cl, nlen) < 0) return -3;
                if (build(t.d, tf.d, D_FAST_BITS, cl + nlen, ndist) < 0) return -3;
            }
            for (;;) {
                const int sym = decode(b, t.ll, tf.ll, LL_FAST_BITS);
                if (sym < 0) return -4;
                if (sym < 256) { if (o >= out_len) return -5; out[o++] = (uint8_t)sym; continue; }
                if (sym == 256) break;
                if (sym > 285) return -4;
                const int ls = sym - 257;
                const uint32_t len = LBASE[ls] + bits_get(b, LEXT[ls]);
                const int ds = decode(b, t.d, tf.d, D_FAST_BITS);
                if (ds < 0 || ds > 29) return -4;
                const uint32_t dist = DBASE[ds] + bits_get(b, DEXT[ds]);
                if (dist > o) return -6;
                if (o + len > out_len) return -5;
                const uint8_t *src = out + o - dist;
                uint32_t i = 0;
                if (dist >= 4)                                          // no overlap inside a 4-byte step
                    for (; i + 4 <= len; i += 4) { uint32_t w; __builtin_memcpy(&w, src + i, 4); __builtin_memcpy(out + o + i, &w, 4); }
                for (; i < len; ++i) out[o + i] = src[i];               // forward byte copy: overlap repeats the pattern
                o += len;
                if (bits_overrun(b)) return -7;
            }
        } else return -1;
        if (bits_overrun(b)) return -7;
        if (last) break;
    }
    return o == out_len ? 0 : -5;
}

} // namespace pdi
#endif
