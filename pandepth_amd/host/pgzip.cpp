// pgzip.cpp — see pgzip.h.  Stage 2 below re-states what zlib's deflate does AFTER the LZ77 parse
// (trees.c of zlib 1.2.11 / 1.2.12, as published): same block boundaries, same Huffman codes, same
// bits.  It is test-pinned against the system zlib on every golden table and on generated text.
#include "pgzip.h"

#include <string.h>
#include <zlib.h>

#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <chrono>
#include <atomic>
#include <functional>
#include <thread>
#include <deque>
#include <condition_variable>
#include <mutex>

#include "../csrc/pd_inflate_core.h"
#include "../csrc/pd_lz77.h"

namespace pgz {
namespace {

// ---------------------------------------------------------------------------------------------
// symbols: literal = byte value; match = (len << 16) | dist, len 3..258, dist 1..32768
// ---------------------------------------------------------------------------------------------
typedef uint32_t Sym;
inline bool is_match(Sym s) { return s >= 65536u; }
inline uint32_t sym_len(Sym s) { return is_match(s) ? (s >> 16) : 1u; }

const int LENGTH_CODES = 29, LITERALS = 256, L_CODES = 286, D_CODES = 30, BL_CODES = 19, HEAP_SIZE = 2 * L_CODES + 1;
const int MAX_BITS = 15, MAX_BL_BITS = 7, END_BLOCK = 256, REP_3_6 = 16, REPZ_3_10 = 17, REPZ_11_138 = 18;
const uint32_t LIT_BUFSIZE = 1u << (8 + 6);           // memLevel 8: a block is flushed at LIT_BUFSIZE - 1 symbols

const int extra_lbits[LENGTH_CODES] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const int extra_dbits[D_CODES] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
const int extra_blbits[BL_CODES] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 3, 7};
const uint8_t bl_order[BL_CODES] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct Ct { uint16_t freq, code, dad, len; };

struct Static {
    Ct ltree[L_CODES + 2], dtree[D_CODES];
    uint8_t dist_code[512], length_code[256];
    int base_length[LENGTH_CODES], base_dist[D_CODES];
    Static();
};

unsigned bi_reverse(unsigned code, int len)
{
    unsigned res = 0;
    do { res |= code & 1; code >>= 1; res <<= 1; } while (--len > 0);
    return res >> 1;
}

void gen_codes(Ct *tree, int max_code, const uint16_t *bl_count)
{
    uint16_t next_code[MAX_BITS + 1];
    unsigned code = 0;
    for (int bits = 1; bits <= MAX_BITS; ++bits) { code = (code + bl_count[bits - 1]) << 1; next_code[bits] = (uint16_t)code; }
    for (int n = 0; n <= max_code; ++n) {
        const int len = tree[n].len;
        if (len == 0) continue;
        tree[n].code = (uint16_t)bi_reverse(next_code[len]++, len);
    }
}

Static::Static()
{
    int length = 0, code;
    for (code = 0; code < LENGTH_CODES - 1; ++code) {
        base_length[code] = length;
        for (int n = 0; n < (1 << extra_lbits[code]); ++n) length_code[length++] = (uint8_t)code;
    }
    base_length[code] = 0;                               // code 28 (length 258) carries no extra bits
    length_code[length - 1] = (uint8_t)code;
    int dist = 0;
    for (code = 0; code < 16; ++code) {
        base_dist[code] = dist;
        for (int n = 0; n < (1 << extra_dbits[code]); ++n) dist_code[dist++] = (uint8_t)code;
    }
    dist >>= 7;                                          // from now on all distances are divided by 128
    for (; code < D_CODES; ++code) {
        base_dist[code] = dist << 7;
        for (int n = 0; n < (1 << (extra_dbits[code] - 7)); ++n) dist_code[256 + dist++] = (uint8_t)code;
    }
    uint16_t bl_count[MAX_BITS + 1] = {0};
    int n = 0;
    while (n <= 143) { ltree[n++].len = 8; bl_count[8]++; }
    while (n <= 255) { ltree[n++].len = 9; bl_count[9]++; }
    while (n <= 279) { ltree[n++].len = 7; bl_count[7]++; }
    while (n <= 287) { ltree[n++].len = 8; bl_count[8]++; }
    gen_codes(ltree, L_CODES + 1, bl_count);
    for (n = 0; n < D_CODES; ++n) { dtree[n].len = 5; dtree[n].code = (uint16_t)bi_reverse((unsigned)n, 5); }
}
const Static ST;

inline int d_code(unsigned dist) { return dist < 256 ? ST.dist_code[dist] : ST.dist_code[256 + (dist >> 7)]; }

struct TreeDesc { Ct *dyn; const Ct *stat; const int *extra; int extra_base, elems, max_length, max_code; };

// ---------------------------------------------------------------------------------------------
// stage 2: blocks, trees, bits
// ---------------------------------------------------------------------------------------------
// One deflate block: the symbols zlib would have had in its buffer when it flushed.  The bits start at bit 0 of
// `out`; *nbits is their number (the caller splices blocks together at arbitrary bit offsets).
class BlockEncoder {
public:
    // false: zlib would have STORED this block (incompressible data) — not re-stated
    struct Piece { const Sym *p; size_t n; };
    bool encode(const Sym *syms, size_t n, bool last, std::vector<uint8_t> *out, uint64_t *nbits) { const Piece pc{syms, n}; return encode_pieces(&pc, 1, last, out, nbits); }
    // the block's symbols as stretches of several arrays (a round's stitched stream is never made contiguous: every block's thread gathers
    // its own 16 383 symbols into the encoder's buffer, in cache, instead of a 50 MB copy through memory per round)
    bool encode_pieces(const Piece *pc, size_t np, bool last, std::vector<uint8_t> *out, uint64_t *nbits)
    {
        init_block();
        size_t n = 0;
        for (size_t k = 0; k < np; ++k) n += pc[k].n;
        // room for the worst case up front (a match: 15 + 5 + 15 + 13 bits; the trees' description: < 1 KB), so that a code costs a shift,
        // an OR and now and then a 4-byte store — not two vector push_backs per 16 bits
        out_.resize(n * 6 + 2048);
        wp_ = out_.data(); acc_ = 0; acc_bits_ = 0;
        syms_.resize(n);
        { size_t o = 0; for (size_t k = 0; k < np; ++k) { if (pc[k].n) memcpy(syms_.data() + o, pc[k].p, pc[k].n * sizeof(Sym)); o += pc[k].n; } }
        const Sym *const syms = syms_.data();
        for (size_t i = 0; i < n; ++i) {
            const Sym s = syms[i];
            if (is_match(s)) {
                const unsigned lc = (s >> 16) - 3, dist = (s & 0xffff) - 1;
                ltree_[ST.length_code[lc] + LITERALS + 1].freq++;
                dtree_[d_code(dist)].freq++;
                span_ += (s >> 16);
            } else {
                ltree_[s].freq++;
                span_ += 1;
            }
        }
        if (!flush_block(last)) return false;
        *nbits = (uint64_t)(wp_ - out_.data()) * 8 + (uint64_t)acc_bits_;
        for (; acc_bits_ > 0; acc_bits_ -= 8) { *wp_++ = (uint8_t)acc_; acc_ >>= 8; }
        acc_bits_ = 0;
        out_.resize((size_t)(wp_ - out_.data()));
        out->swap(out_);
        return true;
    }

private:
    std::vector<uint8_t> out_;
    std::vector<Sym> syms_;
    uint64_t span_ = 0;                                  // input bytes covered by the buffered symbols
    Ct ltree_[HEAP_SIZE], dtree_[2 * D_CODES + 1], bltree_[2 * BL_CODES + 1];
    uint16_t bl_count_[MAX_BITS + 1];
    int heap_[2 * L_CODES + 1], heap_len_ = 0, heap_max_ = 0;
    uint8_t depth_[2 * L_CODES + 1];
    unsigned long opt_len_ = 0, static_len_ = 0;
    uint64_t acc_ = 0; int acc_bits_ = 0; uint8_t *wp_ = nullptr;      // bit accumulator (least significant bit first), write pointer into out_

    void init_block()
    {
        for (int n = 0; n < L_CODES; ++n) ltree_[n].freq = 0;
        for (int n = 0; n < D_CODES; ++n) dtree_[n].freq = 0;
        for (int n = 0; n < BL_CODES; ++n) bltree_[n].freq = 0;
        ltree_[END_BLOCK].freq = 1;
        opt_len_ = static_len_ = 0;
        syms_.clear(); span_ = 0;
    }
    void send_bits(unsigned value, int length)
    {
        // (length <= 16, fewer than 32 bits pending: never more than 47 in the accumulator)
        acc_ |= (uint64_t)value << acc_bits_;
        acc_bits_ += length;
        if (acc_bits_ >= 32) {
            const uint32_t w = (uint32_t)acc_;
            memcpy(wp_, &w, 4); wp_ += 4;
            acc_ >>= 32; acc_bits_ -= 32;
        }
    }
    void send_code(int c, const Ct *tree) { send_bits(tree[c].code, tree[c].len); }

    bool smaller(const Ct *tree, int n, int m) const
    {
        return tree[n].freq < tree[m].freq || (tree[n].freq == tree[m].freq && depth_[n] <= depth_[m]);
    }
    void pqdownheap(const Ct *tree, int k)
    {
        const int v = heap_[k];
        int j = k << 1;
        while (j <= heap_len_) {
            if (j < heap_len_ && smaller(tree, heap_[j + 1], heap_[j])) ++j;
            if (smaller(tree, v, heap_[j])) break;
            heap_[k] = heap_[j]; k = j;
            j <<= 1;
        }
        heap_[k] = v;
    }
    void gen_bitlen(TreeDesc &d)
    {
        Ct *tree = d.dyn;
        const int max_code = d.max_code, base = d.extra_base, max_length = d.max_length;
        int h, overflow = 0;
        for (int bits = 0; bits <= MAX_BITS; ++bits) bl_count_[bits] = 0;
        tree[heap_[heap_max_]].len = 0;                  // root of the heap
        for (h = heap_max_ + 1; h < HEAP_SIZE; ++h) {
            const int n = heap_[h];
            int bits = tree[tree[n].dad].len + 1;
            if (bits > max_length) { bits = max_length; ++overflow; }
            tree[n].len = (uint16_t)bits;
            if (n > max_code) continue;                  // not a leaf node
            bl_count_[bits]++;
            int xbits = 0;
            if (n >= base) xbits = d.extra[n - base];
            const unsigned long f = tree[n].freq;
            opt_len_ += f * (unsigned long)(bits + xbits);
            if (d.stat) static_len_ += f * (unsigned long)(d.stat[n].len + xbits);
        }
        if (overflow == 0) return;
        do {                                             // find the first bit length which could increase
            int bits = max_length - 1;
            while (bl_count_[bits] == 0) --bits;
            bl_count_[bits]--;
            bl_count_[bits + 1] += 2;
            bl_count_[max_length]--;
            overflow -= 2;
        } while (overflow > 0);
        for (int bits = max_length; bits != 0; --bits) {
            int n = bl_count_[bits];
            while (n != 0) {
                const int m = heap_[--h];
                if (m > max_code) continue;
                if (tree[m].len != (unsigned)bits) {
                    opt_len_ += ((unsigned long)bits - tree[m].len) * tree[m].freq;
                    tree[m].len = (uint16_t)bits;
                }
                --n;
            }
        }
    }
    void build_tree(TreeDesc &d)
    {
        Ct *tree = d.dyn;
        const int elems = d.elems;
        int max_code = -1, node;
        heap_len_ = 0; heap_max_ = HEAP_SIZE;
        for (int n = 0; n < elems; ++n) {
            if (tree[n].freq != 0) { heap_[++heap_len_] = max_code = n; depth_[n] = 0; }
            else tree[n].len = 0;
        }
        while (heap_len_ < 2) {                          // force at least two codes of non zero frequency
            node = heap_[++heap_len_] = (max_code < 2 ? ++max_code : 0);
            tree[node].freq = 1;
            depth_[node] = 0;
            opt_len_--;
            if (d.stat) static_len_ -= d.stat[node].len;
        }
        d.max_code = max_code;
        for (int n = heap_len_ / 2; n >= 1; --n) pqdownheap(tree, n);
        node = elems;
        do {
            const int n = heap_[1];
            heap_[1] = heap_[heap_len_--];
            pqdownheap(tree, 1);
            const int m = heap_[1];
            heap_[--heap_max_] = n;
            heap_[--heap_max_] = m;
            tree[node].freq = (uint16_t)(tree[n].freq + tree[m].freq);
            depth_[node] = (uint8_t)((depth_[n] >= depth_[m] ? depth_[n] : depth_[m]) + 1);
            tree[n].dad = tree[m].dad = (uint16_t)node;
            heap_[1] = node++;
            pqdownheap(tree, 1);
        } while (heap_len_ >= 2);
        heap_[--heap_max_] = heap_[1];
        gen_bitlen(d);
        gen_codes(tree, max_code, bl_count_);
    }
    void scan_tree(Ct *tree, int max_code)
    {
        int prevlen = -1, nextlen = tree[0].len, count = 0, max_count = 7, min_count = 4;
        if (nextlen == 0) { max_count = 138; min_count = 3; }
        tree[max_code + 1].len = 0xffff;                 // guard
        for (int n = 0; n <= max_code; ++n) {
            const int curlen = nextlen; nextlen = tree[n + 1].len;
            if (++count < max_count && curlen == nextlen) continue;
            else if (count < min_count) bltree_[curlen].freq += (uint16_t)count;
            else if (curlen != 0) { if (curlen != prevlen) bltree_[curlen].freq++; bltree_[REP_3_6].freq++; }
            else if (count <= 10) bltree_[REPZ_3_10].freq++;
            else bltree_[REPZ_11_138].freq++;
            count = 0; prevlen = curlen;
            if (nextlen == 0) { max_count = 138; min_count = 3; }
            else if (curlen == nextlen) { max_count = 6; min_count = 3; }
            else { max_count = 7; min_count = 4; }
        }
    }
    void send_tree(Ct *tree, int max_code)
    {
        int prevlen = -1, nextlen = tree[0].len, count = 0, max_count = 7, min_count = 4;
        if (nextlen == 0) { max_count = 138; min_count = 3; }
        for (int n = 0; n <= max_code; ++n) {
            const int curlen = nextlen; nextlen = tree[n + 1].len;
            if (++count < max_count && curlen == nextlen) continue;
            else if (count < min_count) { do { send_code(curlen, bltree_); } while (--count != 0); }
            else if (curlen != 0) {
                if (curlen != prevlen) { send_code(curlen, bltree_); --count; }
                send_code(REP_3_6, bltree_); send_bits((unsigned)(count - 3), 2);
            } else if (count <= 10) { send_code(REPZ_3_10, bltree_); send_bits((unsigned)(count - 3), 3); }
            else { send_code(REPZ_11_138, bltree_); send_bits((unsigned)(count - 11), 7); }
            count = 0; prevlen = curlen;
            if (nextlen == 0) { max_count = 138; min_count = 3; }
            else if (curlen == nextlen) { max_count = 6; min_count = 3; }
            else { max_count = 7; min_count = 4; }
        }
    }
    void compress_block(const Ct *ltree, const Ct *dtree)
    {
        for (const Sym s : syms_) {
            if (!is_match(s)) { send_code((int)s, ltree); continue; }
            unsigned lc = (s >> 16) - 3, dist = (s & 0xffff) - 1;
            int code = ST.length_code[lc];
            send_code(code + LITERALS + 1, ltree);
            int extra = extra_lbits[code];
            if (extra != 0) { lc -= (unsigned)ST.base_length[code]; send_bits(lc, extra); }
            code = d_code(dist);
            send_code(code, dtree);
            extra = extra_dbits[code];
            if (extra != 0) { dist -= (unsigned)ST.base_dist[code]; send_bits(dist, extra); }
        }
        send_code(END_BLOCK, ltree);
    }
    bool flush_block(bool last)
    {
        TreeDesc ld{ltree_, ST.ltree, extra_lbits, LITERALS + 1, L_CODES, MAX_BITS, 0};
        TreeDesc dd{dtree_, ST.dtree, extra_dbits, 0, D_CODES, MAX_BITS, 0};
        TreeDesc bd{bltree_, nullptr, extra_blbits, 0, BL_CODES, MAX_BL_BITS, 0};
        build_tree(ld);
        build_tree(dd);
        // build_bl_tree
        scan_tree(ltree_, ld.max_code);
        scan_tree(dtree_, dd.max_code);
        build_tree(bd);
        int max_blindex;
        for (max_blindex = BL_CODES - 1; max_blindex >= 3; --max_blindex)
            if (bltree_[bl_order[max_blindex]].len != 0) break;
        opt_len_ += 3 * ((unsigned long)max_blindex + 1) + 5 + 5 + 4;
        unsigned long opt_lenb = (opt_len_ + 3 + 7) >> 3;
        const unsigned long static_lenb = (static_len_ + 3 + 7) >> 3;
        if (static_lenb <= opt_lenb) opt_lenb = static_lenb;
        if (span_ + 4 <= opt_lenb) return false;         // zlib would store this block (incompressible data): not re-stated
        if (static_lenb == opt_lenb) {
            send_bits((1u << 1) + (last ? 1u : 0u), 3);  // STATIC_TREES
            compress_block(ST.ltree, ST.dtree);
        } else {
            send_bits((2u << 1) + (last ? 1u : 0u), 3);  // DYN_TREES
            const int lcodes = ld.max_code + 1, dcodes = dd.max_code + 1, blcodes = max_blindex + 1;
            send_bits((unsigned)(lcodes - 257), 5);
            send_bits((unsigned)(dcodes - 1), 5);
            send_bits((unsigned)(blcodes - 4), 4);
            for (int rank = 0; rank < blcodes; ++rank) send_bits(bltree_[bl_order[rank]].len, 3);
            send_tree(ltree_, lcodes - 1);
            send_tree(dtree_, dcodes - 1);
            compress_block(ltree_, dtree_);
        }
        return true;
    }
};

// ---------------------------------------------------------------------------------------------
// stage 1: zlib's own parse of one chunk, read back as symbols
// ---------------------------------------------------------------------------------------------
bool parse_symbols(const uint8_t *in, size_t in_len, std::vector<Sym> &syms)
{
    using namespace pdi;
    static const uint16_t LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115,
                                       131, 163, 195, 227, 258};
    static const uint16_t DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537,
                                       2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t CLORD[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    if (in_len > 0xFFFFFFF0u) return false;
    Tables tb;
    Fast &tf = tb.fast; Slow &t = tb.slow;
    Bits b; b.in = in; b.pos = 0; b.end = (uint32_t)in_len; b.buf = 0; b.cnt = 0;
    for (;;) {
        const uint32_t last = bits_get(b, 1), type = bits_get(b, 2);
        if (type == 1) {
            for (int i = 0; i < 144; ++i) t.cl[i] = 8;
            for (int i = 144; i < 256; ++i) t.cl[i] = 9;
            for (int i = 256; i < 280; ++i) t.cl[i] = 7;
            for (int i = 280; i < 288; ++i) t.cl[i] = 8;
            if (build(t.ll, tf.ll, LL_FAST_BITS, t.cl, 288) < 0) return false;
            for (int i = 0; i < 30; ++i) t.cl[i] = 5;
            if (build(t.d, tf.d, D_FAST_BITS, t.cl, 30) < 0) return false;
        } else if (type == 2) {
            const int nlen = (int)bits_get(b, 5) + 257, ndist = (int)bits_get(b, 5) + 1, ncode = (int)bits_get(b, 4) + 4;
            if (nlen > 286 || ndist > 30) return false;
            for (int i = 0; i < 19; ++i) t.small[i] = 0;
            for (int i = 0; i < ncode; ++i) t.small[CLORD[i]] = (uint8_t)bits_get(b, 3);
            if (build(t.d, tf.d, D_FAST_BITS, t.small, 19) < 0) return false;
            int idx = 0;
            uint8_t *cl = t.cl;
            while (idx < nlen + ndist) {
                const int sym = decode(b, t.d, tf.d, D_FAST_BITS);
                if (sym < 0) return false;
                if (sym < 16) cl[idx++] = (uint8_t)sym;
                else {
                    int rep; uint8_t val = 0;
                    if (sym == 16) { if (idx == 0) return false; val = cl[idx - 1]; rep = 3 + (int)bits_get(b, 2); }
                    else if (sym == 17) rep = 3 + (int)bits_get(b, 3);
                    else rep = 11 + (int)bits_get(b, 7);
                    if (idx + rep > nlen + ndist) return false;
                    while (rep--) cl[idx++] = val;
                }
            }
            if (build(t.ll, tf.ll, LL_FAST_BITS, cl, nlen) < 0) return false;
            if (build(t.d, tf.d, D_FAST_BITS, cl + nlen, ndist) < 0) return false;
        } else return false;                             // a stored block: zlib found the chunk incompressible
        for (;;) {
            const int sym = decode(b, t.ll, tf.ll, LL_FAST_BITS);
            if (sym < 0 || sym > 285) return false;
            if (sym < 256) { syms.push_back((Sym)sym); continue; }
            if (sym == 256) break;
            static const uint8_t LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
            static const uint8_t DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
            const int ls = sym - 257;
            const uint32_t len = LBASE[ls] + bits_get(b, LEXT[ls]);
            const int ds = decode(b, t.d, tf.d, D_FAST_BITS);
            if (ds < 0 || ds > 29) return false;
            const uint32_t dist = DBASE[ds] + bits_get(b, DEXT[ds]);
            syms.push_back((len << 16) | dist);
            if (bits_overrun(b)) return false;
        }
        if (bits_overrun(b)) return false;
        if (last) break;
    }
    return true;
}

struct Chunk {
    uint64_t start = 0, end = 0, tail_end = 0;   // absolute text positions
    // zlib's parse of [start, tail_end), primed with the 32 KiB before start: the chunk's own (zlib ran here), or a stretch of a
    // provider's symbols, kept alive by `hold`
    struct View {
        const Sym *p = nullptr; size_t n = 0;
        size_t size() const { return n; }
        const Sym &operator[](size_t i) const { return p[i]; }
        const Sym *begin() const { return p; }
        const Sym *end() const { return p + n; }
        const Sym *data() const { return p; }
    } syms;
    std::vector<Sym> own;
    std::shared_ptr<SymVec> hold;
    uint32_t crc = 0;                // of [start, end)
    bool ok = false;
    uint64_t from = ~0ull;           // where the symbols begin when that is not `start` (a chunk parsed again from a hand-over point, see emit_round)
    uint64_t first() const { return from != ~0ull ? from : start; }
};

// `text` holds the absolute range [base, ...)
void run_chunk(const uint8_t *text, uint64_t base, Chunk &c)
{
    c.ok = false;
    c.own.clear(); c.syms = Chunk::View();
    z_stream z;
    memset(&z, 0, sizeof z);
    if (deflateInit2(&z, Z_DEFAULT_COMPRESSION, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return;
    const uint8_t *p = text + (c.start - base);
    if (c.start > 0) {
        const uint64_t dl = c.start < 32768 ? c.start : 32768;
        if (deflateSetDictionary(&z, p - dl, (uInt)dl) != Z_OK) { deflateEnd(&z); return; }
    }
    const size_t n = (size_t)(c.tail_end - c.start);
    std::vector<uint8_t> raw(deflateBound(&z, (uLong)n) + 64);
    z.next_in = const_cast<Bytef *>(p); z.avail_in = (uInt)n;
    z.next_out = raw.data(); z.avail_out = (uInt)raw.size();
    const int rc = deflate(&z, Z_FINISH);
    const size_t produced = raw.size() - z.avail_out;
    deflateEnd(&z);
    if (rc != Z_STREAM_END) return;
    c.own.reserve(n / 3 + 16);
    if (!parse_symbols(raw.data(), produced, c.own)) return;
    c.syms.p = c.own.data(); c.syms.n = c.own.size();
    c.crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), p, (uInt)(c.end - c.start));
    c.ok = true;
}

// GF(2) polynomial arithmetic modulo the CRC-32 polynomial, bit-reflected (bit 31 = x^0)
uint32_t crc_mulmod(uint32_t a, uint32_t b)
{
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) { p ^= b; if ((a & (m - 1)) == 0) break; }
        m >>= 1;
        b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
    }
    return p;
}
// x^(8 * len) mod P: appending `len` bytes multiplies a CRC by this
uint32_t crc_shift_op(uint64_t len)
{
    uint32_t sq = 1u << 23;                                   // x^8
    uint32_t p = 1u << 31;                                    // x^0
    for (; len; len >>= 1) {
        if (len & 1) p = crc_mulmod(sq, p);
        sq = crc_mulmod(sq, sq);
    }
    return p;
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// fork-join over [0, n): the calling thread takes part
void parallel_for(int threads, size_t n, const std::function<void(size_t)> &fn)
{
    if (n == 0) return;
    std::atomic<size_t> next{0};
    auto body = [&] { for (;;) { const size_t i = next.fetch_add(1); if (i >= n) return; fn(i); } };
    std::vector<std::thread> pool;
    const size_t extra = std::min<size_t>(threads > 1 ? (size_t)threads - 1 : 0, n - 1);
    for (size_t t = 0; t < extra; ++t) pool.emplace_back(body);
    body();
    for (auto &t : pool) t.join();
}

} // namespace

// ---------------------------------------------------------------------------------------------
// the stream: text in, .gz bytes out, in batches of chunks
// ---------------------------------------------------------------------------------------------
struct Stream::Impl {
    int threads = 1;
    uint64_t CH = 1 << 20, TAIL = 1 << 16;
    uint64_t batch_bytes = (uint64_t)96 << 20;
    unsigned calls = 2;
    std::function<bool(const uint8_t *, size_t)> sink;
    ParseFn parse;
    bool failed = false, finished = false, header_done = false;
    std::vector<uint8_t> buf;        // text [base, base + buf.size())
    uint64_t base = 0, total = 0;
    uint64_t pos = 0;                // every byte before pos is covered by emitted symbols
    uint64_t next_chunk = 0;         // first chunk (on the absolute grid) not yet stitched
    bool have_carry = false; Chunk carry;   // parse of chunk next_chunk, made as the look-ahead of the previous batch
    std::vector<Sym> pending;        // stitched symbols not yet in a block (< LIT_BUFSIZE - 1 between batches)
    uint32_t crc = 0;
    // output bit splicing
    std::vector<uint8_t> out; uint8_t part = 0; int part_bits = 0;

    // All blocks of a round at once: block b's bits start at bit (part_bits + sum of the bit counts before it) of the stretch behind
    // `out`'s end.  Every thread shifts its block into place and stores the bytes that belong to it alone; the (at most two) bytes a
    // block shares with its neighbours are OR-ed together afterwards, in order.  Same bytes as put_bits block by block.
    void splice_blocks(const std::vector<std::vector<uint8_t>> &bytes, const std::vector<uint64_t> &nbits)
    {
        const size_t nb = bytes.size();
        if (nb < 4 || threads < 2) { for (size_t b = 0; b < nb; ++b) put_bits(bytes[b].data(), nbits[b]); return; }
        std::vector<uint64_t> start(nb + 1, 0);
        start[0] = (uint64_t)part_bits;
        for (size_t b = 0; b < nb; ++b) start[b + 1] = start[b] + nbits[b];
        const size_t old_sz = out.size();
        const uint64_t end_bit = start[nb];
        const size_t whole = (size_t)(end_bit / 8);                   // complete bytes of the stretch
        out.resize(old_sz + whole + 16);
        uint8_t *const o = out.data() + old_sz;
        struct Edge { size_t at; uint8_t v; };                        // a shared byte's share
        std::vector<Edge> head(nb, Edge{(size_t)-1, 0}), tail(nb, Edge{(size_t)-1, 0});
        parallel_for(threads, nb, [&](size_t b) {
            const uint64_t s0 = start[b], s1 = start[b + 1];
            if (s1 == s0) return;
            const uint8_t *p = bytes[b].data();
            const unsigned sh = (unsigned)(s0 & 7);
            const size_t B0 = (size_t)(s0 / 8), B1 = (size_t)((s1 + 7) / 8);      // bytes [B0, B1) hold bits of this block
            auto byte_at = [&](size_t j) -> uint8_t {               // output byte j's share from this block (bits outside the block are zero)
                // output bit 8 j + t  =  block bit 8 j + t - s0
                const int64_t q = (int64_t)(8 * j) - (int64_t)s0;    // block bit of output bit 0 of byte j (may be negative)
                unsigned v = 0;
                for (int t = 0; t < 8; ++t) {
                    const int64_t bb = q + t;
                    if (bb < 0 || (uint64_t)bb >= s1 - s0) continue;
                    v |= (unsigned)((p[bb >> 3] >> (bb & 7)) & 1u) << t;
                }
                return (uint8_t)v;
            };
            const bool head_shared = sh != 0, tail_shared = (s1 & 7) != 0;
            size_t lo = B0, hi = B1;                                  // bytes this block owns alone: [lo, hi)
            if (head_shared) { head[b] = Edge{B0, byte_at(B0)}; lo = B0 + 1; }
            if (tail_shared && B1 - 1 >= lo) { tail[b] = Edge{B1 - 1, byte_at(B1 - 1)}; hi = B1 - 1; }
            else if (tail_shared) hi = lo;                            // (the block lies inside one shared byte: its head edge carries all of it)
            if (hi <= lo) return;
            // owned bytes: output byte j = (p[k] >> r) | (p[k + 1] << (8 - r)) with 8 j - s0 = 8 k + r — eight at a time
            const uint64_t first_bit = 8 * (uint64_t)lo - s0;        // >= 0
            size_t k = (size_t)(first_bit / 8); const unsigned r = (unsigned)(first_bit & 7);
            const size_t n_src = bytes[b].size();
            size_t j = lo;
            if (r == 0) { memcpy(o + j, p + k, hi - lo); return; }
            for (; j + 8 <= hi && k + 9 <= n_src; j += 8, k += 8) {
                uint64_t w0; memcpy(&w0, p + k, 8);
                const uint64_t v = (w0 >> r) | ((uint64_t)p[k + 8] << (64 - r));
                memcpy(o + j, &v, 8);
            }
            for (; j < hi; ++j, ++k) {
                const unsigned a = p[k], c = k + 1 < n_src ? p[k + 1] : 0u;
                o[j] = (uint8_t)((a >> r) | (c << (8 - r)));
            }
        });
        // shared bytes: the partial byte carried in from before, then every block's edges, in order
        std::vector<size_t> shared;
        auto add = [&](const Edge &e) { if (e.at == (size_t)-1) return; if (shared.empty() || shared.back() != e.at) { shared.push_back(e.at); o[e.at] = 0; } o[e.at] |= e.v; };
        if (part_bits) { shared.push_back(0); o[0] = part; }
        for (size_t b = 0; b < nb; ++b) { add(head[b]); add(tail[b]); }
        // what is left over: the bits of the last, incomplete byte
        part_bits = (int)(end_bit & 7);
        part = part_bits ? (uint8_t)(o[whole] & ((1u << part_bits) - 1)) : 0;
        out.resize(old_sz + whole);
    }

    void put_bits(const uint8_t *p, uint64_t nbits)
    {
        if (part_bits == 0) {
            const size_t whole = (size_t)(nbits / 8);
            out.insert(out.end(), p, p + whole);
            if (nbits & 7) { part = (uint8_t)(p[whole] & ((1u << (nbits & 7)) - 1)); part_bits = (int)(nbits & 7); }
            return;
        }
        // unaligned: shift whole 64-bit words
        const int sh = part_bits;                            // 1..7
        const size_t whole = (size_t)(nbits / 8), old = out.size();
        const unsigned tailbits = (unsigned)(nbits & 7);
        out.resize(old + whole + 9);
        uint8_t *o = out.data() + old;
        uint64_t carry = part;                               // low `sh` bits valid
        size_t i = 0;
        for (; i + 8 <= whole; i += 8) {
            uint64_t w;
            memcpy(&w, p + i, 8);
            const uint64_t v = carry | (w << sh);
            memcpy(o + i, &v, 8);
            carry = w >> (64 - sh);
        }
        unsigned acc = (unsigned)carry; int have = sh;
        size_t produced = i;
        for (; i < whole; ++i) {
            acc |= (unsigned)p[i] << have;
            o[produced++] = (uint8_t)acc; acc >>= 8;
        }
        if (tailbits) {
            acc |= (unsigned)(p[whole] & ((1u << tailbits) - 1)) << have;
            have += (int)tailbits;
            if (have >= 8) { o[produced++] = (uint8_t)acc; acc >>= 8; have -= 8; }
        }
        out.resize(old + produced);
        part = (uint8_t)acc; part_bits = have;
    }
    // The finished bytes go to the sink on a thread of their own (a round's 15 MB took as long to write as its blocks took to encode),
    // in order, at most two rounds behind; drain_writer() waits until the sink has seen everything handed over so far — every path on
    // which the caller may look at what the sink wrote (finish, a failure, wait_idle) goes through it.
    std::thread writer; std::mutex wmu; std::condition_variable wcv;
    std::deque<std::vector<uint8_t>> wq; bool w_on = false, w_stop = false, w_busy = false, w_fail = false;
    void writer_loop()
    {
        for (;;) {
            std::vector<uint8_t> b;
            {
                std::unique_lock<std::mutex> lk(wmu);
                wcv.wait(lk, [&] { return w_stop || !wq.empty(); });
                if (wq.empty()) return;
                b.swap(wq.front()); wq.pop_front(); w_busy = true;
            }
            const bool ok = w_fail ? false : sink(b.data(), b.size());
            { std::lock_guard<std::mutex> lk(wmu); if (!ok) w_fail = true; w_busy = false; }
            wcv.notify_all();
        }
    }
    bool drain_writer()
    {
        std::unique_lock<std::mutex> lk(wmu);
        wcv.wait(lk, [&] { return wq.empty() && !w_busy; });
        return !w_fail;
    }
    void stop_writer()
    {
        if (!w_on) return;
        { std::lock_guard<std::mutex> lk(wmu); w_stop = true; }
        wcv.notify_all();
        writer.join(); w_on = false;
    }
    bool flush_out(bool final)
    {
        if (final && part_bits) { out.push_back(part); part = 0; part_bits = 0; }
        if (!out.empty()) {
            if (!w_on) { w_on = true; writer = std::thread([this] { writer_loop(); }); }
            std::unique_lock<std::mutex> lk(wmu);
            wcv.wait(lk, [&] { return wq.size() < 2; });
            const size_t cap = out.capacity();
            wq.emplace_back();
            wq.back().swap(out);
            out.reserve(cap);
            lk.unlock();
            wcv.notify_all();
        }
        if (final) return drain_writer();
        std::lock_guard<std::mutex> lk(wmu);
        return !w_fail;
    }

    // ---- a round in two stages, so that stage B of one round runs while stage A of the next is being parsed (on the GPU, with a
    // provider): A = the parse of the chunks whose text has arrived; B = stitch + blocks + bits of a parsed round.  B needs no text.
    uint64_t parsed_hi = 0;                                   // first chunk (on the absolute grid) not yet parsed
    const uint64_t MARGIN = 1024;                             // zlib's look-ahead near the artificial end of a chunk
    struct RoundInfo { double parse_s = 0; size_t provided = 0; };

    // the text stage A reads: a round's text is handed over by the writer, which goes on filling `buf` meanwhile
    std::vector<uint8_t> work; uint64_t work_base = 0;
    // ... and the text of the round before, which stage B may want to look at again (a pair of parses that did not meet)
    std::vector<uint8_t> old; uint64_t old_base = 0;
    bool text_of(uint64_t off, size_t n, uint8_t *dst)
    {
        if (is_remote) return remote.fetch && remote.fetch(off, n, dst);
        if (off >= work_base && off + n <= work_base + work.size()) { memcpy(dst, work.data() + (off - work_base), n); return true; }
        if (off >= old_base && off + n <= old_base + old.size()) { memcpy(dst, old.data() + (off - old_base), n); return true; }
        return false;
    }
    // ... or the text is elsewhere (Remote): no buffers here, positions are the source's
    Remote remote; bool is_remote = false;
    // page-locked stretches of the two text buffers (Params::pin; the buffers are reserved once and never move)
    std::function<bool(void *, size_t)> pin; std::function<void(void *)> unpin;
    struct Pin { void *p; size_t n; };
    std::vector<Pin> pins;
    void ensure_pinned(void *p, size_t used, size_t cap)
    {
        if (!pin || !unpin || used < ((size_t)32 << 20)) return;
        for (size_t i = 0; i < pins.size(); ++i)
            if (pins[i].p == p) { if (pins[i].n >= used) return; unpin(p); pins.erase(pins.begin() + (std::ptrdiff_t)i); break; }
        if (pin(p, cap)) pins.push_back({p, cap});
    }
    // symbol buffers of the provider, taken in turn (a buffer is free again when no chunk holds it; fresh memory costs a page
    // fault per 4 KiB, more than the parse itself)
    std::vector<std::shared_ptr<SymVec>> sym_pool;
    std::shared_ptr<SymVec> take_symvec()
    {
        for (auto &p : sym_pool) if (p.use_count() == 1) return p;
        sym_pool.push_back(std::make_shared<SymVec>());
        return sym_pool.back();
    }

    // stage A: parses the chunks [parsed_hi, c_hi) of `work` into `chunks`
    bool parse_round(bool final, uint64_t c_hi, std::vector<Chunk> &chunks, RoundInfo &info)
    {
        if (c_hi <= parsed_hi) { chunks.clear(); return true; }
        const size_t nc = (size_t)(c_hi - parsed_hi);
        chunks.assign(nc, Chunk());
        for (size_t k = 0; k < nc; ++k) {
            const uint64_t i = parsed_hi + k;
            chunks[k].start = i * CH;
            chunks[k].end = final ? std::min<uint64_t>(total, (i + 1) * CH) : (i + 1) * CH;
            chunks[k].tail_end = final ? std::min<uint64_t>(total, chunks[k].end + TAIL) : chunks[k].end + TAIL;
        }
        const size_t zero_ = 0;
        const double tp0 = now_s();
        size_t provided = 0;
        if ((is_remote ? (bool)remote.parse : (bool)parse) && nc > zero_) {
            // stage 1 by the provider for every new chunk except one that runs to the true end of the text (zlib sees the end of
            // its input there); their CRCs on the threads
            std::vector<uint64_t> tri; std::vector<size_t> which;
            for (size_t k = zero_; k < nc; ++k) {
                const Chunk &c = chunks[k];
                if (final && c.tail_end == total) continue;
                const uint64_t dl = c.start < 32768 ? c.start : 32768;
                if (c.start - dl < work_base) continue;                   // (cannot happen: the buffer keeps 32 KiB before the first chunk)
                tri.push_back(c.start - work_base); tri.push_back(c.tail_end - work_base); tri.push_back(c.start - dl - work_base);
                which.push_back(k);
            }
            // two (or four) calls at a time, each on its share of the chunks and the text they cover: a provider that copies the text elsewhere
            // moves one share while it parses another — and a provider whose parse kernels cannot share the device (the engine's LDS parse: one
            // workgroup fills a CU's LDS) sorts, check-sums and returns the symbols of the other calls under the one that parses
            struct Group { size_t lo = 0, hi = 0; std::shared_ptr<SymVec> sy; std::vector<uint64_t> off; std::vector<uint32_t> crc; bool ok = false; };
            const size_t want_groups = calls >= 4 ? 4 : calls >= 2 ? 2 : 1;
            const size_t n_groups = which.size() >= 256 * want_groups ? want_groups : which.size() >= 512 ? 2 : 1;
            Group grp[4];
            for (size_t g = 0; g < n_groups; ++g) { grp[g].lo = which.size() * g / n_groups; grp[g].hi = which.size() * (g + 1) / n_groups; grp[g].sy = take_symvec(); }
            auto call = [&](Group &G) {
                if (G.hi <= G.lo) return;
                const uint64_t t_lo = tri[3 * G.lo + 2], t_hi = tri[3 * (G.hi - 1) + 1];       // first origin .. last tail end
                std::vector<uint64_t> local(tri.begin() + (std::ptrdiff_t)(3 * G.lo), tri.begin() + (std::ptrdiff_t)(3 * G.hi));
                for (auto &x : local) x -= t_lo;
                if (is_remote) {
                    G.crc.assign(G.hi - G.lo, 0);
                    G.ok = remote.parse(work_base + t_lo, (size_t)(t_hi - t_lo), local.data(), G.hi - G.lo, *G.sy, G.off, G.crc.data(), CH);
                } else
                    G.ok = parse(work.data() + t_lo, (size_t)(t_hi - t_lo), local.data(), G.hi - G.lo, *G.sy, G.off);
                G.ok = G.ok && G.off.size() == G.hi - G.lo + 1 && G.off.back() <= G.sy->size();
            };
            {
                std::thread others[3];
                for (size_t g = 1; g < n_groups; ++g) others[g - 1] = std::thread([&, g] { call(grp[g]); });
                call(grp[0]);
                for (auto &t : others) if (t.joinable()) t.join();
            }
            for (size_t g = 0; g < n_groups; ++g) {
                Group &G = grp[g];
                if (!G.ok) continue;
                parallel_for(threads, G.hi - G.lo, [&](size_t j) {
                    Chunk &c = chunks[which[G.lo + j]];
                    c.syms.p = G.sy->data() + G.off[j]; c.syms.n = (size_t)(G.off[j + 1] - G.off[j]); c.hold = G.sy;
                    c.crc = is_remote ? G.crc[j] : (uint32_t)crc32(crc32(0L, Z_NULL, 0), work.data() + (c.start - work_base), (uInt)(c.end - c.start));
                    c.ok = true;
                });
                provided += G.hi - G.lo;
            }
        }
        {
            std::vector<size_t> todo;
            for (size_t k = zero_; k < nc; ++k) if (!chunks[k].ok) todo.push_back(k);
            if (is_remote) {
                // zlib's chunks (the ones that run to the end of the text; any the source did not parse) on text fetched for them
                std::vector<char> fetched(todo.size(), 1);
                parallel_for(threads, todo.size(), [&](size_t j) {
                    Chunk &c = chunks[todo[j]];
                    const uint64_t dl = c.start < 32768 ? c.start : 32768;
                    std::vector<uint8_t> tmp((size_t)(c.tail_end - (c.start - dl)) + 64, 0);
                    if (!remote.fetch || !remote.fetch(c.start - dl, (size_t)(c.tail_end - (c.start - dl)), tmp.data())) { fetched[j] = 0; return; }
                    run_chunk(tmp.data(), c.start - dl, c);
                });
                for (char f : fetched) if (!f) return false;
            } else
                parallel_for(threads, todo.size(), [&](size_t j) { run_chunk(work.data(), work_base, chunks[todo[j]]); });
        }
        info.parse_s = now_s() - tp0; info.provided = provided;
        parsed_hi = c_hi;
        return true;
    }

    // stage B: `fresh` behind the chunk the previous round left waiting for its successor; stitches, cuts blocks, emits
    bool emit_round(std::vector<Chunk> &fresh, bool final, const RoundInfo &info)
    {
        if (!header_done) {
            static const uint8_t HDR[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3};      // deflate, no flags, mtime 0, xfl 0, OS unix
            out.insert(out.end(), HDR, HDR + 10);
            crc = (uint32_t)crc32(0L, Z_NULL, 0);
            header_done = true;
        }
        std::vector<Chunk> chunks;
        chunks.reserve(fresh.size() + 1);
        if (have_carry) { chunks.push_back(std::move(carry)); have_carry = false; }
        for (auto &c : fresh) chunks.push_back(std::move(c));
        fresh.clear();
        const size_t nc = chunks.size();
        if (!final && nc < 2) { if (nc) { carry = std::move(chunks[0]); have_carry = true; } return true; }   // a stitch needs a chunk and its successor
        const double tp1 = now_s(), tp0 = tp1 - info.parse_s;
        const size_t provided = info.provided;
        for (auto &c : chunks) if (!c.ok) return false;
        // stitch: chunk k hands over to chunk k+1 where both parses end a match at the same position.  Where a pair meets depends on
        // the two parses alone (the earlier hand-over lies a tail's length before the stretch that is searched), so every pair is
        // searched on the threads; the hand-overs are then chained, checked and the symbols copied into place, again on the threads.
        const size_t n_stitch = final ? nc : (nc ? nc - 1 : 0);   // the last parsed chunk waits for its successor unless final
        size_t n_eff = n_stitch;                                  // chunks that contribute symbols (the one that runs to the end is the last)
        if (final) for (size_t k = 0; k < nc; ++k) if (chunks[k].tail_end == total) { n_eff = k + 1; break; }
        std::vector<uint64_t> stop(n_eff, 0);
        std::vector<size_t> i_stop(n_eff, 0), i_first(n_eff, 0);
        std::vector<char> st_ok(n_eff, 1);
        size_t mended = 0;
        auto search = [&](size_t k) {
            const Chunk &a = chunks[k];
            // (the results are kept in locals and stored once: neighbouring k belong to different threads, and a store per shared match end
            // — a few hundred per chunk — made the line that holds stop[k] and stop[k + 1] travel between their cores every time)
            uint64_t stop_k = 0; size_t i_stop_k = a.syms.size();
            if (final && a.tail_end == total) { stop[k] = total; i_stop[k] = i_stop_k; st_ok[k] = 1; return; }
            const Chunk &b = chunks[k + 1];
            // match ends of b inside a's tail (b's parse starts at b.start: only its first symbols are walked)
            std::vector<uint64_t> bends;
            uint64_t q = b.first();
            for (const Sym s : b.syms) {
                q += sym_len(s);
                if (q + MARGIN > a.tail_end) break;
                if (is_match(s)) bends.push_back(q);
            }
            // a's symbols in the tail, walked BACKWARDS from its end (a's parse covers [start, tail_end) exactly):
            // the earliest match end beyond b.start that b shares
            uint64_t qe = a.tail_end;                             // end position of symbol i - 1 ... start of symbol i
            size_t j = bends.size();
            for (size_t i = a.syms.size(); i-- > 0;) {
                const Sym sy = a.syms[i];                         // symbol i covers [qe - len, qe)
                if (qe <= b.first()) break;
                if (is_match(sy) && qe + MARGIN <= a.tail_end) {
                    while (j > 0 && bends[j - 1] > qe) --j;
                    if (j > 0 && bends[j - 1] == qe) { stop_k = qe; i_stop_k = i + 1; }     // keep going: an earlier one is better
                }
                qe -= sym_len(sy);
            }
            stop[k] = stop_k; i_stop[k] = i_stop_k;
            st_ok[k] = stop_k != 0;                               // 0: the two parses did not meet inside the tail
        };
        parallel_for(threads, n_eff, search);
        const double tq1 = now_s();
        // A pair that did not meet (the two parses can stay out of step for longer than a tail where the text repeats with a long
        // period) is mended in place: the successor is parsed AGAIN by zlib from a's last match end before it — after a match zlib is
        // in its start state, so that parse is a's own continuation — and takes over there; its own hand-over to the chunk behind
        // it is then searched afresh.  Needs the text: a Remote source keeps a round's text until the round behind it is done, a host
        // text stays in `old` that long.
        // Round 5: the pairs that did not meet are mended TOGETHER — with 8 + 2 KiB chunks (twice the waves per CU on the device) about
        // one pair in a hundred needs it, a hundred per round: their restart points are found, the text they need is fetched in ONE
        // piece (a Remote source answers a fetch with a device synchronise: one per pair was 0.1 ms each, in a row), zlib parses the
        // successors again on the threads, and every mended successor's own hand-over is searched afresh; what that turns up (a
        // mended parse that no longer meets ITS successor) is the next pass's work.  Runs of neighbouring failures stay in order.
        for (int pass = 0; pass < 64; ++pass) {
            std::vector<size_t> bad;
            for (size_t k = 0; k < n_eff; ++k) if (!st_ok[k]) bad.push_back(k);
            if (bad.empty()) break;
            if (pass == 63) return false;
            // runs of consecutive failures [bad[r0], bad[r1]) are mended front to back by one thread: pair k + 1 restarts inside the parse pair k has just made
            std::vector<std::pair<size_t, size_t>> runs;
            for (size_t i = 0; i < bad.size();) { size_t j = i + 1; while (j < bad.size() && bad[j] == bad[j - 1] + 1) ++j; runs.emplace_back(i, j); i = j; }
            // the text: from 32 KiB before the first pair's predecessor to the last successor's end, one fetch when that is a round's worth or less
            uint64_t lo = ~0ull, hi = 0;
            for (size_t k : bad) {
                // (a restart point lies behind a.start + TAIL — checked below —, its dictionary at most 32 KiB before it: what a source keeps)
                const uint64_t a0 = chunks[k].start + TAIL > 32768 ? chunks[k].start + TAIL - 32768 : 0;
                lo = std::min(lo, a0); hi = std::max(hi, chunks[k + 1].tail_end);
            }
            const bool one_fetch = hi - lo <= ((uint64_t)512 << 20);
            std::vector<uint8_t> &mt = mend_text;
            if (one_fetch) {
                if (mt.size() < (size_t)(hi - lo) + 64) mt.resize((size_t)(hi - lo) + 64);
                if (!text_of(lo, (size_t)(hi - lo), mt.data())) return false;
            }
            std::vector<char> run_ok(runs.size(), 1);
            std::mutex fetch_mu;
            parallel_for(threads, runs.size(), [&](size_t r) {
                for (size_t bi = runs[r].first; bi < runs[r].second; ++bi) {
                    const size_t k = bad[bi];
                    if (st_ok[k]) continue;                           // (met after all: its predecessor's mend changed the parse it is searched against)
                    const Chunk &a = chunks[k];
                    Chunk &b = chunks[k + 1];
                    uint64_t q = a.first(), q0 = 0; size_t i0 = 0;
                    for (size_t i = 0; i < a.syms.size(); ++i) {
                        q += sym_len(a.syms[i]);
                        if (q > b.start) break;
                        if (is_match(a.syms[i])) { q0 = q; i0 = i + 1; }
                    }
                    if (q0 <= a.start + TAIL || q0 <= a.first()) { run_ok[r] = 0; return; }      // (no match to restart from behind the stretch a's own hand-over lies in)
                    const uint64_t dl = q0 < 32768 ? q0 : 32768;
                    Chunk again;
                    again.start = q0; again.end = b.tail_end; again.tail_end = b.tail_end;
                    if (one_fetch && q0 - dl >= lo) run_chunk(mt.data(), lo, again);
                    else {
                        std::vector<uint8_t> tmp((size_t)(b.tail_end - (q0 - dl)) + 64, 0);
                        { std::lock_guard<std::mutex> lk(fetch_mu); if (!text_of(q0 - dl, (size_t)(b.tail_end - (q0 - dl)), tmp.data())) { run_ok[r] = 0; return; } }
                        run_chunk(tmp.data(), q0 - dl, again);
                    }
                    if (!again.ok) { run_ok[r] = 0; return; }
                    b.own.swap(again.own); b.syms.p = b.own.data(); b.syms.n = b.own.size(); b.hold.reset(); b.from = q0;
                    stop[k] = q0; i_stop[k] = i0; st_ok[k] = 1;
                    if (k + 1 < n_eff) search(k + 1);                 // (pair k + 1: in this run if it had failed before, else found failing by the next pass)
                }
            });
            for (char ok1 : run_ok) if (!ok1) return false;
            mended += bad.size();
        }
        const double tq2 = now_s();
        // the chain of hand-overs: chunk k's symbols start at the previous hand-over
        std::vector<uint64_t> from(n_eff, 0);
        { uint64_t p = pos; for (size_t k = 0; k < n_eff; ++k) { from[k] = p; if (stop[k] <= p && !(final && stop[k] == p && p == total)) return false; p = stop[k]; } }
        parallel_for(threads, n_eff, [&](size_t k) {
            const Chunk &a = chunks[k];
            uint64_t qa = a.first();
            size_t i = 0;
            while (i < i_stop[k] && qa < from[k]) { qa += sym_len(a.syms[i]); ++i; }
            if (qa != from[k]) st_ok[k] = 0;                      // the hand-over is not a symbol boundary of this parse
            i_first[k] = i;
        });
        for (size_t k = 0; k < n_eff; ++k) if (!st_ok[k]) return false;
        const double tq3 = now_s();
        // the round's symbol stream = what the round before left over + every chunk's stretch between its two hand-overs — kept as PIECES
        // (piece 0: the left-over; piece k + 1: chunk k), with their start offsets in the stream
        std::vector<BlockEncoder::Piece> piece(n_eff + 1);
        std::vector<size_t> at(n_eff + 2, 0);
        const std::vector<Sym> carried(pending.begin(), pending.end());
        pending.clear();
        piece[0] = BlockEncoder::Piece{carried.data(), carried.size()};
        at[1] = carried.size();
        for (size_t k = 0; k < n_eff; ++k) {
            piece[k + 1] = BlockEncoder::Piece{chunks[k].syms.data() + i_first[k], i_stop[k] > i_first[k] ? i_stop[k] - i_first[k] : 0};
            at[k + 2] = at[k + 1] + piece[k + 1].n;
        }
        const size_t n_syms = at[n_eff + 1];
        // the stretch [lo, hi) of the stream as pieces
        auto stretch = [&](size_t lo, size_t hi, std::vector<BlockEncoder::Piece> *out_pc) {
            out_pc->clear();
            if (hi <= lo) return;
            size_t k = (size_t)(std::upper_bound(at.begin(), at.end(), lo) - at.begin()) - 1;      // at[k] <= lo < at[k + 1]
            for (; k <= n_eff && at[k] < hi; ++k) {
                const size_t a0 = std::max(lo, at[k]), a1 = std::min(hi, at[k + 1]);
                if (a1 > a0) out_pc->push_back(BlockEncoder::Piece{piece[k].p + (a0 - at[k]), a1 - a0});
            }
        };
        const double tq4 = now_s();
        // CRC-32 of the concatenation: crc(A B) = crc(A) * x^(8 |B|) + crc(B) in GF(2)[x] / P.  Nearly all chunks have one length, so the
        // power is computed once per length (zlib 1.2.11's crc32_combine squares a 32 x 32 matrix per call: 10 us x thousands of chunks)
        {
            uint64_t op_len = ~0ull; uint32_t op = 0;
            for (size_t k = 0; k < (final ? nc : n_stitch); ++k) {
                const uint64_t len = chunks[k].end - chunks[k].start;
                if (len != op_len) { op = crc_shift_op(len); op_len = len; }
                crc = crc_mulmod(op, crc) ^ chunks[k].crc;
            }
        }
        if (n_eff) pos = stop[n_eff - 1];
        if (final && pos != total) return false;
        if (!final && nc) { carry = std::move(chunks[nc - 1]); have_carry = true; }
        next_chunk += n_stitch;
        // blocks: every LIT_BUFSIZE - 1 symbols; at the end the remainder (possibly empty) closes the stream
        const double tp2 = now_s();
        const size_t BS = LIT_BUFSIZE - 1;
        const size_t full = n_syms / BS;
        const size_t nblocks = full + (final ? 1 : 0);
        std::vector<std::vector<uint8_t>> bytes(nblocks);
        std::vector<uint64_t> nbits(nblocks, 0);
        std::vector<char> good(nblocks, 1);
        parallel_for(threads, nblocks, [&](size_t b) {
            BlockEncoder enc;
            std::vector<BlockEncoder::Piece> pc;
            const size_t lo = b * BS, hi = b < full ? lo + BS : n_syms;
            stretch(lo, hi, &pc);
            good[b] = enc.encode_pieces(pc.data(), pc.size(), final && b + 1 == nblocks, &bytes[b], &nbits[b]) ? 1 : 0;
        });
        const double tp3 = now_s();
        for (size_t b = 0; b < nblocks; ++b) if (!good[b]) return false;
        splice_blocks(bytes, nbits);
        const double tp4 = now_s();
        if (!final) {
            std::vector<BlockEncoder::Piece> pc;
            stretch(full * BS, n_syms, &pc);
            for (const auto &q : pc) pending.insert(pending.end(), q.p, q.p + q.n);
        }
        if (final) {
            if (part_bits) { out.push_back(part); part = 0; part_bits = 0; }
            for (int k = 0; k < 4; ++k) out.push_back((uint8_t)(crc >> (8 * k)));
            for (int k = 0; k < 4; ++k) out.push_back((uint8_t)((uint32_t)total >> (8 * k)));   // ISIZE = length mod 2^32
        }
        if (!flush_out(final)) return false;
        if (getenv("PGZ_DEBUG"))
            fprintf(stderr, "[pgz] round: %zu chunks (%zu parsed by the provider, %zu parsed again where a pair did not meet), %zu blocks: parse %.3f s, stitch %.3f s, encode %.3f s, splice %.3f s\n", nc, provided, mended, nblocks,
                    tp1 - tp0, tp2 - tp1, tp3 - tp2, tp4 - tp3);
        if (getenv("PGZ_DEBUG")) fprintf(stderr, "[pgz]   stitch: search %.4f, mend %.4f, chain + first %.4f, copy %.4f, crc + rest %.4f; flush %.4f\n", tq1 - tp1, tq2 - tq1, tq3 - tq2, tq4 - tq3, tp2 - tq4, now_s() - tp4);
        return true;
    }

    std::vector<uint8_t> mend_text;                           // the text of a round's pairs that are parsed again (kept between rounds)
    SymVec stitched;                                          // a round's stitched symbols (kept: see sym_pool)
    std::vector<Chunk> waiting; RoundInfo waiting_info;       // a parsed round whose stage B has not run yet
    bool have_waiting = false;
    std::thread worker; bool worker_on = false, worker_ok = true;

    // one round: stage A of the text in `work` while stage B of the round before runs
    bool round_body(bool final, uint64_t c_hi)
    {
        std::vector<Chunk> fresh; RoundInfo fresh_info;
        bool a_ok = true, b_ok = true;
        const bool dbg = getenv("PGZ_DEBUG") != nullptr;
        static double t_first = 0, t_prev_end = 0;
        const double r0 = now_s();
        if (t_first == 0) t_first = t_prev_end = r0;
        double a_s = 0, b_s = 0;
        std::thread ta([&] { const double t = now_s(); a_ok = parse_round(final, c_hi, fresh, fresh_info); a_s = now_s() - t; });
        if (have_waiting) { const double t = now_s(); b_ok = emit_round(waiting, false, waiting_info); have_waiting = false; b_s = now_s() - t; }
        ta.join();
        if (dbg) { const double r1 = now_s(); fprintf(stderr, "[pgz] round body at %.4f s (%.4f s after the one before ended): stage A %.4f s, stage B (the round before) %.4f s, body %.4f s\n", r0 - t_first, r0 - t_prev_end, a_s, b_s, r1 - r0); t_prev_end = r1; }
        if (!a_ok || !b_ok) return false;
        if (final) return emit_round(fresh, true, fresh_info);
        if (!fresh.empty()) { waiting = std::move(fresh); waiting_info = fresh_info; have_waiting = true; }
        return true;
    }
    bool join_worker() { if (worker_on) { worker.join(); worker_on = false; } return worker_ok; }

    // everything that can be decided with the text seen so far (all of it when final).  Not final: the round runs on a worker
    // thread behind the caller, who goes on writing; a failure is reported by the next call.
    bool run(bool final, size_t room)
    {
        if (!join_worker()) return false;
        const uint64_t avail = is_remote ? total : base + buf.size();   // == total
        uint64_t c_hi = parsed_hi;                            // exclusive
        if (final) c_hi = total == 0 ? 1 : (total + CH - 1) / CH;
        // (strictly inside the text seen so far: a chunk parsed in a round that is not the last one then never has tail_end == total,
        // so the "runs to the end of the stream" branches of the final round only ever meet chunks that zlib itself parsed with the
        // end of its input in sight — the provider's parse is not zlib's within MIN_LOOKAHEAD of a chunk's end)
        else while ((c_hi + 1) * CH + TAIL < avail) ++c_hi;
        if (!final && c_hi <= parsed_hi) return true;
        if (is_remote) {
            work_base = 0;
            if (final) return round_body(true, c_hi);
            // what the next round still needs; released one round late: stage B of a round (run with the next round's stage A) may
            // fetch text of its chunks (emit_round: a pair of parses that did not meet)
            const uint64_t rel = base;
            base = c_hi * CH > 32768 + CH ? c_hi * CH - 32768 - CH : 0;    // (the history of the chunk that waits for its successor too: it may have to be mended)
            worker_ok = true; worker_on = true;
            worker = std::thread([this, c_hi, rel] { worker_ok = round_body(false, c_hi); if (worker_ok && remote.release && rel) remote.release(rel); });
            return true;
        }
        if (work.capacity() < room) work.reserve(room);      // (first round: the buffers get their final size before any is locked)
        if (!final && old.capacity() < room) old.reserve(room);
        old.swap(work); old_base = work_base;                // the text of the round whose stage B is still to come
        work.swap(buf); work_base = base;
        if (final) { buf.clear(); return round_body(true, c_hi); }
        if (parse) ensure_pinned(work.data(), work.size(), work.capacity());
        // the writer keeps what the next round needs: everything from the dictionary of the next chunk to parse
        const uint64_t keep = std::max<uint64_t>(work_base, c_hi * CH > 32768 + CH ? c_hi * CH - 32768 - CH : 0);   // (+ the waiting chunk and its history: see emit_round)
        buf.clear();
        if (buf.capacity() < room) buf.reserve(room);
        buf.insert(buf.end(), work.begin() + (std::ptrdiff_t)(keep - work_base), work.end());
        base = keep;
        worker_ok = true; worker_on = true;
        worker = std::thread([this, c_hi] { worker_ok = round_body(false, c_hi); });
        return true;
    }
    ~Impl() { if (worker_on) worker.join(); stop_writer(); for (auto &x : pins) unpin(x.p); }
};

Stream::Stream(int threads, std::function<bool(const uint8_t *, size_t)> sink, const Params &p) : p_(new Impl)
{
    p_->threads = threads < 1 ? 1 : threads;
    p_->sink = std::move(sink);
    p_->parse = p.parse; p_->pin = p.pin; p_->unpin = p.unpin;
    p_->CH = p.chunk < 8192 ? 8192 : p.chunk;
    p_->TAIL = p.tail < 2048 ? 2048 : p.tail;
    if (p_->TAIL > p_->CH / 2) p_->TAIL = p_->CH / 2;
    if (p.batch) p_->batch_bytes = p.batch;
    p_->calls = p.calls;
}
Stream::Stream(int threads, std::function<bool(const uint8_t *, size_t)> sink, const Params &p, const Remote &source) : Stream(threads, std::move(sink), p)
{
    p_->remote = source; p_->is_remote = true;
}
Stream::~Stream() { delete p_; }

bool Stream::announce(uint64_t n)
{
    if (p_->failed || p_->finished || !p_->is_remote) return false;
    p_->total += n;
    const uint64_t limit = p_->batch_bytes + 2 * p_->CH + p_->TAIL;
    if (p_->total - p_->base >= limit && !p_->run(false, 0)) { p_->failed = true; (void)p_->drain_writer(); return false; }
    return true;
}

bool Stream::wait_idle()
{
    if (!p_->join_worker()) { p_->failed = true; (void)p_->drain_writer(); return false; }
    if (!p_->drain_writer()) p_->failed = true;
    return !p_->failed;
}

bool Stream::write(const void *data, size_t n)
{
    if (p_->failed || p_->finished || p_->is_remote) return false;
    const uint8_t *q = (const uint8_t *)data;
    const size_t limit = (size_t)(p_->batch_bytes + 2 * p_->CH + p_->TAIL);
    if (p_->buf.capacity() < limit + (size_t)p_->CH) p_->buf.reserve(limit + (size_t)p_->CH);   // (address space; pages come as the text does)
    while (n) {
        const size_t have = p_->buf.size();
        const size_t k = std::min(n, have < limit ? limit - have : (size_t)p_->CH);
        p_->buf.insert(p_->buf.end(), q, q + k);
        p_->total += k; q += k; n -= k;
        if (p_->buf.size() >= limit && !p_->run(false, limit + (size_t)p_->CH)) { p_->failed = true; (void)p_->drain_writer(); return false; }
    }
    return true;
}

bool Stream::finish()
{
    if (p_->failed || p_->finished) return false;
    p_->finished = true;
    if (!p_->run(true, 0)) { p_->failed = true; (void)p_->join_worker(); (void)p_->drain_writer(); return false; }
    return true;
}

Params Params::for_device(ParseFn fn)
{
    Params p;
    // one wave parses one chunk, and the parse is bound by the latency of dependent loads: many small chunks beat few large ones.  Round 5:
    // 8 KiB chunks with 2 KiB of overlap — sixteen of them share a CU's LDS with their text (the engine's "lz_group" default), twice the waves
    // of 16 + 4 KiB.  The parses of tables and per-site rows meet within a few lines; about one pair in a few thousand does not meet inside
    // 2 KiB and is parsed again by zlib from the hand-over point (emit_round: a round's pairs together, on the threads, from one text fetch).
    p.chunk = (size_t)1 << 13; p.tail = (size_t)1 << 11; p.batch = (size_t)96 << 20;
    // (measurement knobs: the geometry in KiB / KiB / MiB)
    if (const char *e = getenv("PGZ_DEV_CHUNK_KB")) { const size_t v = strtoull(e, nullptr, 10); if (v >= 8 && v <= 4096) p.chunk = v << 10; }
    if (const char *e = getenv("PGZ_DEV_TAIL_KB")) { const size_t v = strtoull(e, nullptr, 10); if (v >= 2 && (v << 10) < p.chunk) p.tail = v << 10; }
    if (const char *e = getenv("PGZ_DEV_BATCH_MB")) { const size_t v = strtoull(e, nullptr, 10); if (v >= 1 && v <= 2048) p.batch = v << 20; }
    p.parse = std::move(fn);
    return p;
}

ParseFn host_emulation_parse()
{
    return [](const uint8_t *text, size_t n, const uint64_t *chunks, size_t n_chunks, SymVec &syms, std::vector<uint64_t> &off) -> bool {
        if (n < 3) return false;
        // positions sorted by (hash, position): a stable counting sort (the device does this with a radix sort)
        std::vector<uint8_t> padded(text, text + n);
        padded.resize(n + 32, 0);
        const size_t np = n - 2;
        std::vector<uint32_t> bucket((1u << pdz::HASH_BITS) + 1, 0), S(np), R(n + 1, 0);
        for (size_t p = 0; p < np; ++p) bucket[pdz::hash3(&padded[p]) + 1]++;
        for (size_t h = 0; h < (1u << pdz::HASH_BITS); ++h) bucket[h + 1] += bucket[h];
        { std::vector<uint32_t> cur(bucket.begin(), bucket.end() - 1);
          for (size_t p = 0; p < np; ++p) { const uint32_t i = cur[pdz::hash3(&padded[p])]++; S[i] = (uint32_t)p; R[p] = i; } }
        const pdz::Text T{padded.data(), S.data(), R.data(), bucket.data(), n};
        off.assign(n_chunks + 1, 0);
        syms.clear();
        for (size_t k = 0; k < n_chunks; ++k) {
            const uint64_t start = chunks[3 * k], end = chunks[3 * k + 1], origin = chunks[3 * k + 2];
            const size_t at = syms.size();
            syms.resize(at + (size_t)(end - start) + 8);
            pdz::Out o{syms.data() + at, 0u, (uint32_t)(end - start + 8)};
            if (!pdz::parse_chunk<pdz::HostWave>(T, start, end, origin, o)) return false;
            syms.resize(at + o.n);
            off[k + 1] = syms.size();
        }
        return true;
    };
}

// zlib's own parse of data[dict, dict + n) primed with data[0, dict) (what run_chunk computes for one chunk): tests compare
// the engine's wave-parallel parse (csrc/pd_lz77.h) with it
bool zlib_chunk_symbols(const uint8_t *data, size_t dict, size_t n, std::vector<uint32_t> &syms)
{
    Chunk c;
    c.start = dict; c.end = dict + n; c.tail_end = dict + n;
    if (dict > 32768) { data += dict - 32768; c.start = 32768; c.end = c.tail_end = 32768 + n; }
    run_chunk(data, 0, c);
    if (!c.ok) return false;
    syms.swap(c.own);
    return true;
}

bool gzip_identical(const uint8_t *data, size_t n, int threads, std::vector<uint8_t> &out, const Params &p)
{
    std::vector<uint8_t> img;
    img.reserve(n / 4 + 1024);
    Stream st(threads, [&](const uint8_t *b, size_t k) { img.insert(img.end(), b, b + k); return true; }, p);
    if (!st.write(data, n) || !st.finish()) return false;
    out.insert(out.end(), img.begin(), img.end());
    return true;
}

bool gzip_identical(const uint8_t *data, size_t n, int threads, std::vector<uint8_t> &out)
{
    return gzip_identical(data, n, threads, out, Params());
}

} // namespace pgz
