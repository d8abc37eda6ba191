// pd_inflate_wave.h — WAVE-COOPERATIVE raw DEFLATE (RFC 1951) decoder for one BGZF block: one 64-lane
// wavefront per block, all 64 lanes decoding the SAME block at once.
//
// Huffman decoding is serial by nature (a symbol's length says where the next one starts), so the wave
// decodes SPECULATIVELY and then synchronises:
//   * the compressed bits of a deflate block are cut into 64 subsequences of S bits; lane l decodes
//     subsequence l starting at a guessed bit position (lane 0's start is the true one);
//   * a Huffman decoder started at a wrong position falls back onto the true symbol boundaries after a few
//     symbols (self-synchronisation), so most lanes END at the true position even when they started wrong.
//     Round after round every lane takes the end of its left neighbour as its start and re-decodes if that
//     changed; by induction lanes 0..k are exact after round k, and in practice all lanes agree after 2-3
//     rounds.  No output is written while speculating: a round only counts the bytes a subsequence emits;
//   * an exclusive wave scan of those counts gives every lane its output offset; a last pass writes the
//     literals and copies the matches.  A match may read bytes another lane of the same pass has not written
//     yet, so the pass runs in rounds: a lane stalls at a match whose source is not final — final = below the
//     frontier F (the write cursor of the first unfinished lane at the start of the round) or inside the
//     lane's own region — and the first unfinished lane never stalls, so at most 64 rounds are needed.
// Table construction (canonical codes -> two-level lookup tables: a 10-bit / 8-bit root plus sub-tables for
// longer codes, one or two LDS lookups per symbol) is lane-parallel as well.
//
// The SAME source compiles for gfx950 (class W = the hardware wave: ballots, DPP / shuffles, LDS) and for the
// host (class W = 64 emulated lanes in a loop), where tests/harness/inflate_wave_check.cpp compares it with
// zlib on every block of the test files before it runs on a GPU.  Code outside W::each() is wave-uniform
// (every lane computes the same values); code inside runs per lane; collectives are W:: functions.
//
// Anything this decoder does not want to judge (incomplete Huffman codes other than the usual "one distance
// code", sub-table overflow) returns PD_W_HOST: the caller hands that block's unit back to the host decoder.
// Every loop is bounded by the input / output sizes: corrupt data ends in an error code, never in a hang or
// an out-of-bounds write.
#ifndef PD_INFLATE_WAVE_H_
#define PD_INFLATE_WAVE_H_
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define PW_FN __host__ __device__ __forceinline__
#else
#define PW_FN inline
#endif

#ifndef PW_MARK
#define PW_MARK(code, val) do { } while (0)     /* development hooks (tools/ubench/wave_debug.hip): progress marks, */
#endif
#ifndef PW_TICK
#define PW_TICK(phase) do { } while (0)         /* cycles since the previous tick are charged to `phase` */
#endif

namespace pdw {

// Root widths of the two lookup tables.  The tables are most of a wave's LDS (one wave per workgroup), and LDS is what limits the waves
// per CU of this latency-bound kernel: a 9-bit literal/length root instead of 10 (2 KiB less; codes of 10+ bits take the second lookup) lets
// 20 waves share a CU instead of 16 — measured on 61 220 members of a 50x payload BAM: 178 -> 209 GB/s of inflated bytes
// (tools/ubench/inflate_ab.py, profiles/r04_inflate_ab.txt).
#ifndef PD_LL_ROOT
#define PD_LL_ROOT 9
#endif
#ifndef PD_D_ROOT
#define PD_D_ROOT 8
#endif
// phase 2 (decode_body): the parts of the subsequences behind their second checkpoints handed to idle lanes
#ifndef PD_P2_BALANCE
#define PD_P2_BALANCE 1
#endif
#ifndef PD_CK_N
#define PD_CK_N 3                         /* checkpoints per subsequence (3 or 4) */
#endif
// Match tokens of one superstep a wave's scratch holds; a superstep with more goes to the host decoder (PD_W_HOST).  A superstep cannot have more than
// out_len / 3 + 1 (a match emits three bytes or more): the default never declines.
#ifndef PD_TOK_CAP
#define PD_TOK_CAP (65536 / 3 + 1)
#endif
#ifndef PD_CRC_STEP
#define PD_CRC_STEP 64                    /* bytes of a lane's CRC step (the next step's bytes are in flight meanwhile: 2 x PD_CRC_STEP / 4 registers) */
#endif
#ifndef PD_EACH_OPAQUE
#define PD_EACH_OPAQUE 1
#endif
#ifndef PD_LIT2
#define PD_LIT2 1                         /* two literals per trip of phases 1 and 2 where the second follows the first without reaching a bound */
#endif
#ifndef PD_LIT2_P2
#define PD_LIT2_P2 PD_LIT2               /* ... in phase 2 as well */
#endif
#ifndef PD_P2_LITS
#define PD_P2_LITS 2                      /* literals per trip of phase 2 (2 .. 4: a symbol is at most 15 bits, the window holds 64) */
#endif
#ifndef PD_LIT3
#define PD_LIT3 0                         /* a third literal per trip of phase 1 */
#endif
#ifndef PD_P2_HANDOVER
#define PD_P2_HANDOVER 16                 /* idle lanes it takes for a hand-over */
#endif
// sub-table areas: zlib's `enough` bounds — 286 literal/length symbols of at most 15 bits need 852 entries with a 9-bit root (340 behind the
// root) and 820 with a 10-bit one (308 are needed, 320 kept); a code that would need more goes to the host (build_table checks)
enum { LL_ROOT = PD_LL_ROOT, LL_SUBCAP = PD_LL_ROOT <= 9 ? 340 : 320, D_ROOT = PD_D_ROOT, D_SUBCAP = 256 };
enum { TOK_SCRATCH = PD_TOK_CAP + 63 };   // entries of a wave's token scratch
enum { PD_W_OK = 0, PD_W_HOST = 1 };      // negative values: corrupt stream (same codes as pd_inflate_core.h)
enum { KIND_LIT = 0, KIND_LEN = 1, KIND_EOB = 2, KIND_BAD = 3 };
enum { F_EOB = 1, F_INVALID = 2, F_OVERRUN = 4 };

// Table entry: [3:0] code length in bits (sub-table pointer: index width)  [5:4] kind  [6] sub-table pointer
//              [15:8] literal byte / number of extra bits  [31:16] base value / sub-table offset
struct alignas(16) Tables {
    uint32_t ll[(1 << LL_ROOT) + LL_SUBCAP];
    uint32_t d[(1 << D_ROOT) + D_SUBCAP];
    uint16_t sorted[320];                 // symbols ordered by (code length, symbol)
    uint32_t work[160];                   // table building: uint16 rank[320] (position of a symbol inside its length class);
                                          // phase 3: bdst[64] | bend[64] (destination range of every match of the current batch)
    uint8_t cl[320];                      // code lengths of the deflate block being set up
    uint32_t ck_more[PD_CK_N * 128 > 400 ? PD_CK_N * 128 - 400 : 4];   // the rest of phase 1's checkpoints (PD_CK_N x 512 B over sorted | work | cl | ck_more)
    PW_FN uint16_t *rank() { return reinterpret_cast<uint16_t *>(work); }
    PW_FN uint32_t *bdst() { return work; }
    PW_FN uint32_t *bend() { return work + 64; }
    PW_FN uint64_t *psel() { return reinterpret_cast<uint64_t *>(sorted + 64); }
    PW_FN uint32_t *ckpt() { return reinterpret_cast<uint32_t *>(sorted); }       // phase 1: PD_CK_N checkpoints x 2 fields x 64 lanes over sorted | work | cl | ck_more   // phase 3: 56 selectors (sorted[0 .. 64): a chunk's marks)
};

struct Stats {                            // host builds only (tuning): how much redundant work the speculation costs
    uint64_t blocks = 0, dblocks = 0, steps = 0, sync_rounds = 0, emit_rounds = 0, sym_true = 0, sym_decoded = 0, lanes_redecoded = 0;
    uint64_t copy_serial = 0, long_matches = 0;     // per round: the longest ready match's 8-byte pieces (what the wave waits for); matches > 16 bytes
    uint64_t wave_iters_sync = 0, wave_iters_emit = 0, copy_iters = 0, hdr_syms = 0, matches = 0, batches = 0, tmp_max = 0, handovers = 0;
};

PW_FN uint64_t ld64(const uint8_t *p) { uint64_t w; __builtin_memcpy(&w, p, 8); return w; }
PW_FN void st64(uint8_t *p, uint64_t w) { __builtin_memcpy(p, &w, 8); }
struct V16 { uint64_t a, b; };
PW_FN V16 ld128(const uint8_t *p) { V16 w; __builtin_memcpy(&w, p, 16); return w; }
PW_FN void st128(uint8_t *p, V16 w) { __builtin_memcpy(p, &w, 16); }
PW_FN uint32_t bitrev32(uint32_t x)
{
#if defined(__clang__)
    return __builtin_bitreverse32(x);
#else
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    return __builtin_bswap32(x);
#endif
}
PW_FN uint32_t bitrev(uint32_t x, int n) { return n ? bitrev32(x) >> (32 - n) : 0u; }
PW_FN uint32_t peek_bits(const uint8_t *in, uint32_t q) { return (uint32_t)(ld64(in + (q >> 3)) >> (q & 7)); }   // >= 32 valid bits

template <int MODE> PW_FN uint32_t make_entry(uint32_t sym, uint32_t len)
{
    if (MODE == 0) {                      // literal / length alphabet
        if (sym < 256) return len | (KIND_LIT << 4) | (sym << 8);
        if (sym == 256) return len | (KIND_EOB << 4);
        if (sym > 285) return KIND_BAD << 4;
        const uint32_t ls = sym - 257;
        uint32_t ext = 0, base = 3 + ls;
        if (ls == 28) base = 258;
        else if (ls >= 8) { ext = (ls >> 2) - 1; base = ((4 + (ls & 3)) << ext) + 3; }
        return len | (KIND_LEN << 4) | (ext << 8) | (base << 16);
    }
    if (MODE == 1) {                      // distance alphabet
        if (sym > 29) return KIND_BAD << 4;
        uint32_t ext = 0, base = sym + 1;
        if (sym >= 4) { ext = (sym >> 1) - 1; base = ((2 + (sym & 1)) << ext) + 1; }
        return len | (KIND_LEN << 4) | (ext << 8) | (base << 16);
    }
    return len | (KIND_LIT << 4) | (sym << 8);     // the code-length code
}

// Canonical Huffman code -> lookup table `tab` (root of 2^ROOT entries + sub-tables for longer codes).
// Returns 0, PD_W_HOST (incomplete code / sub-table area too small) or -3 (over-subscribed, no codes).
template <class W, int ROOT, int SUBCAP, int MODE>
PW_FN int build_table(uint32_t *tab, const uint8_t *cl, int n, uint16_t *sorted, uint16_t *rank)
{
    typedef typename W::template Var<uint32_t> U;
    uint32_t count[16], first[16], offs[16];                 // wave-uniform; only ever indexed by unrolled constants
#pragma unroll
    for (int v = 0; v < 16; ++v) count[v] = 0;
    // class sizes and every symbol's position inside its class, by ballots (symbol order is kept inside a class)
    for (int b = 0; b < n; b += 64) {
        U L;
        W::each([&](int l) { const int s = b + l; L[l] = s < n ? cl[s] : 0u; });
#pragma unroll
        for (int v = 1; v <= 15; ++v) {
            const uint64_t m = W::ballot_eq(L, (uint32_t)v);
            if (m) {
                const uint32_t c0 = count[v];
                W::each([&](int l) { if (L[l] == (uint32_t)v) rank[b + l] = (uint16_t)(c0 + W::prefix_count(m, l)); });
                count[v] += (uint32_t)__builtin_popcountll(m);
            }
        }
    }
    uint32_t total = 0;
#pragma unroll
    for (int v = 1; v <= 15; ++v) total += count[v];
    if (total == 0) {
        if (MODE != 1) return -3;
        W::each([&](int l) { for (uint32_t i = l; i < (1u << ROOT); i += 64) tab[i] = KIND_BAD << 4; });   // a block without matches
        W::sync();
        return 0;
    }
    int left = 1;
#pragma unroll
    for (int v = 1; v <= 15; ++v) { left <<= 1; left -= (int)count[v]; if (left < 0) return -3; }
    // incomplete codes: only "one distance code of one bit" is decoded here (zlib accepts exactly that)
    if (left > 0 && !(MODE == 1 && total == 1 && count[1] == 1)) return PD_W_HOST;
    {
        uint32_t code = 0, off = 0;
#pragma unroll
        for (int v = 1; v <= 15; ++v) { code = (code + (v > 1 ? count[v - 1] : 0u)) << 1; first[v] = code; offs[v] = off; off += count[v]; }
    }
    first[0] = offs[0] = 0;
    W::sync();
    W::each([&](int l) {
        for (int s = l; s < n; s += 64) {
            const uint32_t len = cl[s];
            if (!len) continue;
            uint32_t o = 0;
#pragma unroll
            for (int v = 1; v <= 15; ++v) if (len == (uint32_t)v) o = offs[v];
            sorted[o + rank[s]] = (uint16_t)s;
        }
    });
    W::sync();
    // root: every index asks which code (of at most ROOT bits) its bits start with
    W::each([&](int l) {
        for (uint32_t i = l; i < (1u << ROOT); i += 64) {
            const uint32_t rc = bitrev(i, ROOT);
            uint32_t flen = 0, fidx = 0;
#pragma unroll
            for (int v = 1; v <= ROOT; ++v) {
                const uint32_t c = rc >> (ROOT - v);
                if (!flen && c >= first[v] && c - first[v] < count[v]) { flen = (uint32_t)v; fidx = offs[v] + (c - first[v]); }
            }
            tab[i] = flen ? make_entry<MODE>(sorted[fidx], flen) : (uint32_t)(KIND_BAD << 4);
        }
    });
    uint32_t n_long = 0;
#pragma unroll
    for (int v = ROOT + 1; v <= 15; ++v) n_long += count[v];
    if (n_long) {
        // ROOT-bit prefixes (most significant bit first) pmin .. 2^ROOT - 1 lead to codes longer than ROOT bits; the
        // sub-table of a prefix is indexed by as many further bits as its longest code needs
        const uint32_t pmin = first[ROOT] + count[ROOT];
        const uint32_t np = (1u << ROOT) - pmin;
        const uint32_t ch = (np + 63) / 64;
        auto sub_bits = [&](uint32_t P) -> uint32_t {
            uint32_t sb = 0;
#pragma unroll
            for (int v = ROOT + 1; v <= 15; ++v)
                if (count[v] && P >= (first[v] >> (v - ROOT)) && P <= ((first[v] + count[v] - 1) >> (v - ROOT))) sb = (uint32_t)(v - ROOT);
            return sb;
        };
        U size;
        W::each([&](int l) {
            uint32_t s = 0;
            for (uint32_t j = 0; j < ch; ++j) {
                const uint32_t P = pmin + (uint32_t)l * ch + j;
                if (P < (1u << ROOT)) { const uint32_t sb = sub_bits(P); if (sb) s += 1u << sb; }
            }
            size[l] = s;
        });
        uint32_t sub_total = 0;
        const U ex = W::excl_scan(size, &sub_total);
        if (sub_total > (uint32_t)SUBCAP) return PD_W_HOST;
        W::each([&](int l) {
            uint32_t run = (1u << ROOT) + ex[l];
            for (uint32_t j = 0; j < ch; ++j) {
                const uint32_t P = pmin + (uint32_t)l * ch + j;
                if (P >= (1u << ROOT)) break;
                const uint32_t sb = sub_bits(P);
                if (!sb) continue;
                tab[bitrev(P, ROOT)] = sb | (KIND_BAD << 4) | 0x40u | (run << 16);
                run += 1u << sb;
            }
        });
        W::sync();
        const uint32_t t0 = total - n_long;
        W::each([&](int l) {
            for (uint32_t t = t0 + (uint32_t)l; t < total; t += 64) {
                const uint32_t sym = sorted[t];
                const uint32_t len = cl[sym];
                uint32_t code = 0;
#pragma unroll
                for (int v = ROOT + 1; v <= 15; ++v) if (len == (uint32_t)v) code = first[v] + (t - offs[v]);
                const uint32_t lowbits = len - ROOT;
                const uint32_t e = tab[bitrev(code >> lowbits, ROOT)];
                const uint32_t off = e >> 16, sb = e & 15;
                const uint32_t r = bitrev(code & ((1u << lowbits) - 1), (int)lowbits);
                const uint32_t ent = make_entry<MODE>(sym, len);
                for (uint32_t k = 0; k < (1u << (sb - lowbits)); ++k) tab[off + r + (k << lowbits)] = ent;
            }
        });
    }
    W::sync();
    return 0;
}

struct Sym { uint32_t kind, val, dist, used; };

// A lane's view of the compressed bits: 16 bytes (cur | nxt) starting at byte `base` of the stream plus the next 8
// (`ahead`), loaded one step before they are needed — the load is issued when the window moves on, and the window
// moves on only every 64 bits (several symbols), so no symbol waits for memory.  Reads stay below in + lim.
struct BitWin { uint64_t cur, nxt, ahead; uint32_t base; };

// (a load from a clamped address, not a predicated one: bits at and beyond lim are never a symbol that counts — the passes stop at their
// bounds — and an unconditional load can land in the window's own register, where nothing waits for it until the window moves again;
// the predicated form went through a temporary whose copy waited for the load the moment it was issued, in every trip of phases 1 and 2)
PW_FN uint64_t win_load(const uint8_t *in, uint32_t at, uint32_t lim) { return ld64(in + (at + 8 <= lim ? at : lim - 8)); }
PW_FN void win_init(BitWin &b, const uint8_t *in, uint32_t q, uint32_t lim)
{
    b.base = q >> 3;
    b.cur = win_load(in, b.base, lim); b.nxt = win_load(in, b.base + 8, lim); b.ahead = win_load(in, b.base + 16, lim);
}
// 64 bits starting at bit q (q >= 8 * base; at most 64 bits beyond the window start after the caller's last step)
PW_FN uint64_t win_bits(BitWin &b, const uint8_t *in, uint32_t q, uint32_t lim)
{
    uint32_t off = q - 8 * b.base;
    if (off >= 64) {                                                      // moved past `cur` (a symbol is <= 48 bits: at most once)
        b.cur = b.nxt; b.nxt = b.ahead; b.base += 8; off -= 64;
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_sched_barrier(0);                                 // (the load is issued AFTER the two copies, into the register they freed)
#endif
        b.ahead = win_load(in, b.base + 16, lim);
    }
    return off ? (b.cur >> off) | (b.nxt << (64 - off)) : b.cur;
}

// One literal/length symbol (with its distance) from the 64 bits `w` that start at its first bit (<= 48 are used).
// Straight-line on purpose (the lanes of a wave sit on different symbols): the distance lookup is done for every
// symbol and ignored unless a length was decoded; only the rare second-level lookups branch.
PW_FN Sym decode_sym(const Tables &T, uint64_t w)
{
    uint32_t e = T.ll[(uint32_t)w & ((1u << LL_ROOT) - 1)];
    if (e & 0x40u) e = T.ll[(e >> 16) + (((uint32_t)w >> LL_ROOT) & ((1u << (e & 15)) - 1))];
    const uint32_t n = e & 15, kind = (e >> 4) & 3;
    const bool is_len = kind == KIND_LEN;
    w >>= n;
    const uint32_t ext = is_len ? (e >> 8) & 0xff : 0u;
    const uint32_t lenv = (e >> 16) + ((uint32_t)w & ((1u << ext) - 1));
    w >>= ext;
    uint32_t f = T.d[(uint32_t)w & ((1u << D_ROOT) - 1)];
    if (is_len && (f & 0x40u)) f = T.d[(f >> 16) + (((uint32_t)w >> D_ROOT) & ((1u << (f & 15)) - 1))];
    const uint32_t dn = f & 15, dext = (f >> 8) & 0xff;
    const uint32_t dist = (f >> 16) + ((uint32_t)(w >> dn) & ((1u << dext) - 1));
    Sym s;
    s.kind = is_len && ((f >> 4) & 3) != KIND_LEN ? (uint32_t)KIND_BAD : kind;
    s.val = is_len ? lenv : (e >> 8) & 0xff;
    s.dist = is_len ? dist : 0u;
    s.used = n + (is_len ? ext + dn + dext : 0u);
    return s;
}

// A plain literal right behind a symbol (PD_LIT2, round 6): its code length, 0 when what follows is anything else (a length, the end of the block, a code of more
// than LL_ROOT bits, an invalid code); w = the bits from that second symbol on.  The lanes that make a wave's loops long are the ones inside stretches of
// literals (a record's packed bases: ~75 literals in a row) — two of them per trip halve those lanes' trips, for one more table look-up per trip of everybody.
PW_FN uint32_t peek_literal(const Tables &T, uint64_t w, uint32_t *lit)
{
    const uint32_t e = T.ll[(uint32_t)w & ((1u << LL_ROOT) - 1)];
    *lit = (e >> 8) & 0xffu;
    return (e & 0x70u) == (uint32_t)(KIND_LIT << 4) ? (e & 15u) : 0u;     // (bit 6: a sub-table pointer; bits 5:4: the kind)
}

// Phase 1 state of one lane.  The speculative pass over a subsequence runs from bit p until the first symbol boundary
// at or after `bound` (or a stop: end of block, invalid code, end of input); nothing is written: e = where it ended,
// n = bytes it would emit, m = matches among its symbols.
// Two checkpoints (S/8 and S/2 bits into the subsequence) remember where the previous pass of this lane crossed
// them: a re-decode from a corrected start that crosses a checkpoint at the SAME bit has merged with the previous
// pass (from a common symbol boundary on, two passes are identical), so it stops there and keeps the old tail.  A corrected start lies a
// few bits behind the old one and the codes re-synchronise within a few symbols, so the first checkpoint catches nearly every merge.
// The checkpoints live in LDS (Tables::ckpt(): the table builder's scratch, idle while a block is decoded), a column per lane:
// word (k * 3 + f) * 64 of a lane's column = field f (0 position, 1 bytes, 2 matches) of checkpoint k.  (Until round 5 there were three
// checkpoints in nine registers, chosen by chains of selects: most of the vector instructions of a symbol step, and the reason the kernel spilled.)
struct SubCount {
    uint32_t e, n, m, f, ns;                     // results of the last complete pass
    uint32_t q, out, nm, stage, next_t;          // the pass in progress
    BitWin win;
};
enum { CK_STRIDE = 2 * 64, CK_N = PD_CK_N, CK_NONE = 0xFFFFFFFFu };
// where checkpoint k lies in a subsequence of S bits that nominally starts at bit `nominal`: S/8 (the one that catches the merges), then — four
// checkpoints — 11S/32, 9S/16 and 25S/32, or — three — 7S/16 and 23S/32: they cut a subsequence into a first part (11/32; 7/16) and equal further
// parts (7/32; 9/32), which phase 2 hands to whoever is idle (round 6)
PW_FN uint32_t ck_at(uint32_t nominal, uint32_t S, uint32_t k)
{
    if (CK_N >= 4) return k == 0 ? nominal + (S >> 3) : k == 1 ? nominal + ((11u * S) >> 5) : k == 2 ? nominal + ((9u * S) >> 4) : k == 3 ? nominal + ((25u * S) >> 5) : 0xFFFFFFFFu;
    return k == 0 ? nominal + (S >> 3) : k == 1 ? nominal + ((7u * S) >> 4) : k == 2 ? nominal + ((23u * S) >> 5) : 0xFFFFFFFFu;
}

PW_FN void count_begin(SubCount &c, const uint8_t *in, uint32_t in_lim, uint32_t p, uint32_t nominal, uint32_t S)
{
    c.q = p; c.out = 0; c.nm = 0; c.stage = 0; c.next_t = ck_at(nominal, S, 0); c.ns = 0;
    win_init(c.win, in, p, in_lim);
}

// one symbol of the pass in progress; returns false when the pass is over
// (lim = min(bound, in_bits): a pass that stops at lim before reaching its bound ran out of input; ck = the lane's checkpoint column)
PW_FN bool count_step(const Tables &T, const uint8_t *in, uint32_t in_lim, uint32_t nominal, uint32_t S, uint32_t bound, uint32_t lim, bool have_prev,
                      SubCount &c, uint32_t *ck)
{
    uint32_t flag = 0;
    const uint32_t q = c.q;
    bool stop = q >= lim;
    if (stop && q < bound) flag = F_OVERRUN;
    if (!stop && q >= c.next_t) {                                         // crossing a checkpoint (CK_N times per pass)
        const uint32_t st = c.stage;                                      // 0 .. CK_N - 1
        uint32_t *const e = ck + st * CK_STRIDE;
        if (have_prev && e[0] == q) {
            // same tail as before; the counts remembered from here on were relative to the old start
            // (bytes in 17 bits | matches in 15: a pass from a wrong start may have counted anything, what is kept is kept modulo the fields' widths —
            // the TRUE counts of a subsequence fit them, and sums and differences of counts are exact modulo a power of two)
            const uint32_t was = e[64], d_o = c.out - (was & 0x1ffffu), d_m = c.nm - (was >> 17);
#pragma unroll
            for (uint32_t k = 0; k < (uint32_t)CK_N; ++k) if (k >= st) {
                const uint32_t w = ck[k * CK_STRIDE + 64];
                ck[k * CK_STRIDE + 64] = ((w + d_o) & 0x1ffffu) | (((w >> 17) + d_m) << 17);
            }
            c.n = (c.n + d_o) & 0x1ffffu; c.m = (c.m + d_m) & 0x7fffu;
            return false;                                                 // e, f and the later checkpoints stay
        }
        e[0] = q; e[64] = (c.out & 0x1ffffu) | (c.nm << 17);
        c.next_t = ck_at(nominal, S, st + 1);
        c.stage = st + 1;
    }
    if (!stop) {
        const uint64_t w = win_bits(c.win, in, q, in_lim);
        const Sym s = decode_sym(T, w);
        const bool bad = s.kind == KIND_BAD, eob = s.kind == KIND_EOB;
        uint32_t used = bad ? 0u : s.used, n_sym = bad ? 0u : 1u, n_out = bad ? 0u : s.kind == KIND_LIT ? 1u : s.kind == KIND_LEN ? s.val : 0u;
#if PD_LIT2
        if (s.kind == KIND_LIT) {
            // a second literal in the same trip — exactly what the next trip would have done, as long as that trip would not have stopped or crossed
            // a checkpoint at its start (q1 below the limit and below the next checkpoint): the pass visits the same boundaries either way
            uint32_t b2;
            const uint32_t q1 = q + s.used, n2 = peek_literal(T, w >> s.used, &b2);
            if (n2 && q1 < lim && q1 < c.next_t) {
                used += n2; n_sym = 2; n_out = 2;
#if PD_LIT3
                uint32_t b3;
                const uint32_t q2 = q1 + n2, n3 = peek_literal(T, w >> used, &b3);
                if (n3 && q2 < lim && q2 < c.next_t) { used += n3; n_sym = 3; n_out = 3; }
#endif
            }
        }
#endif
        c.q = q + used;
        c.ns += n_sym;
        c.out += n_out;
        c.nm += !bad && s.kind == KIND_LEN ? 1u : 0u;
        flag = bad ? (uint32_t)F_INVALID : eob ? (uint32_t)F_EOB : 0u;
        stop = bad || eob;
    }
    if (!stop) return true;
    // the pass ran to its end: a checkpoint it did not reach must not stay behind for the next comparison
    const uint32_t st = c.stage;
#pragma unroll
    for (uint32_t k = 0; k < (uint32_t)CK_N; ++k) if (st <= k) ck[k * CK_STRIDE] = CK_NONE;
    c.e = c.q; c.f = flag; c.n = c.out; c.m = c.nm;
    return false;
}

PW_FN uint32_t ld32(const uint8_t *p) { uint32_t w; __builtin_memcpy(&w, p, 4); return w; }
PW_FN void st32(uint8_t *p, uint32_t w) { __builtin_memcpy(p, &w, 4); }
PW_FN void st16(uint8_t *p, uint32_t w) { const uint16_t h = (uint16_t)w; __builtin_memcpy(p, &h, 2); }

// exactly n (1..8) low bytes of v, as at most two (overlapping) stores
PW_FN void store_bytes(uint8_t *p, uint64_t v, uint32_t n)
{
    if (n >= 8) st64(p, v);
    else if (n >= 4) { st32(p, (uint32_t)v); st32(p + n - 4, (uint32_t)(v >> (8 * (n - 4)))); }
    else if (n >= 2) { st16(p, (uint32_t)v); st16(p + n - 2, (uint32_t)(v >> (8 * (n - 2)))); }
    else p[0] = (uint8_t)v;
}

// off mod d for 0 <= off < 2^16, 1 <= d < 2^16, without an integer division: a float quotient from the reciprocal (the one-instruction
// approximation on the GPU — a full-precision 1/d is a dozen instructions), corrected by one step either way
PW_FN uint32_t small_mod(uint32_t off, uint32_t d)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float r = __builtin_amdgcn_rcpf((float)d);
#else
    const float r = 1.0f / (float)d;
#endif
    const uint32_t q = (uint32_t)((float)off * r);
    int32_t o = (int32_t)(off - q * d);
    if (o < 0) o += (int32_t)d;
    if ((uint32_t)o >= d) o -= (int32_t)d;
    return (uint32_t)o;
}

// The bytes of `raw` picked by the eight 3-bit indices in the bytes of `sel` (byte k of the result = byte sel_k of raw)
PW_FN uint64_t gather8(uint64_t raw, uint64_t sel)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t lo = (uint32_t)raw, hi = (uint32_t)(raw >> 32);
    return ((uint64_t)__builtin_amdgcn_perm(hi, lo, (uint32_t)(sel >> 32)) << 32) | __builtin_amdgcn_perm(hi, lo, (uint32_t)sel);
#else
    uint64_t v = 0;
    for (int k = 0; k < 8; ++k) v |= ((raw >> (8 * ((sel >> (8 * k)) & 7))) & 0xff) << (8 * k);
    return v;
#endif
}
// Selectors of the short periods: psel[(d - 1) * 8 + o], d = 1 .. 7, o = 0 .. d - 1: byte k = (o + k) mod d — the eight bytes of a
// sequence of period d from its o-th byte on, as indices into the period.  56 words of LDS, filled once per superstep.
PW_FN uint64_t period_selector(uint32_t d, uint32_t o)
{
    uint64_t sel = 0;
    uint32_t q = o;
    for (int k = 0; k < 8; ++k) { sel |= (uint64_t)q << (8 * k); q = q + 1 == d ? 0 : q + 1; }
    return sel;
}

// A match waiting to be copied: out[dst .. dst+len) = out[dst-dist ..] (positions inside the member's output).
struct Token { uint32_t dst; uint32_t len_dist; };             // len_dist = len | dist << 16

// The compressed data of one deflate block (Huffman tables already in T), starting at bit q_io.  On success the
// block's end-of-block code has been consumed: q_io is the bit after it, o_io the output position.
// `tok` = scratch for out_len / 3 + 64 tokens (global memory on the GPU).
template <class W>
PW_FN int decode_body(const uint8_t *in, uint32_t in_bits, uint32_t &q_io, uint8_t *out, uint32_t out_len, uint32_t &o_io, Tables &T,
                      Token *tok, Stats *st)
{
    typedef typename W::template Var<uint32_t> U;
    uint32_t base = q_io, o = o_io;
    const uint32_t max_steps = in_bits / 64 + 2;
    const uint32_t in_lim = in_bits / 8 + 8;                              // readable bytes: the payload and its 8-byte trailer
    for (uint32_t step = 0; step < max_steps; ++step) {
        // subsequence width: the rest of the BGZF payload spread over the wave — a block that fills its BGZF member
        // (the usual case) is ONE superstep; wide subsequences also re-synchronise inside themselves more often
        uint32_t S = (in_bits - base + 63) / 64;
        if (S < 64) S = 64;
        typename W::template Var<SubCount> c;
        U p, need, bound;
        W::each([&](int l) {
            bound[l] = base + (uint32_t)(l + 1) * S;
            p[l] = base + (uint32_t)l * S;
            need[l] = 1;
            SubCount &x = c[l];
            x.e = x.n = x.m = x.f = x.ns = 0;
            x.q = x.out = x.nm = x.stage = x.next_t = 0;
            for (int k = 0; k < 2 * CK_N; ++k) T.ckpt()[k * 64 + l] = k % 2 ? 0u : (uint32_t)CK_NONE;
        });
        // ---- phase 1: speculate and synchronise (nothing is written) ----
        uint32_t kend = 64;
        for (int round = 0;; ++round) {
            if (round > 66) return -9;
            PW_MARK(30, (uint32_t)round);
            // a wave-uniform loop around one predicated symbol step per lane (lanes run out at different trips)
            U act;
            W::each([&](int l) { act[l] = need[l]; if (need[l]) count_begin(c[l], in, in_lim, p[l], base + (uint32_t)l * S, S); });
            uint32_t trips = 0;
            while (W::ballot_ne(act, 0u)) {
                if (++trips > 2 * S + 64) return -9;
                W::each([&](int l) {
                    if (act[l]) act[l] = count_step(T, in, in_lim, base + (uint32_t)l * S, S, bound[l], bound[l] < in_bits ? bound[l] : in_bits, round > 0, c[l], T.ckpt() + l);
                });
            }
            if (round == 0) PW_TICK(7);
            if (st) { st->sync_rounds++; st->wave_iters_sync += trips;
                      W::each([&](int l) { if (need[l]) { st->sym_decoded += c[l].ns; st->lanes_redecoded++; } }); }
            U e, f;
            W::each([&](int l) { e[l] = c[l].e; f[l] = c[l].f; });
            const U pe = W::shift_up1(e, base);
            W::each([&](int l) { const uint32_t np = l == 0 ? base : pe[l]; need[l] = np != p[l]; p[l] = np; });
            const uint64_t nm = W::ballot_ne(need, 0u);
            const int settled = nm ? __builtin_ctzll(nm) : 64;            // lanes [0, settled) started at their true position
            const uint64_t sm = W::ballot_ne(f, 0u) & (settled >= 64 ? ~0ull : ((1ull << settled) - 1));
            if (sm) { kend = (uint32_t)__builtin_ctzll(sm); break; }      // the block (or the valid data) ends in lane kend
            if (!nm) break;
        }
        const uint32_t n_valid = kend < 64 ? kend + 1 : 64;
        U e, n, m;
        W::each([&](int l) { e[l] = c[l].e; n[l] = (uint32_t)l < n_valid ? c[l].n : 0u; m[l] = (uint32_t)l < n_valid ? c[l].m : 0u; });
        if (kend < 64) {
            U f;
            W::each([&](int l) { f[l] = c[l].f; });
            const uint32_t fk = W::bcast(f, (int)kend);
            if (fk & F_INVALID) return -4;
            if (fk & F_OVERRUN) return -7;
        }
        uint32_t total = 0, n_tok = 0;
        const U ex = W::excl_scan(n, &total);
        const U exm = W::excl_scan(m, &n_tok);
        if (total > out_len - o) return -5;
        if (n_tok > out_len / 3 + 1) return -5;                          // (cannot happen: a match emits >= 3 bytes)
        if (n_tok > (uint32_t)PD_TOK_CAP) return PD_W_HOST;              // (more than the scratch was made for)
        if (st) st->steps++;
        PW_MARK(40, total);
        PW_TICK(2);
        // ---- phase 2: every lane writes its literals and lists its matches, in output order (no lane waits) ----
        // The subsequences are equal in BITS, not in symbols: a stretch of literals has three times the symbols of a stretch of matches, and a
        // loop of this kind lasts as long as its busiest lane (207 trips against 99 on average on a 50x short-read file).  Phase 1 left the
        // means to even that out (round 6): a lane's checkpoints behind the first are symbol boundaries of its TRUE pass (ck_at), with the bytes and
        // matches before them — so a lane stops at its second checkpoint, the parts behind it go to a POOL
        // (part k of the pool = what follows checkpoint 1 + k / 64 of lane k % 64), and whoever has nothing to do takes the next part from the
        // pool, sixteen lanes at a time (a hand-over costs the newcomers' first window loads, which everybody waits for).
        U err, q2, w2, t2, act2, end2;
        typename W::template Var<BitWin> bw;
        uint32_t pool_n = 0, pool_next = 0; (void)pool_n; (void)pool_next;
        W::each([&](int l) {
            err[l] = 0; q2[l] = p[l]; w2[l] = o + ex[l]; t2[l] = exm[l]; act2[l] = (uint32_t)l < n_valid; end2[l] = bound[l];
            if (act2[l]) win_init(bw[l], in, p[l], in_lim); else { bw[l].cur = bw[l].nxt = bw[l].ahead = 0; bw[l].base = 0; }
        });
#if PD_P2_BALANCE
        W::each([&](int l) {
            uint32_t *const ck = T.ckpt() + l;
            ck[0] = o + ex[l]; ck[64] = exm[l];                                // (the first checkpoint has done its work: its words hold the lane's bases)
#pragma unroll
            for (uint32_t k = 1; k < (uint32_t)CK_N; ++k) {
                const uint32_t mq = ck[k * CK_STRIDE];
                const bool ok = (uint32_t)l < n_valid && mq != (uint32_t)CK_NONE && mq > p[l] && mq < bound[l];
                if (!ok) ck[k * CK_STRIDE] = CK_NONE;
                else if (end2[l] == bound[l]) end2[l] = mq;                    // the lane's own part ends at its first usable checkpoint behind the first
            }
        });
        W::sync();
        pool_n = (uint32_t)(CK_N - 1) * 64u;
#endif
        {
            uint32_t trips = 0;
            for (;;) {
                const uint64_t am = W::ballot_ne(act2, 0u);
#if PD_P2_BALANCE
                if (pool_next < pool_n) {
                    const uint64_t idle = ~am;
                    const uint32_t n_idle = (uint32_t)__builtin_popcountll(idle);
                    if (n_idle >= (uint32_t)PD_P2_HANDOVER) {                  // (am == 0: all 64)
                        W::each([&](int l) {
                            if (act2[l]) return;
                            const uint32_t k = pool_next + W::prefix_count(idle, l);
                            if (k >= pool_n) return;
                            const uint32_t j = k & 63u, part = 1u + (k >> 6);
                            const uint32_t *const ck = T.ckpt() + j;
                            const uint32_t from = ck[part * CK_STRIDE];
                            if (from == (uint32_t)CK_NONE) return;
                            uint32_t to = base + (j + 1u) * S;
#pragma unroll
                            for (uint32_t n = (uint32_t)CK_N - 1u; n >= 2u; --n) if (n > part && ck[n * CK_STRIDE] != (uint32_t)CK_NONE) to = ck[n * CK_STRIDE];
                            const uint32_t cnt = ck[part * CK_STRIDE + 64];
                            q2[l] = from; end2[l] = to; w2[l] = ck[0] + (cnt & 0x1ffffu); t2[l] = ck[64] + (cnt >> 17); act2[l] = 1;
                            win_init(bw[l], in, from, in_lim);
                        });
                        pool_next = pool_n - pool_next < n_idle ? pool_n : pool_next + n_idle;
                        if (st) st->handovers++;
                        continue;
                    }
                }
#endif
                if (!am) break;
                if (++trips > 3 * S + 128) return -9;
                W::each([&](int l) {
                    if (!act2[l]) return;
                    if (q2[l] >= end2[l]) {
                        act2[l] = 0; return;
                    }
                    const uint64_t wb = win_bits(bw[l], in, q2[l], in_lim);
                    const Sym s = decode_sym(T, wb);
                    const uint32_t wl = w2[l];
                    if (st) st->sym_true++;
                    if (s.kind == KIND_LIT) {
#if PD_LIT2_P2
                        // (a further literal that begins before the part's end belongs to the part: its end is a symbol boundary; up to PD_P2_LITS per trip,
                        // their bytes in one store)
                        uint32_t word = s.val, nb = 1, used = s.used;
#pragma unroll
                        for (int k = 1; k < PD_P2_LITS; ++k) {
                            uint32_t b2;
                            const uint32_t n2 = nb == (uint32_t)k ? peek_literal(T, wb >> used, &b2) : 0u;
                            if (n2 && q2[l] + used < end2[l]) { word |= b2 << (8 * k); used += n2; ++nb; if (st) st->sym_true++; }
                        }
                        if (nb > 1) {
                            if (nb == 2) st16(out + wl, word);
                            else if (nb == 3) { st16(out + wl, word); out[wl + 2] = (uint8_t)(word >> 16); }
                            else st32(out + wl, word);
                            w2[l] = wl + nb; q2[l] += used;
                            return;
                        }
#endif
                        out[wl] = (uint8_t)s.val;
                        w2[l] = wl + 1;
                    } else if (s.kind == KIND_LEN) {
                        if (s.dist > wl) { err[l] = 6; act2[l] = 0; return; }
                        tok[t2[l]].dst = wl; tok[t2[l]].len_dist = s.val | (s.dist << 16); t2[l] += 1;
                        w2[l] = wl + s.val;
                    } else {                                              // end of block
                        act2[l] = 0; return;
                    }
                    q2[l] += s.used;
                });
            }
            if (st) { st->wave_iters_emit += trips; st->matches += n_tok; }
        }
        if (W::ballot_ne(err, 0u)) return -6;
        W::fence();
        PW_TICK(3);
        // ---- phase 3: the matches, up to 64 CONSECUTIVE ones at a time, one per lane.  Neighbouring matches depend on each other (a
        // record repeats the record before it, and a batch spans about four records), so a batch takes three to four rounds: a lane
        // copies once the earlier matches of the batch that overlap its source are done (the exact set, as a lane mask); everything
        // before the batch — literals of phase 2, earlier batches — is final.  The lowest unfinished lane is always ready.
        //
        // The rounds run in LDS.  Through the output in global memory every round read what the round before it had just stored, and on
        // this hardware a load's data cannot be waited for without waiting for the stores issued before it as well: a round trip to L2 per
        // 64 pieces, 185 rounds per member, 40 % of the kernel (profiles/r05_inflate_ticks.txt).  The code tables are free once the block's
        // LAST superstep has emitted its tokens (the usual case: a superstep covers the rest of the payload), so their 5.4 KB hold a
        // WINDOW of the output: out[wbase, wend) is loaded (the literals are in it; the matches' bytes are not yet), the batches whose
        // matches end inside it are resolved there — sources below wbase are final in the output and read from there — and when the
        // next batch reaches past wend the stretch the batches covered is stored to the output and the window starts again at that
        // batch: about four batches per window, so the wait for the stores and the load is paid once per four batches.
        // In a superstep that is not the block's last there is no window and everything goes through the output.
        uint8_t *const win = reinterpret_cast<uint8_t *>(T.ll);
        enum { WIN_CAP = (int)(sizeof(T.ll) + sizeof(T.d)) - 8 };            // (8-byte accesses may reach 7 bytes past what they need)
        const bool use_win = kend < 64;
        const uint32_t sup_end = o + total;                                 // the output of this superstep ends here; its literals are all written
        uint32_t wbase = 0xFFFFFFFFu, wend = 0, wb_lo = 0, whi = 0;         // window out[wbase, wend); [wb_lo, whi): resolved there, not yet in the output
        // 8 bytes of the output from `pos` on, of which the caller needs only bytes that are final
        auto fetch8 = [&](uint32_t pos) -> uint64_t {
            if (pos >= wbase) return ld64(win + (pos - wbase));
            if (pos + 8 <= wbase) return ld64(out + pos);
            const uint32_t k8 = 8 * (wbase - pos);                           // 1 .. 7 bytes from below the window, the rest from its start
            return (ld64(out + pos) & ((1ull << k8) - 1ull)) | (ld64(win) << k8);
        };
        // (both ways 16 bytes per lane and three steps in flight: a step at a time, every step waited for the one before — eleven
        // round trips to fill a window)
        auto write_back = [&]() {                                           // out[wb_lo, whi) = the window's bytes, exactly; wb_lo == wbase
            if (whi <= wb_lo) return;
            const uint32_t n_wb = whi - wb_lo;
            uint8_t *const to = out + wb_lo;
            W::each([&](int l) {
                for (uint32_t g = 16u * (uint32_t)l; g < n_wb; g += 3072) {
                    V16 v[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) v[k] = ld128(win + (g + 1024u * k < (uint32_t)WIN_CAP ? g + 1024u * k : 0u));
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const uint32_t i = g + 1024u * k;
                        if (i + 16 <= n_wb) st128(to + i, v[k]);
                        else if (i + 8 <= n_wb) { st64(to + i, v[k].a); if (i + 8 < n_wb) store_bytes(to + i + 8, v[k].b, n_wb - i - 8); }
                        else if (i < n_wb) store_bytes(to + i, v[k].a, n_wb - i);
                    }
                }
            });
            wb_lo = whi;
        };
        auto fill_window = [&]() {                                          // win[0, wend - wbase) = out[wbase, wend)
            const uint32_t n_new = wend - wbase;
            const uint8_t *const from = out + wbase;
            W::each([&](int l) {
                for (uint32_t g = 16u * (uint32_t)l; g < n_new; g += 3072) {
                    V16 v[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const uint32_t i = g + 1024u * k;
                        v[k].a = v[k].b = 0;
                        if (i + 8 < n_new) v[k] = ld128(from + i);         // (reads at most 7 bytes past the superstep's output: the buffer is padded by 8)
                        else if (i < n_new) v[k].a = ld64(from + i);
                    }
#pragma unroll
                    for (int k = 0; k < 3; ++k) { const uint32_t i = g + 1024u * k; if (i < n_new) st128(win + i, v[k]); }
                }
            });
        };
        W::each([&](int l) { const int lo = W::opaque(l); if (lo < 56) T.psel()[lo] = period_selector((uint32_t)(lo >> 3) + 1, (uint32_t)(lo & 7)); T.sorted[l] = 0; });   // (opaque: computed HERE, not kept in registers from the kernel's first instruction on)
        uint32_t gen = 0;                                                   // chunks of this superstep so far (mod 1024): the tag of own[]'s marks
        U nx_dst, nx_ld;                                                  // the next batch's tokens are fetched a batch ahead
        W::each([&](int l) { nx_dst[l] = nx_ld[l] = 0; if ((uint32_t)l < n_tok) { const Token k = tok[l]; nx_dst[l] = k.dst; nx_ld[l] = k.len_dist; } });
        for (uint32_t b0 = 0; b0 < n_tok;) {
            PW_MARK(50, b0);
            U dst, len, dist, dep_lo, dep_hi, valid;
            W::each([&](int l) {
                const uint32_t i = b0 + (uint32_t)l;
                valid[l] = i < n_tok;
                dst[l] = valid[l] ? nx_dst[l] : 0u; len[l] = valid[l] ? nx_ld[l] & 0xffff : 0u; dist[l] = valid[l] ? nx_ld[l] >> 16 : 0u;
            });
            const uint32_t d0 = W::bcast_u(dst, 0);
            uint32_t take = n_tok - b0 < 64u ? n_tok - b0 : 64u;            // matches of this batch
            if (use_win) {
                U end;
                W::each([&](int l) { end[l] = dst[l] + len[l]; });
                uint32_t e = W::bcast_u(end, take - 1);
                if (wbase == 0xFFFFFFFFu || e > wend) {                     // the batch reaches past the window: move it here
                    W::sync();
                    write_back();
                    wbase = wb_lo = d0;
                    wend = sup_end - d0 > (uint32_t)WIN_CAP ? d0 + (uint32_t)WIN_CAP : sup_end;
                    fill_window();
                    if (e > wend) {                                         // long matches: the first `take` of them fit (one always does)
                        U fits;
                        W::each([&](int l) { fits[l] = valid[l] && end[l] <= wend; });
                        take = (uint32_t)__builtin_ctzll(~W::ballot_ne(fits, 0u));
                        W::each([&](int l) { if ((uint32_t)l >= take) valid[l] = 0; });
                        e = W::bcast_u(end, take - 1);
                    }
                }
                whi = e;
            }
            W::each([&](int l) {
                T.bdst()[l] = valid[l] ? dst[l] : 0xFFFFFFFFu;
                T.bend()[l] = valid[l] ? dst[l] + len[l] : 0xFFFFFFFFu;
            });
            // (the next batch's tokens; a batch cut short leaves them to be fetched when it is done)
            if (take == 64) W::each([&](int l) { const uint32_t i = b0 + 64u + (uint32_t)l; if (i < n_tok) { const Token k = tok[i]; nx_dst[l] = k.dst; nx_ld[l] = k.len_dist; } });
            W::sync();
            W::each([&](int l) {
                dep_lo[l] = 1; dep_hi[l] = 0;                              // empty
                if (!valid[l] || l == 0) return;
                const uint32_t src = dst[l] - dist[l];
                const uint32_t send = src + (len[l] < dist[l] ? len[l] : dist[l]);
                if (send <= d0) return;                                   // the source ends before the batch's first match
                // two binary searches over [0, l), side by side (their LDS reads do not wait for each other), six halvings for l < 64:
                // lo1 = the first i with bend[i] > src (l if none), lo2 = the number of i with bdst[i] < send
                int lo1 = 0, hi1 = l, lo2 = 0, hi2 = l;
#pragma unroll
                for (int it = 0; it < 6; ++it) {
                    const int m1 = (lo1 + hi1) >> 1, m2 = (lo2 + hi2) >> 1;
                    const uint32_t e1 = T.bend()[m1], d2 = T.bdst()[m2];
                    if (lo1 < hi1) { if (e1 > src) hi1 = m1; else lo1 = m1 + 1; }
                    if (lo2 < hi2) { if (d2 < send) lo2 = m2 + 1; else hi2 = m2; }
                }
                dep_lo[l] = (uint32_t)lo1; dep_hi[l] = (uint32_t)lo2;     // matches [lo1, lo2) overlap the source
            });
            PW_TICK(8);
            // The ready matches of a round are copied in 8-byte PIECES dealt out over the whole wave — piece p of the round belongs to the
            // match whose first piece is the last one at or before p — so a round costs its bytes / 512 trips of one uniform step, not the
            // trips of its longest match through a divergent per-lane copy loop (the wave used to wait for 1 350 pieces per member that
            // way; dealt out, a member's 8 900 pieces are 140 trips + one per round).  A piece is a full 8-byte load and store (the last
            // piece of a match overlaps the one before it so that it ends where the match ends; matches shorter than 8 bytes are one
            // byte-exact piece); a piece of a self-overlapping match (dist < len) takes its bytes from the period in front of the match.
            const U packB = [&] { U x; W::each([&](int l) { x[l] = len[l] | (dist[l] << 16); }); return x; }();
            uint16_t *const own = T.sorted;                               // (free since the tables were built) 64 entries: a chunk's piece -> match marks
            uint64_t done = W::ballot_eq(valid, 0u);
            for (int round = 0; done != ~0ull; ++round) {
                if (round > 64) return -9;
                U ready, np;
                W::each([&](int l) {
                    ready[l] = 0; np[l] = 0;
                    if ((done >> l) & 1) return;
                    uint64_t dm = 0;
                    if (dep_lo[l] < dep_hi[l]) dm = (dep_hi[l] >= 64 ? ~0ull : ((1ull << dep_hi[l]) - 1)) & ~((1ull << dep_lo[l]) - 1);
                    if (dm & ~done) return;
                    ready[l] = 1; np[l] = (len[l] + 7) >> 3;
                    if (st) st->copy_iters += (len[l] + 31) / 32;
                });
                uint32_t P = 0;
                const U ps = W::excl_scan(np, &P);                        // a ready match's first piece
                U packA;
                W::each([&](int l) { packA[l] = dst[l] | (ps[l] << 16); });
                PW_TICK(9);
                uint32_t carry = 0;                                       // (1 + match) of the piece in front of the chunk
                for (uint32_t c0 = 0; c0 < P; c0 += 64) {
                    // (a mark carries the number of its chunk, so that the marks of earlier chunks need not be cleared: one LDS trip less)
                    if (++gen == 1024u) { W::each([&](int l) { own[l] = 0; }); W::sync(); gen = 1; }
                    W::each([&](int l) { if (ready[l] && ps[l] - c0 < 64u) own[ps[l] - c0] = (uint16_t)((gen << 6) | (uint32_t)l); });
                    W::sync();
                    U id;
                    W::each([&](int l) { const uint32_t m = own[l]; id[l] = (m >> 6) == gen ? (m & 63u) + 1u : 0u; });
                    id = W::incl_scan_max(id);
                    W::each([&](int l) { if (id[l] < carry) id[l] = carry; });
                    carry = W::bcast_u(id, 63);
                    U src_lane;
                    W::each([&](int l) { src_lane[l] = id[l] ? id[l] - 1 : 0u; });
                    const U a = W::shfl(packA, src_lane), b = W::shfl(packB, src_lane);
                    W::each([&](int l) {
                        const uint32_t p = c0 + (uint32_t)l;
                        if (p >= P) return;
                        const uint32_t dstm = a[l] & 0xffff, lenm = b[l] & 0xffff, distm = b[l] >> 16;
                        uint32_t off = (p - (a[l] >> 16)) * 8;
                        if (lenm >= 8 && off > lenm - 8) off = lenm - 8;
                        const uint32_t sp = dstm - distm;                  // where the match's bytes (its period, if it overlaps itself) begin
                        uint64_t v;
                        if (distm >= lenm) v = fetch8(sp + off);
                        else {
                            // byte j of a self-overlapping match is byte (j mod dist) of the `dist` bytes in front of it
                            const uint32_t om = small_mod(off, distm);
                            if (distm >= 8) {
                                v = fetch8(sp + om);
                                if (om + 8 > distm) { const uint32_t k1 = distm - om; v = (v & ((1ull << (8 * k1)) - 1ull)) | (fetch8(sp) << (8 * k1)); }
                            } else v = gather8(fetch8(sp), T.psel()[(distm - 1) * 8 + om]);
                        }
                        if (use_win) { if (lenm >= 8) st64(win + (dstm - wbase) + off, v); else store_bytes(win + (dstm - wbase), v, lenm); }
                        else { if (lenm >= 8) st64(out + dstm + off, v); else store_bytes(out + dstm, v, lenm); }
                    });
                    if (st) st->copy_serial += 1;
                    W::sync();
                }
                if (st) W::each([&](int l) { if (ready[l] && len[l] > 16) st->long_matches++; });
                PW_TICK(10);
                W::fence();
                done |= W::ballot_ne(ready, 0u);
                PW_TICK(11);
                if (st) st->emit_rounds++;
            }
            if (st) st->batches++;
            b0 += take;
            if (take != 64 && b0 < n_tok)
                W::each([&](int l) { const uint32_t i = b0 + (uint32_t)l; if (i < n_tok) { const Token k = tok[i]; nx_dst[l] = k.dst; nx_ld[l] = k.len_dist; } });
            W::sync();
        }
        if (use_win) { W::sync(); write_back(); W::sync(); }
        PW_TICK(4);
        o += total;
        if (kend < 64) { q_io = W::bcast(e, (int)kend); o_io = o; return 0; }
        base = W::bcast(e, 63);
    }
    return -9;
}

// One whole BGZF member payload (raw DEFLATE) of exactly out_len bytes.  Returns 0, PD_W_HOST or a negative error.
// `in` must be readable for 8 bytes past in_len (a BGZF payload is followed by its 8-byte trailer).
template <class W>
PW_FN int inflate_block(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_len, Tables &T, Token *tok, Stats *st)
{
    // the order of the code-length code's lengths in the header: 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 — five bits each in two
    // constants (a table in memory cost the kernel an address register for its whole life)
    auto clord = [](int l) -> uint32_t {
        const uint64_t lo = 16ull | 17ull << 5 | 18ull << 10 | 0ull << 15 | 8ull << 20 | 7ull << 25 | 9ull << 30 | 6ull << 35 | 10ull << 40 | 5ull << 45 | 11ull << 50 | 4ull << 55;
        const uint64_t hi = 12ull | 3ull << 5 | 13ull << 10 | 2ull << 15 | 14ull << 20 | 1ull << 25 | 15ull << 30;
        return (uint32_t)((l < 12 ? lo >> (5 * l) : hi >> (5 * (l - 12))) & 31u);
    };
    typedef typename W::template Var<uint32_t> U;
    if (in_len > (1u << 17) || out_len > (1u << 16)) return PD_W_HOST;        // not a BGZF member: 16-bit token fields
    const uint32_t in_bits = in_len * 8;
    uint32_t q = 0, o = 0;
    if (st) st->blocks++;
    for (int guard = 0; guard < 70000; ++guard) {
        if (q + 3 > in_bits) return -7;
        const uint32_t hdr = peek_bits(in, q) & 7; q += 3;
        const uint32_t last = hdr & 1, type = hdr >> 1;
        PW_MARK(10 + type, q);
        PW_TICK(5);
        if (st) st->dblocks++;
        if (type == 0) {                                                  // stored: the wave copies the bytes
            q = (q + 7) & ~7u;
            if (q + 32 > in_bits) return -7;
            const uint32_t w = peek_bits(in, q); q += 32;
            const uint32_t len = w & 0xffff, nlen = w >> 16;
            if ((len ^ 0xffff) != nlen) return -2;
            if (len > out_len - o) return -5;
            if ((uint64_t)q + (uint64_t)len * 8 > in_bits) return -7;
            const uint8_t *src = in + (q >> 3);
            uint8_t *dst = out + o;
            W::each([&](int l) { for (uint32_t i = l; i < len; i += 64) dst[i] = src[i]; });
            W::sync();
            q += len * 8; o += len;
        } else if (type == 3) return -1;
        else {
            uint32_t nlen = 288, ndist = 32;
            if (type == 1) {
                W::each([&](int l) {
                    for (int s = l; s < 320; s += 64) T.cl[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : s < 288 ? 8 : 5);
                });
                W::sync();
            } else {
                if (q + 14 > in_bits) return -7;
                const uint32_t h = peek_bits(in, q); q += 14;
                nlen = (h & 31) + 257; ndist = ((h >> 5) & 31) + 1;
                const uint32_t ncode = ((h >> 10) & 15) + 4;
                if (nlen > 286 || ndist > 30) return -3;
                if (q + ncode * 3 > in_bits) return -7;
                W::each([&](int l) { if (l < 19) T.cl[l] = 0; });
                W::sync();
                {
                    const uint64_t w64 = ld64(in + (q >> 3)) >> (q & 7);   // 19 x 3 = 57 bits
                    W::each([&](int l) { if ((uint32_t)l < ncode) T.cl[clord(W::opaque(l))] = (uint8_t)((w64 >> (3 * l)) & 7); });
                    q += ncode * 3;
                }
                W::sync();
                PW_MARK(15, ncode);
                int rc = build_table<W, 7, 0, 2>(T.d, T.cl, 19, T.sorted, T.rank());
                PW_MARK(16, (uint32_t)rc);
                PW_TICK(1);       // borrows the distance table's space
                if (rc) return rc < 0 ? -3 : rc;
                // the code lengths themselves: a short serial stream (every lane runs it, lane l stores).  Its bytes are
                // fetched ONCE: lane l holds the 8 bytes at l * 8 of a 512-byte window, and the bit buffer is fed by
                // readlane (a register access) instead of a dependent memory load per symbol
                // The code lengths themselves: a stream of symbols of 1 .. 14 bits, decoded 64 bit positions at a time.  Lane l decodes the
                // symbol that WOULD start at bit q + l (one table look-up; 63 of 64 lanes guess wrong and nobody minds), the wave follows
                // the chain of true starts through the 64 positions with readlane (a register access per symbol — the loop that decoded one
                // symbol per trip, every lane alike, took 860 cycles per symbol: a sixth of a member's time for 360 symbols), and the true
                // symbols then store their lengths together: places from a prefix sum of the repeat counts, "repeat the previous length"
                // resolved by a prefix maximum.  The array starts out zero, so runs of zeros (symbols 17 and 18) store nothing.
                // The stream's bytes are fetched ONCE: lane l holds the 8 bytes at l * 8 of a 512-byte window.
                W::each([&](int l) { for (int s2 = l; s2 < 320; s2 += 64) T.cl[s2] = 0; });   // (the code-length code's own lengths are in its table now)
                uint32_t idx = 0, prev = 0;
                const uint32_t want = W::uni(nlen + ndist);
                uint32_t wbase = W::uni(q >> 3);                           // byte position of the window in `in`
                q = W::uni(q);
                typename W::template Var<uint64_t> hw;
                W::each([&](int l) { const uint32_t at = wbase + 8u * (uint32_t)l; hw[l] = at <= in_len ? ld64(in + at) : 0ull; });
                W::sync();
                while (idx < want) {
                    uint32_t r = q - 8 * wbase;                            // bit offset inside the window
                    if (r >= 61 * 64) {                                    // (a header longer than the window: move it)
                        wbase = q >> 3;
                        W::each([&](int l) { const uint32_t at = wbase + 8u * (uint32_t)l; hw[l] = at <= in_len ? ld64(in + at) : 0ull; });
                        r = q - 8 * wbase;
                    }
                    const uint32_t k = r >> 6, sh = r & 63;
                    const uint64_t a = W::bcast64(hw, (int)k), b = W::bcast64(hw, (int)k + 1), c = W::bcast64(hw, (int)k + 2);
                    const uint64_t lo = sh ? (a >> sh) | (b << (64 - sh)) : a, hi = sh ? (b >> sh) | (c << (64 - sh)) : b;   // 128 bits from q
                    U nxt, rep, val, bad;
                    W::each([&](int l) {
                        const uint32_t w = (uint32_t)(l ? (lo >> l) | (hi << (64 - l)) : lo);
                        const uint32_t e = T.d[w & 127];
                        const uint32_t n = e & 15, sym = (e >> 8) & 0xff;
                        const uint32_t x = w >> n;
                        const uint32_t ext = sym == 16 ? 2u : sym == 17 ? 3u : sym == 18 ? 7u : 0u;
                        const uint32_t xv = x & ((1u << ext) - 1u);
                        bad[l] = !n || ((e >> 4) & 3) != KIND_LIT;
                        rep[l] = sym < 16 ? 1u : sym == 18 ? 11u + xv : 3u + xv;
                        val[l] = sym;                                      // 16: the previous length; 17, 18: zeros
                        nxt[l] = bad[l] ? 255u : (uint32_t)l + n + ext;
                    });
                    uint64_t starts = 0;
                    uint32_t pos = 0;
                    while (pos < 64) { starts |= 1ull << pos; pos = W::bcast_u(nxt, pos); }
                    // the symbols of this stretch that are still wanted, with their places; the first one at fault decides
                    U cnt, err, used;
                    W::each([&](int l) { cnt[l] = (starts >> l) & 1 ? rep[l] : 0u; });
                    uint32_t tot = 0;
                    const U ex = W::excl_scan(cnt, &tot);
                    W::each([&](int l) {
                        const uint32_t at = idx + ex[l];
                        used[l] = ((starts >> l) & 1) && at < want;
                        err[l] = 0;
                        if (!used[l]) return;
                        if (q + (uint32_t)l >= in_bits) err[l] = 7;
                        else if (bad[l]) err[l] = 4;
                        else if ((val[l] == 16 && at == 0) || at + rep[l] > want) err[l] = 3;
                    });
                    const uint64_t em = W::ballot_ne(err, 0u);
                    if (em) return -(int)W::bcast_u(err, (uint32_t)__builtin_ctzll(em));
                    if (st) st->hdr_syms += (uint64_t)__builtin_popcountll(W::ballot_ne(used, 0u));
                    // "the previous length" = the length of the nearest symbol before that is not itself a repeat (across stretches: prev)
                    U key;
                    W::each([&](int l) { key[l] = used[l] && val[l] != 16 ? ((uint32_t)(l + 1) << 8) | (val[l] < 16 ? val[l] : 0u) : 0u; });
                    key = W::incl_scan_max(key);
                    W::each([&](int l) {
                        if (!used[l]) return;
                        const uint32_t v = val[l] < 16 ? val[l] : val[l] == 16 ? (key[l] ? key[l] & 0xffu : prev) : 0u;
                        if (!v) return;
                        const uint32_t at = idx + ex[l];
                        for (uint32_t j = 0; j < rep[l]; ++j) T.cl[at + j] = (uint8_t)v;   // rep <= 6 here
                    });
                    const uint32_t k63 = W::bcast_u(key, 63);
                    if (k63) prev = k63 & 0xffu;
                    const uint64_t um = W::ballot_ne(used, 0u);
                    const uint64_t over = starts & ~um;                    // true starts behind the last wanted symbol
                    if (idx + tot >= want) {                                // the stream ends inside this stretch
                        q += over ? (uint32_t)__builtin_ctzll(over) : pos;
                        idx = want;
                    } else { idx += tot; q += pos; }
                }
                if (q > in_bits) return -7;
                W::sync();
                if (W::uniform_u8(&T.cl[256]) == 0) return -3;             // no end-of-block code
            }
            PW_MARK(20, nlen);
            PW_TICK(0);
            int rc = build_table<W, LL_ROOT, LL_SUBCAP, 0>(T.ll, T.cl, (int)nlen, T.sorted, T.rank());
            PW_MARK(21, (uint32_t)rc);
            if (rc) return rc;
            rc = build_table<W, D_ROOT, D_SUBCAP, 1>(T.d, T.cl + nlen, (int)ndist, T.sorted, T.rank());
            if (rc) return rc;
            PW_MARK(22, (uint32_t)rc);
            PW_TICK(1);
            rc = decode_body<W>(in, in_bits, q, out, out_len, o, T, tok, st);
            PW_MARK(23, (uint32_t)rc);
            if (rc) return rc;
            PW_MARK(24, last);
        }
        PW_MARK(25, last);
        if (last) break;
        PW_MARK(26, q);
    }
    PW_MARK(27, o);
    return o == out_len ? 0 : -5;
}

// ---- CRC-32 of a member's inflated bytes (RFC 1952; htslib checks it for every BGZF block it reads, so a flipped bit that
// still inflates must not go unnoticed here either).  One wave; the output is cut into 1 KiB chunks of which the FIRST is the short
// one, and the chunks sit in the LAST lanes (chunk L - 1 in lane 63), so that lane l's bytes are followed by exactly 1024 (63 - l)
// more.  The register update is linear over GF(2) — R(s, data) = R(s, zeros) ^ R(0, data), and running over n zero bytes is the
// multiplication by x^(8n) modulo the CRC polynomial — so every lane runs its chunk through the tables from a zero register (the
// first chunk from the start value 0xFFFFFFFF), and the wave adds the 64 results up as the polynomial sum(part_l X^(63 - l)),
// X = x^8192: six pairing steps, each a multiplication by a CONSTANT (X, X^2, X^4 ...).  (Until round 5 the last chunk was the
// short one: every lane then multiplied by its own power of x — seventeen squarings and as many products, 2 900 instructions —
// which cost more than the bytes.)
static const uint32_t CRC_POLY = 0xEDB88320u;                            // reflected: bit 31 is the coefficient of x^0
constexpr uint32_t crc_mul(uint32_t a, uint32_t b)                        // a(x) b(x) mod P(x)
{
    uint32_t p = 0;
    for (int k = 0; k < 32; ++k) {
        p ^= b & (0u - ((a >> (31 - k)) & 1u));
        b = (b >> 1) ^ (0xEDB88320u & (0u - (b & 1u)));
    }
    return p;
}
constexpr uint32_t crc_x8192_pow(int s)                                   // (x^8192)^(2^s) mod P(x)
{
    uint32_t p = 0x00800000u;                                             // x^8
    for (int j = 0; j < 10 + s; ++j) p = crc_mul(p, p);
    return p;
}
template <uint32_t B>
PW_FN uint32_t crc_mul_by(uint32_t a)                                     // a(x) B(x) mod P(x): B's 32 shifts are literals
{
    uint32_t p = 0, b = B;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        p ^= b & (0u - ((a >> (31 - k)) & 1u));
        b = (b >> 1) ^ (CRC_POLY & (0u - (b & 1u)));
    }
    return p;
}
template <class W, int S>
PW_FN void crc_pair_step(typename W::template Var<uint32_t> &part)
{
    typedef typename W::template Var<uint32_t> U;
    U t, from;
    W::each([&](int l) { t[l] = crc_mul_by<crc_x8192_pow(S)>(part[l]); from[l] = (uint32_t)(W::opaque(l) - (1 << S)) & 63u; });   // (opaque: six index registers were kept from the kernel's first instruction to here)
    const U got = W::shfl(t, from);
    W::each([&](int l) { if ((l & ((2 << S) - 1)) == (2 << S) - 1) part[l] ^= got[l]; });
}
template <class W>
PW_FN uint32_t crc32_wave(const uint8_t *data, uint32_t n, uint32_t *tab /* 1024 words of LDS */)
{
    typedef typename W::template Var<uint32_t> U;
    if (n == 0) return 0;
    // Four tables ("slicing by 4"): tab[256 k + i] = the register after byte i followed by k zero bytes, so that four bytes are one step —
    // four INDEPENDENT look-ups instead of four that wait for each other
    W::each([&](int l0) {
        const int l = W::opaque(l0);                                          // (opaque: the entries and their address are computed HERE)
        for (int k = 0; k < 4; ++k) {
            uint32_t c = (uint32_t)(l * 4 + k);
            for (int b = 0; b < 8; ++b) c = (c >> 1) ^ (CRC_POLY & (0u - (c & 1u)));
            tab[l * 4 + k] = c;
        }
    });
    W::sync();
    for (int t = 1; t < 4; ++t) {
        W::each([&](int l) {
            for (int k = 0; k < 4; ++k) { const uint32_t c = tab[256 * (t - 1) + l * 4 + k]; tab[256 * t + l * 4 + k] = (c >> 8) ^ tab[c & 0xffu]; }
        });
        W::sync();
    }
    const uint32_t L = (n + 1023u) >> 10;                                 // chunks (<= 64); the first has n - 1024 (L - 1) bytes
    const uint32_t first = n - 1024u * (L - 1u);
    U part;
    W::each([&](int l) {
        part[l] = 0;
        const int c = l - (int)(64u - L);
        if (c < 0) return;
        const uint32_t a = c == 0 ? 0u : first + 1024u * (uint32_t)(c - 1), b = c == 0 ? first : a + 1024u;
        uint32_t r = c == 0 ? 0xFFFFFFFFu : 0u, i = a;
        auto step4 = [&](uint32_t w) {
            const uint32_t x = r ^ w;
            r = tab[768 + (x & 0xffu)] ^ tab[512 + ((x >> 8) & 0xffu)] ^ tab[256 + ((x >> 16) & 0xffu)] ^ tab[x >> 24];
        };
        // PD_CRC_STEP (64) bytes a step, the NEXT step's bytes on their way while this one's go through the tables (a step that waited for its own 16 bytes
        // was a round trip to memory per 16 bytes: 64 of them per lane, most of the CRC's time)
        enum { CS = PD_CRC_STEP, CW = PD_CRC_STEP / 4 };
        uint32_t cur[CW], nxt[CW];
        if (i + CS <= b) __builtin_memcpy(cur, data + i, CS);
        for (; i + CS <= b; i += CS) {
            // (always a load, from a clamped address when nothing follows: a load under a condition makes the compiler wait for it at once)
            __builtin_memcpy(nxt, data + (i + 2 * CS <= b ? i + CS : b - CS), CS);
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_sched_barrier(0);                             // (the loads stay HERE: sunk to where their data is used they are no prefetch)
#endif
#pragma unroll
            for (int k = 0; k < CW; ++k) step4(cur[k]);
#pragma unroll
            for (int k = 0; k < CW; ++k) cur[k] = nxt[k];
        }
        for (; i + 16 <= b; i += 16) {
            uint32_t w[4]; __builtin_memcpy(w, data + i, 16);
#pragma unroll
            for (int k = 0; k < 4; ++k) step4(w[k]);
        }
        for (; i < b; ++i) r = tab[(r ^ data[i]) & 0xffu] ^ (r >> 8);
        part[l] = r;
    });
    crc_pair_step<W, 0>(part); crc_pair_step<W, 1>(part); crc_pair_step<W, 2>(part);
    crc_pair_step<W, 3>(part); crc_pair_step<W, 4>(part); crc_pair_step<W, 5>(part);
    const uint32_t body = W::bcast_u(part, 63);
    W::sync();                                                            // (the tables' LDS is the caller's again)
    return body ^ 0xFFFFFFFFu;
}

// A whole BGZF member: inflate, then the CRC-32 of the output against the member's trailer (the 4 bytes behind the payload).
// Returns 0, PD_W_HOST, a negative inflate error, or -20: the bytes inflate but are not the bytes that were compressed.
template <class W>
PW_FN int inflate_member(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_len, Tables &T, Token *tok, Stats *st, bool check_crc = true)
{
    if (out_len) {                                                        // (an empty member — the EOF marker — has nothing to inflate)
        const int rc = inflate_block<W>(in, in_len, out, out_len, T, tok, st);
        if (rc) return rc;
    }
    if (!check_crc) return 0;
    W::fence();
    PW_TICK(6);
    uint32_t want; __builtin_memcpy(&want, in + in_len, 4);
    static_assert(offsetof(Tables, work) == offsetof(Tables, sorted) + sizeof(T.sorted) && offsetof(Tables, cl) == offsetof(Tables, work) + sizeof(T.work) &&
                  offsetof(Tables, ck_more) == offsetof(Tables, cl) + sizeof(T.cl) &&
                  sizeof(T.sorted) + sizeof(T.work) + sizeof(T.cl) + sizeof(T.ck_more) >= CK_N * CK_STRIDE * 4 && sizeof(Tables) <= 7680, "phase 1's checkpoints lie over sorted | work | cl | ck_more");
    static_assert(sizeof(T.ll) + sizeof(T.d) >= 4096 && offsetof(Tables, d) == sizeof(T.ll), "the CRC tables take the first 4 KiB of the (now free) code tables");
    return crc32_wave<W>(out, out_len, T.ll) == want ? 0 : -20;
}

// ---- the two wave implementations -------------------------------------------------------------------------------
struct HostWave {                         // 64 emulated lanes
    template <class T> struct Var { T v[64]; T &operator[](int l) { return v[l]; } const T &operator[](int l) const { return v[l]; } };
    template <class F> static void each(F f) { for (int l = 0; l < 64; ++l) f(l); }
    static void sync() {}
    static void fence() {}
    static uint64_t ballot_eq(const Var<uint32_t> &x, uint32_t v) { uint64_t m = 0; for (int l = 0; l < 64; ++l) if (x.v[l] == v) m |= 1ull << l; return m; }
    static uint64_t ballot_ne(const Var<uint32_t> &x, uint32_t v) { uint64_t m = 0; for (int l = 0; l < 64; ++l) if (x.v[l] != v) m |= 1ull << l; return m; }
    static uint32_t prefix_count(uint64_t m, int l) { return (uint32_t)__builtin_popcountll(m & ((1ull << l) - 1)); }
    static Var<uint32_t> excl_scan(const Var<uint32_t> &x, uint32_t *total) { Var<uint32_t> r; uint32_t a = 0; for (int l = 0; l < 64; ++l) { r.v[l] = a; a += x.v[l]; } *total = a; return r; }
    static Var<uint32_t> shift_up1(const Var<uint32_t> &x, uint32_t fill) { Var<uint32_t> r; r.v[0] = fill; for (int l = 1; l < 64; ++l) r.v[l] = x.v[l - 1]; return r; }
    static uint32_t bcast(const Var<uint32_t> &x, int lane) { return x.v[lane]; }
    static Var<uint32_t> incl_scan_max(const Var<uint32_t> &x) { Var<uint32_t> r; uint32_t a = 0; for (int l = 0; l < 64; ++l) { if (x.v[l] > a) a = x.v[l]; r.v[l] = a; } return r; }
    static Var<uint32_t> shfl(const Var<uint32_t> &x, const Var<uint32_t> &from) { Var<uint32_t> r; for (int l = 0; l < 64; ++l) r.v[l] = x.v[from.v[l] & 63]; return r; }
    static uint64_t bcast64(const Var<uint64_t> &x, int lane) { return x.v[lane]; }
    static uint32_t uni(uint32_t x) { return x; }                                               // a value every lane holds alike
    static int opaque(int l) { return l; }                                                      // the lane number, in a form the GPU compiler does not hoist computations on out of their loops
    static uint32_t bcast_u(const Var<uint32_t> &x, uint32_t lane) { return x.v[lane & 63]; }      // lane must be wave-uniform
    static uint32_t min_where(const Var<uint32_t> &x, const Var<uint32_t> &skip) { uint32_t m = 0xFFFFFFFFu; for (int l = 0; l < 64; ++l) if (!skip.v[l] && x.v[l] < m) m = x.v[l]; return m; }
    static uint32_t uniform_u8(const uint8_t *p) { return *p; }
    static Var<uint64_t> excl_scan_max64(const Var<uint64_t> &x) { Var<uint64_t> r; uint64_t a = 0; for (int l = 0; l < 64; ++l) { r.v[l] = a; if (x.v[l] > a) a = x.v[l]; } return r; }
    // segmented exclusive prefix maximum: head1[l] = 1 + the lane where l's segment begins (0: it began before lane 0); lane l gets the maximum over its segment's lanes before it (0: none)
    static Var<uint64_t> seg_excl_scan_max64(const Var<uint64_t> &x, const Var<uint32_t> &head1)
    {
        Var<uint64_t> r;
        for (int l = 0; l < 64; ++l) { const int h = head1.v[l] ? (int)head1.v[l] - 1 : 0; uint64_t a = 0; for (int i = h; i < l; ++i) if (x.v[i] > a) a = x.v[i]; r.v[l] = a; }
        return r;
    }
    static uint32_t reduce_or(const Var<uint32_t> &x) { uint32_t a = 0; for (int l = 0; l < 64; ++l) a |= x.v[l]; return a; }
    static uint32_t reduce_xor(const Var<uint32_t> &x) { uint32_t a = 0; for (int l = 0; l < 64; ++l) a ^= x.v[l]; return a; }
    static uint32_t reduce_max(const Var<uint32_t> &x) { uint32_t a = 0; for (int l = 0; l < 64; ++l) if (x.v[l] > a) a = x.v[l]; return a; }
    static uint64_t reduce_max64(const Var<uint64_t> &x) { uint64_t a = 0; for (int l = 0; l < 64; ++l) if (x.v[l] > a) a = x.v[l]; return a; }
    static uint64_t reduce_min64(const Var<uint64_t> &x) { uint64_t a = ~0ull; for (int l = 0; l < 64; ++l) if (x.v[l] < a) a = x.v[l]; return a; }
};

#if defined(__HIPCC__)
struct DevWave {                          // the hardware wavefront (one wave per workgroup)
    template <class T> struct Var { T v; __device__ T &operator[](int) { return v; } __device__ const T &operator[](int) const { return v; } };
    // (the lane number through an empty asm: what a lambda computes from it is computed where the lambda stands — the compiler otherwise hoists every
    // lane-only expression of the whole kernel (16 * lane, 3 * lane, (lane - 8) & 63 ...) to its first instructions and keeps them in registers to its last;
    // with 96 registers for 5 waves per SIMD that meant spills, and a kernel with scratch costs 7 ms per hardware queue at its first launch)
#if PD_EACH_OPAQUE
    template <class F> __device__ static __forceinline__ void each(F f) { int l = (int)(threadIdx.x & 63); asm volatile("" : "+v"(l)); f(l); }
#else
    template <class F> __device__ static __forceinline__ void each(F f) { f((int)(threadIdx.x & 63)); }
#endif
    // One wave IS the workgroup: its LDS (and global) accesses execute in program order, so what one lane wrote another lane's later
    // read sees — nothing to wait for, the compiler only must not move accesses across (__syncthreads() also waits for every global
    // store the wave has in flight)
    __device__ static __forceinline__ void sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }
    __device__ static __forceinline__ void fence() { asm volatile("" ::: "memory"); }   // compiler-only: the wave issues in order
    __device__ static __forceinline__ uint64_t ballot_eq(const Var<uint32_t> &x, uint32_t v) { return __ballot(x.v == v); }
    __device__ static __forceinline__ uint64_t ballot_ne(const Var<uint32_t> &x, uint32_t v) { return __ballot(x.v != v); }
    __device__ static __forceinline__ uint32_t prefix_count(uint64_t m, int) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
    __device__ static __forceinline__ Var<uint32_t> excl_scan(const Var<uint32_t> &x, uint32_t *total)
    {
        int s = (int)x.v;
        s += __builtin_amdgcn_update_dpp(0, s, 0x111, 0xf, 0xf, false);
        s += __builtin_amdgcn_update_dpp(0, s, 0x112, 0xf, 0xf, false);
        s += __builtin_amdgcn_update_dpp(0, s, 0x114, 0xf, 0xf, false);
        s += __builtin_amdgcn_update_dpp(0, s, 0x118, 0xf, 0xf, false);
        s += __builtin_amdgcn_update_dpp(0, s, 0x142, 0xa, 0xf, false);
        s += __builtin_amdgcn_update_dpp(0, s, 0x143, 0xc, 0xf, false);
        *total = (uint32_t)__builtin_amdgcn_readlane(s, 63);
        Var<uint32_t> r; r.v = (uint32_t)s - x.v; return r;
    }
    __device__ static __forceinline__ Var<uint32_t> shift_up1(const Var<uint32_t> &x, uint32_t fill)
    {
        // (wave_shr:1 — lane 0 keeps `old`, which is the fill value; no index register, no LDS crossbar)
        Var<uint32_t> r; r.v = (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)x.v, 0x138, 0xf, 0xf, false); return r;
    }
    __device__ static __forceinline__ uint32_t bcast(const Var<uint32_t> &x, int lane) { return (uint32_t)__shfl((int)x.v, lane); }
    __device__ static __forceinline__ Var<uint32_t> incl_scan_max(const Var<uint32_t> &x)
    {
        // (unsigned maximum with the row-shift / row-broadcast pattern of the prefix sum; lanes that receive nothing keep their value: 0 is the identity)
        uint32_t m = x.v;
        auto mx = [](uint32_t p, uint32_t q) { return p > q ? p : q; };
        m = mx(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x111, 0xf, 0xf, false));
        m = mx(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x112, 0xf, 0xf, false));
        m = mx(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x114, 0xf, 0xf, false));
        m = mx(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x118, 0xf, 0xf, false));
        m = mx(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x142, 0xa, 0xf, false));
        m = mx(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x143, 0xc, 0xf, false));
        Var<uint32_t> r; r.v = m; return r;
    }
    __device__ static __forceinline__ Var<uint32_t> shfl(const Var<uint32_t> &x, const Var<uint32_t> &from) { Var<uint32_t> r; r.v = (uint32_t)__shfl((int)x.v, (int)(from.v & 63u)); return r; }
    __device__ static __forceinline__ uint64_t bcast64(const Var<uint64_t> &x, int lane)        // lane must be wave-uniform
    {
        const int sl = __builtin_amdgcn_readfirstlane(lane);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x.v, sl), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x.v >> 32), sl);
        return ((uint64_t)hi << 32) | lo;
    }
    __device__ static __forceinline__ int opaque(int l) { asm volatile("" : "+v"(l)); return l; }
    __device__ static __forceinline__ uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }   // (tells the compiler: scalar)
    __device__ static __forceinline__ uint32_t bcast_u(const Var<uint32_t> &x, uint32_t lane)
    {
        return (uint32_t)__builtin_amdgcn_readlane((int)x.v, __builtin_amdgcn_readfirstlane((int)lane));
    }
    __device__ static __forceinline__ uint32_t min_where(const Var<uint32_t> &x, const Var<uint32_t> &skip)
    {
        uint32_t m = skip.v ? 0xFFFFFFFFu : x.v;
#pragma unroll
        for (int o = 32; o; o >>= 1) { const uint32_t y = (uint32_t)__shfl_xor((int)m, o); m = y < m ? y : m; }
        return m;
    }
    __device__ static __forceinline__ uint32_t uniform_u8(const uint8_t *p) { return *p; }
    __device__ static __forceinline__ uint64_t shfl_up64(uint64_t v, int d)
    {
        const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)v, d), hi = (uint32_t)__shfl_up((int)(uint32_t)(v >> 32), d);
        return ((uint64_t)hi << 32) | lo;
    }
    __device__ static __forceinline__ uint64_t shfl_xor64(uint64_t v, int d)
    {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, d), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), d);
        return ((uint64_t)hi << 32) | lo;
    }
    __device__ static __forceinline__ Var<uint64_t> excl_scan_max64(const Var<uint64_t> &x)
    {
        const int lane = (int)(threadIdx.x & 63);
        uint64_t m = x.v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint64_t y = shfl_up64(m, d); if (lane >= d && y > m) m = y; }
        const uint64_t up = shfl_up64(m, 1);
        Var<uint64_t> r; r.v = lane ? up : 0ull; return r;
    }
    __device__ static __forceinline__ Var<uint64_t> seg_excl_scan_max64(const Var<uint64_t> &x, const Var<uint32_t> &head1)
    {
        // Hillis-Steele with the segment's first lane as the fence: a partner d lanes down counts only if it lies in the same segment
        const int lane = (int)(threadIdx.x & 63), h = head1.v ? (int)head1.v - 1 : 0;
        uint64_t m = x.v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint64_t y = shfl_up64(m, d); if (lane - d >= h && y > m) m = y; }
        const uint64_t up = shfl_up64(m, 1);
        Var<uint64_t> r; r.v = lane > h ? up : 0ull; return r;
    }
    __device__ static __forceinline__ uint32_t reduce_or(const Var<uint32_t> &x)
    {
        uint32_t m = x.v;
#pragma unroll
        for (int o = 32; o; o >>= 1) m |= (uint32_t)__shfl_xor((int)m, o);
        return m;
    }
    __device__ static __forceinline__ uint32_t reduce_xor(const Var<uint32_t> &x)
    {
        uint32_t m = x.v;
#pragma unroll
        for (int o = 32; o; o >>= 1) m ^= (uint32_t)__shfl_xor((int)m, o);
        return m;
    }
    __device__ static __forceinline__ uint32_t reduce_max(const Var<uint32_t> &x)
    {
        uint32_t m = x.v;
#pragma unroll
        for (int o = 32; o; o >>= 1) { const uint32_t y = (uint32_t)__shfl_xor((int)m, o); m = y > m ? y : m; }
        return m;
    }
    __device__ static __forceinline__ uint64_t reduce_max64(const Var<uint64_t> &x)
    {
        uint64_t m = x.v;
#pragma unroll
        for (int o = 32; o; o >>= 1) { const uint64_t y = shfl_xor64(m, o); m = y > m ? y : m; }
        return m;
    }
    __device__ static __forceinline__ uint64_t reduce_min64(const Var<uint64_t> &x)
    {
        uint64_t m = x.v;
#pragma unroll
        for (int o = 32; o; o >>= 1) { const uint64_t y = shfl_xor64(m, o); m = y < m ? y : m; }
        return m;
    }
};
#endif

} // namespace pdw
#endif
