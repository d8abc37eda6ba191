// pd_lz77.h — zlib's level-6 LZ77 parse (deflate.c: deflate_slow + longest_match; good_length 8, max_lazy 16, nice_length 128,
// max_chain 128, hash of 3 bytes in 15 bits, MAX_DIST 32506, TOO_FAR 4096), re-stated so that ONE WAVE evaluates the candidates
// of a position at once instead of chasing a hash chain.
//
// What makes that possible: deflate_slow inserts EVERY position into its hash chains as it passes it (inside matches too), so
// at the moment position p is searched its chain is "all earlier positions with p's hash, most recent first" — a function of the
// text alone, not of the parse.  With the positions sorted by (hash, position) (S[], and R[p] = where p stands in S) the chain of p
// is S[R[p] - 1], S[R[p] - 2], ... down to the start of its hash's bucket: a wave loads 64 candidates with one access, every lane
// measures its candidate's match length, and the wave takes what zlib's loop would have taken:
//   * candidates are valid while they are nearer than MAX_DIST (the first one: <= MAX_DIST) and not position 0 (zlib's NIL);
//   * at most `budget` of them are looked at (128; 32 once the previous match reached good_length);
//   * the loop stops behind the first candidate of nice_length or more; the result is the FIRST candidate with the greatest
//     length among those looked at, if that is longer than the previous match (strictly: zlib updates on len > best_len);
//   * a 3-byte match further than TOO_FAR away is dropped.
// The lazy-evaluation state machine around it (match_available, prev_length, the emission of the previous match when the
// current one is not longer) is sequential and tiny; it runs wave-uniform.
//
// A CHUNK is parsed as zlib parses it when primed with deflateSetDictionary: `dict` bytes before `start` are history only
// (their positions are candidates), the parse starts fresh at `start` and covers [start, end).  The text's buffer carries at least
// 32 bytes behind its end (reads of whole words).  The bytes of the last
// MIN_LOOKAHEAD of a chunk are parsed as if more text followed (zlib there sees the end of its input): callers stitch chunks
// well before that (host/pgzip.cpp takes a chunk's symbols only up to 1 KiB before its end) and leave a stream's true end to zlib.
//
// The same source compiles for gfx950 (W = the hardware wave) and for the host (W = 64 lanes in a loop; tests/harness/
// lz77_check.cpp compares its symbols with the ones zlib itself produced, chunk by chunk).
#ifndef PD_LZ77_H_
#define PD_LZ77_H_
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define PZ_FN __host__ __device__ __forceinline__
#else
#define PZ_FN inline
#endif

namespace pdz {

enum { MIN_MATCH = 3, MAX_MATCH = 258, MAX_DIST = 32768 - 262, TOO_FAR = 4096, GOOD_LEN = 8, MAX_LAZY = 16, NICE_LEN = 128, MAX_CHAIN = 128,
       HASH_BITS = 15 };

PZ_FN uint32_t hash3(const uint8_t *p) { return (((uint32_t)p[0] << 10) ^ ((uint32_t)p[1] << 5) ^ (uint32_t)p[2]) & ((1u << HASH_BITS) - 1); }

PZ_FN uint64_t ld64(const uint8_t *p) { uint64_t w; __builtin_memcpy(&w, p, 8); return w; }

// common prefix of tp[..] and tq[..], at most cap bytes, 16 bytes per step (tq < tp; reads stay below tp + cap + 16: buffers carry
// at least 32 bytes of slack behind the text)
PZ_FN uint32_t lcp(const uint8_t *tp, const uint8_t *tq, uint32_t cap)
{
    uint32_t n = 0;
    while (n < cap) {
        const uint64_t x0 = ld64(tp + n) ^ ld64(tq + n), x1 = ld64(tp + n + 8) ^ ld64(tq + n + 8);
        if (x0) { n += (uint32_t)(__builtin_ctzll(x0) >> 3); break; }
        if (x1) { n += 8u + (uint32_t)(__builtin_ctzll(x1) >> 3); break; }
        n += 16;
    }
    return n < cap ? n : cap;
}

struct Text {
    const uint8_t *text;          // the batch's text, or a window of it held nearer (LDS): the byte of position p is text[p - lo]
    const uint32_t *S;            // positions with at least 3 bytes left, sorted by (hash3, position)
    const uint32_t *R;            // R[p]: index of p in S
    const uint32_t *bucket;       // bucket[h]: index in S of the first position with hash h (1 << HASH_BITS entries + 1)
    uint64_t n;                   // bytes of text
    uint64_t lo = 0;              // the position text[0] stands for (positions below it are never read: a chunk's candidates lie in its history)
    PZ_FN const uint8_t *at(uint64_t p) const { return text + (uint32_t)(p - lo); }
};

// R[] of the positions around the parse, 128 at a time in two registers per lane (one coalesced load per 64 positions): the index of
// a position the parse may visit next is then a lane read instead of a load that the loads of its candidates would have to wait for.
template <class W> struct RWin {
    typedef typename W::template Var<uint32_t> U;
    uint64_t wb = ~0ull << 32;    // cur = R[wb .. wb + 64), nxt = R[wb + 64 .. wb + 128)
    U cur, nxt;
    PZ_FN void load(const Text &T, U &v, uint64_t base) { W::each([&](int l) { const uint64_t i = base + (uint64_t)l; v[l] = i < T.n ? T.R[i] : 0u; }); }
    PZ_FN void seek(const Text &T, uint64_t s)
    {
        const uint64_t d = s - wb;
        if (d < 64) return;
        if (d < 128) { cur = nxt; wb += 64; } else { wb = s & ~63ull; load(T, cur, wb); }
        load(T, nxt, wb + 64);
        // (once per 64 positions: with the window's loads known to have landed, the lane reads of the visits that follow need not wait for
        // whatever else is in flight — the compiler cannot tell the age of a load across the loop and would wait for everything)
        W::loads_landed();
    }
    PZ_FN uint32_t get(const Text &T, uint64_t c) const
    {
        const uint64_t d = c - wb;
        if (d < 64) return W::bcast(cur, (int)d);
        if (d < 128) return W::bcast(nxt, (int)(d - 64));
        return W::uni(T.R[c]);                                  // (a jump of more than a window: the one load a fetch then waits for)
    }
};

// What a visit of position p needs from memory besides the text: where p stands in S, where its bucket begins, and its chain —
// the 128 entries of S before it, lane l holding the l-th and the (64 + l)-th most recent.  The parse fetches this for the (at most
// two) positions it can visit next while it works on the current one, so that a visit's own chain of dependent accesses is text only.
template <class W> struct Cand {
    typedef typename W::template Var<uint32_t> U;
    uint64_t p = ~0ull;
    uint32_t r = 0;
    U b;                          // the bucket's start, the same in every lane (left in a vector register: reading it out would wait for its load)
    U s0, s1;
};
template <class W>
PZ_FN void fetch(const Text &T, const RWin<W> &rw, uint64_t c, Cand<W> &x)
{
    x.p = c;
    const uint32_t r = rw.get(T, c);
    x.r = r;
    const uint32_t h = hash3(T.at(c));
    W::each([&](int l) {
        x.b[l] = T.bucket[h];
        x.s0[l] = r > (uint32_t)l ? T.S[r - 1u - (uint32_t)l] : 0u;
        x.s1[l] = r > 64u + (uint32_t)l ? T.S[r - 65u - (uint32_t)l] : 0u;
    });
}

// zlib's longest_match for position p (prev_len = length of the match found at p - 1, the bar to beat), `look` bytes left from p;
// x = what fetch() brought for p, a0 / a1 = the 16 bytes at p.  Returns the match length (prev_len if nothing longer was found, as zlib
// does; MIN_MATCH - 1 when the head of the chain is no candidate at all: zlib does not even call longest_match then) and sets *start.
// Wave-uniform in, wave-uniform out.
//
// A lane measures TWO candidates at once — the k-th and the (64 + k)-th most recent — with all their loads in flight together;
// what zlib's loop would have done with them (budget, the end of the chain, nice_length) is then decided group by group, in order.
template <class W>
PZ_FN uint32_t longest_match(const Text &T, uint64_t p, uint32_t prev_len, uint64_t look, uint64_t origin, const Cand<W> &x, uint64_t a0, uint64_t a1,
                             uint32_t *start)
{
    typedef typename W::template Var<uint32_t> U;
    const uint32_t avail = x.r - W::uni(W::bcast(x.b, 0));     // earlier positions with this hash
    const uint32_t budget = prev_len >= GOOD_LEN ? MAX_CHAIN >> 2 : MAX_CHAIN;
    const uint32_t nice = look < (uint64_t)NICE_LEN ? (uint32_t)look : (uint32_t)NICE_LEN;
    const uint32_t cap = look < (uint64_t)MAX_MATCH ? (uint32_t)look : (uint32_t)MAX_MATCH;
    const bool two = budget > 64u && avail > 64u;
    const uint8_t *tp = T.at(p);
    U len0, pos0, len1, pos1;
    const uint32_t p32 = (uint32_t)p, org = (uint32_t)origin;   // (a round's text is below 4 GB)
    W::each([&](int l) {
        // zlib: the head of the chain may be exactly MAX_DIST away, the others must be nearer; position `origin` is NIL
        const uint32_t k0 = (uint32_t)l;
        const bool in0 = k0 < avail && k0 < budget;
        const uint32_t q0 = in0 ? x.s0[l] : 0u;
        const bool ok0 = in0 && q0 != org && (l == 0 ? p32 - q0 <= (uint32_t)MAX_DIST : p32 - q0 < (uint32_t)MAX_DIST);
        // the first 16 bytes at once (a lane without a candidate compares p with itself and drops the result)
        const uint8_t *t0 = ok0 ? T.at(q0) : tp;
        uint32_t q1 = 0; bool ok1 = false;
        const uint8_t *t1 = tp;
        if (two) {                                             // (wave-uniform) the second 64 candidates, their loads in flight with the first
            const uint32_t k1 = 64u + (uint32_t)l;
            const bool in1 = k1 < avail && k1 < budget;
            q1 = in1 ? x.s1[l] : 0u;
            ok1 = in1 && q1 != org && p32 - q1 < (uint32_t)MAX_DIST;
            t1 = ok1 ? T.at(q1) : tp;
        }
        const uint64_t x00 = a0 ^ ld64(t0), x01 = a1 ^ ld64(t0 + 8);
        uint64_t x10 = 0, x11 = 0;
        if (two) { x10 = a0 ^ ld64(t1); x11 = a1 ^ ld64(t1 + 8); }
        uint32_t n0 = x00 ? (uint32_t)(__builtin_ctzll(x00) >> 3) : x01 ? 8u + (uint32_t)(__builtin_ctzll(x01) >> 3) : 16u;
        if (ok0 && n0 == 16u && cap > 16u) n0 = 16u + lcp(tp + 16, t0 + 16, cap - 16u);
        pos0[l] = q0;
        len0[l] = ok0 ? (n0 < cap ? n0 : cap) + 1u : 0u;      // + 1: 0 marks "the chain ends here"
        pos1[l] = q1; len1[l] = 0u;
        if (two) {
            uint32_t n1 = x10 ? (uint32_t)(__builtin_ctzll(x10) >> 3) : x11 ? 8u + (uint32_t)(__builtin_ctzll(x11) >> 3) : 16u;
            if (ok1 && n1 == 16u && cap > 16u) n1 = 16u + lcp(tp + 16, t1 + 16, cap - 16u);
            len1[l] = ok1 ? (n1 < cap ? n1 : cap) + 1u : 0u;
        }
    });
    uint32_t best = prev_len, best_q = 0;
    bool head_ok = true;
    // one group of 64 candidates as zlib's loop takes them; true = the loop goes on behind them
    auto take = [&](const U &len, const U &pos, uint32_t bud, bool first) -> bool {
        // the chain ends at the first invalid candidate (positions only get older)
        const uint64_t dead = W::ballot_eq(len, 0u);
        const uint32_t n_here = dead ? (uint32_t)__builtin_ctzll(dead) : 64u;
        const uint32_t n_look = n_here < bud ? n_here : bud;
        if (n_look == 0) { if (first) head_ok = false; return false; }
        // the first candidate of nice length or more ends the loop (it is looked at)
        U nice_hit;
        W::each([&](int l) { nice_hit[l] = (uint32_t)l < n_look && len[l] - 1u >= nice ? 1u : 0u; });
        const uint64_t nm = W::ballot_ne(nice_hit, 0u);
        const uint32_t n_eff = nm ? (uint32_t)__builtin_ctzll(nm) + 1u : n_look;
        // the greatest length among the first n_eff, earliest lane first
        U cand;
        W::each([&](int l) { cand[l] = (uint32_t)l < n_eff ? len[l] - 1u : 0u; });
        const uint32_t mx = W::uni(W::reduce_max(cand));
        if (mx > best) {
            const uint64_t at = W::ballot_eq(cand, mx);
            best = mx; best_q = W::uni(W::bcast(pos, (int)__builtin_ctzll(at)));
        }
        return !(nm || n_here < 64u || n_look < 64u);          // nice hit, chain ended, or budget used up inside this group
    };
    if (take(len0, pos0, budget, true) && two) (void)take(len1, pos1, budget - 64u, false);
    if (!head_ok) { *start = 0; return MIN_MATCH - 1; }
    *start = best_q;
    return (uint64_t)best <= look ? best : (uint32_t)look;
}

struct Out { uint32_t *syms; uint32_t n, cap; };               // literal = byte; match = len << 16 | dist
template <class W> PZ_FN bool put(Out &o, uint32_t s) { if (o.n >= o.cap) return false; if (W::lead()) o.syms[o.n] = s; ++o.n; return true; }

// The parse of text[start, end) with text[start - dict, start) as history.  `origin` = the position zlib's window index 0
// stands for (start - dict).  Returns false when the symbol buffer is too small.  Only lane 0 writes the symbols.
template <class W>
PZ_FN bool parse_chunk(const Text &T, uint64_t start, uint64_t end, uint64_t origin, Out &o)
{
    uint64_t s = start;                                         // strstart
    uint32_t match_len = MIN_MATCH - 1, match_start = 0;
    bool avail = false;
    bool ok = true;
    RWin<W> rw;
    Cand<W> A, B, cur;                                          // fetched for the two positions that could follow the previous visit; the current one
    uint32_t prev_byte = 0;                                     // text[s - 1] when the previous visit was there
    while (s < end) {
        const uint64_t look = end - s;
        const uint32_t prev_len = match_len, prev_start = match_start;
        match_len = MIN_MATCH - 1;
        const bool search = look >= MIN_MATCH && prev_len < MAX_LAZY;
        rw.seek(T, s);
        if (s == A.p) cur = A;
        else if (s == B.p) cur = B;
        else if (search) fetch<W>(T, rw, s, cur);
        // the next visit: s + 1, or — when this visit emits the previous match — the position behind that match
        const uint64_t p1 = s + 1, p2 = prev_len >= MIN_MATCH ? s + prev_len - 1 : p1;
        if (p1 + MIN_MATCH <= T.n) fetch<W>(T, rw, p1, A); else A.p = ~0ull;      // (behind the text: never visited with a search)
        if (p2 != p1 && p2 + MIN_MATCH <= T.n) fetch<W>(T, rw, p2, B); else B.p = ~0ull;
        const uint64_t a0 = ld64(T.at(s)), a1 = ld64(T.at(s) + 8);
        // (the string at s enters the chains here in zlib: the chain of s is every earlier position of its bucket)
        if (search && cur.r > W::uni(W::bcast(cur.b, 0))) {
            match_len = longest_match<W>(T, s, prev_len, look, origin, cur, a0, a1, &match_start);
            if (match_len <= prev_len) match_start = prev_start;               // (zlib leaves match_start alone unless it found something longer)
            if (match_len == MIN_MATCH && s - match_start > (uint64_t)TOO_FAR) match_len = MIN_MATCH - 1;
        }
        if (prev_len >= MIN_MATCH && match_len <= prev_len) {
            ok = ok && put<W>(o, (prev_len << 16) | (uint32_t)(s - 1 - prev_start));
            s += prev_len - 1;                                  // the previous match began at s - 1
            avail = false; match_len = MIN_MATCH - 1;
        } else if (avail) {
            ok = ok && put<W>(o, prev_byte);
            ++s;
        } else {
            avail = true; ++s;
        }
        prev_byte = (uint32_t)(a0 & 0xffu);
    }
    if (avail) ok = ok && put<W>(o, prev_byte);
    return ok;
}

// ---- the host wave (64 lanes in a loop); the device wave lives with the kernels -----------------------------------
struct HostWave {
    template <class T> struct Var { T v[64]; T &operator[](int l) { return v[l]; } const T &operator[](int l) const { return v[l]; } };
    template <class F> static void each(F f) { for (int l = 0; l < 64; ++l) f(l); }
    static uint64_t ballot_eq(const Var<uint32_t> &x, uint32_t v) { uint64_t m = 0; for (int l = 0; l < 64; ++l) if (x.v[l] == v) m |= 1ull << l; return m; }
    static uint64_t ballot_ne(const Var<uint32_t> &x, uint32_t v) { uint64_t m = 0; for (int l = 0; l < 64; ++l) if (x.v[l] != v) m |= 1ull << l; return m; }
    static uint32_t reduce_max(const Var<uint32_t> &x) { uint32_t a = 0; for (int l = 0; l < 64; ++l) if (x.v[l] > a) a = x.v[l]; return a; }
    static uint32_t bcast(const Var<uint32_t> &x, int lane) { return x.v[lane]; }
    static void loads_landed() {}
    static uint32_t uni(uint32_t x) { return x; }             // (the device wave: the value is the same in every lane — keep it in a scalar register)
    static bool lead() { return true; }
};

} // namespace pdz
#endif
