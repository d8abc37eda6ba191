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
// (their positions are candidates), the parse starts fresh at `start` and covers [start, end).  The bytes of the last
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

// common prefix of text[p ..] and text[q ..], at most cap bytes (q < p; reads stay below p + cap + 8: buffers carry 8 bytes of slack)
PZ_FN uint32_t lcp(const uint8_t *text, uint64_t p, uint64_t q, uint32_t cap)
{
    uint32_t n = 0;
    while (n < cap) {
        const uint64_t x = ld64(text + p + n) ^ ld64(text + q + n);
        if (x) { n += (uint32_t)(__builtin_ctzll(x) >> 3); break; }
        n += 8;
    }
    return n < cap ? n : cap;
}

struct Text {
    const uint8_t *text;          // the batch's text; positions are indices into it
    const uint32_t *S;            // positions with at least 3 bytes left, sorted by (hash3, position)
    const uint32_t *R;            // R[p]: index of p in S
    const uint32_t *bucket;       // bucket[h]: index in S of the first position with hash h (1 << HASH_BITS entries + 1)
    uint64_t n;                   // bytes of text
};

// zlib's longest_match for position p (prev_len = length of the match found at p - 1, the bar to beat), `look` bytes left from p.
// Returns the match length (prev_len if nothing longer was found, as zlib does) and sets *start.  Wave-uniform in, wave-uniform out.
template <class W>
PZ_FN uint32_t longest_match(const Text &T, uint64_t p, uint32_t prev_len, uint64_t look, uint64_t origin, uint32_t *start)
{
    typedef typename W::template Var<uint32_t> U;
    const uint32_t h = hash3(T.text + p);
    const uint32_t r = T.R[p], b0 = T.bucket[h];
    uint32_t avail = r - b0;                                   // earlier positions with this hash
    uint32_t budget = prev_len >= GOOD_LEN ? MAX_CHAIN >> 2 : MAX_CHAIN;
    const uint32_t nice = look < (uint64_t)NICE_LEN ? (uint32_t)look : (uint32_t)NICE_LEN;
    const uint32_t cap = look < (uint64_t)MAX_MATCH ? (uint32_t)look : (uint32_t)MAX_MATCH;
    uint32_t best = prev_len, best_q = 0;
    bool first = true;
    uint32_t k0 = 0;                                           // candidates already looked at
    while (budget && k0 < avail) {
        U len, pos;
        W::each([&](int l) {
            len[l] = 0; pos[l] = 0;
            const uint32_t k = k0 + (uint32_t)l;               // k-th most recent
            if (k >= avail || (uint32_t)l >= budget) return;
            const uint32_t q = T.S[r - 1 - k];
            pos[l] = q;
            // zlib: the head of the chain may be exactly MAX_DIST away, the others must be nearer; position `origin` is NIL
            const uint64_t dist = p - q;
            const bool ok = q != (uint32_t)origin && (first && l == 0 ? dist <= (uint64_t)MAX_DIST : dist < (uint64_t)MAX_DIST);
            len[l] = ok ? lcp(T.text, p, q, cap) + 1u : 0u;    // + 1: 0 marks "the chain ends here"
        });
        // the chain ends at the first invalid candidate (positions only get older)
        const uint64_t dead = W::ballot_eq(len, 0u);
        const uint32_t n_here = dead ? (uint32_t)__builtin_ctzll(dead) : 64u;
        const uint32_t n_look = n_here < budget ? n_here : budget;
        if (n_look == 0) break;
        // the first candidate of nice length or more ends the loop (it is looked at)
        U nice_hit;
        W::each([&](int l) { nice_hit[l] = (uint32_t)l < n_look && len[l] - 1u >= nice ? 1u : 0u; });
        const uint64_t nm = W::ballot_ne(nice_hit, 0u);
        const uint32_t n_eff = nm ? (uint32_t)__builtin_ctzll(nm) + 1u : n_look;
        // the greatest length among the first n_eff, earliest lane first
        U cand;
        W::each([&](int l) { cand[l] = (uint32_t)l < n_eff ? len[l] - 1u : 0u; });
        const uint32_t mx = W::reduce_max(cand);
        if (mx > best) {
            const uint64_t at = W::ballot_eq(cand, mx);
            best = mx; best_q = W::bcast(pos, (int)__builtin_ctzll(at));
        }
        if (nm || n_here < 64u || n_look < 64u) break;         // nice hit, chain ended, or budget used up inside this group
        budget -= 64u; k0 += 64u; first = false;
    }
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
    while (s < end) {
        const uint64_t look = end - s;
        const uint32_t prev_len = match_len, prev_start = match_start;
        match_len = MIN_MATCH - 1;
        // (the string at s enters the chains here in zlib: the chain of s is every earlier position of its bucket)
        if (look >= MIN_MATCH && prev_len < MAX_LAZY && T.R[s] > T.bucket[hash3(T.text + s)]) {
            const uint32_t head = T.S[T.R[s] - 1];
            if (head != (uint32_t)origin && s - head <= (uint64_t)MAX_DIST) {
                match_len = longest_match<W>(T, s, prev_len, look, origin, &match_start);
                if (match_len <= prev_len) match_start = prev_start;       // (zlib leaves match_start alone unless it found something longer)
                if (match_len == MIN_MATCH && s - match_start > (uint64_t)TOO_FAR) match_len = MIN_MATCH - 1;
            }
        }
        if (prev_len >= MIN_MATCH && match_len <= prev_len) {
            ok = ok && put<W>(o, (prev_len << 16) | (uint32_t)(s - 1 - prev_start));
            s += prev_len - 1;                                  // the previous match began at s - 1
            avail = false; match_len = MIN_MATCH - 1;
        } else if (avail) {
            ok = ok && put<W>(o, (uint32_t)T.text[s - 1]);
            ++s;
        } else {
            avail = true; ++s;
        }
    }
    if (avail) ok = ok && put<W>(o, (uint32_t)T.text[s - 1]);
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
    static bool lead() { return true; }
};

} // namespace pdz
#endif
