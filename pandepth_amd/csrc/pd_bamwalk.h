// pd_bamwalk.h — finding and parsing BAM records in inflated bytes with a whole wavefront, replacing the one-thread-
// per-unit walk of round 1.  Restates PD:434-460 (filter + CIGAR walk) on raw BAM bytes (SAM spec §4.2): block_size,
// refID, pos, l_read_name, mapq, bin, n_cigar_op, flag, l_seq, next_refID, next_pos, tlen at fixed offsets, the CIGAR
// after the read name.
//
// Records form a chain (each record's length says where the next one starts), so a stretch of the inflated stream is
// walked SPECULATIVELY, like the Huffman streams in pd_inflate_wave.h:
//   * a SEGMENT is up to 64 x PD_WALK_SUB bytes (256 KiB) whose first record start is known (an index offset) or guessed; lane l of the wave owns
//     the records that start in its l-th stretch of PD_WALK_SUB bytes ("its KiB" below, from the days when that was 1024);
//   * every lane looks for the first position in its KiB that passes a strict record-header test (sizes consistent with
//     each other, ids inside the header's tables, a NUL where the name ends, and the same for the record it points to),
//     and walks from there to the end of its KiB;
//   * the true start of lane l is where the chain of lanes 0..l-1 ends = the prefix maximum of their end positions; lanes
//     whose guess was different walk again, until nothing changes.  A stable state is the exact chain by induction from
//     lane 0, whatever the guesses were: the header test only makes it come quickly.  A segment whose own start was a
//     guess is checked the same way against the segments before it (k_walk_finish), and redone with the corrected start.
// Pass 1 counts what every lane will emit (reads that pass the filter; their M/=/X runs), pass 2 writes the runs to
// dense arrays: the first run of every read in file order (position sorted for a coordinate-sorted BAM), the other runs
// behind them in the same order.
//
// The same source compiles for gfx950 and for the host (tests/harness/bamwalk_check.cpp compares it with the host reader).
#ifndef PD_BAMWALK_H_
#define PD_BAMWALK_H_
#include <stdint.h>
#include "pd_inflate_wave.h"              // PW_FN, the wave classes
#include "../../include/pandepth_amd.h"

namespace pdb2 {

#ifndef PD_WALK_SUB
// bytes of a segment a lane owns (a segment = 64 of them = one wave).  Round 5, measured on the 3e8-record file (profiles/r05_walk_sub.txt): a lane pays for
// its header search and the three-records-deep test of its guess once, whatever it owns — at 1 KiB (three records of a short-read file) that is
// half its work; 4 KiB: pass 1 takes 256 ms of kernel time instead of 410, the chain kernel has a quarter of the segments to go through (8 KiB: the same;
// 512 B: 590 ms)
#define PD_WALK_SUB 4096
#endif
enum { SUB = PD_WALK_SUB, SEG_BYTES = 64 * SUB };
// Later runs of a read that begin more than Cfg::near_span bases after its start may go to a separate ("far") stream.  Measured
// on the 50x sample a third stream costs the tile kernels more (per-batch bookkeeping in every tile) than its tighter disorder
// bound saves, so the split is off by default (near_span = 0xFFFFFFFF) and every later run goes to the one "other" stream.
enum { WF_BAD = 1, WF_MORE = 2, WF_HOST = 4 };     // corrupt record / record runs past the batch / CIGAR in the CG tag
static const uint64_t NONE = ~0ull, STOPPED = 1ull << 62;

// Compact emission (whole-contig modes of a coordinate-sorted file, PD_DECODE_COMPACT): pass 2 writes every read's first run straight to
// the batch's segment of the sample's sorted stream as 8 bytes — the low 32 bits of its flat begin (cell index in the engine's buffer, the
// begin clamped to [0, len] as PD:449-452's cells are) and its clamped length —, the first run a lane writes and every run that opens
// a new 512-cell bucket leave (batch, index) in marks[bucket] (an atomic minimum: the buckets' first runs in file order, which is all the
// direct kernels need to find a tile's runs), and the order of the stream is checked on the way: inside a lane, across the lanes of a
// segment, and (by the host, from SegOut) across segments and batches.
struct R8 { uint32_t b, len; };           // = pdk::Run8
struct SegOut { uint64_t first_key, last_key; uint32_t unsorted, n_long; };   // keys: flat begins (order like (tid, begin)); first_key = NONE: no first run
struct C8Out {
    R8 *r8 = nullptr;                     // the batch's first runs (index = Seg::base_first + ...); null: 12-byte runs as before
    unsigned long long *marks = nullptr;  // bucket -> min (batch << 32 | index in the batch) of a run that begins there (pre-set to all ones)
    const uint64_t *contig_off = nullptr; // first cell of every contig's slot
    uint32_t cshift = 0;                  // log2(cells per bucket)
    SegOut *seg_out = nullptr;            // one per segment of the batch (also without r8: 12-byte first runs, keys (tid << 32 | begin))
    uint32_t batch = 0;                   // the batch's number in file order
};

struct Cfg {                              // wave-uniform
    const uint8_t *buf; uint64_t avail;   // the batch's inflated bytes
    int32_t n_ref; const uint32_t *contig_len; const uint8_t *contig_on;      // contig_on[tid] != 0: the contig has targets
    uint32_t flag_mask; int32_t min_mapq;
    const uint32_t *span_off; const int32_t *spans;                            // -g / -b: (begin0, end) per contig, sorted; or null
    uint32_t near_span;
    C8Out c8;
};

#if defined(__HIP_DEVICE_COMPILE__)
PW_FN void min_u64(unsigned long long *p, unsigned long long v) { atomicMin(p, v); }
#else
PW_FN void min_u64(unsigned long long *p, unsigned long long v) { if (v < *p) *p = v; }
#endif

struct Seg {
    uint64_t begin, end;                  // records STARTING in [begin, end) belong to the segment
    uint64_t hint;                        // its first record start (>= begin) when known, else NONE
    uint64_t avail;                       // end of the unit's inflated bytes (a record must end at or before it)
    uint32_t unit_first, n_rec;           // first segment of its unit (nothing before it to be checked against); OUT: records it owns
    // results
    uint64_t used_start, e_last;          // where lane 0 started; first record start >= end (0: no record seen)
    uint32_t n_first, n_other, flags, max_span;   // n_other: near runs; n_far: runs that begin > NEAR_SPAN after their read's start
    uint32_t n_far, pad2;
    uint64_t base_first, base_other, base_far;    // written by the host between the passes: where the segment's runs go
};
static const int SCAN = 64;               // positions of the header search tested per step (one group of loads)
struct LaneOut { uint64_t start; uint32_t n_first, n_other, n_far, pad; };
// what one walk of a segment found (wave-uniform; the same values go to the Seg in memory)
struct WalkOut { uint64_t used_start, e_last; uint32_t n_first, n_other, n_far, n_rec, flags, max_span; };

PW_FN int ctz64(uint64_t x) { return __builtin_ctzll(x); }
PW_FN uint32_t rd32(const uint8_t *p) { uint32_t w; __builtin_memcpy(&w, p, 4); return w; }
PW_FN uint32_t rd16(const uint8_t *p) { uint16_t w; __builtin_memcpy(&w, p, 2); return w; }

// strict header test of a candidate record start (everything it reads lies below avail)
PW_FN bool plausible(const Cfg &c, uint64_t p, int depth)
{
    for (int k = 0; k < depth; ++k) {
        if (p + 36 > c.avail) return k > 0;                               // cannot look further: what was seen was consistent
        const uint8_t *r = c.buf + p;
        const uint32_t bs = rd32(r);
        if (bs < 33 || bs > (1u << 27)) return false;                         // 32 fixed bytes + a name of at least its NUL
        const int32_t tid = (int32_t)rd32(r + 4), pos = (int32_t)rd32(r + 8);
        if (tid < -1 || tid >= c.n_ref || pos < -1) return false;
        const uint32_t l_name = r[12], n_cig = rd16(r + 16), l_seq = rd32(r + 20);
        const int32_t ntid = (int32_t)rd32(r + 24), npos = (int32_t)rd32(r + 28);
        if (l_name < 1 || l_seq > (1u << 27) || ntid < -1 || ntid >= c.n_ref || npos < -1) return false;
        if ((uint64_t)32 + l_name + 4ull * n_cig + (l_seq + 1) / 2 + l_seq > bs) return false;
        if (tid >= 0 && pos >= 0 && (uint32_t)pos > c.contig_len[tid]) return false;
        if (p + 36 + l_name <= c.avail && r[36 + l_name - 1] != 0) return false;             // the name ends with NUL
        p += 4 + (uint64_t)bs;
    }
    return true;
}

struct Rec { int32_t tid, pos; uint32_t keep, n_cig; const uint8_t *cig; uint32_t flags, unmapped; };

// header of the record at p (p + 36 <= avail, block size already validated by the caller) + the filter
PW_FN Rec open_record(const Cfg &c, uint64_t p, uint32_t bs)
{
    const uint8_t *r = c.buf + p;
    Rec x;
    x.tid = (int32_t)rd32(r + 4); x.pos = (int32_t)rd32(r + 8);
    const uint32_t l_name = r[12], mapq = r[13], flag = rd16(r + 18), l_seq = rd32(r + 20);
    x.n_cig = rd16(r + 16); x.cig = r + 36 + l_name; x.flags = 0; x.unmapped = flag & 4;
    x.keep = !(flag & c.flag_mask) && (int32_t)mapq >= c.min_mapq && x.tid >= 0 && x.tid < c.n_ref;
    if (x.keep && !c.contig_on[x.tid]) x.keep = 0;
    if ((uint64_t)32 + l_name + 4ull * x.n_cig > bs) { x.keep = 0; x.flags = WF_BAD; }      // the host reader fails such a record
    // CIGAR kept in the CG tag (more than 65 535 operations: long reads).  htslib's test (bam_tag2cigar), as host/bam.cpp restates it: a
    // first operation <l_seq>S on a placed read, and the FIRST CG tag of the record of type B,I or B,i with at least n_cigar entries that
    // lie inside the record — then the tag's operations are the CIGAR; without such a tag the placeholder is kept.  Until round 6 the
    // lane flagged the record (WF_HOST) and the whole unit went to the host reader: every batch of a long-read file.  The record's
    // bytes all lie below c.avail (the caller has checked p + 4 + bs), so the lane may walk its tags here; one record in a hundred.
    else if (x.n_cig >= 1 && x.tid >= 0 && x.pos >= 0 && (rd32(x.cig) & 0xf) == 4 && (rd32(x.cig) >> 4) == l_seq && x.keep) {
        const uint64_t end = p + 4 + (uint64_t)bs;
        uint64_t a = p + 36 + l_name + 4ull * x.n_cig + ((uint64_t)l_seq + 1) / 2 + (uint64_t)l_seq;
        if (a > end) a = end;
        while (a + 3 <= end) {
            const uint8_t t0 = c.buf[a], t1 = c.buf[a + 1], ty = c.buf[a + 2];
            a += 3;
            uint64_t sz = 0;
            if (ty == 'A' || ty == 'c' || ty == 'C') sz = 1;
            else if (ty == 's' || ty == 'S') sz = 2;
            else if (ty == 'i' || ty == 'I' || ty == 'f') sz = 4;
            else if (ty == 'Z' || ty == 'H') {
                uint64_t z = a;
                while (z < end && c.buf[z] != 0) ++z;
                if (z >= end) break;
                sz = z - a + 1;
            } else if (ty == 'B') {
                if (a + 5 > end) break;
                const uint8_t st = c.buf[a];
                const uint32_t cnt = rd32(c.buf + a + 1);
                const uint64_t es = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4;
                if (t0 == 'C' && t1 == 'G') {
                    if ((st == 'I' || st == 'i') && cnt >= x.n_cig && cnt < (1u << 29) && a + 5 + 4ull * cnt <= end) { x.cig = c.buf + a + 5; x.n_cig = cnt; }
                    break;
                }
                sz = 5 + es * cnt;
            } else break;
            a += sz;
        }
    }
    return x;
}

// M/=/X runs of one kept record; emit(first?, beg, end).  Returns the reference end position.
template <class F> PW_FN int32_t walk_cigar(const Rec &x, F emit)
{
    int32_t cur = x.pos;
    bool first_done = false, moved = false;
    for (uint32_t i = 0; i < x.n_cig; ++i) {
        const uint32_t cg = rd32(x.cig + 4 * i), op = cg & 0xf;
        const int32_t len = (int32_t)(cg >> 4);
        // (the reference's int cursor wraps on corrupt lengths; here the wrap is spelled out in unsigned arithmetic)
        const int32_t nxt = (int32_t)((uint32_t)cur + (uint32_t)len);
        if (op == 0 || op == 7 || op == 8) {
            emit(!first_done && !moved, cur, nxt);
            first_done = true; cur = nxt;
        } else if (op == 2 || op == 3) { cur = nxt; moved = true; }
    }
    return cur;
}

// htslib's multi-region test for -g / -b (PD:419-434): pos < span end && endpos > span begin0 for some widened span
PW_FN bool span_hit(const Cfg &c, int32_t tid, int32_t pos, int32_t endpos)
{
    uint32_t lo = c.span_off[tid], hi = c.span_off[tid + 1];
    const uint32_t top = hi;
    while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if (c.spans[2 * m + 1] > pos) hi = m; else lo = m + 1; }
    return lo < top && endpos > c.spans[2 * lo];
}

struct LaneWalk { uint64_t e; uint32_t n_first, n_other, n_far, flags, max_span, n_rec; uint64_t key_first, key_last; uint32_t unsorted, n_long; };

// A record whose CIGAR the whole wave walks (round 6: long reads).  A lane walks a CIGAR one operation — one dependent load — at a
// time; with the ~2 600 operations of a 15 kb HiFi / ONT read that is a millisecond per record while the other 63 lanes of the wave
// idle (only a handful of records start in a 256 KiB segment).  A lane that meets a CIGAR of COOP_MIN operations or more therefore
// stops there and leaves the record to coop_cigar: 64 operations per step, reference positions by a prefix sum of the advancing
// lengths, the places of the runs by a prefix count.
enum { COOP_MIN = 96 };
struct Pending { uint64_t cig_off, next_p; uint32_t n_cig; int32_t tid, pos; };

// one first run: written (EMIT) in the session's form, with the order keys and bucket marks of the compact emission riding along
template <bool EMIT>
PW_FN void first_run(const Cfg &c, LaneWalk &w, int32_t tid, int32_t beg, int32_t end, pd_iv *first, uint64_t of)
{
    if (EMIT && c.c8.r8) {                            // compact emission (wave-uniform choice)
        const uint32_t clen = c.contig_len[tid];
        uint32_t cb = beg < 0 ? 0u : (uint32_t)beg; if (cb > clen) cb = clen;
        uint32_t ce = end < 0 ? 0u : (uint32_t)end; if (ce > clen) ce = clen;
        const uint32_t len = ce > cb ? ce - cb : 0u;
        const uint64_t flat = c.c8.contig_off[tid] + cb, G = of + w.n_first;
        c.c8.r8[G] = R8{(uint32_t)flat, len};
        if (w.key_first == NONE || (w.key_last >> c.c8.cshift) != (flat >> c.c8.cshift))
            min_u64(&c.c8.marks[flat >> c.c8.cshift], ((unsigned long long)c.c8.batch << 32) | (uint32_t)G);
        if (w.key_first == NONE) w.key_first = flat; else if (flat < w.key_last) w.unsorted = 1;
        w.key_last = flat;
        if (len > (1u << c.c8.cshift)) ++w.n_long;
    } else if (EMIT) {
        first[of + w.n_first] = pd_iv{tid, beg, end};
        if (c.c8.seg_out) {                           // (12-byte emission with the order check riding along: key = (tid, begin) as k_runs_sorted's)
            const uint64_t key = ((uint64_t)(uint32_t)tid << 32) | (uint32_t)beg;
            if (w.key_first == NONE) w.key_first = key; else if (key < w.key_last) w.unsorted = 1;
            w.key_last = key;
        }
    }
}

// Records starting in [p, b): counts (EMIT == false) or runs written at first[of..] / other[oo..] (EMIT == true), from where the lane
// stands (p, guard, w: its state, so that the walk can be taken up again).  Returns true when it has stopped AT a record whose CIGAR
// the wave is to walk (`pend` says which; nothing of that record has been counted yet but ++n_rec) — the caller adds the record's
// runs and calls again with p = pend.next_p — and false when the lane is through (w.e = where its chain ends).
template <bool EMIT>
PW_FN bool walk_lane(const Cfg &c, uint64_t &p, uint64_t b, uint32_t &guard, LaneWalk &w, Pending &pend, pd_iv *first, pd_iv *other, pd_iv *far, uint64_t of, uint64_t oo, uint64_t ofar)
{
    for (; p < b && guard < SUB / 36 + 2; ++guard) {
        // a record that cannot be finished here stops the chain: nothing after it may be taken for a record start
        if (p + 36 > c.avail) { w.flags |= WF_MORE; p = STOPPED; break; }
        const uint32_t bs = rd32(c.buf + p);
        if (bs < 32 || bs > (1u << 29)) { w.flags |= WF_BAD; p = STOPPED; break; }
        if (p + 4 + (uint64_t)bs > c.avail) { w.flags |= WF_MORE; p = STOPPED; break; }
        const Rec x = open_record(c, p, bs);
        w.flags |= x.flags; ++w.n_rec;
        if (x.keep && !x.flags) {
            if (x.n_cig >= (uint32_t)COOP_MIN && c.near_span == 0xFFFFFFFFu && !(c.spans && x.unmapped)) {
                pend.cig_off = (uint64_t)(x.cig - c.buf); pend.next_p = p + 4 + (uint64_t)bs; pend.n_cig = x.n_cig; pend.tid = x.tid; pend.pos = x.pos;
                ++guard;
                return true;
            }
            bool take = true;
            if (c.spans) {                                               // endpos first: the filter needs it
                // htslib's bam_endpos: unmapped reads and alignments without reference bases count as one base
                // (pos + reference length in the reference's wrapping int arithmetic, like the host reader's AlnRec::endpos:
                // only a reference length of zero becomes one base — corrupt lengths that wrap select the same reads on both paths)
                const int32_t one = (int32_t)((uint32_t)x.pos + 1u);
                const int32_t endpos = x.unmapped || !x.n_cig ? one : walk_cigar(x, [](bool, int32_t, int32_t) {});
                take = span_hit(c, x.tid, x.pos, endpos == x.pos ? one : endpos);
            }
            if (take) {
                uint32_t nf = 0, no = 0, nfar = 0, span = 0;
                walk_cigar(x, [&](bool is_first, int32_t beg, int32_t end) {
                    if (is_first) { first_run<EMIT>(c, w, x.tid, beg, end, first, of); nf = 1; return; }
                    const uint32_t d = (uint32_t)beg - (uint32_t)x.pos;
                    if (d > span) span = d;
                    if (d <= c.near_span) { if (EMIT) other[oo + w.n_other + no] = pd_iv{x.tid, beg, end}; ++no; }
                    else { if (EMIT) far[ofar + w.n_far + nfar] = pd_iv{x.tid, beg, end}; ++nfar; }
                });
                w.n_first += nf; w.n_other += no; w.n_far += nfar;
                if (span > w.max_span) w.max_span = span;
            }
        }
        p += 4 + (uint64_t)bs;
    }
    w.e = p;
    return false;
}

// The M/=/X runs of ONE record's CIGAR (n operations at c.buf + cig_off), walked by the whole wave — what walk_cigar does one
// operation at a time: a run that is the record's first and comes before any D / N is its FIRST run (reported, not written: the
// lane that owns the record writes it, with the keys and marks of its stream), every other run is written (emit) at
// other[obase + its rank] and counted; span = the largest distance of such a run's begin from the record's position.
struct CoopOut { uint32_t has_first; int32_t fbeg, fend; uint32_t n_other, span; int32_t endpos; };
template <class W>
PW_FN CoopOut coop_cigar(const Cfg &c, uint64_t cig_off, uint32_t n, int32_t tid, int32_t pos, pd_iv *other, uint64_t obase, bool emit)
{
    typedef typename W::template Var<uint32_t> U;
    CoopOut r; r.has_first = 0; r.fbeg = r.fend = 0; r.n_other = 0; r.span = 0;
    uint32_t cur0 = (uint32_t)pos;
    bool first_done = false, moved = false;
    for (uint32_t k0 = 0; k0 < n; k0 += 64) {
        U adv, isrun, isgap, len;
        W::each([&](int l) {
            const uint32_t i = k0 + (uint32_t)l;
            const uint32_t cg = i < n ? rd32(c.buf + cig_off + 4ull * i) : 0xFu;         // (15: no operation)
            const uint32_t op = cg & 0xf;
            len[l] = cg >> 4;
            isrun[l] = (op == 0 || op == 7 || op == 8) ? 1u : 0u;
            isgap[l] = (op == 2 || op == 3) ? 1u : 0u;
            adv[l] = (isrun[l] | isgap[l]) ? len[l] : 0u;
        });
        uint32_t tot = 0;
        const U ea = W::excl_scan(adv, &tot);                                              // (wraps like the reference's int cursor)
        const uint64_t rm = W::ballot_ne(isrun, 0u), gm = W::ballot_ne(isgap, 0u);
        int fl = -1;
        if (!first_done) {
            if (rm) {
                const int f0 = ctz64(rm);
                if (gm & (f0 ? (1ull << f0) - 1 : 0ull)) moved = true;
                if (!moved) fl = f0;
                first_done = true;
            } else if (gm) moved = true;
        }
        const uint64_t om = fl >= 0 ? rm & ~(1ull << fl) : rm;                            // the runs that are not the record's first
        U dd;
        W::each([&](int l) {
            dd[l] = 0;
            if (!((om >> l) & 1)) return;
            const uint32_t beg = cur0 + ea[l], d = beg - (uint32_t)pos;
            dd[l] = d;
            if (emit) other[obase + r.n_other + W::prefix_count(om, l)] = pd_iv{tid, (int32_t)beg, (int32_t)(beg + len[l])};
        });
        if (om) { const uint32_t mx = W::reduce_max(dd); if (mx > r.span) r.span = mx; }
        if (fl >= 0) { const uint32_t fb = cur0 + W::bcast(ea, fl); r.has_first = 1; r.fbeg = (int32_t)fb; r.fend = (int32_t)(fb + W::bcast(len, fl)); }
        r.n_other += (uint32_t)__builtin_popcountll(om);
        cur0 += tot;
    }
    r.endpos = (int32_t)cur0;
    return r;
}

// Every lane's walk of its stretch, to the end: the lanes walk on their own until each is through or stands at a record for the
// wave; those records are walked by the wave one after the other, their runs added to their lanes' counts (or written behind the runs
// the lane has written so far), and the lanes go on.  active[l] = 0: the lane has nothing to walk.
template <class W, bool EMIT>
PW_FN void run_lanes(const Cfg &c, typename W::template Var<uint32_t> &active, typename W::template Var<uint64_t> &p, const typename W::template Var<uint64_t> &b,
                     typename W::template Var<LaneWalk> &lw, pd_iv *first, pd_iv *other, pd_iv *far, const typename W::template Var<uint64_t> &of,
                     const typename W::template Var<uint64_t> &oo, const typename W::template Var<uint64_t> &ofar)
{
    typedef typename W::template Var<uint32_t> U;
    typedef typename W::template Var<uint64_t> U64;
    U guard, pending, pn, ptid, ppos;
    U64 pcig, pnext, pob;
    W::each([&](int l) { guard[l] = 0; pending[l] = 0; pn[l] = 0; ptid[l] = 0; ppos[l] = 0; pcig[l] = 0; pnext[l] = 0; pob[l] = 0; });
    for (;;) {
        W::each([&](int l) {
            if (!active[l]) return;
            Pending pd; pd.cig_off = 0; pd.next_p = 0; pd.n_cig = 0; pd.tid = 0; pd.pos = 0;
            uint64_t pp = p[l]; uint32_t g = guard[l];
            const bool stop = walk_lane<EMIT>(c, pp, b[l], g, lw[l], pd, first, other, far, of[l], oo[l], ofar[l]);
            p[l] = pp; guard[l] = g;
            pending[l] = stop ? 1u : 0u;
            if (stop) { pcig[l] = pd.cig_off; pnext[l] = pd.next_p; pn[l] = pd.n_cig; ptid[l] = (uint32_t)pd.tid; ppos[l] = (uint32_t)pd.pos; pob[l] = oo[l] + lw[l].n_other; }
            else active[l] = 0;
        });
        uint64_t pm = W::ballot_ne(pending, 0u);
        if (!pm) break;
        while (pm) {
            const int j = ctz64(pm);
            pm &= pm - 1;
            const uint64_t cig_off = W::bcast64(pcig, j), obase = W::bcast64(pob, j);
            const uint32_t n = W::bcast(pn, j);
            const int32_t tid = (int32_t)W::bcast(ptid, j), pos = (int32_t)W::bcast(ppos, j);
            // -g / -b sessions: the read counts only when it meets a target span, which takes its end — a pass that only counts, first
            CoopOut r = coop_cigar<W>(c, cig_off, n, tid, pos, other, obase, EMIT && !c.spans);
            U tk;
            W::each([&](int l) {
                tk[l] = 0;
                if (l != j) return;
                bool take = true;
                if (c.spans) { const int32_t one = (int32_t)((uint32_t)pos + 1u); take = span_hit(c, tid, pos, r.endpos == pos ? one : r.endpos); }
                tk[l] = take ? 1u : 0u;
            });
            const bool take = W::ballot_ne(tk, 0u) != 0;
            if (take && EMIT && c.spans) r = coop_cigar<W>(c, cig_off, n, tid, pos, other, obase, true);
            W::each([&](int l) {
                if (l != j) return;
                LaneWalk &w = lw[l];
                if (take) {
                    if (r.has_first) { first_run<EMIT>(c, w, tid, r.fbeg, r.fend, first, of[l]); w.n_first += 1; }
                    w.n_other += r.n_other;
                    if (r.span > w.max_span) w.max_span = r.span;
                }
                p[l] = pnext[l]; pending[l] = 0;
            });
        }
    }
}

// pass 1: one wave, one segment.  lanes[64] receives every lane's start and counts for pass 2.  `hint_in` (or null: the
// segment's own) is the first record start to walk from — the chain kernel passes a corrected start in a register, so that
// no lane has to read what another lane of the wave has just stored.
template <class W>
PW_FN WalkOut walk_segment(const Cfg &cfg, Seg &sg, LaneOut *lanes, const uint64_t *hint_in = nullptr)
{
    Cfg c = cfg; c.avail = sg.avail;
    typedef typename W::template Var<uint64_t> U64;
    typedef typename W::template Var<uint32_t> U;
    U64 a, b, s, e;
    U nf, no, fl, ms, need, nr, nfar;
    const uint64_t hint = hint_in ? *hint_in : sg.hint;
    W::each([&](int l) {
        a[l] = sg.begin + (uint64_t)l * SUB; b[l] = a[l] + SUB < sg.end ? a[l] + SUB : sg.end;
        if (a[l] >= sg.end) { a[l] = b[l] = sg.end; }
        // the guess: first plausible header in the lane's KiB (lane 0 of a segment whose start is known: that)
        uint64_t g = NONE;
        if (l == 0 && hint != NONE) g = hint;
        else {
            // SCAN positions per step from dwords held in registers: the integer fields of a header that have a range (block
            // size, reference ids, sequence length, positions >= -1) reject nearly every position without touching memory
            // again; the survivors of a step are a bit mask per lane and get the full three-records-deep test one per loop
            // trip, every lane its own — so the wave's trip count is the LARGEST number of survivors a lane has to try, not
            // the number of positions at which ANY lane has one (the per-position form ran the deep test ~50 times per
            // step for the wave: 0.42 ms per 200 MB batch, ten times the record walk itself).
            for (uint64_t p = a[l]; p < b[l] && g == NONE; p += SCAN) {
                uint32_t d[SCAN / 4 + 9];
#pragma unroll
                for (int j = 0; j < SCAN / 4 + 9; ++j) d[j] = p + 4 * j + 4 <= c.avail + 32 ? rd32(c.buf + p + 4 * j) : 0u;   // (buffers carry >= 64 bytes of slack)
                const uint32_t lim = b[l] - p < (uint64_t)SCAN ? (uint32_t)(b[l] - p) : (uint32_t)SCAN;
                uint64_t cand = 0;
#pragma unroll
                for (uint32_t k = 0; k < (uint32_t)SCAN; ++k) {
                    const uint32_t sh = (k & 3) * 8, j = k >> 2;
                    auto fld = [&](uint32_t q) { return sh ? (d[j + q] >> sh) | (d[j + q + 1] << (32 - sh)) : d[j + q]; };   // dword at byte p + k + 4 q
                    const bool ok = fld(0) - 33u <= (1u << 27) - 33u              // block size
                                    && fld(1) + 1u <= (uint32_t)c.n_ref           // refID in -1 .. n_ref - 1
                                    && (int32_t)fld(2) >= -1                      // pos
                                    && (fld(3) & 0xffu) != 0u                     // l_read_name >= 1
                                    && fld(5) <= (1u << 27)                       // l_seq
                                    && fld(6) + 1u <= (uint32_t)c.n_ref           // next refID
                                    && (int32_t)fld(7) >= -1;                     // next pos
                    cand |= (uint64_t)(ok && k < lim) << k;
                }
                while (cand != 0 && g == NONE) {
                    const uint32_t k = (uint32_t)ctz64(cand);
                    cand &= cand - 1;
                    if (plausible(c, p + k, 3)) g = p + k;
                }
            }
        }
        s[l] = g; need[l] = 1; e[l] = 0; nf[l] = no[l] = fl[l] = ms[l] = nr[l] = nfar[l] = 0;
    });
    typename W::template Var<LaneWalk> lw;
    U active;
    U64 wp, zero64;
    W::each([&](int l) { zero64[l] = 0; });
    for (int round = 0; round < 70; ++round) {
        W::each([&](int l) {
            active[l] = 0; wp[l] = 0;
            LaneWalk &w = lw[l];
            w.e = 0; w.n_first = w.n_other = w.n_far = w.flags = w.max_span = w.n_rec = 0; w.key_first = NONE; w.key_last = 0; w.unsorted = w.n_long = 0;
            if (!need[l]) return;
            e[l] = 0; nf[l] = no[l] = fl[l] = ms[l] = nr[l] = nfar[l] = 0;
            if (s[l] == NONE) return;                                     // nothing known to start here: no information
            if (s[l] >= b[l]) { e[l] = s[l]; return; }                     // the chain passes over this lane's KiB
            active[l] = 2; wp[l] = s[l];                                   // (2: walks in this round; run_lanes clears the low states)
        });
        U walked;
        W::each([&](int l) { walked[l] = active[l]; });
        run_lanes<W, false>(c, active, wp, b, lw, nullptr, nullptr, nullptr, zero64, zero64, zero64);
        W::each([&](int l) {
            if (!walked[l]) return;
            const LaneWalk &w = lw[l];
            e[l] = w.e; nf[l] = w.n_first; no[l] = w.n_other; fl[l] = w.flags; ms[l] = w.max_span; nr[l] = w.n_rec; nfar[l] = w.n_far;
        });
        // where the chain of the lanes before l ends = prefix maximum of their ends
        const U64 pm = W::excl_scan_max64(e);
        W::each([&](int l) {
            uint64_t t = s[l];
            if (l > 0 && pm[l] != 0) t = pm[l];                           // (0: no lane before this one knows anything — keep the guess)
            need[l] = t != s[l]; s[l] = t;
        });
        if (!W::ballot_ne(need, 0u)) break;
    }
    uint32_t tf = 0, to = 0;
    const U ef = W::excl_scan(nf, &tf);
    const U eo = W::excl_scan(no, &to);
    uint32_t tr = 0, tfar = 0;
    const U er = W::excl_scan(nr, &tr);
    const U efar = W::excl_scan(nfar, &tfar);
    (void)ef; (void)eo; (void)er; (void)efar;
    const uint64_t flags = W::ballot_ne(fl, 0u);
    uint32_t allf = 0, mspan = 0;
    if (flags) allf = W::reduce_or(fl);
    mspan = W::reduce_max(ms);
    const uint64_t last_e = W::reduce_max64(e);
    U64 own;
    W::each([&](int l) { own[l] = s[l] != NONE && s[l] < b[l] ? s[l] : NONE; });
    const uint64_t first_start = W::reduce_min64(own);                   // the first record the segment owns (NONE: none starts here)
    W::each([&](int l) {
        lanes[l].start = s[l]; lanes[l].n_first = nf[l]; lanes[l].n_other = no[l]; lanes[l].n_far = nfar[l]; lanes[l].pad = 0;
        if (l == 0) {
            sg.used_start = first_start; sg.e_last = last_e; sg.n_first = tf; sg.n_other = to; sg.flags = allf; sg.max_span = mspan; sg.n_rec = tr; sg.n_far = tfar;
            if (hint_in) sg.hint = hint;
        }
    });
    return WalkOut{first_start, last_e, tf, to, tfar, tr, allf, mspan};
}

// pass 2: the same lanes write their runs; sg.base_first / base_other say where the segment's runs go
template <class W>
PW_FN void emit_segment(const Cfg &cfg, const Seg &sg, const LaneOut *lanes, pd_iv *first, pd_iv *other, pd_iv *far, SegOut *seg_out = nullptr)
{
    Cfg c = cfg; c.avail = sg.avail;
    typedef typename W::template Var<uint32_t> U;
    U nf, no, nfar;
    W::each([&](int l) { nf[l] = lanes[l].n_first; no[l] = lanes[l].n_other; nfar[l] = lanes[l].n_far; });
    uint32_t tf = 0, to = 0, tfar = 0;
    const U ef = W::excl_scan(nf, &tf);
    const U eo = W::excl_scan(no, &to);
    const U efar = W::excl_scan(nfar, &tfar);
    typedef typename W::template Var<uint64_t> U64;
    U64 kf, kl; U bad, nl;
    typename W::template Var<LaneWalk> lw;
    U active;
    U64 wp, bb, of, oo, ofr;
    W::each([&](int l) {
        kf[l] = NONE; kl[l] = 0; bad[l] = 0; nl[l] = 0;
        LaneWalk &w = lw[l];
        w.e = 0; w.n_first = w.n_other = w.n_far = w.flags = w.max_span = w.n_rec = 0; w.key_first = NONE; w.key_last = 0; w.unsorted = w.n_long = 0;
        const uint64_t a = sg.begin + (uint64_t)l * SUB;
        const uint64_t b = a + SUB < sg.end ? a + SUB : sg.end;
        const uint64_t s = lanes[l].start;
        active[l] = 0; wp[l] = 0; bb[l] = b;
        of[l] = sg.base_first + ef[l]; oo[l] = sg.base_other + eo[l]; ofr[l] = sg.base_far + efar[l];
        if (a >= sg.end || s == NONE || s >= b || (nf[l] | no[l] | nfar[l]) == 0) return;
        active[l] = 1; wp[l] = s;
    });
    U walked;
    W::each([&](int l) { walked[l] = active[l]; });
    run_lanes<W, true>(c, active, wp, bb, lw, first, other, far, of, oo, ofr);
    W::each([&](int l) {
        if (!walked[l]) return;
        const LaneWalk &w = lw[l];
        kf[l] = w.key_first; kl[l] = w.key_last; bad[l] = w.unsorted; nl[l] = w.n_long;
    });
    if (seg_out) {
        // the order across the lanes: a lane's first key against the largest key of the lanes before it
        const U64 pm = W::excl_scan_max64(kl);
        W::each([&](int l) { if (kf[l] != NONE && kf[l] < pm[l]) bad[l] = 1; });
        const uint64_t first_key = W::reduce_min64(kf), last_key = W::reduce_max64(kl);
        const uint32_t any_bad = W::reduce_or(bad);
        uint32_t n_long = 0;
        (void)W::excl_scan(nl, &n_long);
        W::each([&](int l) { if (l == 0) { seg_out->first_key = first_key; seg_out->last_key = last_key; seg_out->unsorted = any_bad; seg_out->n_long = n_long; } });
    }
}

// The host side after pass 1 (and after every repeat): is every segment's speculated first record the one the chain of
// the segments before it arrives at?  Segments that are not get the right start as their hint and their index in `redo`
// (they walk again).  A segment right after one that must walk again cannot be checked in this round — the end of its
// predecessor's chain is not known yet — and is taken at its word until the next round; the segments after it are
// checked against ITS chain, so an isolated wrong guess costs one repeat, not one repeat per segment behind it.
// A confirmed chain that stops inside a unit ends the unit: its remaining segments are marked WF_BAD (a record that
// cannot be one) or WF_MORE (a hand-over to the host: the record runs past the bytes, a CIGAR in the CG tag).
// Returns the number of segments to walk again; 0 = every start is confirmed.
template <class Vec, class Redo>
inline uint32_t check_chain(Vec &segs, Redo *redo)
{
    redo->clear();
    uint64_t E = 0;
    bool known = true;
    uint32_t dead = 0;                                   // flags for the rest of a unit whose chain has stopped
    for (size_t j = 0; j < segs.size(); ++j) {
        Seg &s = segs[j];
        bool confirmed = true;
        if (s.unit_first) { E = 0; known = true; dead = 0; }
        else if (dead) { s.flags = dead; s.n_first = s.n_other = s.n_far = s.n_rec = 0; continue; }
        else if (known) {
            const bool none_expected = E >= s.end;
            const bool ok = none_expected ? s.used_start == NONE : s.used_start == E;
            if (!ok) { s.hint = E; redo->push_back((uint32_t)j); known = false; continue; }       // its own e_last is stale
        } else confirmed = false;
        if (confirmed) {
            if (s.e_last > E) E = s.e_last;
            if (E >= STOPPED) dead = (s.flags & WF_BAD) || !(s.flags & (WF_MORE | WF_HOST)) ? (uint32_t)WF_BAD : (uint32_t)WF_MORE;
        } else if (s.e_last != 0 && s.e_last < STOPPED) { E = s.e_last; known = true; }          // taken at its word for this round
    }
    return (uint32_t)redo->size();
}

// The same confirmation ON THE DEVICE, by one wave (round 5: a batch no longer waits for the host between its two passes).  The
// wave goes through the segments in order, 64 at a time: the end of the chain before every segment is a prefix maximum of the
// ends before it (restarting at the first segment of a unit), so a stretch of segments whose guesses were all right is confirmed
// by ONE scan and compare.  The first segment of a stretch that starts somewhere else walks again at once — `rewalk(j, start)`,
// the whole wave, from the corrected start — and the stretch is compared again: what comes out is the sequential chain, which is
// also the fixed point check_chain's rounds arrive at.  Then every segment gets the places of its runs (running sums of the
// counts: Seg::base_first / base_other), which is all pass 2 needs.
// Anything out of the ordinary is NOT handled here: a member that did not inflate (or that the device leaves to zlib), a flag on
// any segment (a record that cannot be one, that runs past the unit's bytes, a CIGAR in the CG tag), more repeats than
// `max_redo`, far runs, or more runs than the batch's arrays hold set ChainOut::slow — pass 2 then writes nothing and the host
// goes through the batch the way it always did (check_chain, unit outcomes, hand-backs).
struct ChainOut { uint64_t n_first, n_other, n_rec; uint32_t max_span, slow, n_redo, pad; uint64_t first_start, next_start /* unit 0: pd_decode_result's */, pad2; };   // 64 bytes
enum { CH_MEMBER = 1, CH_FLAG = 2, CH_REDO = 4, CH_ROOM = 8, CH_FAR = 16 };

template <class W, class RW>
PW_FN void chain_device(Seg *segs, uint32_t n_seg, const int *member_status, uint32_t n_members, uint64_t cap_first, uint64_t cap_other,
                        uint32_t max_redo, RW rewalk, ChainOut *out)
{
    typedef typename W::template Var<uint64_t> U64;
    typedef typename W::template Var<uint32_t> U;
    uint32_t slow = 0, n_redo = 0, max_span = 0;
    for (uint32_t b0 = 0; b0 < n_members && !slow; b0 += 64) {
        U bad;
        W::each([&](int l) { bad[l] = b0 + (uint32_t)l < n_members && member_status[b0 + (uint32_t)l] != 0 ? 1u : 0u; });
        if (W::ballot_ne(bad, 0u)) slow |= CH_MEMBER;
    }
    uint64_t E = 0, nf = 0, no = 0, nr = 0;
    uint64_t fs0 = NONE, e0 = 0; uint32_t units_seen = 0;           // unit 0: the first record it owns, and where its chain ends
    for (uint32_t j0 = 0; j0 < n_seg && !slow; j0 += 64) {
        const uint32_t cnt = n_seg - j0 < 64u ? n_seg - j0 : 64u;
        U64 us, el, en;
        U uf, fl, cf, co, cr, ms;
        W::each([&](int l) {
            us[l] = NONE; el[l] = 0; en[l] = 0; uf[l] = fl[l] = cf[l] = co[l] = cr[l] = ms[l] = 0;
            if ((uint32_t)l >= cnt) return;
            const Seg &s = segs[j0 + (uint32_t)l];
            us[l] = s.used_start; el[l] = s.e_last; en[l] = s.end; uf[l] = s.unit_first ? 1u : 0u; cf[l] = s.n_first; co[l] = s.n_other; cr[l] = s.n_rec; ms[l] = s.max_span;
            fl[l] = s.flags | (s.n_far ? 0x80000000u : 0u);
        });
        // A group may hold segments of many units (region fetch: a unit per joined index chunk, often one segment long): the chain's end
        // before a segment is the maximum of the ends of ITS unit's segments before it — a SEGMENTED prefix maximum over the group, the
        // unit that was open when the group began continuing with the carried end E.  head1[l] = 1 + the lane of the last first-of-unit
        // segment at or before l (0: none in this group).
        U h1;
        W::each([&](int l) { h1[l] = uf[l] ? (uint32_t)l + 1u : 0u; });
        const U head1 = W::incl_scan_max(h1);
        for (;;) {
            const U64 pm = W::seg_excl_scan_max64(el, head1);
            U64 eb;
            U bad;
            W::each([&](int l) {
                eb[l] = head1[l] == 0u && E > pm[l] ? E : pm[l];
                const bool chk = (uint32_t)l < cnt && !uf[l];             // (a unit's first segment has nothing before it to be checked against)
                const uint64_t expect = eb[l] >= en[l] ? NONE : eb[l];
                bad[l] = chk && us[l] != expect ? 1u : 0u;
            });
            const uint64_t bm = W::ballot_ne(bad, 0u);
            // a flag counts once its segment's start is confirmed (a wrong guess usually ends in garbage and carries one: the repeat
            // clears it); confirmed are the segments before the first one that does not fit
            const int jb = bm ? ctz64(bm) : 64;
            const uint64_t confirmed = jb >= 64 ? ~0ull : (1ull << jb) - 1;
            const uint64_t flm = W::ballot_ne(fl, 0u) & confirmed;
            if (flm) { U f2; W::each([&](int l) { f2[l] = (confirmed >> l) & 1 ? fl[l] : 0u; }); slow |= W::reduce_or(f2) & 0x80000000u ? CH_FAR : CH_FLAG; break; }
            if (!bm) {
                // the end the chain has reached behind this group's last segment (the carry for the next group)
                const int last = (int)cnt - 1;
                const uint64_t pl = W::bcast64(pm, last), ll = W::bcast64(el, last);
                const bool open_last = W::bcast(head1, last) == 0u;
                uint64_t e1 = pl > ll ? pl : ll;
                if (open_last && E > e1) e1 = E;
                // unit 0's first record and chain end, as pd_decode_result reports them
                uint32_t n_heads = 0;
                const U upc = W::excl_scan(uf, &n_heads);                            // units begun in this group before lane l
                U64 o0, e00;
                W::each([&](int l) {
                    const bool in0 = (uint32_t)l < cnt && units_seen + upc[l] + uf[l] == 1u;
                    o0[l] = in0 ? us[l] : NONE; e00[l] = in0 ? el[l] : 0ull;
                });
                const uint64_t f = W::reduce_min64(o0), m0 = W::reduce_max64(e00);
                if (fs0 == NONE) fs0 = f;
                if (m0 > e0) e0 = m0;
                units_seen += n_heads;
                E = e1;
                break;
            }
            if (++n_redo > max_redo) { slow |= CH_REDO; break; }
            const uint64_t start = W::bcast64(eb, jb), seg_end = W::bcast64(en, jb);
            const WalkOut w = rewalk(j0 + (uint32_t)jb, start);
            if (w.flags) { slow |= CH_FLAG; break; }
            if (w.n_far) { slow |= CH_FAR; break; }
            if (w.used_start != (start >= seg_end ? NONE : start)) { slow |= CH_REDO; break; }      // (a walk that does not start where it was told to: the host looks at it)
            W::each([&](int l) { if (l == jb) { us[l] = w.used_start; el[l] = w.e_last; cf[l] = w.n_first; co[l] = w.n_other; cr[l] = w.n_rec; ms[l] = w.max_span; fl[l] = 0; } });
        }
        if (slow) break;
        uint32_t tf = 0, to = 0, tr = 0;
        const U ef = W::excl_scan(cf, &tf);
        const U eo = W::excl_scan(co, &to);
        (void)W::excl_scan(cr, &tr);
        W::each([&](int l) {
            if ((uint32_t)l >= cnt) return;
            Seg &s = segs[j0 + (uint32_t)l];
            s.base_first = nf + ef[l]; s.base_other = no + eo[l]; s.base_far = 0;
        });
        nf += tf; no += to; nr += tr;
        const uint32_t mspan = W::reduce_max(ms);
        if (mspan > max_span) max_span = mspan;
    }
    if (nf > cap_first || no > cap_other) slow |= CH_ROOM;
    W::each([&](int l) {
        if (l == 0) { out->n_first = nf; out->n_other = no; out->n_rec = nr; out->max_span = max_span; out->slow = slow; out->n_redo = n_redo; out->pad = 0; out->first_start = fs0; out->next_start = e0 ? e0 : NONE; out->pad2 = 0; }
    });
}

} // namespace pdb2
#endif
