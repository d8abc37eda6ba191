/* oracle/pd_oracle.c — CPU restatement of the reference's per-base depth hot path.
 *
 * TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library, and only as the checker / the reported CPU baseline.  The product
 * path (pandepth_amd/, include/) never links, loads or calls it.
 *
 * Every function restates one piece of /root/reference/src/PanDepth.cpp ("PD") the way the
 * reference computes it — one `depth[p]++` per covered base, one compare+add per region base —
 * NOT the way the HIP path computes it (difference array + prefix sum), so that agreement
 * between the two is evidence and not tautology.
 *
 * Pinned against the compiled reference: tests/test_oracle_golden.py replays every fixture in
 * tests/golden/ (outputs of oracle/_ref/pandepth_ref) through these functions.
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>

/* PD:436-460 (identical copies at PD:572-596, 732-760, 4611-4670, 4681-4710): record filter,
 * CIGAR walk from core.pos, one increment per reference base of every M/=/X run; D/N advance
 * only; I/S/H/P ignored.  `depth` is the concatenation of per-contig arrays, contig t starting
 * at contig_off[t] (each padded like the reference's new SiteInfo[len+500]).  The position
 * cursor is int32 like the reference's (PD:432-433).  Returns the number of records that
 * passed the filter. */
int64_t pdo_walk_records(int64_t n_rec, const int32_t *tid, const int32_t *pos,
                         const uint16_t *flag, const uint8_t *mapq,
                         const int64_t *cigar_off, const uint32_t *cigar,
                         uint32_t flag_mask, int32_t min_mapq,
                         uint32_t *depth, const int64_t *contig_off)
{
    int64_t kept = 0;
    for (int64_t r = 0; r < n_rec; ++r) {
        if (flag[r] & flag_mask) continue;              /* PD:436 */
        if ((int32_t)mapq[r] < min_mapq) continue;      /* PD:437 */
        if (tid[r] < 0) continue;                       /* unmapped/unplaced: reference UB, skipped */
        ++kept;
        uint32_t *d = depth + contig_off[tid[r]];
        int32_t cur = pos[r];                           /* PD:439 */
        for (int64_t c = cigar_off[r]; c < cigar_off[r + 1]; ++c) {
            int op = (int)(cigar[c] & 0xf);             /* bam_cigar_op, sam.h:104-115 */
            int32_t len = (int32_t)(cigar[c] >> 4);     /* bam_cigar_oplen */
            switch (op) {
            case 0: case 7: case 8: {                   /* M = X  PD:446-454 */
                int32_t stop = cur + len;
                for (; cur < stop; ++cur) d[cur]++;
                break;
            }
            case 2: case 3:                             /* D N    PD:455-458 */
                cur += len;
                break;
            default:
                break;
            }
        }
    }
    return kept;
}

/* The same increment loop driven by pre-expanded (tid, beg, end) runs — the unit the C-ABI
 * boundary carries (include/pandepth_amd.h, pd_iv).  [beg,end) is exactly the range PD:449-452
 * covers for one M/=/X run. */
void pdo_add_intervals(int64_t n, const int32_t *iv /* n x {tid,beg,end} */,
                       uint32_t *depth, const int64_t *contig_off)
{
    for (int64_t i = 0; i < n; ++i) {
        int32_t t = iv[3 * i], b = iv[3 * i + 1], e = iv[3 * i + 2];
        uint32_t *d = depth + contig_off[t];
        for (int32_t p = b; p < e; ++p) d[p]++;
    }
}

/* DataClass.h:85-88: `unsigned Depth:18` — every increment wraps modulo 2^18.  Because 2^18
 * divides 2^32, masking the 32-bit count afterwards gives the same cell value. */
void pdo_wrap18(uint32_t *depth, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) depth[i] &= 0x3FFFFu;
}

/* PD:329-348 StatChrDepthLowMEM / PD:295-327 StatChrDepthWin, per CDS entry: for the 1-based
 * inclusive pair (first, second) visit cells [first-1, second) of one contig; where
 * depth >= minDep count the site and add the depth.  cover is `int` in the reference
 * (DataClass.h:63) — int32 here, wrap-around identical; depth sum is uint64. */
void pdo_stat_regions(const uint32_t *depth, const int64_t *contig_off, int64_t n_reg,
                      const int32_t *reg /* n_reg x {tid, first, second} */, uint32_t min_dep,
                      int32_t *cover, uint64_t *sum)
{
    for (int64_t r = 0; r < n_reg; ++r) {
        const uint32_t *d = depth + contig_off[reg[3 * r]];
        int32_t c = 0; uint64_t s = 0;
        for (int32_t i = reg[3 * r + 1] - 1; i < reg[3 * r + 2]; ++i) {
            if (d[i] >= min_dep) { c++; s += d[i]; }
        }
        cover[r] = c; sum[r] = s;
    }
}

/* PD:4366-4389 (copies at PD:4835-4858, 3119-3142): the mode-6 sweep for -w < 150.
 * `for (j = 1; j < len; j += w)` visits cells [j-1, min(j-1+w, len)); per-window accumulators
 * are `int`.  Returns the number of windows written (a trailing 1-base window is not
 * produced because of `j < len`). */
int64_t pdo_sweep_windows(const uint32_t *d, int32_t len, int32_t w, uint32_t min_dep,
                          int32_t *win_start1, int32_t *win_end, int32_t *cover, int32_t *sum)
{
    int64_t k = 0;
    for (int32_t j = 1; j < len; j += w) {
        int32_t s = j - 1, e = s + w;
        if (e > len) e = len;
        int32_t c = 0, t = 0;
        for (; s < e; ++s) {
            if (d[s] >= min_dep) { c++; t += (int32_t)d[s]; }
        }
        win_start1[k] = j; win_end[k] = e; cover[k] = c; sum[k] = t;
        ++k;
    }
    return k;
}

/* bam_endpos as htslib 1.19 computes it (sam.c: rlen = unmapped ? 1 : bam_cigar2rlen; a zero
 * reference length counts as 1).  Needed by the read-selection rules that decide WHICH reads
 * the reference feeds to the loop above (PD:430-434 index fetch, PD:4616-4646 sorted stream). */
int32_t pdo_endpos(int32_t pos, uint16_t flag, int64_t n_cigar, const uint32_t *cigar)
{
    int32_t rlen = 0;
    if (!(flag & 4) && n_cigar > 0) {
        for (int64_t c = 0; c < n_cigar; ++c) {
            int op = (int)(cigar[c] & 0xf);
            if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rlen += (int32_t)(cigar[c] >> 4);
        }
    } else {
        rlen = 1;
    }
    if (rlen == 0) rlen = 1;
    return pos + rlen;
}
