// pd_bamdev_core.h — BAM record walk + run extraction on inflated bytes, shared by the gfx950 kernels
// (pd_bgzf.hip) and the host (tests/harness/bamdev_check.cpp validates it against the host reader
// before it runs on a GPU).  Restates PD:436-460 (filter, CIGAR walk) on raw BAM bytes (SAM spec
// §4.2): block_size, refID, pos, l_read_name, mapq, bin, n_cigar_op, flag, l_seq, ... at fixed
// offsets, CIGAR after the read name.
#ifndef PD_BAMDEV_CORE_H_
#define PD_BAMDEV_CORE_H_
#include <stdint.h>
#include "../../include/pandepth_amd.h"

#if defined(__HIPCC__)
#define PDB_FN __host__ __device__ __forceinline__
#else
#define PDB_FN inline
#endif

namespace pdb {

// A unit = a record-aligned stretch of one file range inside the batch's inflated buffer:
// records START in [start, stop); the bytes up to avail (end of the last inflated block of the unit)
// may be read to finish the last record.
struct Unit {
    uint64_t start, stop, avail;
    uint64_t rec_base;        // first slot of this unit in the record-offset array (capacity-based)
    uint32_t n_rec;           // OUT: records found
    int32_t status;           // OUT: 0 ok, 1 = a record runs past `avail` (host decodes this unit), 2 = corrupt
};

PDB_FN uint32_t ld32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
PDB_FN uint32_t ld16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

// Walks the records of one unit, writing their offsets; bounded by the unit's byte range.
PDB_FN void walk_unit(const uint8_t *buf, Unit &u, uint64_t *rec_off, uint64_t rec_cap)
{
    uint64_t p = u.start;
    uint32_t n = 0;
    int32_t st = 0;
    while (p < u.stop) {
        if (p + 4 > u.avail) { st = 1; break; }
        const uint32_t bs = ld32(buf + p);
        if (bs < 32 || bs > (1u << 29)) { st = 2; break; }
        if (p + 4 + (uint64_t)bs > u.avail) { st = 1; break; }
        if ((uint64_t)n >= rec_cap) { st = 2; break; }
        rec_off[u.rec_base + n++] = p;
        p += 4 + (uint64_t)bs;
    }
    u.n_rec = n; u.status = st;
}

struct Filter { uint32_t flag_mask; int32_t min_mapq; int32_t n_contigs; };

// One record -> its first-run slot `first` (always written: an empty run (tid,pos,pos) when the
// record is filtered out or starts with a deletion/skip, so that the dense first-run array stays
// sorted by position and complete) and its remaining runs through `emit_other`.  Returns 0, or 1
// when the record needs the host (CIGAR kept in the CG tag, SAM spec §4.2.2).
template <class EmitOther>
PDB_FN int parse_record(const uint8_t *rec, const Filter &f, const uint32_t *contig_len, pd_iv *first, EmitOther emit_other)
{
    const uint32_t bs = ld32(rec);
    const int32_t tid = (int32_t)ld32(rec + 4);
    const int32_t pos = (int32_t)ld32(rec + 8);
    const uint32_t l_name = rec[12];
    const uint32_t mapq = rec[13];
    const uint32_t n_cigar = ld16(rec + 16);
    const uint32_t flag = ld16(rec + 18);
    const uint32_t l_seq = ld32(rec + 20);
    if (tid < 0 || tid >= f.n_contigs) {
        // unplaced reads (end of a sorted file): an empty run at the very end of the last contig keeps
        // the array sorted and every entry owned by a tile
        first->tid = f.n_contigs - 1; first->beg = first->end = (int32_t)contig_len[f.n_contigs - 1];
        return 0;
    }
    first->tid = tid; first->beg = pos; first->end = pos;
    if ((flag & f.flag_mask) || (int32_t)mapq < f.min_mapq) return 0;
    if (contig_len[tid] < 2) return 0;            // whole-contig modes give contigs shorter than 2 no bins (PD:4000)
    if (36 + l_name + 4 * (uint64_t)n_cigar > 4 + (uint64_t)bs) return 0;
    const uint8_t *cg = rec + 36 + l_name;
    if (n_cigar == 2 && (ld32(cg) & 0xf) == 4 && (ld32(cg) >> 4) == l_seq && (ld32(cg + 4) & 0xf) == 3) return 1;
    int32_t cur = pos;
    bool first_done = false, moved = false;
    for (uint32_t i = 0; i < n_cigar; ++i) {
        const uint32_t c = ld32(cg + 4 * i);
        const uint32_t op = c & 0xf;
        const int32_t len = (int32_t)(c >> 4);
        if (op == 0 || op == 7 || op == 8) {
            if (!first_done && !moved) { first->end = cur + len; first_done = true; }
            else emit_other(pd_iv{tid, cur, cur + len});
            first_done = true;
            cur += len;
        } else if (op == 2 || op == 3) { cur += len; moved = true; }
    }
    return 0;
}

} // namespace pdb
#endif
