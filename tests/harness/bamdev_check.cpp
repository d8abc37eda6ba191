// tests/harness/bamdev_check.cpp — TEST INFRASTRUCTURE: the record walk / run extraction that the
// gfx950 kernels use (pandepth_amd/csrc/pd_bamdev_core.h) run on the host over a whole inflated BAM
// and compared with the host reader (pandepth_amd/host/bam.cpp) + the CIGAR walk of PD:438-460.
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <tuple>
#include <vector>
#include "../../pandepth_amd/host/bam.h"
#include "../../pandepth_amd/csrc/pd_bamdev_core.h"

typedef std::tuple<int32_t, int32_t, int32_t> Run;

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: bamdev_check in.bam flag_mask min_mapq\n"); return 2; }
    const uint32_t flag_mask = (uint32_t)atoi(argv[2]); const int min_mapq = atoi(argv[3]);
    std::string err;
    pdh::AlnReader rd;
    if (!rd.open(argv[1], &err)) { fprintf(stderr, "%s\n", err.c_str()); return 2; }
    const uint64_t hdr_end_voff = rd.tell();
    const pdh::AlnHeader &h = rd.header();
    // reference: host reader
    std::vector<Run> ref;
    pdh::AlnRec r; long nrec = 0;
    while (rd.next(&r) > 0) {
        ++nrec;
        if ((r.flag & flag_mask) || (int)r.mapq < min_mapq || r.tid < 0 || r.tid >= (int)h.lens.size() || h.lens[r.tid] < 2) continue;
        int32_t cur = r.pos;
        for (uint32_t i = 0; i < r.n_cigar; ++i) {
            const uint32_t op = r.cigar[i] & 0xf; const int32_t len = (int32_t)(r.cigar[i] >> 4);
            if (op == 0 || op == 7 || op == 8) { ref.emplace_back(r.tid, cur, cur + len); cur += len; }
            else if (op == 2 || op == 3) cur += len;
        }
    }
    // inflate the whole file into one buffer, remembering where the first record starts
    pdh::BgzfReader bg;
    if (!bg.open(argv[1], &err)) return 2;
    std::vector<uint8_t> all;
    uint64_t first_rec = 0; bool found = false;
    for (;;) {
        size_t av = 0;
        const uint64_t v = bg.tell();
        const uint8_t *p = bg.peek(&av);
        if (!p || !av) break;
        if (!found && (v >> 16) == (hdr_end_voff >> 16)) { first_rec = all.size() - (v & 0xffff) + (hdr_end_voff & 0xffff); found = true; }
        all.insert(all.end(), p, p + av);
        bg.consume(av);
    }
    if (!found) first_rec = all.size();
    pdb::Unit u; u.start = first_rec; u.stop = all.size(); u.avail = all.size(); u.rec_base = 0; u.n_rec = 0; u.status = 0;
    std::vector<uint64_t> off(all.size() / 36 + 8);
    pdb::walk_unit(all.data(), u, off.data(), off.size());
    if (u.status != 0 || (long)u.n_rec != nrec) { fprintf(stderr, "walk: status %d, %u records vs %ld\n", u.status, u.n_rec, nrec); return 1; }
    pdb::Filter f{flag_mask, min_mapq, (int32_t)h.lens.size()};
    std::vector<Run> got;
    std::vector<pd_iv> firsts(u.n_rec);
    for (uint32_t i = 0; i < u.n_rec; ++i) {
        const int rc = pdb::parse_record(all.data() + off[i], f, h.lens.data(), &firsts[i],
                                         [&](pd_iv v) { got.emplace_back(v.tid, v.beg, v.end); });
        if (rc) { fprintf(stderr, "record %u needs the host\n", i); return 1; }
        if (firsts[i].beg < firsts[i].end) got.emplace_back(firsts[i].tid, firsts[i].beg, firsts[i].end);
        if (i && std::make_pair(firsts[i].tid, firsts[i].beg) < std::make_pair(firsts[i - 1].tid, firsts[i - 1].beg) && h.sorted_coordinate()) {
            fprintf(stderr, "first-run array not sorted at record %u\n", i); return 1;
        }
    }
    std::sort(ref.begin(), ref.end()); std::sort(got.begin(), got.end());
    printf("%ld records, %zu runs (reference %zu): %s\n", nrec, got.size(), ref.size(), ref == got ? "identical" : "DIFFERENT");
    return ref == got ? 0 : 1;
}
