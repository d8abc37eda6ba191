// tests/harness/bamwalk_check.cpp — TEST INFRASTRUCTURE: the product's wave-parallel BAM record walk
// (pandepth_amd/csrc/pd_bamwalk.h, the source the gfx950 kernels compile) with its 64 lanes emulated on the host,
// against the product's sequential host reader (pandepth_amd/host/bam.cpp) on the same file: the same reads must
// pass the filter and the same M/=/X runs must come out, first runs in file order.
//   bamwalk_check [-x flagmask] [-q mapq] [-noguess] file.bam ...
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include <algorithm>
#include <tuple>
#include <vector>
#include "../../pandepth_amd/host/bam.h"
#include "../../pandepth_amd/csrc/pd_bamwalk.h"

using namespace pdb2;
typedef std::tuple<int32_t, int32_t, int32_t> Run;

static bool gunzip_all(const char *path, std::vector<uint8_t> &out)
{
    gzFile g = gzopen(path, "rb");
    if (!g) return false;
    uint8_t buf[1 << 16]; int n;
    while ((n = gzread(g, buf, sizeof buf)) > 0) out.insert(out.end(), buf, buf + n);
    gzclose(g);
    return true;
}

// every segment's output offsets and the totals, once the chain is confirmed (pdb2::check_chain)
static void offsets(std::vector<Seg> &segs, uint64_t *tot_first, uint64_t *tot_other, uint64_t *tot_far, uint32_t *flags, uint32_t *max_span)
{
    uint64_t f = 0, o = 0, fr = 0; *flags = 0; *max_span = 0;
    for (auto &s : segs) {
        s.base_first = f; s.base_other = o; s.base_far = fr; f += s.n_first; o += s.n_other; fr += s.n_far;
        *flags |= s.flags; if (s.max_span > *max_span) *max_span = s.max_span;
    }
    *tot_first = f; *tot_other = o; *tot_far = fr;
}

int main(int argc, char **argv)
{
    uint32_t mask = 1796, near_span = 0xFFFFFFFFu; int mapq = 0; bool guess_all = true; int seg_kb = 64;
    int bad_files = 0;
    for (int a = 1; a < argc; ++a) {
        if (!strcmp(argv[a], "-x")) { mask = (uint32_t)atoi(argv[++a]); continue; }
        if (!strcmp(argv[a], "-q")) { mapq = atoi(argv[++a]); continue; }
        if (!strcmp(argv[a], "-noguess")) { guess_all = false; continue; }
        if (!strcmp(argv[a], "-seg")) { seg_kb = atoi(argv[++a]); continue; }
        if (!strcmp(argv[a], "-near")) { near_span = (uint32_t)atoi(argv[++a]); continue; }
        const char *path = argv[a];
        std::vector<uint8_t> d;
        if (!gunzip_all(path, d)) { fprintf(stderr, "cannot read %s\n", path); return 2; }
        // expected: the host reader
        pdh::AlnReader rd; std::string err;
        if (!rd.open(path, &err)) { fprintf(stderr, "%s: %s\n", path, err.c_str()); return 2; }
        const pdh::AlnHeader &h = rd.header();
        std::vector<Run> exp_first, exp_other;
        pdh::AlnRec r; int k; uint64_t n_rec = 0;
        while ((k = rd.next(&r)) == 1) {
            ++n_rec;
            if ((r.flag & mask) || (int)r.mapq < mapq || r.tid < 0 || r.tid >= (int32_t)h.names.size()) continue;
            int32_t cur = r.pos; bool first_done = false, moved = false;
            for (uint32_t i = 0; i < r.n_cigar; ++i) {
                const uint32_t op = r.cigar[i] & 0xf; const int32_t len = (int32_t)(r.cigar[i] >> 4);
                if (op == 0 || op == 7 || op == 8) { (!first_done && !moved ? exp_first : exp_other).push_back(Run(r.tid, cur, cur + len)); first_done = true; cur += len; }
                else if (op == 2 || op == 3) { cur += len; moved = true; }
            }
        }
        // the BAM header inside the inflated bytes: magic, l_text, text, n_ref, (l_name, name, l_ref)*
        size_t o = 4; uint32_t l_text = rd32(d.data() + o); o += 4 + l_text;
        const uint32_t n_ref = rd32(d.data() + o); o += 4;
        std::vector<uint32_t> lens(n_ref); std::vector<uint8_t> on(n_ref, 1);
        for (uint32_t i = 0; i < n_ref; ++i) { const uint32_t ln = rd32(d.data() + o); o += 4 + ln; lens[i] = rd32(d.data() + o); o += 4; }
        d.resize(d.size() + 64, 0);
        Cfg c; c.buf = d.data(); c.avail = d.size() - 64; c.n_ref = (int32_t)n_ref; c.contig_len = lens.data(); c.contig_on = on.data();
        c.flag_mask = mask; c.min_mapq = mapq; c.span_off = nullptr; c.spans = nullptr; c.near_span = near_span;
        std::vector<Seg> segs;
        const uint64_t SB = (uint64_t)seg_kb * 1024;
        for (uint64_t b = o; b < c.avail; b += SB) {
            Seg s; memset(&s, 0, sizeof s);
            s.begin = b; s.end = std::min<uint64_t>(b + SB, c.avail); s.hint = b == o ? o : NONE; s.unit_first = b == o; s.avail = c.avail;
            segs.push_back(s);
        }
        if (segs.empty()) { printf("%s: no records\n", path); continue; }
        std::vector<LaneOut> lanes(segs.size() * 64);
        for (size_t j = 0; j < segs.size(); ++j) walk_segment<pdw::HostWave>(c, segs[j], &lanes[j * 64]);
        uint64_t tf = 0, to = 0, tfar = 0; uint32_t fl = 0, ms = 0; int rounds = 0;
        std::vector<uint32_t> redo;
        while (check_chain(segs, &redo) > 0 && rounds < 24) {
            ++rounds;
            for (uint32_t j : redo) walk_segment<pdw::HostWave>(c, segs[j], &lanes[(size_t)j * 64]);
        }
        offsets(segs, &tf, &to, &tfar, &fl, &ms);
        (void)guess_all;
        std::vector<pd_iv> first(tf + 1), other(to + 1), far(tfar + 1);
        for (size_t j = 0; j < segs.size(); ++j) emit_segment<pdw::HostWave>(c, segs[j], &lanes[j * 64], first.data(), other.data(), far.data());
        std::vector<Run> got_first, got_other;
        for (uint64_t i = 0; i < tf; ++i) got_first.push_back(Run(first[i].tid, first[i].beg, first[i].end));
        for (uint64_t i = 0; i < to; ++i) got_other.push_back(Run(other[i].tid, other[i].beg, other[i].end));
        for (uint64_t i = 0; i < tfar; ++i) got_other.push_back(Run(far[i].tid, far[i].beg, far[i].end));
        const bool same_first = got_first == exp_first;                  // file order
        // the later runs come out as two streams (near / far from their read's start), each in file order: same multiset
        std::sort(got_other.begin(), got_other.end()); std::sort(exp_other.begin(), exp_other.end());
        const bool same_other = got_other == exp_other;
        to += tfar;
        printf("%s: %llu records, %zu segments (%d corrected in %d extra rounds), first runs %llu (expected %zu) %s, other runs %llu (expected %zu) %s, flags %u, max span %u\n",
               path, (unsigned long long)n_rec, segs.size(), (int)redo.size(), rounds, (unsigned long long)tf, exp_first.size(), same_first ? "identical" : "DIFFERENT",
               (unsigned long long)to, exp_other.size(), same_other ? "identical" : "DIFFERENT", fl, ms);
        if (!same_first || !same_other || fl) ++bad_files;
    }
    return bad_files ? 1 : 0;
}
