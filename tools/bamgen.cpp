// tools/bamgen.cpp — TEST / BENCH TOOLING (not product code): a fast multi-threaded generator of coordinate-sorted
// short-read BAM files with a realistic payload, plus their .bai, for the end-to-end benchmarks (a 4.6 GB file in
// seconds instead of minutes with the numpy writer in tools/synth.py).
//
//   bamgen -o out.bam -n RECORDS [-s genome_scale] [-t threads] [-l level] [-S seed] [-z] [-Q 40] [--long]
//
//   -Q 40    unbinned qualities (40 levels with a position-dependent decay, HiSeq-2500-like) instead of the NovaSeq-like four bins in
//            runs: the same records in about twice the compressed bytes (~95 B per record instead of 53)
//   --long   long reads (HiFi / ONT shape) instead of 150-base ones: 10-20 kb, an insertion or deletion every 8-15 bases (~2 600 CIGAR
//            operations per read), 5 % soft-clipped, 2 % with one N gap of 50-180 kb (reference span up to 200 kb), 1 % with MORE THAN
//            65 535 operations — stored the way htslib writes them: <l_seq>S<ref_len>N in the record, the real CIGAR in a CG:B,I tag —
//            on the chromosomes of >= 1 Mb only; records are 20-450 KB and span BGZF members (a record starts a fresh member when it
//            does not fit the current one, as htslib's writer does)
//
// Workload (SURVEY.md §8d, configs[1] shape): the Capsicum-like genome of tools/synth.py scaled by -s (12 chromosomes +
// scaffolds), 150-base reads at uniformly drawn sorted positions with 1 % of the 100 kb blocks at depth multiplier
// {0, 0, 0.25, 2, 4}; CIGAR mix 85 % 150M, 5 % aM dD bM, 5 % aM iI bM, 4 % sS (150-s)M, 1 % aM nN bM; flags 2 % duplicate,
// 1 % secondary, 1 % QC-fail, 0.5 % unmapped-flagged; mapq {0, 20, 60}.  Payload like a real Illumina BAM: instrument-
// style read names, SEQ taken from a pseudo-random reference (overlapping reads share bases; 0.5 % mismatches), binned
// qualities with runs, NM / MD / AS / XS / RG tags.  BGZF members of <= 0xff00 bytes that end at record boundaries (as
// htslib writes them), compressed with libdeflate (dlopen; the compressor samtools uses) or zlib (-z) at level -l.
// The BAI (bins + 16 kb linear index, SAM spec §5.2) is built from the records' virtual offsets while writing.
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include <algorithm>
#include <map>
#include <string>
#include <thread>
#include <vector>

static const uint64_t CHR[12] = {332615375, 177319215, 289790774, 248932513, 254874144, 253233553, 266382521, 174326481, 278410012, 215000000, 200000000, 182000000};

struct Rng { uint64_t s; explicit Rng(uint64_t x) : s(x * 0x9E3779B97F4A7C15ull + 1) {} uint64_t next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s * 0x2545F4914F6CDD1Dull; }
             double uni() { return (next() >> 11) * (1.0 / 9007199254740992.0); } uint32_t below(uint32_t n) { return (uint32_t)((next() >> 32) * (uint64_t)n >> 32); } };

static inline uint32_t base_at(uint32_t tid, uint32_t pos) { uint64_t h = ((uint64_t)tid << 32 | pos) * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; return (uint32_t)(h >> 61) & 3; }
static inline int reg2bin(int64_t beg, int64_t end) { --end; if (beg >> 14 == end >> 14) return 4681 + (int)(beg >> 14); if (beg >> 17 == end >> 17) return 585 + (int)(beg >> 17);
    if (beg >> 20 == end >> 20) return 73 + (int)(beg >> 20); if (beg >> 23 == end >> 23) return 9 + (int)(beg >> 23); if (beg >> 26 == end >> 26) return 1 + (int)(beg >> 26); return 0; }

typedef void *(*ld_alloc_t)(int); typedef size_t (*ld_comp_t)(void *, const void *, size_t, void *, size_t); typedef void (*ld_free_t)(void *);
static ld_alloc_t ld_alloc; static ld_comp_t ld_comp; static ld_free_t ld_free;

struct Part {                                 // one thread's slice: records [lo, hi) of contig tid
    int tid; uint64_t lo, hi, n_contig; uint64_t seed;
    std::vector<uint8_t> bgzf;                // its members
    struct Rec { int bin; uint32_t beg; uint64_t v0, v1; uint32_t span; };       // local virtual offsets (member offset << 16 | within)
    std::vector<Rec> recs;
};

static void put32(std::vector<uint8_t> &v, uint32_t x) { v.push_back(x); v.push_back(x >> 8); v.push_back(x >> 16); v.push_back(x >> 24); }
static void put16(std::vector<uint8_t> &v, uint32_t x) { v.push_back(x); v.push_back(x >> 8); }

static void bgzf_member(std::vector<uint8_t> &out, const uint8_t *d, size_t n, int level, bool use_zlib, void *ldc)
{
    static const uint8_t head[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
    std::vector<uint8_t> c(n + n / 8 + 256);
    size_t cn = 0;
    if (!use_zlib && ldc) cn = ld_comp(ldc, d, n, c.data(), c.size());
    if (!cn) {
        z_stream zs; memset(&zs, 0, sizeof zs); deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
        zs.next_in = (Bytef *)d; zs.avail_in = (uInt)n; zs.next_out = c.data(); zs.avail_out = (uInt)c.size();
        deflate(&zs, Z_FINISH); cn = zs.total_out; deflateEnd(&zs);
    }
    const size_t o = out.size();
    out.insert(out.end(), head, head + 16); put16(out, (uint32_t)(cn + 25));
    out.insert(out.end(), c.begin(), c.begin() + (long)cn);
    put32(out, (uint32_t)crc32(crc32(0, nullptr, 0), d, (uInt)n)); put32(out, (uint32_t)n);
    (void)o;
}

int main(int argc, char **argv)
{
    std::string outp; uint64_t R = 1000000; double scale = -1; int threads = 8, level = 6; uint64_t seed = 42; bool use_zlib = false;
    bool long_mode = false; int qlevels = 4;
    for (int a = 1; a < argc; ++a) {
        if (!strcmp(argv[a], "-o")) outp = argv[++a]; else if (!strcmp(argv[a], "-n")) R = (uint64_t)atof(argv[++a]);
        else if (!strcmp(argv[a], "-s")) scale = atof(argv[++a]); else if (!strcmp(argv[a], "-t")) threads = atoi(argv[++a]);
        else if (!strcmp(argv[a], "-l")) level = atoi(argv[++a]); else if (!strcmp(argv[a], "-S")) seed = strtoull(argv[++a], nullptr, 10);
        else if (!strcmp(argv[a], "-z")) use_zlib = true;
        else if (!strcmp(argv[a], "--long")) long_mode = true; else if (!strcmp(argv[a], "-Q")) qlevels = atoi(argv[++a]);
    }
    if (outp.empty()) { fprintf(stderr, "usage: bamgen -o out.bam -n records [-s scale] [-t threads] [-l level] [-S seed] [-z]\n"); return 2; }
    if (scale < 0) scale = long_mode ? std::min(1.0, (double)R / 1e7) : (double)R / 1e9;      // (long reads: 1e7 records of 15 kb = 50x of the 3 Gb genome)
    if (!use_zlib) {
        void *h = nullptr;
        for (const char *p : {"libdeflate.so.0", "/usr/lib/x86_64-linux-gnu/libdeflate.so.0", "libdeflate.so"}) if ((h = dlopen(p, RTLD_NOW))) break;
        if (h) { ld_alloc = (ld_alloc_t)dlsym(h, "libdeflate_alloc_compressor"); ld_comp = (ld_comp_t)dlsym(h, "libdeflate_deflate_compress"); ld_free = (ld_free_t)dlsym(h, "libdeflate_free_compressor"); }
        if (!ld_alloc || !ld_comp) { fprintf(stderr, "bamgen: libdeflate not found, using zlib\n"); use_zlib = true; }
    }
    // genome
    std::vector<std::string> names; std::vector<uint32_t> lens;
    Rng g(seed);
    for (int i = 0; i < 12; ++i) { char nm[16]; snprintf(nm, sizeof nm, "Chr%02d", i + 1); names.push_back(nm); lens.push_back((uint32_t)std::max<double>(20000, CHR[i] * scale)); }
    const int n_scaf = scale >= 0.05 ? 500 : std::max(4, (int)(500 * scale * 10));
    for (int i = 0; i < n_scaf; ++i) { char nm[16]; snprintf(nm, sizeof nm, "scaf%04d", i); names.push_back(nm);
        double l = 10000 + g.uni() * 490000; if (scale < 0.25) l *= std::min(1.0, scale * 4); lens.push_back((uint32_t)std::max(10000.0, l)); }
    double tot = 0; for (uint32_t l : lens) tot += l;
    std::vector<uint64_t> per(lens.size()); uint64_t acc = 0;
    if (long_mode) { double big = 0; for (uint32_t l : lens) if (l >= 1000000) big += l;
        if (big == 0) { fprintf(stderr, "bamgen: --long needs a genome scale with chromosomes of >= 1 Mb (-s >= 0.006)\n"); return 2; }
        for (size_t i = 0; i < lens.size(); ++i) { per[i] = lens[i] >= 1000000 ? (uint64_t)(lens[i] / big * R) : 0; acc += per[i]; } }
    else
    for (size_t i = 0; i < lens.size(); ++i) { per[i] = (uint64_t)(lens[i] / tot * R); acc += per[i]; }
    per[0] += R - acc;
    // parts: ~200 K records each, in file order
    std::vector<Part> parts;
    for (size_t t = 0; t < lens.size(); ++t) {
        const uint64_t n = per[t]; if (!n) continue;
        const uint64_t per_part = long_mode ? 2000 : 200000;
        const uint64_t np = (n + per_part - 1) / per_part;
        for (uint64_t k = 0; k < np; ++k) { Part p; p.tid = (int)t; p.lo = n * k / np; p.hi = n * (k + 1) / np; p.n_contig = n; p.seed = seed * 1000003 + parts.size(); parts.push_back(p); }
    }
    // header
    std::vector<uint8_t> hdr;
    { std::string text = "@HD\tVN:1.6\tSO:coordinate\n";
      for (size_t i = 0; i < names.size(); ++i) text += "@SQ\tSN:" + names[i] + "\tLN:" + std::to_string(lens[i]) + "\n";
      text += "@RG\tID:grp1\tSM:sample1\tPL:ILLUMINA\n@PG\tID:bamgen\tPN:bamgen\n";
      hdr.insert(hdr.end(), {'B', 'A', 'M', 1}); put32(hdr, (uint32_t)text.size()); hdr.insert(hdr.end(), text.begin(), text.end()); put32(hdr, (uint32_t)names.size());
      for (size_t i = 0; i < names.size(); ++i) { put32(hdr, (uint32_t)names[i].size() + 1); hdr.insert(hdr.end(), names[i].begin(), names[i].end()); hdr.push_back(0); put32(hdr, lens[i]); } }
    std::vector<uint8_t> hdr_bgzf;
    { void *ldc = use_zlib ? nullptr : ld_alloc(level);
      for (size_t o = 0; o < hdr.size(); o += 0xff00) bgzf_member(hdr_bgzf, hdr.data() + o, std::min<size_t>(0xff00, hdr.size() - o), level, use_zlib, ldc);
      if (ldc) ld_free(ldc); }
    // the parts, in parallel; written in order
    FILE *fo = fopen(outp.c_str(), "wb");
    if (!fo) { perror(outp.c_str()); return 2; }
    fwrite(hdr_bgzf.data(), 1, hdr_bgzf.size(), fo);
    uint64_t file_off = hdr_bgzf.size();
    std::vector<uint64_t> part_base(parts.size());
    auto work = [&](Part &p) {
        Rng r(p.seed);
        void *ldc = use_zlib ? nullptr : ld_alloc(level);
        const uint32_t L = lens[(size_t)p.tid];
        const uint64_t n = p.hi - p.lo;
        // sorted positions inside this part's share of the contig, with the depth modulation per 100 kb block
        const double a0 = (double)L * p.lo / p.n_contig, a1 = (double)L * p.hi / p.n_contig;
        std::vector<uint32_t> pos(n);
        { std::vector<double> u(n); for (auto &x : u) x = r.uni(); std::sort(u.begin(), u.end());
          for (uint64_t i = 0; i < n; ++i) { uint32_t q = (uint32_t)(a0 + u[i] * (a1 - a0));
              const uint32_t blk = q / 100000; uint64_t hb = ((uint64_t)p.tid << 32 | blk) * 0xD6E8FEB86659FD93ull; hb ^= hb >> 32;
              if ((hb & 127) == 0) { const int m = (hb >> 8) % 5; if (m < 2) q = blk * 100000 + 100000 + (q % 1000); }      // empty blocks: push the reads out
              const uint32_t room = long_mode ? 300000u : 5300u;
              if (q + room > L) q = L > room ? L - room : 0;
              if (q >= (uint32_t)a1 && (uint32_t)a1 > 0) q = (uint32_t)a1 - 1;                       // stay inside this part's share: parts are written in order
              if (q < (uint32_t)a0) q = (uint32_t)a0;
              pos[i] = q; }
          std::sort(pos.begin(), pos.end()); }
        std::vector<uint8_t> blk; blk.reserve(0x10000);
        std::vector<uint8_t> rec; rec.reserve(600);
        uint64_t members = 0;
        auto flush = [&]() { if (blk.empty()) return; bgzf_member(p.bgzf, blk.data(), blk.size(), level, use_zlib, ldc); blk.clear(); ++members; };
        // a record may be larger than a member: it starts a fresh member when it does not fit the current one (htslib: bgzf_flush_try) and then
        // fills members of 0xff00 bytes one after the other
        auto emit_record = [&](uint32_t beg, uint32_t span) {
            if (!blk.empty() && blk.size() + rec.size() > 0xff00) flush();
            Part::Rec ir; ir.bin = reg2bin(beg, beg + span); ir.beg = beg; ir.span = span;
            ir.v0 = (uint64_t)p.bgzf.size() << 16 | blk.size();
            for (size_t o = 0; o < rec.size();) {
                const size_t take = std::min<size_t>(0xff00 - blk.size(), rec.size() - o);
                blk.insert(blk.end(), rec.begin() + (long)o, rec.begin() + (long)(o + take));
                o += take;
                if (blk.size() == 0xff00 && o < rec.size()) flush();
            }
            ir.v1 = (uint64_t)p.bgzf.size() << 16 | blk.size();
            p.recs.push_back(ir);
        };
        std::vector<uint32_t> ops; std::vector<uint8_t> sq;
        for (uint64_t i = 0; long_mode && i < n; ++i) {
            ops.clear();
            uint32_t qlen = 0, span = 0;
            const double x = r.uni();
            if (x < 0.01) {
                // more than 65 535 operations: 1M 1D pairs and a tail
                const uint32_t pairs = 33000 + r.below(12001);
                for (uint32_t k = 0; k < pairs; ++k) { ops.push_back(1u << 4); ops.push_back(1u << 4 | 2u); }
                ops.push_back(20u << 4);
                qlen = pairs + 20; span = 2 * pairs + 20;
            } else {
                const uint32_t want = 10000 + r.below(10001);
                if (r.below(100) < 5) { const uint32_t sc = 10 + r.below(191); ops.push_back(sc << 4 | 4u); qlen += sc; }
                const bool gap = r.below(100) < 2; bool gap_done = false;
                while (qlen < want) {
                    uint32_t m = 8 + r.below(8); if (qlen + m > want) m = want - qlen;
                    ops.push_back(m << 4); qlen += m; span += m;
                    if (qlen >= want) break;
                    if (gap && !gap_done && qlen > want / 2) { const uint32_t g2 = 50000 + r.below(130001); ops.push_back(g2 << 4 | 3u); span += g2; gap_done = true; continue; }
                    const uint32_t u = r.below(100);
                    if (u < 45 && qlen + 4 < want) { const uint32_t k = 1 + r.below(3); ops.push_back(k << 4 | 1u); qlen += k; }
                    else { const uint32_t k = 1 + r.below(5); ops.push_back(k << 4 | 2u); span += k; }
                }
            }
            const bool fake = ops.size() > 65535;
            const double f = r.uni();
            uint32_t flag = (r.next() & 1) ? 16 : 0;
            if (f < 0.02) flag |= 1024; else if (f < 0.03) flag |= 256; else if (f < 0.04) flag |= 512; else if (f < 0.05) flag |= 2048;
            const uint32_t mq = (r.next() % 10) < 1 ? 0 : (r.next() % 10) < 2 ? 20 : 60;
            char name[64]; const uint64_t serial = p.lo + i;
            const int ln = snprintf(name, sizeof name, "m64011_190830_220126/%u/ccs", (uint32_t)(serial * 2654435761u >> 8) + p.tid) + 1;
            rec.clear();
            put32(rec, 0);
            put32(rec, (uint32_t)p.tid); put32(rec, pos[i]);
            rec.push_back((uint8_t)ln); rec.push_back((uint8_t)mq); put16(rec, (uint32_t)reg2bin(pos[i], pos[i] + span));
            put16(rec, fake ? 2u : (uint32_t)ops.size()); put16(rec, flag); put32(rec, qlen);
            put32(rec, 0xFFFFFFFFu); put32(rec, 0xFFFFFFFFu); put32(rec, 0);
            rec.insert(rec.end(), name, name + ln);
            if (fake) { put32(rec, qlen << 4 | 4u); put32(rec, span << 4 | 3u); } else for (uint32_t c : ops) put32(rec, c);
            { sq.resize(qlen + 1); uint32_t rp = pos[i]; uint32_t qi = 0;
              for (uint32_t c : ops) { const uint32_t op = c & 15, len = c >> 4;
                  if (op == 0) for (uint32_t j = 0; j < len; ++j, ++rp) sq[qi++] = (uint8_t)((r.next() % 500) ? base_at((uint32_t)p.tid, rp) : r.below(4));
                  else if (op == 1 || op == 4) for (uint32_t j = 0; j < len; ++j) sq[qi++] = (uint8_t)r.below(4);
                  else rp += len; }
              sq[qlen] = 0;
              static const uint8_t nib[4] = {1, 2, 4, 8};
              for (uint32_t j = 0; j < qlen; j += 2) rec.push_back((uint8_t)(nib[sq[j]] << 4 | (j + 1 < qlen ? nib[sq[j + 1]] : 0))); }
            { uint32_t j = 0; while (j < qlen) { const uint32_t qv = (r.next() % 100) < 70 ? 93 : 20 + r.below(60); uint32_t run = 1 + r.below(qv == 93 ? 40 : 6);
                  for (; run > 0 && j < qlen; --run, ++j) rec.push_back((uint8_t)qv); } }
            rec.insert(rec.end(), {'N', 'M', 'I'}); put32(rec, (uint32_t)(ops.size() / 2));
            rec.insert(rec.end(), {'R', 'G', 'Z', 'g', 'r', 'p', '1', 0});
            if (fake) { rec.insert(rec.end(), {'C', 'G', 'B', 'I'}); put32(rec, (uint32_t)ops.size()); for (uint32_t c : ops) put32(rec, c); }
            const uint32_t bs = (uint32_t)rec.size() - 4;
            rec[0] = (uint8_t)bs; rec[1] = (uint8_t)(bs >> 8); rec[2] = (uint8_t)(bs >> 16); rec[3] = (uint8_t)(bs >> 24);
            emit_record(pos[i], span);
        }
        for (uint64_t i = 0; !long_mode && i < n; ++i) {
            const double x = r.uni();
            const int kind = x < 0.85 ? 0 : x < 0.90 ? 1 : x < 0.95 ? 2 : x < 0.99 ? 3 : 4;
            const uint32_t a = 10 + r.below(121);
            const uint32_t xv = kind == 1 ? 1 + r.below(20) : kind == 2 ? 1 + r.below(10) : kind == 3 ? 1 + r.below(40) : kind == 4 ? 100 + r.below(4901) : 0;
            uint32_t cig[3]; int nc = 1;
            if (kind == 0) cig[0] = 150 << 4; else if (kind == 3) { cig[0] = xv << 4 | 4; cig[1] = (150 - xv) << 4; nc = 2; }
            else { cig[0] = a << 4; cig[1] = xv << 4 | (kind == 1 ? 2u : kind == 2 ? 1u : 3u); cig[2] = (kind == 2 ? 150 - a - xv : 150 - a) << 4; nc = 3; }
            const uint32_t span = kind == 0 ? 150 : kind == 3 ? 150 - xv : kind == 2 ? 150 - xv : 150 + xv;
            const double f = r.uni();
            uint32_t flag = (r.next() & 1) ? 16 : 0; flag |= 1 | 2 | ((r.next() & 1) ? 64 : 128);
            if (f < 0.02) flag |= 1024; else if (f < 0.03) flag |= 256; else if (f < 0.04) flag |= 512; else if (f < 0.045) flag |= 4;
            const uint32_t mq = (r.next() % 10) < 1 ? 0 : (r.next() % 10) < 2 ? 20 : 60;
            char name[64]; const uint64_t serial = p.lo + i;
            const int ln = snprintf(name, sizeof name, "A00123:45:HXXXXXXX:%u:%u:%u:%u", 1 + (uint32_t)(serial % 4), 1101 + (uint32_t)((serial / 7) % 578), 1000 + r.below(31000), 1000 + r.below(36000)) + 1;
            rec.clear();
            put32(rec, 0);                                            // block_size, patched below
            put32(rec, (uint32_t)p.tid); put32(rec, pos[i]);
            rec.push_back((uint8_t)ln); rec.push_back((uint8_t)mq); put16(rec, (uint32_t)reg2bin(pos[i], pos[i] + span));
            put16(rec, (uint32_t)nc); put16(rec, flag); put32(rec, 150);
            put32(rec, (uint32_t)p.tid); const int32_t mpos = (int32_t)pos[i] + (int32_t)r.below(400) - 100; put32(rec, (uint32_t)(mpos < 0 ? 0 : mpos)); put32(rec, (uint32_t)(300 + r.below(200)));
            rec.insert(rec.end(), name, name + ln);
            for (int k = 0; k < nc; ++k) put32(rec, cig[k]);
            // SEQ: the reference under the aligned bases (inserted / clipped bases random), 0.5 % mismatches
            { uint8_t b[150]; uint32_t rp = pos[i]; int qi = 0;
              for (int k = 0; k < nc; ++k) { const uint32_t op = cig[k] & 15, len = cig[k] >> 4;
                  if (op == 0) for (uint32_t j = 0; j < len; ++j, ++rp) b[qi++] = (uint8_t)((r.next() % 200) ? base_at((uint32_t)p.tid, rp) : r.below(4));
                  else if (op == 1 || op == 4) for (uint32_t j = 0; j < len; ++j) b[qi++] = (uint8_t)r.below(4);
                  else rp += len; }
              static const uint8_t nib[4] = {1, 2, 4, 8};
              for (int j = 0; j < 150; j += 2) rec.push_back((uint8_t)(nib[b[j]] << 4 | nib[b[j + 1]])); }
            // QUAL: binned, in runs (-Q 40: unbinned, decaying along the read)
            if (qlevels == 40) { int q0 = 36 + (int)r.below(6); for (int j = 0; j < 150; ++j) { if ((r.next() % 16) == 0 && q0 > 12) q0 -= (int)r.below(3);
                  int qv = q0 - (int)(r.next() % 100 < 70 ? r.below(4) : r.below(22)); if (qv < 2) qv = 2; if (qv > 41) qv = 41; rec.push_back((uint8_t)qv); } }
            else
            { int j = 0; while (j < 150) { const uint32_t qv = (r.next() % 100) < 78 ? 37 : (r.next() % 10) < 6 ? 25 : (r.next() % 10) < 7 ? 11 : 2; int run = 1 + (int)r.below(qv == 37 ? 14 : 3);
                  for (; run > 0 && j < 150; --run, ++j) rec.push_back((uint8_t)qv); } }
            // tags
            { const uint32_t nm = kind == 0 ? ((r.next() % 10) < 7 ? 0 : 1 + r.below(3)) : xv > 20 ? 20 : xv;
              rec.insert(rec.end(), {'N', 'M', 'C'}); rec.push_back((uint8_t)nm);
              char md[32]; const int ml = snprintf(md, sizeof md, "%u%s", nm ? a : 150u, nm ? "A" : "") ; rec.insert(rec.end(), {'M', 'D', 'Z'}); rec.insert(rec.end(), md, md + ml); if (nm) { char t2[16]; const int l2 = snprintf(t2, sizeof t2, "%u", 149 - a); rec.insert(rec.end(), t2, t2 + l2); } rec.push_back(0);
              rec.insert(rec.end(), {'A', 'S', 'C'}); rec.push_back((uint8_t)(150 - 5 * nm > 0 ? 150 - 5 * nm : 0));
              rec.insert(rec.end(), {'X', 'S', 'C'}); rec.push_back((uint8_t)r.below(60));
              rec.insert(rec.end(), {'R', 'G', 'Z', 'g', 'r', 'p', '1', 0}); }
            const uint32_t bs = (uint32_t)rec.size() - 4;
            rec[0] = (uint8_t)bs; rec[1] = (uint8_t)(bs >> 8); rec[2] = (uint8_t)(bs >> 16); rec[3] = (uint8_t)(bs >> 24);
            if (blk.size() + rec.size() > 0xff00) flush();
            Part::Rec ir; ir.bin = reg2bin(pos[i], pos[i] + span); ir.beg = pos[i]; ir.span = span;
            ir.v0 = (uint64_t)p.bgzf.size() << 16 | blk.size();
            blk.insert(blk.end(), rec.begin(), rec.end());
            ir.v1 = (uint64_t)p.bgzf.size() << 16 | blk.size();
            p.recs.push_back(ir);
        }
        flush();
        if (ldc) ld_free(ldc);
    };
    // the BAI accumulators
    struct Ref { std::map<int, std::vector<std::pair<uint64_t, uint64_t>>> bins; std::vector<uint64_t> lin; uint64_t v_first = 0, v_last = 0, n_map = 0; };
    std::vector<Ref> refs(lens.size());
    size_t done = 0;
    while (done < parts.size()) {
        const size_t nb = std::min<size_t>((size_t)threads, parts.size() - done);
        std::vector<std::thread> th;
        for (size_t k = 0; k < nb; ++k) th.emplace_back(work, std::ref(parts[done + k]));
        for (auto &t : th) t.join();
        for (size_t k = 0; k < nb; ++k) {
            Part &p = parts[done + k];
            part_base[done + k] = file_off;
            fwrite(p.bgzf.data(), 1, p.bgzf.size(), fo);
            Ref &rf = refs[(size_t)p.tid];
            for (const Part::Rec &x : p.recs) {
                // a record's end offset: the end of a member is written as the start of the next one, like htslib does
                const uint64_t v0 = ((x.v0 >> 16) + file_off) << 16 | (x.v0 & 0xffff);
                const uint64_t v1 = ((x.v1 >> 16) + file_off) << 16 | (x.v1 & 0xffff);
                const int bin = x.bin; const uint32_t span = x.span;
                auto &ch = rf.bins[bin];
                if (!ch.empty() && ch.back().second >> 16 == v0 >> 16) ch.back().second = v1; else ch.emplace_back(v0, v1);
                const uint32_t w0 = x.beg >> 14, w1 = (x.beg + (span ? span : 1) - 1) >> 14;
                if (rf.lin.size() <= w1) rf.lin.resize(w1 + 1, 0);
                for (uint32_t w = w0; w <= w1; ++w) if (!rf.lin[w]) rf.lin[w] = v0;
                if (!rf.n_map) rf.v_first = v0;
                rf.v_last = v1; ++rf.n_map;
            }
            file_off += p.bgzf.size();
            std::vector<uint8_t>().swap(p.bgzf); std::vector<Part::Rec>().swap(p.recs);
        }
        done += nb;
    }
    static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    fwrite(eof, 1, 28, fo); fclose(fo);
    // .bai
    { std::vector<uint8_t> b; b.insert(b.end(), {'B', 'A', 'I', 1}); put32(b, (uint32_t)lens.size());
      auto put64 = [&](uint64_t v) { put32(b, (uint32_t)v); put32(b, (uint32_t)(v >> 32)); };
      for (Ref &rf : refs) {
          put32(b, (uint32_t)rf.bins.size() + (rf.n_map ? 1 : 0));
          for (auto &kv : rf.bins) { put32(b, (uint32_t)kv.first); put32(b, (uint32_t)kv.second.size()); for (auto &c : kv.second) { put64(c.first); put64(c.second); } }
          if (rf.n_map) { put32(b, 37450); put32(b, 2); put64(rf.v_first); put64(rf.v_last); put64(rf.n_map); put64(0); }
          // htslib fills the holes of the linear index with the next known offset from the left
          for (size_t w = 1; w < rf.lin.size(); ++w) if (!rf.lin[w]) rf.lin[w] = rf.lin[w - 1];
          put32(b, (uint32_t)rf.lin.size()); for (uint64_t v : rf.lin) put64(v);
      }
      put64(0);
      FILE *fb = fopen((outp + ".bai").c_str(), "wb"); if (!fb) { perror("bai"); return 2; } fwrite(b.data(), 1, b.size(), fb); fclose(fb); }
    fprintf(stderr, "bamgen: %llu records%s%s, %zu contigs (%.1f Mb), %.2f GB, %s level %d\n", (unsigned long long)R, long_mode ? " (long reads: 10-20 kb, ~2 600 CIGAR operations each, 1 % in the CG tag)" : "",
            qlevels == 40 ? " (40-level qualities)" : "", lens.size(), tot / 1e6, (file_off + 28) / 1e9, use_zlib ? "zlib" : "libdeflate", level);
    return 0;
}
