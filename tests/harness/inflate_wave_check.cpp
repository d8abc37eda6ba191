// tests/harness/inflate_wave_check.cpp — TEST INFRASTRUCTURE: runs the product's wave-cooperative DEFLATE decoder
// (pandepth_amd/csrc/pd_inflate_wave.h, the same source the gfx950 kernel compiles) on the host, with the 64 lanes
// of the wave emulated in a loop, over every BGZF block of the given files (or every member of plain multi-member
// input made by the generator mode) and compares each block with zlib's inflate.
//   inflate_wave_check [-s] [-f N] file...     -s: print speculation statistics   -f N: N mutated copies per block
//   inflate_wave_check -g                     generated streams: every block type, levels 0-9, pathological inputs
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include <random>
#include <vector>
#include "../../pandepth_amd/csrc/pd_inflate_wave.h"

static pdw::Tables g_T;
static pdw::Stats g_st;
static pdw::Token g_tok[65536 / 3 + 64];

static bool zinflate(const unsigned char *in, size_t n, std::vector<unsigned char> &out, size_t want)
{
    out.assign(want + 1, 0);
    z_stream zs; memset(&zs, 0, sizeof zs); inflateInit2(&zs, -15);
    zs.next_in = (Bytef *)in; zs.avail_in = (uInt)n; zs.next_out = out.data(); zs.avail_out = (uInt)want;
    const int zr = inflate(&zs, Z_FINISH);
    const bool ok = zr == Z_STREAM_END && zs.avail_out == 0;
    inflateEnd(&zs);
    return ok;
}

// one raw deflate stream against zlib; returns 0 ok, 1 mismatch
static int check_stream(const unsigned char *in, size_t n, size_t isize, const char *what, bool must_decode)
{
    std::vector<unsigned char> padded(in, in + n);
    padded.resize(n + 16, 0xA5);
    std::vector<unsigned char> mine(isize + 32, 0xEE), ref;      // (the span path reads up to 15 bytes past a stretch: the product buffers carry 256 bytes of slack)
    const bool zok = zinflate(in, n, ref, isize);
    const int rc = pdw::inflate_block<pdw::HostWave>(padded.data(), (uint32_t)n, mine.data(), (uint32_t)isize, g_T, g_tok, &g_st);
    if (must_decode && !zok) { fprintf(stderr, "%s: zlib itself failed\n", what); return 1; }
    if (rc == pdw::PD_W_HOST) return must_decode ? (fprintf(stderr, "%s: handed to the host (rc 1)\n", what), 1) : 0;
    if (rc == 0) {
        if (!zok) { fprintf(stderr, "%s: accepted a stream zlib rejects\n", what); return 1; }
        if (memcmp(mine.data(), ref.data(), isize) != 0) { fprintf(stderr, "%s: output differs from zlib\n", what); return 1; }
    } else if (zok) { fprintf(stderr, "%s: rc=%d on a stream zlib inflates\n", what, rc); return 1; }
    for (int k = 0; k < 9; ++k) if (mine[isize + k] != 0xEE) { fprintf(stderr, "%s: wrote past the output\n", what); return 1; }
    return 0;
}

static std::vector<unsigned char> zdeflate(const std::vector<unsigned char> &d, int level, int strategy, int memlevel = 8)
{
    z_stream zs; memset(&zs, 0, sizeof zs);
    deflateInit2(&zs, level, Z_DEFLATED, -15, memlevel, strategy);
    std::vector<unsigned char> out(deflateBound(&zs, d.size()) + 64);
    zs.next_in = (Bytef *)d.data(); zs.avail_in = (uInt)d.size(); zs.next_out = out.data(); zs.avail_out = (uInt)out.size();
    deflate(&zs, Z_FINISH);
    out.resize(zs.total_out);
    deflateEnd(&zs);
    return out;
}

static int generated()
{
    std::mt19937 rng(11);
    int bad = 0, n = 0;
    auto run = [&](const std::vector<unsigned char> &d, const char *name) {
        for (int level : {0, 1, 4, 6, 9})
            for (int strat : {Z_DEFAULT_STRATEGY, Z_FIXED, Z_HUFFMAN_ONLY, Z_RLE, Z_FILTERED})
                for (int ml : {8, 1}) {
                    const std::vector<unsigned char> c = zdeflate(d, level, strat, ml);
                    if (c.size() > 65536 * 2) continue;
                    char what[128]; snprintf(what, sizeof what, "%s level %d strategy %d memlevel %d (%zu -> %zu)", name, level, strat, ml, d.size(), c.size());
                    bad += check_stream(c.data(), c.size(), d.size(), what, true); ++n;
                }
    };
    for (size_t len : {0ul, 1ul, 2ul, 3ul, 7ul, 8ul, 9ul, 63ul, 64ul, 65ul, 257ul, 258ul, 259ul, 1000ul, 4096ul, 32768ul, 32769ul, 65280ul, 65535ul}) {
        std::vector<unsigned char> d(len);
        for (auto &x : d) x = rng() & 0xff; run(d, "random");
        for (auto &x : d) x = 0; run(d, "zeros");
        for (size_t i = 0; i < len; ++i) d[i] = (unsigned char)(i % 3); run(d, "period3");
        for (size_t i = 0; i < len; ++i) d[i] = (unsigned char)(i % 7 == 0 ? rng() : 'A' + i % 9); run(d, "period7+noise");
        for (size_t i = 0; i < len; ++i) d[i] = "ACGT"[rng() & 3]; run(d, "acgt");
        for (size_t i = 0; i < len; ++i) d[i] = (unsigned char)(i < len / 2 ? rng() : d[i - len / 2]); run(d, "second half repeats");
        for (size_t i = 0; i < len; ++i) d[i] = (unsigned char)((rng() % 1000) ? 'F' : '#' + rng() % 40); run(d, "long runs");
        {   // skewed alphabet: code lengths up to 15 bits (sub-tables)
            for (size_t i = 0; i < len; ++i) { unsigned r = rng(), k = 0; while ((r & 1) && k < 30) { r >>= 1; ++k; } d[i] = (unsigned char)(k * 7); }
            run(d, "geometric");
        }
        {   // records-like: fixed header fields + names that share prefixes + noise
            size_t i = 0; unsigned rec = 0;
            while (i < len) {
                char nm[64]; const int k = snprintf(nm, sizeof nm, "A00123:45:HXXXXXXX:1:%u:%u:%u", 1101 + rec / 999, 1000 + rec * 13 % 30000, 2000 + rec * 7 % 30000);
                for (int j = 0; j < k && i < len; ++j) d[i++] = (unsigned char)nm[j];
                for (int j = 0; j < 36 && i < len; ++j) d[i++] = (unsigned char)(j < 8 ? rec >> (j * 2) : 0);
                for (int j = 0; j < 75 && i < len; ++j) d[i++] = (unsigned char)(0x11 << (rng() & 3));
                for (int j = 0; j < 150 && i < len; ++j) d[i++] = (unsigned char)((rng() % 10) < 6 ? 37 : 25);
                ++rec;
            }
            run(d, "records");
        }
    }
    // the wave CRC-32 against zlib's, every length class (empty, partial first KiB, exact KiBs, the 64 KiB maximum)
    for (size_t len : {0ul, 1ul, 2ul, 1023ul, 1024ul, 1025ul, 2048ul, 4097ul, 30000ul, 65535ul, 65536ul})
        for (int rep = 0; rep < 3; ++rep) {
            std::vector<unsigned char> d(len + 8);
            for (auto &x : d) x = rep == 0 ? 0 : rep == 1 ? 0xff : (unsigned char)rng();
            const uint32_t mine = pdw::crc32_wave<pdw::HostWave>(d.data(), (uint32_t)len, g_T.ll);
            const uint32_t ref = (uint32_t)crc32(crc32(0L, Z_NULL, 0), d.data(), (uInt)len);
            if (mine != ref) { fprintf(stderr, "crc32_wave: length %zu fill %d: %08x, zlib %08x\n", len, rep, mine, ref); ++bad; }
            ++n;
        }
    // the small pieces of phase 3: the remainder without a division, the byte gather of a short period through its selector
    for (uint32_t d = 1; d <= 300; ++d)
        for (uint32_t off = 0; off < 65536; off += (d < 9 ? 1 : 37)) {
            if (pdw::small_mod(off, d) != off % d) { fprintf(stderr, "small_mod(%u, %u) = %u\n", off, d, pdw::small_mod(off, d)); ++bad; }
            ++n;
        }
    for (uint32_t d = 1; d <= 7; ++d)
        for (uint32_t o = 0; o < d; ++o) {
            const uint64_t raw = ((uint64_t)rng() << 32) | rng();
            uint64_t want = 0;
            for (int k = 0; k < 8; ++k) want |= ((raw >> (8 * ((o + k) % d))) & 0xff) << (8 * k);
            if (pdw::gather8(raw, pdw::period_selector(d, o)) != want) { fprintf(stderr, "gather8: period %u from %u\n", d, o); ++bad; }
            ++n;
        }
    printf("generated: %d streams and unit checks, %d failures\n", n, bad);
    return bad;
}

int main(int argc, char **argv)
{
    bool stats = false; int fuzz = 0; bool gen = false;
    std::vector<const char *> files;
    for (int a = 1; a < argc; ++a) {
        if (!strcmp(argv[a], "-s")) stats = true;
        else if (!strcmp(argv[a], "-g")) gen = true;
        else if (!strcmp(argv[a], "-f") && a + 1 < argc) fuzz = atoi(argv[++a]);
        else files.push_back(argv[a]);
    }
    long blocks = 0, bytes = 0, bad = 0, fuzzed = 0, fuzz_ok = 0;
    if (gen) bad += generated();
    std::mt19937 rng(5);
    for (const char *path : files) {
        FILE *f = fopen(path, "rb");
        if (!f) { fprintf(stderr, "cannot open %s\n", path); return 2; }
        std::vector<unsigned char> d;
        unsigned char buf[1 << 16]; size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + n);
        fclose(f);
        size_t o = 0;
        while (o + 18 <= d.size()) {
            const unsigned char *p = d.data() + o;
            if (p[0] != 0x1f || p[1] != 0x8b || !(p[3] & 4)) { fprintf(stderr, "%s: not BGZF at %zu\n", path, o); return 2; }
            const unsigned xlen = p[10] | (p[11] << 8);
            const unsigned bsize = (p[16] | (p[17] << 8)) + 1;
            const unsigned doff = 12 + xlen;
            const unsigned isize = p[bsize - 4] | (p[bsize - 3] << 8) | (p[bsize - 2] << 16) | ((unsigned)p[bsize - 1] << 24);
            char what[256]; snprintf(what, sizeof what, "%s: block at %zu (csize %u, isize %u)", path, o, bsize, isize);
            if (check_stream(p + doff, bsize - doff - 8, isize, what, true)) ++bad;
            {   // the whole member with its CRC-32; one flipped output-affecting bit in the trailer must be noticed
                std::vector<unsigned char> mem(p + doff, p + bsize); mem.resize(mem.size() + 16, 0);
                std::vector<unsigned char> o2(isize + 9);
                int r = pdw::inflate_member<pdw::HostWave>(mem.data(), bsize - doff - 8, o2.data(), isize, g_T, g_tok, nullptr);
                if (r != 0 && r != pdw::PD_W_HOST) { fprintf(stderr, "%s: inflate_member rc %d\n", what, r); ++bad; }
                if (r == 0) {
                    mem[bsize - doff - 8] ^= 0x10;             // the stored CRC
                    r = pdw::inflate_member<pdw::HostWave>(mem.data(), bsize - doff - 8, o2.data(), isize, g_T, g_tok, nullptr);
                    if (r != -20) { fprintf(stderr, "%s: a wrong CRC-32 went unnoticed (rc %d)\n", what, r); ++bad; }
                }
            }
            for (int k = 0; k < fuzz; ++k) {
                std::vector<unsigned char> c(p + doff, p + bsize - 8);
                const int flips = 1 + (int)(rng() % 3);
                for (int j = 0; j < flips && !c.empty(); ++j) {
                    const size_t at = (rng() % 4 == 0) ? rng() % std::min<size_t>(c.size(), 64) : rng() % c.size();
                    c[at] ^= (unsigned char)(1u << (rng() & 7));
                }
                const pdw::Stats keep = g_st;
                char w2[300]; snprintf(w2, sizeof w2, "%s mutation %d", what, k);
                const int r = check_stream(c.data(), c.size(), isize, w2, false);
                g_st = keep;
                ++fuzzed; if (!r) ++fuzz_ok; else ++bad;
            }
            ++blocks; bytes += isize; o += bsize;
        }
    }
    printf("%ld blocks, %ld bytes, %ld mismatches", blocks, bytes, bad);
    if (fuzz) printf(", %ld mutated streams (%ld consistent with zlib)", fuzzed, fuzz_ok);
    printf("\n");
    if (stats && g_st.steps) {
        const pdw::Stats &s = g_st;
        printf("deflate blocks %.2f per member, supersteps %.2f per deflate block, sync rounds %.2f per superstep; matches %.0f per member in %.1f batches, %.2f rounds per batch\n",
               (double)s.dblocks / s.blocks, (double)s.steps / s.dblocks, (double)s.sync_rounds / s.steps, (double)s.matches / s.blocks, (double)s.batches / s.blocks,
               s.batches ? (double)s.emit_rounds / s.batches : 0.0);
        printf("symbols: %.0f true per member, decoded %.2fx while speculating; lanes re-decoded per sync round %.1f\n", (double)s.sym_true / s.blocks,
               (double)s.sym_decoded / s.sym_true, (double)s.lanes_redecoded / s.sync_rounds);
        printf("wave-serial loop trips per member: sync %.0f, emit %.0f (+ %.1f hand-overs), header symbols %.0f, copy chunk-iterations %.0f; bytes per member %.0f\n",
               (double)s.wave_iters_sync / s.blocks, (double)s.wave_iters_emit / s.blocks, (double)s.handovers / s.blocks, (double)s.hdr_syms / s.blocks, (double)s.copy_iters / s.blocks,
               (double)bytes / blocks);
        printf("match copies: the wave waits for %.0f 8-byte pieces per member (the longest ready match of every round); %.0f matches per member are longer than 16 bytes\n",
               (double)s.copy_serial / s.blocks, (double)s.long_matches / s.blocks);
    }
    return bad ? 1 : 0;
}
