// tests/harness/inflate_check.cpp — TEST INFRASTRUCTURE: runs the product's DEFLATE decoder
// (pandepth_amd/csrc/pd_inflate_core.h, the same source the gfx950 kernel compiles) on the host over
// every BGZF block of the given files and compares each block with zlib's inflate.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include <vector>
#include "../../pandepth_amd/csrc/pd_inflate_core.h"

int main(int argc, char **argv)
{
    long blocks = 0, bytes = 0, bad = 0;
    pdi::Tables *t = new pdi::Tables;
    for (int a = 1; a < argc; ++a) {
        FILE *f = fopen(argv[a], "rb");
        if (!f) { fprintf(stderr, "cannot open %s\n", argv[a]); return 2; }
        std::vector<unsigned char> d;
        unsigned char buf[1 << 16]; size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + n);
        fclose(f);
        size_t o = 0;
        while (o + 18 <= d.size()) {
            const unsigned char *p = d.data() + o;
            if (p[0] != 0x1f || p[1] != 0x8b || !(p[3] & 4)) { fprintf(stderr, "%s: not BGZF at %zu\n", argv[a], o); return 2; }
            const unsigned xlen = p[10] | (p[11] << 8);
            const unsigned bsize = (p[16] | (p[17] << 8)) + 1;          // BC is the first subfield in every writer seen
            const unsigned doff = 12 + xlen;
            const unsigned isize = p[bsize - 4] | (p[bsize - 3] << 8) | (p[bsize - 2] << 16) | ((unsigned)p[bsize - 1] << 24);
            std::vector<unsigned char> mine(isize + 1, 0xEE), ref(isize + 1, 0);
            const int rc = pdi::inflate_block(p + doff, bsize - doff - 8, mine.data(), isize, t->fast, t->slow);
            z_stream zs; memset(&zs, 0, sizeof zs); inflateInit2(&zs, -15);
            zs.next_in = (Bytef *)(p + doff); zs.avail_in = bsize - doff - 8; zs.next_out = ref.data(); zs.avail_out = isize;
            const int zr = inflate(&zs, Z_FINISH); inflateEnd(&zs);
            if (zr != Z_STREAM_END && isize) { fprintf(stderr, "zlib failed on block at %zu\n", o); return 2; }
            if (rc != 0 || memcmp(mine.data(), ref.data(), isize) != 0 || mine[isize] != 0xEE) {
                if (bad < 5) fprintf(stderr, "%s: block at %zu (csize %u, isize %u): rc=%d\n", argv[a], o, bsize, isize, rc);
                ++bad;
            }
            ++blocks; bytes += isize; o += bsize;
        }
        // corrupt-input robustness: flip bytes in the first blocks and make sure the decoder returns
        if (d.size() > 4096) {
            std::vector<unsigned char> c(d.begin(), d.begin() + 4096);
            std::vector<unsigned char> out(70000);
            for (int k = 0; k < 2000; ++k) {
                c[18 + (k * 7919) % 4000] ^= (unsigned char)(1 + k % 255);
                (void)pdi::inflate_block(c.data() + 18, 4096 - 18, out.data(), 65536, t->fast, t->slow);
            }
        }
    }
    printf("%ld blocks, %ld bytes, %ld mismatches\n", blocks, bytes, bad);
    return bad ? 1 : 0;
}
