// tests/harness/lz77_check.cpp — TEST INFRASTRUCTURE: the engine's wave-parallel restatement of zlib's level-6 LZ77 parse
// (pandepth_amd/csrc/pd_lz77.h, the source the gfx950 kernel compiles, here with the 64 lanes in a loop) against the symbols
// zlib ITSELF produces (pgz::zlib_chunk_symbols: deflate on the chunk primed with its dictionary, symbols read back out of the
// stream).  usage: lz77_check <file> [chunk tail]  — the file is cut into chunks like host/pgzip.cpp cuts its text; every chunk's
// symbols must be zlib's up to 1 KiB before the chunk's end (where zlib sees the end of its input and the callers never look).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "../../pandepth_amd/csrc/pd_lz77.h"
#include "../../pandepth_amd/host/pgzip.h"

static std::vector<uint8_t> slurp(const char *path)
{
    std::vector<uint8_t> v;
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    uint8_t buf[1 << 16];
    size_t k;
    while ((k = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + k);
    fclose(f);
    return v;
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: lz77_check <file> [chunk tail]\n"); return 2; }
    std::vector<uint8_t> text = slurp(argv[1]);
    const size_t CH = argc > 2 ? (size_t)atoll(argv[2]) : (size_t)1 << 18, TAIL = argc > 3 ? (size_t)atoll(argv[3]) : (size_t)1 << 14;
    const size_t N = text.size();
    text.resize(N + 32, 0);
    // positions sorted by (hash, position): a stable counting sort
    const size_t NP = N >= 3 ? N - 2 : 0;
    std::vector<uint32_t> bucket((1u << pdz::HASH_BITS) + 1, 0), S(NP ? NP : 1), R(N + 1, 0);
    for (size_t p = 0; p < NP; ++p) bucket[pdz::hash3(&text[p]) + 1]++;
    for (size_t h = 0; h < (1u << pdz::HASH_BITS); ++h) bucket[h + 1] += bucket[h];
    { std::vector<uint32_t> cur(bucket.begin(), bucket.end() - 1);
      for (size_t p = 0; p < NP; ++p) { const uint32_t i = cur[pdz::hash3(&text[p])]++; S[i] = (uint32_t)p; R[p] = i; } }
    pdz::Text T{text.data(), S.data(), R.data(), bucket.data(), N};
    size_t bad = 0, chunks = 0, syms_total = 0;
    for (size_t start = 0; start < N; start += CH) {
        const size_t end = std::min(N, start + CH + TAIL);
        const size_t dict = start < 32768 ? start : 32768;
        std::vector<uint32_t> ref;
        if (!pgz::zlib_chunk_symbols(text.data() + (start - dict), dict, end - start, ref)) { fprintf(stderr, "chunk at %zu: zlib's stream could not be read back\n", start); return 2; }
        std::vector<uint32_t> mine(end - start + 8);
        pdz::Out o{mine.data(), 0, (uint32_t)mine.size()};
        if (!pdz::parse_chunk<pdz::HostWave>(T, start, end, start - dict, o)) { fprintf(stderr, "chunk at %zu: symbol buffer too small\n", start); return 2; }
        // compare up to the last symbol that ends at least 1 KiB before the chunk's end (the true end of the text: all of it but the last 300 bytes)
        const size_t margin = end == N ? 300 : 1024;
        size_t q = start, i = 0;
        bool same = true;
        for (; i < ref.size() && i < o.n; ++i) {
            const size_t len = ref[i] >= 65536u ? (ref[i] >> 16) : 1u;
            if (q + len + margin > end) break;
            if (ref[i] != mine[i]) { same = false; break; }
            q += len;
        }
        if (!same) {
            if (bad < 5) fprintf(stderr, "chunk at %zu: symbol %zu at text position %zu differs: zlib %08x, wave parse %08x\n", start, i, q, ref[i], mine[i]);
            ++bad;
        }
        ++chunks; syms_total += i;
    }
    printf("%zu chunks, %zu symbols compared, %zu chunks differ\n", chunks, syms_total, bad);
    return bad ? 1 : 0;
}
