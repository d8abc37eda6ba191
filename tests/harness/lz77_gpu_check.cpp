// tests/harness/lz77_gpu_check.cpp — TEST INFRASTRUCTURE: pd_deflate_parse (the engine's LZ77 parse ON THE DEVICE, through the C-ABI)
// against the symbols zlib itself produces for the same chunks (pgz::zlib_chunk_symbols), the way host/pgzip.cpp cuts a text.
//   lz77_gpu_check <file> [chunk tail]   -> "N chunks, M symbols compared, K chunks differ" + the device times of the call
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <thread>
#include <vector>
#include "../../include/pandepth_amd.h"
#include "../../include/pandepth_amd_dev.h"
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
    if (argc < 2) { fprintf(stderr, "usage: lz77_gpu_check <file> [chunk tail]\n"); return 2; }
    std::vector<uint8_t> text = slurp(argv[1]);
    const size_t CH = argc > 2 ? (size_t)atoll(argv[2]) : (size_t)1 << 18, TAIL = argc > 3 ? (size_t)atoll(argv[3]) : (size_t)1 << 14;
    const size_t N = text.size();
    std::vector<pd_lz_chunk> chunks;
    for (size_t start = 0; start < N; start += CH) chunks.push_back(pd_lz_chunk{start, std::min(N, start + CH + TAIL), start < 32768 ? 0 : start - 32768});
    const uint32_t len1[1] = {1000};
    pd_ctx *ctx = nullptr;
    if (pd_create(0, 1, len1, &ctx) != 0) { fprintf(stderr, "pd_create: %s\n", pd_strerror(nullptr)); return 3; }
    if (const char *e = getenv("LZ_GROUP")) if (pd_set_param(ctx, "lz_group", (uint64_t)atoll(e)) != 0) { fprintf(stderr, "lz_group: %s\n", pd_strerror(ctx)); return 3; }   // (A/B of the LDS parse)
    size_t cap = 16;
    for (auto &c : chunks) cap += c.end - c.start;
    std::vector<uint32_t> syms(cap);
    std::vector<uint64_t> off(chunks.size() + 1);
    pd_profile(ctx, 1);
    double best = 1e9;
    for (int rep = 0; rep < 2; ++rep) {
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = pd_deflate_parse(ctx, text.data(), N, chunks.data(), (uint32_t)chunks.size(), syms.data(), cap, off.data());
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (rc != 0) { fprintf(stderr, "pd_deflate_parse: %d %s\n", rc, pd_strerror(ctx)); return 3; }
        best = std::min(best, dt);
    }
    double ms_sort = 0, ms_parse = 0, ms_g = 0; uint64_t n1 = 0, n2 = 0, n3 = 0;
    pd_profile_get(ctx, "lz_sort", &ms_sort, &n1); pd_profile_get(ctx, "lz_parse", &ms_parse, &n2); pd_profile_get(ctx, "lz_gather", &ms_g, &n3);
    // zlib's own symbols, chunk by chunk, on the host threads
    std::vector<std::vector<uint32_t>> ref(chunks.size());
    std::vector<int> okv(chunks.size(), 1);
    {
        std::vector<std::thread> th;
        const unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        for (unsigned t = 0; t < nt; ++t) th.emplace_back([&, t] {
            for (size_t k = t; k < chunks.size(); k += nt) {
                const pd_lz_chunk &c = chunks[k];
                okv[k] = pgz::zlib_chunk_symbols(text.data() + c.origin, (size_t)(c.start - c.origin), (size_t)(c.end - c.start), ref[k]) ? 1 : 0;
            }
        });
        for (auto &t : th) t.join();
    }
    size_t bad = 0, compared = 0, unread = 0;
    for (size_t k = 0; k < chunks.size(); ++k) {
        if (!okv[k]) { ++unread; continue; }                     // a stored block: zlib found the chunk incompressible (pgzip declines there)
        const pd_lz_chunk &c = chunks[k];
        const uint32_t *mine = syms.data() + off[k];
        const size_t nm = (size_t)(off[k + 1] - off[k]);
        const size_t margin = c.end == N ? 300 : 1024;
        size_t q = c.start, i = 0; bool same = true;
        for (; i < ref[k].size() && i < nm; ++i) {
            const size_t len = ref[k][i] >= 65536u ? (ref[k][i] >> 16) : 1u;
            if (q + len + margin > c.end) break;
            if (ref[k][i] != mine[i]) { same = false; break; }
            q += len;
        }
        if (!same) { if (bad < 5) fprintf(stderr, "chunk at %llu: symbol %zu at text position %zu differs: zlib %08x, device %08x\n", (unsigned long long)c.start, i, q, ref[k][i], mine[i]); ++bad; }
        compared += i;
    }
    printf("%zu chunks, %zu symbols compared, %zu chunks differ, %zu chunks stored by zlib; %.1f MB: call %.3f s (%.0f MB/s), device: sort %.2f ms, parse %.2f ms, gather %.2f ms\n",
           chunks.size(), compared, bad, unread, N / 1e6, best, N / 1e6 / best, n1 ? ms_sort / n1 : 0, n2 ? ms_parse / n2 : 0, n3 ? ms_g / n3 : 0);
    pd_destroy(ctx);
    return bad ? 1 : 0;
}
