// tests/harness/pgzip_check.cpp — TEST INFRASTRUCTURE: pgz::gzip_identical against the system zlib's gzopen/gzwrite/gzclose
// on a file's bytes.  usage: pgzip_check <file> [threads chunk tail batch]  -> prints "identical" / where the streams diverge.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <zlib.h>

#include <atomic>
#include <chrono>
#include <string>
#include <vector>

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
    if (argc < 2) { fprintf(stderr, "usage: pgzip_check <file> [threads chunk tail batch]\n"); return 2; }
    const std::vector<uint8_t> data = slurp(argv[1]);
    const int threads = argc > 2 ? atoi(argv[2]) : 4;
    pgz::Params p;
    if (argc > 3) p.chunk = (size_t)atoll(argv[3]);
    if (argc > 4) p.tail = (size_t)atoll(argv[4]);
    if (argc > 5) p.batch = (size_t)atoll(argv[5]);
    if (getenv("PGZ_NATIVE")) {                              // stage 1 by the engine's parse (host emulation), the geometry a device provider gets
        const pgz::Params d = pgz::Params::for_device(pgz::host_emulation_parse());
        p.parse = d.parse;
        if (argc <= 3) p.chunk = d.chunk;
        if (argc <= 4) p.tail = d.tail;
        if (argc <= 5) p.batch = (size_t)24 << 20;           // (small rounds: several per test file)
    }
    if (const char *e = getenv("PGZ_CALLS")) p.calls = (unsigned)atoi(e);       // provider calls of a round in flight (1, 2, 4)
    // the reference stream: the way GzWriter (and the reference's gzstream) writes
    char tmp[] = "/tmp/pgzcheckXXXXXX";
    const int fd = mkstemp(tmp);
    if (fd < 0) { perror("mkstemp"); return 2; }
    close(fd);
    const auto t0 = std::chrono::steady_clock::now();
    gzFile g = gzopen(tmp, "wb");
    gzbuffer(g, 1 << 18);
    for (size_t o = 0; o < data.size(); o += 100000) gzwrite(g, data.data() + o, (unsigned)std::min<size_t>(100000, data.size() - o));
    gzclose(g);
    const auto t1 = std::chrono::steady_clock::now();
    const std::vector<uint8_t> ref = slurp(tmp);
    unlink(tmp);
    std::vector<uint8_t> out;
    bool ok;
    if (getenv("PGZ_REMOTE")) {
        // a Remote source (the engine's resident text, pd_text_*): the stream is only told how many bytes exist; parse / fetch answer from `data`, and
        // what has been RELEASED is gone — a parse or a fetch below the release mark fails, as it does on the device's ring
        std::atomic<uint64_t> released{0}, n_fetch{0};
        const pgz::ParseFn emu = pgz::host_emulation_parse();
        pgz::Remote src;
        src.parse = [&](uint64_t off, size_t n, const uint64_t *chunks, size_t n_chunks, pgz::SymVec &syms, std::vector<uint64_t> &soff, uint32_t *crc, uint64_t crc_span) -> bool {
            if (off < released.load() || off + n > data.size()) return false;
            if (!emu(data.data() + off, n, chunks, n_chunks, syms, soff)) return false;
            for (size_t k = 0; k < n_chunks; ++k)
                crc[k] = (uint32_t)crc32(crc32(0L, Z_NULL, 0), data.data() + off + chunks[3 * k], (uInt)std::min<uint64_t>(crc_span, chunks[3 * k + 1] - chunks[3 * k]));
            return true;
        };
        src.fetch = [&](uint64_t off, size_t n, uint8_t *dst) { if (off < released.load() || off + n > data.size()) return false; memcpy(dst, data.data() + off, n); ++n_fetch; return true; };
        src.release = [&](uint64_t off) { if (off > released.load()) released = off; };
        pgz::Params d = pgz::Params::for_device(nullptr);
        if (argc > 3) d.chunk = p.chunk;
        if (argc > 4) d.tail = p.tail;
        d.batch = argc > 5 ? p.batch : (size_t)2 << 20;
        pgz::Stream st(threads, [&](const uint8_t *b, size_t k) { out.insert(out.end(), b, b + k); return true; }, d, src);
        ok = true;
        for (size_t o = 0; o < data.size() && ok; o += 700000) ok = st.announce(std::min<size_t>(700000, data.size() - o));
        ok = ok && st.finish();
        fprintf(stderr, "remote source: %llu fetches, released up to %llu of %zu\n", (unsigned long long)n_fetch.load(), (unsigned long long)released.load(), data.size());
    } else if (getenv("PGZ_PIECES")) {
        // the streaming interface, fed in pieces of irregular size (as the per-site writer does)
        pgz::Stream st(threads, [&](const uint8_t *b, size_t k) { out.insert(out.end(), b, b + k); return true; }, p);
        unsigned long long x = 88172645463325252ull;
        ok = true;
        for (size_t o = 0; o < data.size() && ok;) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            const size_t k = std::min<size_t>(data.size() - o, 1 + (size_t)(x % (x % 5 == 0 ? 3000000 : 70000)));
            ok = st.write(data.data() + o, k);
            o += k;
        }
        ok = ok && st.finish();
    } else ok = pgz::gzip_identical(data.data(), data.size(), threads, out, p);
    const auto t2 = std::chrono::steady_clock::now();
    const double ts = std::chrono::duration<double>(t1 - t0).count(), tp = std::chrono::duration<double>(t2 - t1).count();
    if (!ok) { printf("declined (%zu bytes, zlib %.3f s)\n", data.size(), ts); return 3; }
    if (out == ref) { printf("identical %zu -> %zu bytes, zlib %.3f s, parallel(%d) %.3f s\n", data.size(), ref.size(), ts, threads, tp); return 0; }
    size_t i = 0;
    while (i < out.size() && i < ref.size() && out[i] == ref[i]) ++i;
    printf("DIFFERENT at byte %zu (sizes %zu vs %zu)\n", i, out.size(), ref.size());
    return 1;
}
