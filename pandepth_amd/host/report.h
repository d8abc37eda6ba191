// report.h — gz text output with the reference's byte stream: one zlib gz member, default level,
// written through gzopen/gzwrite like its gzstream wrapper (include/gzstream.c:46-67,133-142), so
// the .gz bytes depend only on the text.  Number formatting = iostream fixed/setprecision(2),
// i.e. printf("%.2f").
#ifndef PD_REPORT_H_
#define PD_REPORT_H_
#include <stdint.h>
#include <functional>
#include <string>
#include "pgzip.h"

namespace pdh {

class GzWriter {
public:
    ~GzWriter() { close(); }
    bool open(const std::string &path);
    // threads > 1: the text is collected and deflated at close() by pgz::gzip_identical (host/pgzip.h) —
    // the SAME bytes as the single zlib stream, with the LZ77 parse spread over the threads; texts
    // below PANDEPTH_PGZ_MIN bytes (default 1 MiB), and texts pgz declines, take zlib's serial stream
    void set_threads(int threads) { threads_ = threads; }
    // stage 1 of that stream (zlib's LZ77 parse) done by the engine instead of the host threads (pgz::ParseFn; pd_deflate_parse)
    void set_parse(pgz::ParseFn fn) { parse_ = std::move(fn); }
    void write(const char *p, size_t n);
    void write(const std::string &s) { write(s.data(), s.size()); }
    // the finished gzip stream's bytes from elsewhere (the engine parsed text that never came to the host: pipeline.cpp's resident
    // table writer) instead of text; collecting mode only (threads > 1), nothing written as text before.  raw_rewind(): forget them.
    bool raw(const uint8_t *p, size_t n);
    bool raw_rewind();
    bool collecting() const { return fp_ != nullptr && text_.empty(); }
    bool close();
    bool good() const { return f_ != nullptr || fp_ != nullptr; }
private:
    void *f_ = nullptr;          // gzFile: streaming mode
    void *fp_ = nullptr;         // FILE*: collecting mode
    int threads_ = 1;
    pgz::ParseFn parse_;
    std::string path_, text_;
    bool raw_ = false;
};

// Large per-site outputs: text chunks are produced AND deflated on worker threads, each chunk
// becoming its own gzip member; members are written in submission order.  Concatenated members are
// one valid .gz whose decompressed content is exactly the concatenated text (the byte stream of the
// .gz differs from a single-member file, which is why small outputs do not use this).
class ParallelGzWriter {
public:
    ParallelGzWriter();
    ~ParallelGzWriter();
    bool open(const std::string &path, int threads);
    // `make` fills the text of the next chunk; it runs on a worker thread
    void submit(std::function<void(std::string *)> make);
    bool close();
private:
    struct Impl;
    Impl *p_;
};

std::string fmt2(double v);          // "%.2f" (exact, without printf for the values tables hold)
size_t fmt2_to(char *out, double v); // out >= 64 bytes; returns the length
void append_fmt2(std::string *s, double v);
void append_u64(std::string *s, uint64_t x);
void append_i64(std::string *s, int64_t x);

} // namespace pdh
#endif
