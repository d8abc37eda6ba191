// report.cpp — see report.h
#include "report.h"
#include "pgzip.h"
#include "options.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <zlib.h>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

namespace pdh {

bool GzWriter::open(const std::string &path)
{
    close();
    path_ = path;
    if (threads_ > 1) {
        fp_ = fopen(path.c_str(), "wb");                 // created now, like gzopen would; filled at close()
        return fp_ != nullptr;
    }
    f_ = gzopen(path.c_str(), "wb");
    if (f_) gzbuffer((gzFile)f_, 1 << 18);
    return f_ != nullptr;
}

void GzWriter::write(const char *p, size_t n)
{
    if (fp_) { text_.append(p, n); return; }
    while (n) {
        const unsigned k = n > (1u << 30) ? (1u << 30) : (unsigned)n;
        gzwrite((gzFile)f_, p, k);
        p += k; n -= k;
    }
}

bool GzWriter::raw(const uint8_t *p, size_t n)
{
    if (!fp_ || !text_.empty()) return false;
    raw_ = true;
    return fwrite(p, 1, n, (FILE *)fp_) == n;
}

bool GzWriter::raw_rewind()
{
    if (!fp_) return false;
    FILE *fp = (FILE *)fp_;
    raw_ = false;
    return fflush(fp) == 0 && ftruncate(fileno(fp), 0) == 0 && fseek(fp, 0, SEEK_SET) == 0;
}

bool GzWriter::close()
{
    if (fp_ && raw_) {                                           // the stream's bytes are in the file already
        FILE *fp = (FILE *)fp_;
        fp_ = nullptr; raw_ = false;
        return fclose(fp) == 0;
    }
    if (fp_) {
        FILE *fp = (FILE *)fp_;
        fp_ = nullptr;
        size_t pgz_min = (size_t)1 << 20;
        if (const char *e = tune("pgz_min")) pgz_min = (size_t)strtoull(e, nullptr, 10);
        std::vector<uint8_t> img;
        bool ok;
        // with the engine's parse (small chunks); else zlib's own parse on the threads (1 MiB chunks with 64 KiB of overlap: a text
        // whose parses do not meet inside the small chunks' overlap); else one zlib stream
        bool done = false;
        if (text_.size() >= pgz_min) {
            if (parse_) done = pgz::gzip_identical((const uint8_t *)text_.data(), text_.size(), threads_, img, pgz::Params::for_device(parse_));
            if (!done) { img.clear(); done = pgz::gzip_identical((const uint8_t *)text_.data(), text_.size(), threads_, img, pgz::Params()); }
        }
        if (done) {
            ok = fwrite(img.data(), 1, img.size(), fp) == img.size();
            ok = fclose(fp) == 0 && ok;
        } else {
            // zlib's own stream (small texts; data pgz does not re-state, e.g. incompressible blocks)
            fclose(fp);
            gzFile g = gzopen(path_.c_str(), "wb");
            ok = g != nullptr;
            if (g) {
                gzbuffer(g, 1 << 18);
                const char *p = text_.data();
                size_t n = text_.size();
                while (n) {
                    const unsigned k = n > (1u << 30) ? (1u << 30) : (unsigned)n;
                    if (gzwrite(g, p, k) != (int)k) { ok = false; break; }
                    p += k; n -= k;
                }
                ok = gzclose(g) == Z_OK && ok;
            }
        }
        std::string().swap(text_);
        return ok;
    }
    if (!f_) return true;
    const int r = gzclose((gzFile)f_);
    f_ = nullptr;
    return r == Z_OK;
}

struct ParallelGzWriter::Impl {
    FILE *f = nullptr;
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_job, cv_done;      // cv_done: a member finished OR was written
    std::deque<std::pair<uint64_t, std::function<void(std::string *)>>> jobs;
    std::map<uint64_t, std::string> done;
    uint64_t next_submit = 0, next_write = 0;
    size_t max_inflight = 8;
    bool stop = false, ok = true;

    void worker()
    {
        std::string text, comp;
        for (;;) {
            std::pair<uint64_t, std::function<void(std::string *)>> job;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_job.wait(lk, [&] { return stop || !jobs.empty(); });
                if (jobs.empty()) return;
                job = std::move(jobs.front()); jobs.pop_front();
            }
            text.clear();
            job.second(&text);
            z_stream zs;
            memset(&zs, 0, sizeof zs);
            bool good = deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) == Z_OK;
            comp.resize(good ? deflateBound(&zs, (uLong)text.size()) + 64 : 0);
            if (good) {
                zs.next_in = (Bytef *)text.data(); zs.avail_in = (uInt)text.size();
                zs.next_out = (Bytef *)&comp[0]; zs.avail_out = (uInt)comp.size();
                good = deflate(&zs, Z_FINISH) == Z_STREAM_END;
                comp.resize(comp.size() - zs.avail_out);
                deflateEnd(&zs);
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                if (!good) ok = false;
                done.emplace(job.first, good ? comp : std::string());
            }
            cv_done.notify_all();
        }
    }
    // caller holds mu: write every member that is next in line
    void drain(std::unique_lock<std::mutex> &lk)
    {
        for (;;) {
            auto it = done.find(next_write);
            if (it == done.end()) return;
            std::string m = std::move(it->second);
            done.erase(it);
            ++next_write;
            lk.unlock();
            if (!m.empty() && fwrite(m.data(), 1, m.size(), f) != m.size()) ok = false;
            lk.lock();
        }
    }
};

ParallelGzWriter::ParallelGzWriter() : p_(new Impl) {}
ParallelGzWriter::~ParallelGzWriter() { close(); delete p_; }

bool ParallelGzWriter::open(const std::string &path, int threads)
{
    p_->f = fopen(path.c_str(), "wb");
    if (!p_->f) return false;
    if (threads < 1) threads = 1;
    p_->max_inflight = (size_t)threads * 2 + 2;
    for (int i = 0; i < threads; ++i) p_->th.emplace_back([this] { p_->worker(); });
    return true;
}

void ParallelGzWriter::submit(std::function<void(std::string *)> make)
{
    std::unique_lock<std::mutex> lk(p_->mu);
    for (;;) {
        p_->drain(lk);                                   // the submitting thread is the writer
        if (p_->next_submit - p_->next_write < p_->max_inflight) break;
        p_->cv_done.wait(lk);                            // woken by every finished member
    }
    p_->jobs.emplace_back(p_->next_submit++, std::move(make));
    lk.unlock();
    p_->cv_job.notify_one();
}

bool ParallelGzWriter::close()
{
    if (!p_->f) return p_->ok;
    {
        std::unique_lock<std::mutex> lk(p_->mu);
        while (p_->next_write < p_->next_submit) {
            p_->drain(lk);
            if (p_->next_write < p_->next_submit) p_->cv_done.wait(lk);
        }
        p_->stop = true;
    }
    p_->cv_job.notify_all();
    for (auto &t : p_->th) t.join();
    p_->th.clear();
    if (p_->next_submit == 0) {                 // nothing written: still a valid (empty) gzip file
        static const unsigned char empty_gz[20] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (fwrite(empty_gz, 1, sizeof empty_gz, p_->f) != sizeof empty_gz) p_->ok = false;
    }
    if (fclose(p_->f) != 0) p_->ok = false;
    p_->f = nullptr;
    return p_->ok;
}

// printf("%.2f") for the finite non-negative doubles the tables hold, without printf: v = M * 2^e exactly, so
// v * 100 = (M * 100) >> -e with an exact remainder, rounded half-to-even on the exact binary value — what
// glibc's correctly rounded printf does.  Anything else (negative, non-finite, >= 2^52) goes through snprintf.
size_t fmt2_to(char *out, double v)
{
    uint64_t bits;
    memcpy(&bits, &v, 8);
    const int be = (int)((bits >> 52) & 0x7ff);
    if ((bits >> 63) || be == 0x7ff || be >= 1023 + 52) return (size_t)snprintf(out, 64, "%.2f", v);
    uint64_t m = bits & ((1ull << 52) - 1);
    int e;                                               // v = m * 2^e
    if (be == 0) e = -1074; else { m |= 1ull << 52; e = be - 1075; }
    const uint64_t n = m * 100;                          // < 2^60
    const int sh = -e;                                   // >= 1 here (v < 2^52)
    uint64_t q;
    if (sh >= 64) q = 0;                                 // n < 2^60 is below half of 2^sh
    else {
        q = n >> sh;
        const uint64_t rem = n & ((1ull << sh) - 1), half = 1ull << (sh - 1);
        if (rem > half || (rem == half && (q & 1))) ++q;
    }
    const uint64_t ip = q / 100;
    const unsigned fp = (unsigned)(q % 100);
    char tmp[24];
    int k = 0;
    uint64_t x = ip;
    do { tmp[k++] = (char)('0' + x % 10); x /= 10; } while (x);
    size_t o = 0;
    while (k) out[o++] = tmp[--k];
    out[o++] = '.'; out[o++] = (char)('0' + fp / 10); out[o++] = (char)('0' + fp % 10);
    out[o] = 0;
    return o;
}

std::string fmt2(double v)
{
    char b[64];
    const size_t n = fmt2_to(b, v);
    return std::string(b, n);
}

void append_u64(std::string *s, uint64_t x)
{
    char tmp[24];
    int k = 0;
    do { tmp[k++] = (char)('0' + x % 10); x /= 10; } while (x);
    while (k) s->push_back(tmp[--k]);
}

void append_i64(std::string *s, int64_t x)
{
    if (x < 0) { s->push_back('-'); append_u64(s, (uint64_t)0 - (uint64_t)x); } else append_u64(s, (uint64_t)x);
}

void append_fmt2(std::string *s, double v)
{
    char b[64];
    const size_t n = fmt2_to(b, v);
    s->append(b, n);
}

} // namespace pdh
