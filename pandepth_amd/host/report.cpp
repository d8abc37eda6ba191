// report.cpp — see report.h
#include "report.h"
#include <stdio.h>
#include <string.h>
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
    f_ = gzopen(path.c_str(), "wb");
    if (f_) gzbuffer((gzFile)f_, 1 << 18);
    return f_ != nullptr;
}

void GzWriter::write(const char *p, size_t n)
{
    while (n) {
        const unsigned k = n > (1u << 30) ? (1u << 30) : (unsigned)n;
        gzwrite((gzFile)f_, p, k);
        p += k; n -= k;
    }
}

bool GzWriter::close()
{
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

std::string fmt2(double v)
{
    char b[64];
    snprintf(b, sizeof b, "%.2f", v);
    return b;
}

} // namespace pdh
