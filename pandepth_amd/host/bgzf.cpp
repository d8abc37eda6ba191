// bgzf.cpp — see bgzf.h
#include "bgzf.h"
#include <dlfcn.h>
#include <fcntl.h>
#include <string.h>
#include <unistd.h>
#include <zlib.h>
#include <condition_variable>
#include <mutex>
#include <thread>

namespace pdh {

namespace {
typedef void *(*ld_alloc_t)(void);
typedef int (*ld_decomp_t)(void *, const void *, size_t, void *, size_t, size_t *);
typedef void (*ld_free_t)(void *);
typedef uint32_t (*ld_crc32_t)(uint32_t, const void *, size_t);
ld_crc32_t g_ld_crc32 = nullptr;
ld_alloc_t g_ld_alloc = nullptr;
ld_decomp_t g_ld_decomp = nullptr;
ld_free_t g_ld_free = nullptr;
std::once_flag g_ld_once;

void load_libdeflate()
{
    const char *names[] = {"libdeflate.so.0", "libdeflate.so", "/usr/lib/x86_64-linux-gnu/libdeflate.so.0"};
    for (const char *n : names) {
        void *h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!h) continue;
        g_ld_alloc = (ld_alloc_t)dlsym(h, "libdeflate_alloc_decompressor");
        g_ld_decomp = (ld_decomp_t)dlsym(h, "libdeflate_deflate_decompress");
        g_ld_free = (ld_free_t)dlsym(h, "libdeflate_free_decompressor");
        g_ld_crc32 = (ld_crc32_t)dlsym(h, "libdeflate_crc32");
        if (g_ld_alloc && g_ld_decomp && g_ld_free) return;
        g_ld_alloc = nullptr; g_ld_decomp = nullptr; g_ld_free = nullptr;
    }
}
} // namespace

Inflater::Inflater()
{
    std::call_once(g_ld_once, load_libdeflate);
    if (g_ld_alloc) ld_ = g_ld_alloc();
    if (!ld_) {
        z_stream *zs = new z_stream;
        memset(zs, 0, sizeof *zs);
        inflateInit2(zs, -15);
        zs_ = zs;
    }
}

Inflater::~Inflater()
{
    if (ld_) g_ld_free(ld_);
    if (zs_) { inflateEnd((z_stream *)zs_); delete (z_stream *)zs_; }
}

const char *Inflater::backend()
{
    std::call_once(g_ld_once, load_libdeflate);
    return g_ld_alloc ? "libdeflate" : "zlib";
}

bool Inflater::inflate_raw(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len)
{
    if (out_len == 0) return true;
    if (ld_) {
        size_t got = 0;
        const int r = g_ld_decomp(ld_, in, in_len, out, out_len, &got);
        return r == 0 && got == out_len;
    }
    z_stream *zs = (z_stream *)zs_;
    inflateReset(zs);
    zs->next_in = const_cast<Bytef *>(in); zs->avail_in = (uInt)in_len;
    zs->next_out = out; zs->avail_out = (uInt)out_len;
    const int r = inflate(zs, Z_FINISH);
    return r == Z_STREAM_END && zs->avail_out == 0;
}

// a whole BGZF member: the payload must inflate to exactly out_len bytes AND those bytes must have the CRC-32 stored behind
// the payload (RFC 1952; htslib's bgzf_read_block checks it too: a flipped bit that still inflates is a corrupt block)
bool Inflater::inflate_member(const uint8_t *payload, size_t in_len, uint8_t *out, size_t out_len)
{
    if (!inflate_raw(payload, in_len, out, out_len)) return false;
    const uint8_t *t = payload + in_len;
    const uint32_t want = t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24);
    const uint32_t got = g_ld_crc32 ? g_ld_crc32(0, out, out_len) : (uint32_t)crc32(crc32(0L, Z_NULL, 0), out, (uInt)out_len);
    return got == want;
}

uint32_t bgzf_block_size(const uint8_t *p, size_t avail, uint32_t *data_off)
{
    if (avail < 18) return 0;
    if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return 0;
    const uint32_t xlen = p[10] | (p[11] << 8);
    if (avail < 12 + xlen) return 0;
    uint32_t o = 12;
    const uint32_t xend = 12 + xlen;
    while (o + 4 <= xend) {
        const uint32_t slen = p[o + 2] | (p[o + 3] << 8);
        if (p[o] == 'B' && p[o + 1] == 'C' && slen == 2) {
            if (o + 6 > xend) return 0;                          // the BC payload itself must lie inside the extra field
            const uint32_t bsize = p[o + 4] | (p[o + 5] << 8);
            // a block holds its header, the extra field and the 8-byte CRC32/ISIZE trailer: callers compute
            // bs - data_off - 8 and read p[bs - 4 ..], so anything shorter is a corrupt block (0), not a size
            if (bsize + 1 < xend + 8) return 0;
            if (data_off) *data_off = xend;
            return bsize + 1;
        }
        o += 4 + slen;
    }
    return 0;
}

// ---- threaded read-ahead ------------------------------------------------------------------
struct BgzfReader::Pipe {
    struct Blk { uint32_t coff, csize, doff, usize; size_t uoff; };
    struct Job {
        std::vector<uint8_t> comp, out;
        std::vector<Blk> blks;
        uint64_t file_off = 0;
        int state = 0;                    // 0 empty, 1 ready to inflate, 2 inflating, 3 done
        bool eof = false, bad = false;
    };
    static constexpr size_t CHUNK = (size_t)1 << 20;
    int fd;
    uint64_t off;
    std::vector<Job> jobs;
    size_t head = 0;                      // consumer position (job sequence number)
    size_t tail = 0;                      // reader position
    size_t next_inflate = 0;              // next job sequence number a worker may take
    size_t cur_blk = 0;
    bool started_job = false;
    bool stop = false;
    std::mutex mu;
    std::condition_variable cv_empty, cv_ready, cv_done;
    std::vector<std::thread> th;

    Pipe(int fd_, uint64_t start, int n) : fd(fd_), off(start), jobs((size_t)n * 3 + 2)
    {
        th.emplace_back([this] { reader(); });
        for (int i = 0; i < n; ++i) th.emplace_back([this] { worker(); });
    }
    ~Pipe()
    {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv_empty.notify_all(); cv_ready.notify_all(); cv_done.notify_all();
        for (auto &t : th) t.join();
    }
    void reader()
    {
        for (;;) {
            Job *j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_empty.wait(lk, [&] { return stop || jobs[tail % jobs.size()].state == 0; });
                if (stop) return;
                j = &jobs[tail % jobs.size()];
            }
            j->comp.resize(CHUNK + 65536);
            const ssize_t n = pread(fd, j->comp.data(), j->comp.size(), (off_t)off);
            j->blks.clear(); j->eof = false; j->bad = false; j->file_off = off;
            size_t p = 0, u = 0;
            if (n <= 0) j->eof = true;
            else {
                while (p + 18 <= (size_t)n && p < CHUNK) {
                    uint32_t doff = 0;
                    const uint32_t bs = bgzf_block_size(j->comp.data() + p, (size_t)n - p, &doff);
                    if (bs == 0) { if (p == 0) j->bad = true; break; }
                    if (p + bs > (size_t)n) break;
                    const uint8_t *q = j->comp.data() + p;
                    const uint32_t isize = q[bs - 4] | (q[bs - 3] << 8) | (q[bs - 2] << 16) | ((uint32_t)q[bs - 1] << 24);
                    j->blks.push_back(Blk{(uint32_t)p, bs, doff, isize, u});
                    u += isize; p += bs;
                }
                if (p == 0 && !j->bad) j->bad = true;        // a partial block at end of file
                off += p;
            }
            j->out.resize(u);
            const bool last = j->eof || j->bad;
            {
                std::lock_guard<std::mutex> lk(mu);
                j->state = 1; ++tail;
            }
            cv_ready.notify_all();
            if (last) return;
        }
    }
    void worker()
    {
        Inflater inf;
        for (;;) {
            Job *j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_ready.wait(lk, [&] { return stop || (next_inflate < tail && jobs[next_inflate % jobs.size()].state == 1); });
                if (stop) return;
                j = &jobs[next_inflate % jobs.size()];
                j->state = 2; ++next_inflate;
            }
            for (const Blk &b : j->blks)
                if (!inf.inflate_member(j->comp.data() + b.coff + b.doff, b.csize - b.doff - 8, j->out.data() + b.uoff, b.usize))
                    j->bad = true;
            {
                std::lock_guard<std::mutex> lk(mu);
                j->state = 3;
            }
            cv_done.notify_all();
            cv_ready.notify_all();
        }
    }
};

BgzfReader::BgzfReader() {}
BgzfReader::~BgzfReader() { close(); }

void BgzfReader::set_threads(int n)
{
    if (!is_bgzf_ || n < 1 || pipe_) return;
    pipe_ = new Pipe(fd_, next_coff_, n);
}

bool BgzfReader::load_block_threaded()
{
    Pipe &P = *pipe_;
    for (;;) {
        Pipe::Job *j;
        {
            std::unique_lock<std::mutex> lk(P.mu);
            if (P.started_job && P.cur_blk >= P.jobs[P.head % P.jobs.size()].blks.size()) {
                P.jobs[P.head % P.jobs.size()].state = 0;      // chunk fully served: recycle
                ++P.head; P.cur_blk = 0; P.started_job = false;
                P.cv_empty.notify_all();
            }
            P.cv_done.wait(lk, [&] { return P.jobs[P.head % P.jobs.size()].state == 3 && P.head < P.tail; });
            j = &P.jobs[P.head % P.jobs.size()];
            P.started_job = true;
        }
        if (j->bad) { err_ = "truncated or corrupt BGZF block"; at_eof_ = true; return false; }
        if (j->eof) { at_eof_ = true; return false; }
        while (P.cur_blk < j->blks.size()) {
            const Pipe::Blk &b = j->blks[P.cur_blk++];
            block_coff_ = j->file_off + b.coff;
            next_coff_ = block_coff_ + b.csize;
            if (b.usize == 0) continue;
            udata_ = j->out.data() + b.uoff;
            upos_ = 0; ulen_ = b.usize;
            return true;
        }
    }
}

void BgzfReader::close()
{
    if (pipe_) { delete pipe_; pipe_ = nullptr; }
    if (gz_) { gzclose((gzFile)gz_); gz_ = nullptr; }
    if (fd_ >= 0) { ::close(fd_); fd_ = -1; }
}

bool BgzfReader::open(const std::string &path, std::string *err)
{
    close();
    fd_ = ::open(path.c_str(), O_RDONLY);
    if (fd_ < 0) { if (err) *err = "cannot open " + path; return false; }
    uint8_t h[18];
    const ssize_t n = pread(fd_, h, sizeof h, 0);
    is_bgzf_ = n == 18 && bgzf_block_size(h, 18, nullptr) != 0;
    at_eof_ = false; upos_ = ulen_ = 0; cpos_ = cend_ = 0; cfile_off_ = 0; block_coff_ = next_coff_ = 0;
    if (!is_bgzf_) {
        // plain text or ordinary gzip: zlib's gz layer reads both transparently
        gz_ = gzdopen(dup(fd_), "rb");
        if (!gz_) { if (err) *err = "cannot read " + path; return false; }
        gzbuffer((gzFile)gz_, 1 << 20);
        ubuf_.resize(1 << 16);
    } else {
        cbuf_.resize(1 << 22);
        ubuf_.resize(1 << 16);
    }
    return true;
}

bool BgzfReader::load_block()
{
    upos_ = ulen_ = 0;
    if (at_eof_) return false;
    if (pipe_) return load_block_threaded();
    udata_ = ubuf_.data();
    if (!is_bgzf_) {
        const int n = gzread((gzFile)gz_, ubuf_.data(), (unsigned)ubuf_.size());
        if (n < 0) { err_ = "read error"; at_eof_ = true; return false; }
        if (n == 0) { at_eof_ = true; return false; }
        block_coff_ = next_coff_; next_coff_ += (uint64_t)n;      // plain offsets (no virtual offsets)
        ulen_ = (size_t)n;
        return true;
    }
    for (;;) {
        uint32_t doff = 0;
        uint32_t bs = cend_ - cpos_ >= 18 ? bgzf_block_size(cbuf_.data() + cpos_, cend_ - cpos_, &doff) : 0;
        if (bs == 0 || cend_ - cpos_ < bs) {
            // refill the read-ahead buffer starting at the current block
            const uint64_t want_off = cfile_off_ + cpos_;
            const ssize_t n = pread(fd_, cbuf_.data(), cbuf_.size(), (off_t)want_off);
            if (n < 0) { err_ = "read error"; at_eof_ = true; return false; }
            cfile_off_ = want_off; cpos_ = 0; cend_ = (size_t)n;
            if (n == 0) { at_eof_ = true; return false; }
            bs = bgzf_block_size(cbuf_.data(), cend_, &doff);
            if (bs == 0 || cend_ < bs) { err_ = "truncated or corrupt BGZF block"; at_eof_ = true; return false; }
        }
        const uint8_t *p = cbuf_.data() + cpos_;
        const uint32_t isize = p[bs - 4] | (p[bs - 3] << 8) | (p[bs - 2] << 16) | ((uint32_t)p[bs - 1] << 24);
        block_coff_ = cfile_off_ + cpos_;
        next_coff_ = block_coff_ + bs;
        cpos_ += bs;
        if (isize == 0) continue;                               // empty block (e.g. the EOF marker)
        if (isize > ubuf_.size()) { ubuf_.resize(isize); }
        udata_ = ubuf_.data();
        if (!inf_.inflate_member(p + doff, bs - doff - 8, ubuf_.data(), isize)) {
            err_ = "corrupt BGZF block (inflate failed)"; at_eof_ = true; return false;
        }
        ulen_ = isize;
        return true;
    }
}

long BgzfReader::read(void *dst, size_t n)
{
    uint8_t *d = (uint8_t *)dst;
    size_t got = 0;
    while (got < n) {
        if (upos_ == ulen_) { if (!load_block()) break; }
        const size_t k = std::min(n - got, ulen_ - upos_);
        memcpy(d + got, udata_ + upos_, k);
        upos_ += k; got += k;
    }
    if (got == 0 && !err_.empty()) return -1;
    return (long)got;
}

const uint8_t *BgzfReader::peek(size_t *avail)
{
    if (upos_ == ulen_) { if (!load_block()) { *avail = 0; return nullptr; } }
    *avail = ulen_ - upos_;
    return udata_ + upos_;
}

void BgzfReader::consume(size_t n) { upos_ += n; }

bool BgzfReader::eof()
{
    if (upos_ < ulen_) return false;
    return !load_block();
}

uint64_t BgzfReader::tell() const
{
    if (upos_ == ulen_) return next_coff_ << 16;
    return (block_coff_ << 16) | (uint64_t)upos_;
}

bool BgzfReader::seek(uint64_t voffset)
{
    if (!is_bgzf_ || pipe_) return false;
    const uint64_t coff = voffset >> 16;
    const uint32_t uoff = (uint32_t)(voffset & 0xffff);
    cfile_off_ = coff; cpos_ = cend_ = 0; at_eof_ = false; upos_ = ulen_ = 0;
    block_coff_ = next_coff_ = coff;
    if (uoff) {
        if (!load_block()) return false;
        if (uoff > ulen_) return false;
        upos_ = uoff;
    }
    return true;
}

} // namespace pdh
