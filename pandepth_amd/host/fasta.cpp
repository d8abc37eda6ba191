// fasta.cpp — see fasta.h
#include "fasta.h"
#include <ctype.h>
#include <string.h>
#include <zlib.h>
#include <fcntl.h>
#include <unistd.h>
#include <sys/stat.h>
#include <algorithm>
#include <vector>

namespace pdh {

namespace {

// byte source over gzread (plain and gzip files alike, like the reference's gzopen)
struct Bytes {
    gzFile f = nullptr;
    int fd = -1;                                     // plain files are read directly (zlib's pass-through copies twice)
    std::vector<unsigned char> buf;
    int beg = 0, end = 0;
    bool eof = false;
    Bytes(gzFile g, int plain_fd) : f(g), fd(plain_fd), buf((size_t)4 << 20) {}
    bool fill()
    {
        if (eof) return false;
        beg = 0;
        end = f ? gzread(f, buf.data(), (unsigned)buf.size()) : (int)read(fd, buf.data(), buf.size());
        if (end <= 0) { end = 0; eof = true; return false; }
        return true;
    }
    int getc() { if (beg >= end && !fill()) return -1; return buf[beg++]; }
    // appends up to (not including) the next '\n' (or white space when `space`); returns the delimiter or -1 at EOF;
    // *any reports whether the stream still had data
    int until(bool space, std::string *s, bool *any)
    {
        *any = false;
        for (;;) {
            if (beg >= end && !fill()) return -1;
            *any = true;
            int i = beg;
            if (space) { while (i < end && !isspace(buf[i])) ++i; }
            else { const void *p = memchr(buf.data() + beg, '\n', (size_t)(end - beg)); i = p ? (int)((const unsigned char *)p - buf.data()) : end; }
            s->append((const char *)buf.data() + beg, (size_t)(i - beg));
            beg = i + 1;
            if (i < end) return buf[i];
        }
    }
};

} // namespace

bool read_fasta_records(const std::string &path, std::string *dst, const std::function<void(const std::string &, size_t, size_t)> &rec)
{
    int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    unsigned char magic[2] = {0, 0};
    const bool gz = pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
    struct stat st;
    if (!gz && fstat(fd, &st) == 0) dst->reserve(dst->size() + (size_t)st.st_size);
    gzFile g = nullptr;
    if (gz) {
        g = gzdopen(fd, "r");
        if (!g) { close(fd); return false; }
        gzbuffer(g, 1u << 20);
    }
    Bytes in(g, fd);
    int last = 0;                                        // header character already consumed
    std::string name, qual;
    std::string &seq = *dst;
    for (;;) {
        int c;
        if (last == 0) {
            while ((c = in.getc()) != -1 && c != '>' && c != '@') {}
            if (c == -1) break;
        }
        last = 0;
        name.clear();
        const size_t off = seq.size();
        bool any;
        c = in.until(true, &name, &any);
        if (!any) break;
        if (c != '\n' && c != -1) { std::string comment; in.until(false, &comment, &any); }
        while ((c = in.getc()) != -1 && c != '>' && c != '+' && c != '@') {
            if (c == '\n') continue;
            seq.push_back((char)c);
            in.until(false, &seq, &any);
            if (seq.size() - off > 1 && seq.back() == '\r') seq.pop_back();
        }
        if (c == '>' || c == '@') last = c;
        bool ok = true;
        if (c == '+') {
            while ((c = in.getc()) != -1 && c != '\n') {}
            if (c == -1) ok = false;
            else {
                qual.clear();
                for (;;) {
                    const int d = in.until(false, &qual, &any);
                    if (!any) break;
                    if (qual.size() > 1 && qual.back() == '\r') qual.pop_back();
                    if (qual.size() >= seq.size() - off || d == -1) break;
                }
                if (qual.size() != seq.size() - off) ok = false;
            }
        }
        if (!ok) { seq.resize(off); break; }             // kseq_read < 0 ends the caller's loop
        rec(name, off, seq.size() - off);
    }
    if (g) gzclose(g); else close(fd);
    return true;
}

bool load_reference(const std::string &path, std::map<std::string, int32_t> *chr2tid, RefSeqs *out)
{
    out->loaded = true;
    std::string *arena = out->arena();
    return read_fasta_records(path, arena, [&](const std::string &name, size_t off, size_t n) {
        const void *z = memchr(arena->data() + off, 0, n);     // `string seqBB = seq->seq.s` stops at a NUL
        if (z) n = (size_t)((const char *)z - (arena->data() + off));
        auto it = chr2tid->find(name);
        int32_t id = 0;
        if (it == chr2tid->end()) (*chr2tid)[name] = 0; else id = it->second;
        out->claim(id, off, n);                              // first claim wins
    });
}

bool RefSeqs::claim(int32_t tid, size_t off, size_t n)
{
    if (span_.find(tid) != span_.end()) { arena_.resize(off); return false; }
    span_[tid] = {off, n};
    arena_.resize(off + n);
    return true;
}

uint64_t RefSeqs::gc(int32_t tid, int64_t first, int64_t last) const
{
    auto it = span_.find(tid);
    if (it == span_.end()) return 0;
    int64_t lo = first - 1, hi = last;                   // cells [first-1, last)
    if (lo < 0) lo = 0;
    if (hi > (int64_t)it->second.second) hi = (int64_t)it->second.second;
    const unsigned char *p = (const unsigned char *)arena_.data() + it->second.first;
    // 8 bases per step: byte == 'c' or 'g' after folding the case bit; (x ^ pattern) has a zero byte exactly there
    uint64_t n = 0;
    int64_t i = lo;
    const uint64_t ones = 0x0101010101010101ull, low7 = 0x7f7f7f7f7f7f7f7full;
    auto zero_bytes = [&](uint64_t v) { return ~(((v & low7) + low7) | v | low7); };      // 0x80 in every byte that is zero
    while (i + 8 <= hi) {
        // per-byte counters for up to 255 words, then one horizontal sum (no popcount instruction is assumed)
        uint64_t acc = 0;
        const int64_t stop = std::min<int64_t>(hi - 7, i + 8 * 255);
        for (; i < stop; i += 8) {
            uint64_t w;
            memcpy(&w, p + i, 8);
            w |= 0x2020202020202020ull;
            acc += (zero_bytes(w ^ (ones * 'c')) | zero_bytes(w ^ (ones * 'g'))) >> 7;
        }
        n += ((acc & 0x00ff00ff00ff00ffull) * 0x0001000100010001ull >> 48) + (((acc >> 8) & 0x00ff00ff00ff00ffull) * 0x0001000100010001ull >> 48);
    }
    for (; i < hi; ++i) {
        const unsigned c = p[i] | 0x20u;
        n += (unsigned)(c == 'c') | (unsigned)(c == 'g');
    }
    return n;
}

} // namespace pdh
