// fasta.cpp — see fasta.h
#include "fasta.h"
#include <ctype.h>
#include <string.h>
#include <zlib.h>
#include <vector>

namespace pdh {

namespace {

// byte source over gzread (plain and gzip files alike, like the reference's gzopen)
struct Bytes {
    gzFile f = nullptr;
    std::vector<unsigned char> buf;
    int beg = 0, end = 0;
    bool eof = false;
    explicit Bytes(gzFile g) : f(g), buf((size_t)1 << 20) {}
    bool fill()
    {
        if (eof) return false;
        beg = 0;
        end = gzread(f, buf.data(), (unsigned)buf.size());
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

bool read_fasta_records(const std::string &path, const std::function<void(const std::string &, std::string &)> &rec)
{
    gzFile g = gzopen(path.c_str(), "r");
    if (!g) return false;
    gzbuffer(g, 1u << 20);
    Bytes in(g);
    int last = 0;                                        // header character already consumed
    std::string name, seq, qual;
    for (;;) {
        int c;
        if (last == 0) {
            while ((c = in.getc()) != -1 && c != '>' && c != '@') {}
            if (c == -1) break;
        }
        last = 0;
        name.clear(); seq.clear();
        bool any;
        c = in.until(true, &name, &any);
        if (!any) break;
        if (c != '\n' && c != -1) { std::string comment; in.until(false, &comment, &any); }
        while ((c = in.getc()) != -1 && c != '>' && c != '+' && c != '@') {
            if (c == '\n') continue;
            seq.push_back((char)c);
            in.until(false, &seq, &any);
            if (seq.size() > 1 && seq.back() == '\r') seq.pop_back();
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
                    if (qual.size() >= seq.size() || d == -1) break;
                }
                if (qual.size() != seq.size()) ok = false;
            }
        }
        if (!ok) break;                                  // kseq_read < 0 ends the caller's loop
        rec(name, seq);
    }
    gzclose(g);
    return true;
}

bool load_reference(const std::string &path, std::map<std::string, int32_t> *chr2tid, RefSeqs *out)
{
    out->loaded = true;
    return read_fasta_records(path, [&](const std::string &name, std::string &seq) {
        const size_t z = seq.find('\0');                 // `string seqBB = seq->seq.s` stops at a NUL
        if (z != std::string::npos) seq.resize(z);
        auto it = chr2tid->find(name);
        int32_t id = 0;
        if (it == chr2tid->end()) (*chr2tid)[name] = 0; else id = it->second;
        if (out->seq.find(id) == out->seq.end()) out->seq.emplace(id, std::move(seq));      // first claim wins
    });
}

uint64_t RefSeqs::gc(int32_t tid, int64_t first, int64_t last) const
{
    auto it = seq.find(tid);
    if (it == seq.end()) return 0;
    const std::string &s = it->second;
    int64_t lo = first - 1, hi = last;                   // cells [first-1, last)
    if (lo < 0) lo = 0;
    if (hi > (int64_t)s.size()) hi = (int64_t)s.size();
    uint64_t n = 0;
    const unsigned char *p = (const unsigned char *)s.data();
    for (int64_t i = lo; i < hi; ++i) {
        const unsigned c = p[i] | 0x20u;
        n += (unsigned)(c == 'c') | (unsigned)(c == 'g');
    }
    return n;
}

} // namespace pdh
