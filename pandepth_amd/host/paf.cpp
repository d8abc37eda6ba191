// paf.cpp — see paf.h
#include "paf.h"
#include <ctype.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include <fcntl.h>
#include <unistd.h>
#include <vector>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

namespace pdh {

namespace {

// lines of a plain or gzip file, the way `while (!in.eof()) getline(in, line)` sees them, without holding the file
struct GzLines {
    gzFile f = nullptr;
    std::vector<char> buf;
    size_t beg = 0, end = 0;
    bool eof = false;
    int fd = -1;                                     // plain files bypass zlib's copy
    explicit GzLines(const std::string &path) : buf((size_t)4 << 20)
    {
        fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) return;
        unsigned char magic[2] = {0, 0};
        const ssize_t got = pread(fd, magic, 2, 0);
        if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
            f = gzdopen(fd, "rb");
            if (f) { gzbuffer(f, 1u << 20); fd = -1; } else { close(fd); fd = -1; }
        }
    }
    ~GzLines() { if (f) gzclose(f); if (fd >= 0) close(fd); }
    int fill() { return read_raw(buf.data(), buf.size()); }
    int read_raw(char *dst, size_t n)
    {
        if (n > ((size_t)1 << 30)) n = (size_t)1 << 30;
        return f ? gzread(f, dst, (unsigned)n) : (int)read(fd, dst, n);
    }
    bool next(std::string *line)
    {
        line->clear();
        if (!f && fd < 0) return false;
        bool any = false;
        for (;;) {
            if (beg >= end) {
                if (eof) return any;
                const int n = fill();
                if (n <= 0) { eof = true; return any; }
                beg = 0; end = (size_t)n;
            }
            any = true;
            const char *p = (const char *)memchr(buf.data() + beg, '\n', end - beg);
            if (p) { line->append(buf.data() + beg, (size_t)(p - (buf.data() + beg))); beg = (size_t)(p - buf.data()) + 1; return true; }
            line->append(buf.data() + beg, end - beg);
            beg = end;
        }
    }
};

} // namespace

bool is_paf_path(const std::string &path)
{
    auto ext_of = [](const std::string &p) { const size_t d = p.rfind('.'); return d == std::string::npos ? std::string() : p.substr(d + 1); };
    std::string ext = ext_of(path);
    if (ext == "gz") ext = ext_of(path.substr(0, path.rfind('.')));
    return ext == "paf" || ext == "PAF";
}

bool paf_targets(const Options &o, AlnHeader *hdr, std::map<std::string, int32_t> *chr2tid, RefSeqs *ref)
{
    hdr->names.clear(); hdr->lens.clear(); chr2tid->clear();
    if (!o.reference.empty()) {
        // PD:873-907: one target per FASTA record, ids in file order; a repeated name keeps both targets and the
        // name points at the later one.  (Without -c the reference still switches its GC column on and then counts
        // in strings it never filled; the sequences are kept here either way and the column is real.)
        ref->loaded = true;
        std::string *arena = ref->arena();
        return read_fasta_records(o.reference, arena, [&](const std::string &name, size_t off, size_t n) {
            const int32_t id = (int32_t)hdr->names.size();
            (*chr2tid)[name] = id;
            hdr->names.push_back(name);
            hdr->lens.push_back((uint32_t)n);
            const void *z = memchr(arena->data() + off, 0, n);
            ref->claim(id, off, z ? (size_t)((const char *)z - (arena->data() + off)) : n);
        });
    }
    // PD:917-942: columns 6 and 7 of the first file; a short line re-uses what the previous line left in the variables
    GzLines in(o.input);
    std::string line, chr;
    int len = 0;
    while (in.next(&line)) {
        if (line.empty()) continue;
        // `isone >> tmp1 >> tmp2 >> tmp3 >> tmp4 >> tmp5 >> chr >> chrlength`: tokens between white space; what a short
        // line does not reach keeps its previous value; the number is read the way num_get reads an int
        const char *p = line.data(), *e = p + line.size();
        int k = 0;
        for (; k < 7; ++k) {
            while (p < e && isspace((unsigned char)*p)) ++p;
            if (p >= e) break;
            const char *b = p;
            if (k == 6) {
                const char *q = p;
                bool neg = false;
                if (q < e && (*q == '+' || *q == '-')) { neg = *q == '-'; ++q; }
                const char *d0 = q;
                long long v = 0;
                while (q < e && *q >= '0' && *q <= '9') { if (v < (1LL << 40)) v = v * 10 + (*q - '0'); ++q; }
                if (q == d0) len = 0;                      // no digits: failbit, and C++11 stores 0
                else { if (neg) v = -v; len = v > 2147483647LL ? 2147483647 : v < -2147483648LL ? (int)-2147483648LL : (int)v; }
                break;
            }
            while (p < e && !isspace((unsigned char)*p)) ++p;
            if (k == 5) chr.assign(b, (size_t)(p - b));
        }
        if (chr2tid->find(chr) == chr2tid->end()) {
            (*chr2tid)[chr] = (int32_t)hdr->names.size();
            hdr->names.push_back(chr);
            hdr->lens.push_back((uint32_t)len);
        }
    }
    return true;
}

namespace {

// one PAF line -> runs (PD:1549-1612)
struct LineParser {
    const Options &o;
    const std::map<std::string, int32_t> &tab;
    RunEmitter *out;
    uint64_t n_records = 0;
    const bool skip_secondary;
    std::vector<std::pair<const char *, size_t>> f;
    std::string key;
    struct Op { int32_t n; char c; };
    std::vector<Op> ops;
    LineParser(const Options &opt, const std::map<std::string, int32_t> &t, RunEmitter *e)
        : o(opt), tab(t), out(e), skip_secondary((opt.flag_mask & 0x100u) != 0) {}

    void line(const char *b, const char *e)
    {
        if (b == e) return;
        if (skip_secondary && memmem(b, (size_t)(e - b), "tp:A:S", 6)) return;
        f.clear();
        for (const char *p = b; p < e;) {
            while (p < e && (*p == ' ' || *p == '\t')) ++p;
            if (p >= e) break;
            const char *t = p;
            while (p < e && *p != ' ' && *p != '\t') ++p;
            f.emplace_back(t, (size_t)(p - t));
        }
        if (f.size() < 12) return;
        // the reference looks the name up with operator[]: a name the table does not know is ENTERED as target 0 — the
        // same answer as not finding it, so the table stays read-only here (and may be shared by parser threads)
        key.assign(f[5].first, f[5].second);
        auto it = tab.find(key);
        const int32_t tid = it == tab.end() ? 0 : it->second;
        auto num = [&](size_t k) { char tmp[32]; const size_t n = f[k].second < 31 ? f[k].second : 31; memcpy(tmp, f[k].first, n); tmp[n] = 0; return atoi(tmp); };
        if (num(11) < o.min_mapq) return;
        int32_t s = num(7), en = num(8);
        if (s > en) { const int32_t t = s; s = en; en = t; }
        size_t cg = 0;
        for (size_t k = 0; k < f.size(); ++k) if (f[k].second >= 5 && memcmp(f[k].first, "cg:Z:", 5) == 0) { cg = k; break; }
        if (cg > 1) {
            // PD:806-833 + PD:1585-1608: <number><op> pairs; M/=/X are runs, D/N advance, everything else is ignored
            const char *p = f[cg].first + 5, *pe = f[cg].first + f[cg].second;
            ops.clear();
            while (p < pe) {
                if (!(*p >= '0' && *p <= '9')) return;
                int64_t v = 0;
                while (p < pe && *p >= '0' && *p <= '9') { v = v * 10 + (*p - '0'); if (v > 0x7fffffffLL) return; ++p; }
                ops.push_back(Op{(int32_t)v, p < pe ? *p : '\0'});
                ++p;
            }
            ++n_records;
            int32_t cur = s;
            for (const Op &x : ops) {
                if (x.c == 'M' || x.c == '=' || x.c == 'X') { out->emit(tid, cur, cur + x.n); cur += x.n; }
                else if (x.c == 'D' || x.c == 'N') cur += x.n;
            }
        } else {
            ++n_records;
            if (en > s - 1) out->emit(tid, s - 1, en);
        }
    }
    void block(const char *b, const char *e)
    {
        while (b < e) {
            const char *nl = (const char *)memchr(b, '\n', (size_t)(e - b));
            if (!nl) nl = e;
            line(b, nl);
            b = nl + 1;
        }
    }
};

} // namespace

bool read_paf(const std::string &path, const Options &o, const std::map<std::string, int32_t> &chr2tid,
              const std::function<std::unique_ptr<RunEmitter>()> &make_emitter, int threads, uint64_t *n_records)
{
    GzLines in(path);
    if (!in.f && in.fd < 0) return true;                   // (the reference's gzstream reports nothing either)
    // The reader cuts the byte stream into blocks of whole lines; parser threads (each with its own emitter) take them.
    constexpr size_t BLOCK = (size_t)8 << 20;
    const int T = std::max(1, std::min(threads, 16));
    std::mutex mu;
    std::condition_variable cv_put, cv_get;
    std::deque<std::vector<char>> queue;
    bool done = false;
    std::atomic<uint64_t> total{0};
    auto worker = [&]() {
        std::unique_ptr<RunEmitter> em = make_emitter();
        LineParser lp(o, chr2tid, em.get());
        for (;;) {
            std::vector<char> blk;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_get.wait(lk, [&] { return done || !queue.empty(); });
                if (queue.empty()) break;
                blk.swap(queue.front()); queue.pop_front();
            }
            cv_put.notify_one();
            lp.block(blk.data(), blk.data() + blk.size());
        }
        total += lp.n_records;
    };
    std::vector<std::thread> th;
    for (int k = 0; k < T; ++k) th.emplace_back(worker);
    std::vector<char> carry;
    for (;;) {
        std::vector<char> blk(carry.size() + BLOCK);
        if (!carry.empty()) memcpy(blk.data(), carry.data(), carry.size());
        size_t have = carry.size();
        carry.clear();
        bool eof = false;
        while (have < blk.size()) {
            const int n = in.read_raw(blk.data() + have, blk.size() - have);
            if (n <= 0) { eof = true; break; }
            have += (size_t)n;
        }
        size_t cut = have;
        if (!eof) {                                        // keep the unfinished last line for the next block
            while (cut > 0 && blk[cut - 1] != '\n') --cut;
            if (cut == 0) { carry.assign(blk.begin(), blk.begin() + have); if (carry.size() > ((size_t)1 << 30)) eof = true; else continue; }
            carry.assign(blk.begin() + cut, blk.begin() + have);
        }
        blk.resize(cut);
        if (!blk.empty()) {
            std::unique_lock<std::mutex> lk(mu);
            cv_put.wait(lk, [&] { return queue.size() < (size_t)T * 2; });
            queue.emplace_back(std::move(blk));
            lk.unlock();
            cv_get.notify_one();
        }
        if (eof) break;
    }
    { std::lock_guard<std::mutex> lk(mu); done = true; }
    cv_get.notify_all();
    for (auto &t : th) t.join();
    *n_records += total.load();
    return true;
}

} // namespace pdh
