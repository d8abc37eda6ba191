// bam.cpp — see bam.h
#include "bam.h"
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include <algorithm>
#include <map>

namespace pdh {

bool file_exists(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0; }
uint64_t file_size(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0 ? (uint64_t)st.st_size : 0; }

static inline uint32_t le32(const uint8_t *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
static inline uint64_t le64(const uint8_t *p) { return (uint64_t)le32(p) | ((uint64_t)le32(p + 4) << 32); }

bool AlnHeader::sorted_coordinate() const
{
    const std::string::size_type a = text.find("\tSO:");
    if (a == std::string::npos) return false;
    const std::string::size_type b = text.find_first_of("\n\t", a + 4);
    return text.substr(a + 4, b == std::string::npos ? std::string::npos : b - (a + 4)) == "coordinate";
}

int32_t AlnRec::endpos() const
{
    int32_t rlen = 0;
    if (!(flag & 4) && n_cigar > 0) {
        for (uint32_t i = 0; i < n_cigar; ++i) {
            const uint32_t op = cigar[i] & 0xf;
            if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rlen = (int32_t)((uint32_t)rlen + (cigar[i] >> 4));   // wraps like the reference's int on corrupt lengths, without UB
        }
    } else {
        rlen = 1;
    }
    if (rlen == 0) rlen = 1;
    return (int32_t)((uint32_t)pos + (uint32_t)rlen);
}

bool AlnReader::open(const std::string &path, std::string *err)
{
    is_cram_ = CramReader::is_cram(path);
    if (is_cram_) {
        is_bam_ = false;
        if (cram_.open(path, &hdr_, &err_)) return true;
        if (err) *err = err_;
        return false;
    }
    if (!bg_.open(path, err)) return false;
    size_t av = 0;
    const uint8_t *p = bg_.peek(&av);
    is_bam_ = p && av >= 4 && memcmp(p, "BAM\1", 4) == 0;
    const bool ok = is_bam_ ? read_bam_header() : read_sam_header();
    if (!ok && err) *err = err_.empty() ? "cannot read the header of " + path : err_;
    return ok;
}

bool AlnReader::read_bam_header()
{
    uint8_t b[8];
    if (!bg_.read_exact(b, 8)) { err_ = "truncated BAM header"; return false; }
    const uint32_t l_text = le32(b + 4);
    if (l_text > 0x7fffffffu) { err_ = "damaged BAM header (text length)"; return false; }      // int32 in the specification
    hdr_.text.resize(l_text);
    if (l_text && !bg_.read_exact(&hdr_.text[0], l_text)) { err_ = "truncated BAM header"; return false; }
    const std::string::size_type z = hdr_.text.find('\0');
    if (z != std::string::npos) hdr_.text.resize(z);
    if (!bg_.read_exact(b, 4)) { err_ = "truncated BAM header"; return false; }
    const uint32_t n_ref = le32(b);
    for (uint32_t i = 0; i < n_ref; ++i) {
        if (!bg_.read_exact(b, 4)) { err_ = "truncated BAM header"; return false; }
        const uint32_t l_name = le32(b);
        if (l_name > (1u << 20)) { err_ = "damaged BAM header (reference name length)"; return false; }
        std::string nm(l_name, '\0');
        if (l_name && !bg_.read_exact(&nm[0], l_name)) { err_ = "truncated BAM header"; return false; }
        if (!nm.empty() && nm.back() == '\0') nm.pop_back();
        if (!bg_.read_exact(b, 4)) { err_ = "truncated BAM header"; return false; }
        hdr_.names.push_back(nm);
        hdr_.lens.push_back(le32(b));
    }
    return true;
}

bool AlnReader::getline(std::string *line)
{
    line->clear();
    for (;;) {
        size_t av = 0;
        const uint8_t *p = bg_.peek(&av);
        if (!p || av == 0) return !line->empty();
        const uint8_t *nl = (const uint8_t *)memchr(p, '\n', av);
        if (nl) {
            line->append((const char *)p, (size_t)(nl - p));
            bg_.consume((size_t)(nl - p) + 1);
            if (!line->empty() && line->back() == '\r') line->pop_back();
            return true;
        }
        line->append((const char *)p, av);
        bg_.consume(av);
    }
}

bool AlnReader::read_sam_header()
{
    std::string line;
    while (getline(&line)) {
        if (line.empty()) continue;
        if (line[0] != '@') { pending_line_ = line; have_pending_ = true; break; }
        hdr_.text += line; hdr_.text += '\n';
        if (line.compare(0, 3, "@SQ") == 0) {
            std::string sn; uint32_t ln = 0;
            size_t o = 0;
            while (o < line.size()) {
                size_t e = line.find('\t', o); if (e == std::string::npos) e = line.size();
                if (line.compare(o, 3, "SN:") == 0) sn = line.substr(o + 3, e - o - 3);
                else if (line.compare(o, 3, "LN:") == 0) ln = (uint32_t)strtoul(line.c_str() + o + 3, nullptr, 10);
                o = e + 1;
            }
            name2tid_[sn] = (int32_t)hdr_.names.size();
            hdr_.names.push_back(sn); hdr_.lens.push_back(ln);
        }
    }
    return true;
}

static inline int cigar_op_code(char c)
{
    switch (c) {
    case 'M': return 0; case 'I': return 1; case 'D': return 2; case 'N': return 3; case 'S': return 4;
    case 'H': return 5; case 'P': return 6; case '=': return 7; case 'X': return 8; case 'B': return 9;
    default: return -1;
    }
}

int AlnReader::next_sam(AlnRec *r)
{
    std::string line;
    for (;;) {
        if (have_pending_) { line.swap(pending_line_); have_pending_ = false; }
        else if (!getline(&line)) return 0;
        if (line.empty() || line[0] == '@') continue;
        break;
    }
    // QNAME FLAG RNAME POS MAPQ CIGAR ...
    const char *f[6]; size_t fl[6];
    size_t o = 0; int k = 0;
    while (k < 6) {
        size_t e = line.find('\t', o); if (e == std::string::npos) e = line.size();
        f[k] = line.c_str() + o; fl[k] = e - o; ++k;
        if (e == line.size()) break;
        o = e + 1;
    }
    if (k < 6) { err_ = "malformed SAM line"; return -1; }
    r->flag = (uint16_t)strtoul(f[1], nullptr, 10);
    const std::string rname(f[2], fl[2]);
    if (rname == "*") r->tid = -1;
    else {
        auto it = name2tid_.find(rname);
        if (it == name2tid_.end()) { err_ = "SAM line names an unknown reference: " + rname; return -1; }
        r->tid = it->second;
    }
    r->pos = (int32_t)strtol(f[3], nullptr, 10) - 1;
    r->mapq = (uint8_t)strtoul(f[4], nullptr, 10);
    cig_.clear();
    if (!(fl[5] == 1 && f[5][0] == '*')) {
        uint32_t num = 0;
        for (size_t i = 0; i < fl[5]; ++i) {
            const char c = f[5][i];
            if (c >= '0' && c <= '9') num = num * 10 + (uint32_t)(c - '0');
            else {
                const int op = cigar_op_code(c);
                if (op < 0) { err_ = "malformed CIGAR"; return -1; }
                cig_.push_back((num << 4) | (uint32_t)op); num = 0;
            }
        }
    }
    r->n_cigar = (uint32_t)cig_.size();
    r->cigar = cig_.data();
    return 1;
}

int AlnReader::next(AlnRec *r)
{
    if (is_cram_) { const int k = cram_.next(r); if (k < 0) err_ = cram_.error(); return k; }
    if (!is_bam_) return next_sam(r);
    size_t av = 0;
    const uint8_t *p = bg_.peek(&av);
    if (!p || av == 0) return bg_.error().empty() ? 0 : -1;
    uint8_t szb[4];
    const uint8_t *rec;
    uint32_t bs;
    if (av >= 4 && (bs = le32(p), av >= 4 + (size_t)bs)) {
        rec = p + 4;                                   // whole record inside the current block
        bg_.consume(4 + (size_t)bs);
    } else {
        if (!bg_.read_exact(szb, 4)) { err_ = "truncated BAM record"; return -1; }
        bs = le32(szb);
        if (bs < 32 || bs > (1u << 30)) { err_ = "corrupt BAM record"; return -1; }     // before anything is sized from it
        if (rec_.size() < bs) rec_.resize(bs);
        if (!bg_.read_exact(rec_.data(), bs)) { err_ = "truncated BAM record"; return -1; }
        rec = rec_.data();
    }
    if (bs < 32) { err_ = "corrupt BAM record"; return -1; }
    r->tid = (int32_t)le32(rec);
    r->pos = (int32_t)le32(rec + 4);
    const uint32_t l_read_name = rec[8];
    r->mapq = rec[9];
    uint32_t n_cigar = rec[12] | (rec[13] << 8);
    r->flag = (uint16_t)(rec[14] | (rec[15] << 8));
    if (32 + l_read_name + 4 * (size_t)n_cigar > bs) { err_ = "corrupt BAM record"; return -1; }
    const uint8_t *cg = rec + 32 + l_read_name;
    // CIGARs with > 65535 operations are stored in the CG:B,I tag behind a <l_seq>S<ref_len>N
    // placeholder (SAM spec §4.2.2); long reads need this.
    const uint32_t l_seq = le32(rec + 16);
    // htslib's test (bam_tag2cigar): any CIGAR whose first operation is <l_seq>S on a placed read (tid, pos >= 0), and a
    // CG tag of type B,I or B,i with at least n_cigar entries; the fake CIGAR is kept when no such tag exists.
    if (n_cigar >= 1 && r->tid >= 0 && r->pos >= 0 && (le32(cg) & 0xf) == 4 && (le32(cg) >> 4) == l_seq) {
        const uint8_t *aux = cg + 4 * (size_t)n_cigar + (l_seq + 1) / 2 + (size_t)l_seq, *end = rec + bs;
        if (aux > end) aux = end;
        while (aux + 3 <= end) {
            const char t0 = (char)aux[0], t1 = (char)aux[1], ty = (char)aux[2];
            aux += 3;
            size_t sz = 0;
            if (ty == 'A' || ty == 'c' || ty == 'C') sz = 1;
            else if (ty == 's' || ty == 'S') sz = 2;
            else if (ty == 'i' || ty == 'I' || ty == 'f') sz = 4;
            else if (ty == 'Z' || ty == 'H') { const uint8_t *z = (const uint8_t *)memchr(aux, 0, (size_t)(end - aux)); if (!z) break; sz = (size_t)(z - aux) + 1; }
            else if (ty == 'B') {
                if (aux + 5 > end) break;
                const char st = (char)aux[0];
                const uint32_t cnt = le32(aux + 1);
                const size_t es = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4;
                if (t0 == 'C' && t1 == 'G') {
                    // the first CG tag decides (bam_aux_get): wrong subtype or too short means "keep the fake CIGAR"
                    if (!((st == 'I' || st == 'i') && cnt >= n_cigar && cnt < (1u << 29) && aux + 5 + 4 * (size_t)cnt <= end)) break;
                    cig_.resize(cnt);
                    for (uint32_t i = 0; i < cnt; ++i) cig_[i] = le32(aux + 5 + 4 * i);
                    r->n_cigar = cnt; r->cigar = cig_.data();
                    return 1;
                }
                sz = 5 + es * cnt;
            } else break;
            aux += sz;
        }
    }
    if (((uintptr_t)cg & 3) == 0) r->cigar = (const uint32_t *)cg;     // x86: little-endian, aligned
    else {
        cig_.resize(n_cigar);
        memcpy(cig_.data(), cg, 4 * (size_t)n_cigar);
        r->cigar = cig_.data();
    }
    r->n_cigar = n_cigar;
    return 1;
}

bool BaiIndex::load(const std::string &path, std::string *err)
{
    // the file mapped (a 50x human-sized .bai is 100 MB: no copy, no zeroed pages), read into memory where it cannot be
    raw_.reset(); raw_n_ = 0;
    {
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) { if (err) *err = "cannot open " + path; return false; }
        struct stat sb;
        if (fstat(fd, &sb) == 0 && sb.st_size > 0) {
            void *m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
            if (m != MAP_FAILED) { const size_t n = (size_t)sb.st_size; raw_ = std::shared_ptr<const uint8_t>((const uint8_t *)m, [n](const uint8_t *q) { munmap((void *)q, n); }); raw_n_ = n; }
        }
        if (!raw_) {
            std::vector<uint8_t> tmp; uint8_t buf[1 << 16]; ssize_t n;
            while ((n = ::read(fd, buf, sizeof buf)) > 0) tmp.insert(tmp.end(), buf, buf + n);
            uint8_t *q = new uint8_t[tmp.size() + 1];
            memcpy(q, tmp.data(), tmp.size());
            raw_ = std::shared_ptr<const uint8_t>(q, [](const uint8_t *x) { delete[] x; }); raw_n_ = tmp.size();
        }
        ::close(fd);
    }
    struct View { const uint8_t *p; size_t n; const uint8_t *data() const { return p; } size_t size() const { return n; } } d{raw_.get(), raw_n_};
    size_t o = 0;
    auto need = [&](size_t k) { return o + k <= d.size(); };
    if (!need(8) || memcmp(d.data(), "BAI\1", 4) != 0) { if (err) *err = path + " is not a BAI index"; raw_.reset(); return false; }
    const uint32_t n_ref = le32(d.data() + 4); o = 8;
    if ((size_t)n_ref > (d.size() - 8) / 8) { if (err) *err = path + ": damaged BAI index (reference count)"; raw_.reset(); return false; }   // 8 bytes per reference at least
    linear.assign(n_ref, {}); ref_beg.assign(n_ref, 0); ref_end.assign(n_ref, 0); bins.assign(n_ref, {}); bin_at_.assign(n_ref, UINT64_MAX);
    for (uint32_t r = 0; r < n_ref; ++r) {
        if (!need(4)) goto bad;
        {
            bin_at_[r] = o;
            const uint32_t n_bin = le32(d.data() + o); o += 4;
            uint64_t lo = UINT64_MAX, hi = 0;
            for (uint32_t b = 0; b < n_bin; ++b) {
                if (!need(8)) goto bad;
                const uint32_t bin = le32(d.data() + o), n_chunk = le32(d.data() + o + 4); o += 8;
                if (!need(16 * (size_t)n_chunk)) goto bad;
                if (bin != 37450)
                    for (uint32_t c = 0; c < n_chunk; ++c) {
                        const uint64_t cb = le64(d.data() + o + 16 * c), ce = le64(d.data() + o + 16 * c + 8);
                        lo = std::min(lo, cb); hi = std::max(hi, ce);
                    }
                o += 16 * (size_t)n_chunk;
            }
            if (hi) { ref_beg[r] = lo; ref_end[r] = hi; }
            if (!need(4)) goto bad;
            const uint32_t n_intv = le32(d.data() + o); o += 4;
            if (!need(8 * (size_t)n_intv)) goto bad;
            linear[r].resize(n_intv);
            for (uint32_t i = 0; i < n_intv; ++i) linear[r][i] = le64(d.data() + o + 8 * i);
            o += 8 * (size_t)n_intv;
        }
    }
    return true;
bad:
    if (err) *err = path + ": truncated BAI index";
    raw_.reset(); bin_at_.clear();
    return false;
}

// a reference's bins out of the .bai's bytes (load has been through them: every size is inside the file); not for concurrent first use
void BaiIndex::ensure_bins(size_t r) const
{
    if (r >= bin_at_.size() || bin_at_[r] == UINT64_MAX) return;
    struct View { const uint8_t *p; const uint8_t *data() const { return p; } } d{raw_.get()};
    size_t o = (size_t)bin_at_[r];
    bin_at_[r] = UINT64_MAX;
    const uint32_t n_bin = le32(d.data() + o); o += 4;
    bins[r].reserve(n_bin);
    for (uint32_t b = 0; b < n_bin; ++b) {
        const uint32_t bin = le32(d.data() + o), n_chunk = le32(d.data() + o + 4); o += 8;
        if (bin != 37450) {
            auto &v = bins[r][bin];
            v.reserve(n_chunk);
            for (uint32_t c = 0; c < n_chunk; ++c) v.emplace_back(le64(d.data() + o + 16 * c), le64(d.data() + o + 16 * c + 8));
        }
        o += 16 * (size_t)n_chunk;
    }
}

static inline uint32_t reg2bin(int64_t beg, int64_t end)
{
    --end;
    if (beg >> 14 == end >> 14) return (uint32_t)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (uint32_t)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (uint32_t)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (uint32_t)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (uint32_t)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

bool bai_build(const std::string &bam_path, std::string *err)
{
    AlnReader rd;
    if (!rd.open(bam_path, err)) return false;
    if (!rd.is_bam()) { if (err) *err = bam_path + " is not a BAM file"; return false; }
    const size_t n_ref = rd.header().names.size();
    struct Chunk { uint64_t b, e; };
    struct Ref {
        std::map<uint32_t, std::vector<Chunk>> bins;
        std::vector<uint64_t> lin;
        uint64_t off_beg = 0, off_end = 0, n_mapped = 0, n_unmapped = 0;
        bool any = false;
    };
    std::vector<Ref> refs(n_ref);
    uint64_t n_no_coor = 0;
    AlnRec r;
    int k;
    int32_t last_tid = 0, last_pos = -1;
    for (;;) {
        const uint64_t v0 = rd.tell();
        k = rd.next(&r);
        if (k <= 0) break;
        const uint64_t v1 = rd.tell();
        if (r.tid < 0) { ++n_no_coor; continue; }
        if ((size_t)r.tid >= n_ref) { if (err) *err = "record names a reference outside the header"; return false; }
        if (r.tid < last_tid || (r.tid == last_tid && r.pos < last_pos)) { if (err) *err = bam_path + " is not coordinate sorted"; return false; }
        last_tid = r.tid; last_pos = r.pos;
        Ref &R = refs[r.tid];
        const int64_t beg = r.pos < 0 ? 0 : r.pos;
        int64_t end = r.endpos(); if (end <= beg) end = beg + 1;
        const uint32_t bin = reg2bin(beg, end);
        auto &ch = R.bins[bin];
        if (!ch.empty() && ch.back().e == v0) ch.back().e = v1; else ch.push_back({v0, v1});
        const size_t w0 = (size_t)(beg >> 14), w1 = (size_t)((end - 1) >> 14);
        if (R.lin.size() <= w1) R.lin.resize(w1 + 1, 0);
        for (size_t w = w0; w <= w1; ++w) if (R.lin[w] == 0) R.lin[w] = v0;
        if (!R.any) { R.any = true; R.off_beg = v0; }
        R.off_end = v1;
        if (r.flag & 4) ++R.n_unmapped; else ++R.n_mapped;
    }
    if (k < 0) { if (err) *err = rd.error(); return false; }
    std::vector<uint8_t> out;
    auto p32 = [&](uint32_t v) { for (int i = 0; i < 4; ++i) out.push_back((uint8_t)(v >> (8 * i))); };
    auto p64 = [&](uint64_t v) { for (int i = 0; i < 8; ++i) out.push_back((uint8_t)(v >> (8 * i))); };
    out.insert(out.end(), {'B', 'A', 'I', 1});
    p32((uint32_t)n_ref);
    for (Ref &R : refs) {
        p32((uint32_t)(R.bins.size() + (R.any ? 1 : 0)));
        for (auto &b : R.bins) {
            p32(b.first); p32((uint32_t)b.second.size());
            for (auto &c : b.second) { p64(c.b); p64(c.e); }
        }
        if (R.any) { p32(37450); p32(2); p64(R.off_beg); p64(R.off_end); p64(R.n_mapped); p64(R.n_unmapped); }
        for (size_t i = 1; i < R.lin.size(); ++i) if (R.lin[i] == 0) R.lin[i] = R.lin[i - 1];
        p32((uint32_t)R.lin.size());
        for (uint64_t v : R.lin) p64(v);
    }
    p64(n_no_coor);
    FILE *f = fopen((bam_path + ".bai").c_str(), "wb");
    if (!f) { if (err) *err = "cannot write " + bam_path + ".bai"; return false; }
    const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
    fclose(f);
    if (!ok && err) *err = "short write on " + bam_path + ".bai";
    return ok;
}

bool BaiIndex::load_csi(const std::string &path, std::string *err)
{
    gzFile f = gzopen(path.c_str(), "rb");
    if (!f) { if (err) *err = "cannot open " + path; return false; }
    std::vector<uint8_t> d;
    uint8_t buf[1 << 16];
    int n;
    while ((n = gzread(f, buf, sizeof buf)) > 0) d.insert(d.end(), buf, buf + n);
    gzclose(f);
    size_t o = 0;
    auto need = [&](size_t k) { return o + k <= d.size(); };
    if (!need(16) || memcmp(d.data(), "CSI\1", 4) != 0) { if (err) *err = path + " is not a CSI index"; return false; }
    min_shift = (int)le32(d.data() + 4); depth = (int)le32(d.data() + 8);
    const uint32_t l_aux = le32(d.data() + 12);
    o = 16 + l_aux;
    if (min_shift < 1 || min_shift > 30 || depth < 1 || depth > 9 || !need(4)) { if (err) *err = path + ": unsupported CSI parameters"; return false; }
    const uint32_t n_ref = le32(d.data() + o); o += 4;
    if ((size_t)n_ref > (d.size() - o) / 4 + 1) { if (err) *err = path + ": damaged CSI index (reference count)"; return false; }  // 4 bytes per reference at least
    linear.assign(n_ref, {}); ref_beg.assign(n_ref, 0); ref_end.assign(n_ref, 0); bins.assign(n_ref, {}); loffset.assign(n_ref, {});
    const uint32_t meta_bin = (uint32_t)(((1ull << (3 * (depth + 1))) - 1) / 7 + 1);
    for (uint32_t r = 0; r < n_ref; ++r) {
        if (!need(4)) goto bad;
        {
            const uint32_t n_bin = le32(d.data() + o); o += 4;
            uint64_t lo = UINT64_MAX, hi = 0;
            for (uint32_t b = 0; b < n_bin; ++b) {
                if (!need(16)) goto bad;
                const uint32_t bin = le32(d.data() + o);
                const uint64_t lof = le64(d.data() + o + 4);
                const uint32_t n_chunk = le32(d.data() + o + 12); o += 16;
                if (!need(16 * (size_t)n_chunk)) goto bad;
                if (bin != meta_bin) {
                    auto &v = bins[r][bin];
                    loffset[r][bin] = lof;
                    for (uint32_t c = 0; c < n_chunk; ++c) {
                        const uint64_t cb = le64(d.data() + o + 16 * c), ce = le64(d.data() + o + 16 * c + 8);
                        lo = std::min(lo, cb); hi = std::max(hi, ce);
                        v.emplace_back(cb, ce);
                    }
                }
                o += 16 * (size_t)n_chunk;
            }
            if (hi) { ref_beg[r] = lo; ref_end[r] = hi; }
        }
    }
    return true;
bad:
    if (err) *err = path + ": truncated CSI index";
    return false;
}

bool BaiIndex::load_for(const std::string &bam_path, std::string *err)
{
    if (file_exists(bam_path + ".bai") && load(bam_path + ".bai", err)) return true;
    if (file_exists(bam_path + ".csi") && load_csi(bam_path + ".csi", err)) return true;
    return false;
}

void BaiIndex::query(int32_t tid, int64_t beg0, int64_t end, std::vector<Chunk> *out) const
{
    if (tid < 0 || (size_t)tid >= bins.size() || beg0 >= end) return;
    if (beg0 < 0) beg0 = 0;
    const int64_t maxpos = (int64_t)1 << (min_shift + 3 * depth);
    if (end > maxpos) end = maxpos;
    if (beg0 >= end) return;
    ensure_bins((size_t)tid);
    const auto &bm = bins[tid];
    const int64_t e = end - 1;
    // lower bound on the file offset of anything overlapping [beg0, ...): BAI's linear index, or
    // the loffset of the smallest existing CSI bin that contains beg0
    uint64_t min_off = 0;
    if (!linear.empty() && !linear[tid].empty()) {
        const auto &lin = linear[tid];
        const size_t w = (size_t)(beg0 >> 14);
        min_off = w < lin.size() ? lin[w] : lin.back();
    } else if (!loffset.empty()) {
        for (int l = depth; l >= 0; --l) {
            const uint32_t bin = (uint32_t)(((1ull << (3 * l)) - 1) / 7 + (uint64_t)(beg0 >> (min_shift + 3 * (depth - l))));
            auto it = loffset[tid].find(bin);
            if (it != loffset[tid].end()) { min_off = it->second; break; }
        }
    }
    for (int l = 0; l <= depth; ++l) {
        const int sh = min_shift + 3 * (depth - l);
        const uint64_t base = ((1ull << (3 * l)) - 1) / 7;
        for (int64_t k = beg0 >> sh; k <= e >> sh; ++k) {
            auto it = bm.find((uint32_t)(base + (uint64_t)k));
            if (it == bm.end()) continue;
            for (const Chunk &c : it->second) if (c.second > min_off) out->push_back(c);
        }
    }
}

void BaiIndex::normalise(std::vector<Chunk> *v)
{
    std::sort(v->begin(), v->end());
    size_t w = 0;
    for (size_t i = 0; i < v->size(); ++i) {
        if (w && (*v)[i].first <= (*v)[w - 1].second) { if ((*v)[i].second > (*v)[w - 1].second) (*v)[w - 1].second = (*v)[i].second; }
        else (*v)[w++] = (*v)[i];
    }
    v->resize(w);
}

std::vector<uint64_t> BaiIndex::record_starts() const
{
    std::vector<uint64_t> cand;
    for (size_t r = 0; r < linear.size(); ++r) {
        if (ref_beg[r]) cand.push_back(ref_beg[r]);
        for (uint64_t v : linear[r]) if (v) cand.push_back(v);
        if (linear[r].empty() && r < bins.size()) ensure_bins(r);
        if (linear[r].empty() && r < bins.size())              // CSI: chunk begins are record starts too
            for (auto &b : bins[r]) for (const Chunk &c : b.second) cand.push_back(c.first);
    }
    std::sort(cand.begin(), cand.end());
    cand.erase(std::unique(cand.begin(), cand.end()), cand.end());
    return cand;
}

std::vector<uint64_t> BaiIndex::split(uint64_t first, uint64_t fsize, int n_parts) const
{
    const std::vector<uint64_t> cand = record_starts();
    std::vector<uint64_t> out;
    out.push_back(first);
    if (n_parts < 1) n_parts = 1;
    const uint64_t c0 = first >> 16;
    const uint64_t span = fsize > c0 ? fsize - c0 : 0;
    uint64_t next_target = 1;
    for (uint64_t v : cand) {
        if (v <= out.back()) continue;
        const uint64_t c = v >> 16;
        if (c < c0) continue;
        // place a boundary at the first candidate past each 1/n_parts of the compressed span
        if ((c - c0) * (uint64_t)n_parts >= next_target * span) {
            out.push_back(v);
            next_target = (c - c0) * (uint64_t)n_parts / (span ? span : 1) + 1;
        }
    }
    out.push_back(UINT64_MAX);
    return out;
}

} // namespace pdh
