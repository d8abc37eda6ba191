// bam.cpp — see bam.h
#include "bam.h"
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <algorithm>

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
            if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rlen += (int32_t)(cigar[i] >> 4);
        }
    } else {
        rlen = 1;
    }
    if (rlen == 0) rlen = 1;
    return pos + rlen;
}

bool AlnReader::open(const std::string &path, std::string *err)
{
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
    hdr_.text.resize(l_text);
    if (l_text && !bg_.read_exact(&hdr_.text[0], l_text)) { err_ = "truncated BAM header"; return false; }
    const std::string::size_type z = hdr_.text.find('\0');
    if (z != std::string::npos) hdr_.text.resize(z);
    if (!bg_.read_exact(b, 4)) { err_ = "truncated BAM header"; return false; }
    const uint32_t n_ref = le32(b);
    for (uint32_t i = 0; i < n_ref; ++i) {
        if (!bg_.read_exact(b, 4)) { err_ = "truncated BAM header"; return false; }
        const uint32_t l_name = le32(b);
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
    if (n_cigar == 2 && (le32(cg) & 0xf) == 4 && (le32(cg) >> 4) == l_seq && (le32(cg + 4) & 0xf) == 3) {
        const uint8_t *aux = cg + 8 + (l_seq + 1) / 2 + l_seq, *end = rec + bs;
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
                if (t0 == 'C' && t1 == 'G' && st == 'I' && aux + 5 + 4 * (size_t)cnt <= end) {
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
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) { if (err) *err = "cannot open " + path; return false; }
    std::vector<uint8_t> d;
    uint8_t buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + n);
    fclose(f);
    size_t o = 0;
    auto need = [&](size_t k) { return o + k <= d.size(); };
    if (!need(8) || memcmp(d.data(), "BAI\1", 4) != 0) { if (err) *err = path + " is not a BAI index"; return false; }
    const uint32_t n_ref = le32(d.data() + 4); o = 8;
    linear.assign(n_ref, {}); ref_beg.assign(n_ref, 0); ref_end.assign(n_ref, 0);
    for (uint32_t r = 0; r < n_ref; ++r) {
        if (!need(4)) goto bad;
        {
            const uint32_t n_bin = le32(d.data() + o); o += 4;
            uint64_t lo = UINT64_MAX, hi = 0;
            for (uint32_t b = 0; b < n_bin; ++b) {
                if (!need(8)) goto bad;
                const uint32_t bin = le32(d.data() + o), n_chunk = le32(d.data() + o + 4); o += 8;
                if (!need(16 * (size_t)n_chunk)) goto bad;
                if (bin != 37450) {
                    for (uint32_t c = 0; c < n_chunk; ++c) {
                        const uint64_t cb = le64(d.data() + o + 16 * c), ce = le64(d.data() + o + 16 * c + 8);
                        lo = std::min(lo, cb); hi = std::max(hi, ce);
                    }
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
    return false;
}

std::vector<uint64_t> BaiIndex::split(uint64_t first, uint64_t fsize, int n_parts) const
{
    std::vector<uint64_t> cand;
    for (size_t r = 0; r < linear.size(); ++r) {
        if (ref_beg[r]) cand.push_back(ref_beg[r]);
        for (uint64_t v : linear[r]) if (v) cand.push_back(v);
    }
    std::sort(cand.begin(), cand.end());
    cand.erase(std::unique(cand.begin(), cand.end()), cand.end());
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
