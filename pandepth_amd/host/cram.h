// cram.h — CRAM 2.1 / 3.0 / 3.1 input reduced to the five fields the depth path reads (refID, pos, mapq, flag and the
// reference-consuming shape of the alignment).  The reference gets CRAM from htslib (PD:3486-3492 sets the decoder to
// FLAG | RNAME | POS | MAPQ | CIGAR); this is an independent reader written from the CRAM 3.0 specification
// (samtools/hts-specs CRAMv3 and CRAMcodecs): file definition, containers, blocks (raw / gzip / rANS 4x8 order 0 and 1 /
// rANS Nx16 with its stripe, pack, run-length and stored transforms and the adaptive arithmetic coder — the codecs CRAM 3.1
// uses for the series read here), the
// compression header (preservation map, data-series and tag encodings: EXTERNAL, HUFFMAN, BYTE_ARRAY_LEN,
// BYTE_ARRAY_STOP, BETA, GAMMA, SUBEXP, NULL), slices (single- and multi-reference, delta-coded positions) and the
// record layout.  A CRAM record stores no CIGAR: it is rebuilt from the read features (§10.6 of the specification),
// which needs no reference sequence — only positions matter here, so `-r` is not required for decoding.
// Not read: CRAM 1.0 / 2.0; bzip2 / lzma blocks (and arithmetic-coder blocks that wrap bzip2): the reference's own htslib build
// has neither library; fqzcomp and name-tokeniser blocks only ever hold qualities and names, which are never inflated.
#ifndef PD_CRAM_H_
#define PD_CRAM_H_
#include <stdint.h>
#include <stdio.h>
#include <deque>
#include <functional>
#include <future>
#include <string>
#include <vector>

namespace pdh {

struct AlnHeader;
struct AlnRec;

class CramReader {
public:
    ~CramReader() { close(); }
    // true when the file starts with the CRAM magic (the caller then commits to this reader)
    static bool is_cram(const std::string &path);
    bool open(const std::string &path, AlnHeader *hdr, std::string *err);
    // next record: 1 record, 0 end of file, -1 error (error() has the text)
    int next(AlnRec *r);
    const std::string &error() const { return err_; }
    // containers are independent: with n > 1 up to n of them are decoded ahead on helper threads (records still come
    // back in file order)
    void set_threads(int n) { threads_ = n < 1 ? 1 : n > 64 ? 64 : n; }
    // Region reads without the .crai: a container header names its contig and the stretch its reads cover (§7), so
    // containers the predicate rejects are stepped over unread.  keep(ref, start0, end): [start0, end) 0-based.
    // Multi-reference containers (ref -2) are always decoded, unmapped ones (ref -1) never once a predicate is set.
    void set_container_filter(std::function<bool(int32_t, int64_t, int64_t)> keep) { keep_ = std::move(keep); }
    uint64_t containers_read() const { return n_read_; }
    uint64_t containers_skipped() const { return n_skipped_; }
    void close();
    struct Rec { int32_t tid, pos; uint16_t flag; uint8_t mapq; uint32_t cig_off, n_cig; };
    struct Batch { std::vector<Rec> recs; std::vector<uint32_t> cigs; std::string err; };      // one container
private:
    bool read_body(std::vector<uint8_t> *body);  // the next container that holds records; false at end of file / error
    bool fail(const std::string &m) { if (err_.empty()) err_ = m; return false; }
    FILE *f_ = nullptr;
    bool eof_ = false, v2_ = false;          // v2_: CRAM 2.1 framing (no CRC32 fields, 32-bit record counters)
    int threads_ = 1;
    uint64_t fsize_ = 0;
    std::function<bool(int32_t, int64_t, int64_t)> keep_;
    uint64_t n_read_ = 0, n_skipped_ = 0;
    std::deque<std::future<Batch>> ahead_;
    Batch cur_batch_;
    size_t cur_ = 0;
    std::string err_;
};

} // namespace pdh
#endif
