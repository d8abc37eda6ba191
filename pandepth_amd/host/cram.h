// cram.h — CRAM 3.0 input reduced to the five fields the depth path reads (refID, pos, mapq, flag and the
// reference-consuming shape of the alignment).  The reference gets CRAM from htslib (PD:3486-3492 sets the decoder to
// FLAG | RNAME | POS | MAPQ | CIGAR); this is an independent reader written from the CRAM 3.0 specification
// (samtools/hts-specs CRAMv3): file definition, containers, blocks (raw / gzip / rANS 4x8 order 0 and 1), the
// compression header (preservation map, data-series and tag encodings: EXTERNAL, HUFFMAN, BYTE_ARRAY_LEN,
// BYTE_ARRAY_STOP, BETA, GAMMA, SUBEXP, NULL), slices (single- and multi-reference, delta-coded positions) and the
// record layout.  A CRAM record stores no CIGAR: it is rebuilt from the read features (§10.6 of the specification),
// which needs no reference sequence — only positions matter here, so `-r` is not required for decoding.
// Not read: CRAM 2.x / 3.1 (their codecs), bzip2 / lzma blocks (the reference's own htslib build has neither).
#ifndef PD_CRAM_H_
#define PD_CRAM_H_
#include <stdint.h>
#include <stdio.h>
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
    void close();
private:
    struct Rec { int32_t tid, pos; uint16_t flag; uint8_t mapq; uint32_t cig_off, n_cig; };
    bool load_container();                       // decodes the next data container into recs_
    bool fail(const std::string &m) { if (err_.empty()) err_ = m; return false; }
    FILE *f_ = nullptr;
    bool eof_ = false;
    std::vector<Rec> recs_;
    std::vector<uint32_t> cigs_;
    size_t cur_ = 0;
    std::string err_;
};

} // namespace pdh
#endif
