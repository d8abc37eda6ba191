// bam.h — alignment input: BAM (SAM spec §4.2), text SAM and CRAM 3.0 (cram.h) records reduced to the five fields
// the depth path reads (refID, pos, mapq, flag, CIGAR), plus the BAI index (§5.2) used to cut a
// coordinate-sorted BAM into independent, record-aligned virtual-offset ranges.
// The reference obtains all of this from htslib (sam_hdr_read / sam_read1 / sam_index_load,
// PD:3483-3507, PD:434, PD:378); this is an independent implementation from the specification.
#ifndef PD_BAM_H_
#define PD_BAM_H_
#include <stdint.h>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>
#include "bgzf.h"
#include "cram.h"

namespace pdh {

struct AlnHeader {
    std::string text;                       // @HD/@SQ/... lines
    std::vector<std::string> names;
    std::vector<uint32_t> lens;
    bool sorted_coordinate() const;         // "\tSO:coordinate" test of PD:4537-4549
};

struct AlnRec {
    int32_t tid, pos;                       // pos 0-based
    uint16_t flag;
    uint8_t mapq;
    uint32_t n_cigar;
    const uint32_t *cigar;                  // BAM encoding: len << 4 | op ("MIDNSHP=XB"); valid until next()
    // htslib's bam_endpos: pos + reference length, where unmapped reads and zero-length
    // alignments count as length 1
    int32_t endpos() const;
};

class AlnReader {
public:
    bool open(const std::string &path, std::string *err);
    const AlnHeader &header() const { return hdr_; }
    bool is_bam() const { return is_bam_; }
    bool is_cram() const { return is_cram_; }        // CRAM 3.0 (host/cram.h): sequential reading only
    CramReader &cram() { return cram_; }
    // next record; returns 1 record, 0 end of file, -1 error
    int next(AlnRec *r);
    uint64_t tell() const { return bg_.tell(); }       // BAM only: virtual offset of the next record
    bool seek(uint64_t voff) { return bg_.seek(voff); }
    void set_threads(int n) { if (is_cram_) cram_.set_threads(n); else bg_.set_threads(n); }    // sequential streams: parallel inflate read-ahead
    const std::string &error() const { static const std::string generic = "corrupt BGZF/BAM data"; return err_.empty() ? generic : err_; }
private:
    bool read_bam_header();
    bool read_sam_header();
    int next_sam(AlnRec *r);
    bool getline(std::string *line);
    BgzfReader bg_;
    AlnHeader hdr_;
    bool is_bam_ = false, is_cram_ = false;
    CramReader cram_;
    std::vector<uint8_t> rec_;
    std::vector<uint32_t> cig_;
    std::unordered_map<std::string, int32_t> name2tid_;
    std::string pending_line_;
    bool have_pending_ = false;
    std::string err_;
};

// BAI: only what range partitioning needs.
struct BaiIndex {
    std::vector<std::vector<uint64_t>> linear;    // per reference: 16 kb window -> smallest voffset
    std::vector<uint64_t> ref_beg, ref_end;       // per reference: span of its chunks (0,0 if none)
    typedef std::pair<uint64_t, uint64_t> Chunk;  // [begin, end) virtual offsets
    // per reference.  A .bai's bins are kept as the file's bytes until somebody asks for a reference's bins (query, record_starts of a CSI):
    // the whole-genome modes only use the linear index, and turning the 6.4 million chunks of a 50x human-sized index into hash maps of vectors
    // was 0.15-0.2 s of every run
    mutable std::vector<std::unordered_map<uint32_t, std::vector<Chunk>>> bins;
    std::shared_ptr<const uint8_t> raw_; size_t raw_n_ = 0;   // the .bai's bytes (the file mapped, or read where it cannot be)
    mutable std::vector<uint64_t> bin_at_;        // per reference: where its bins begin in raw_ (UINT64_MAX: made already)
    void ensure_bins(size_t r) const;
    int min_shift = 14, depth = 5;                // BAI's fixed scheme; CSI stores its own
    std::vector<std::unordered_map<uint32_t, uint64_t>> loffset;          // CSI: per-bin lower bound (no linear index)
    bool load(const std::string &path, std::string *err);       // .bai
    bool load_csi(const std::string &path, std::string *err);   // .csi (BGZF-compressed, SAM spec CSIv1)
    // whichever of <bam>.bai / <bam>.csi exists and parses
    bool load_for(const std::string &bam_path, std::string *err);
    // file ranges that can hold reads overlapping [beg0, end) of reference tid (binning scheme of
    // SAM spec §5.3, pruned with the linear index); appended to *out unsorted
    void query(int32_t tid, int64_t beg0, int64_t end, std::vector<Chunk> *out) const;
    // sorts and fuses overlapping / touching ranges
    static void normalise(std::vector<Chunk> *v);
    // record-aligned split points covering [first_record_voff, EOF): about n_parts ranges of similar
    // compressed size.  Returns the boundaries (size = parts + 1, last = UINT64_MAX).
    std::vector<uint64_t> split(uint64_t first_record_voff, uint64_t file_size, int n_parts) const;
    // every virtual offset the index names that is the start of a record (linear-index entries, first chunk of each
    // reference; CSI: chunk begins), sorted
    std::vector<uint64_t> record_starts() const;
};

// Builds <bam_path>.bai for a coordinate-sorted BAM (SAM spec §5.2: binning index + 16 kb linear
// index + the 37450 metadata pseudo-bin).  A companion tool, not part of the reference's CLI: the
// generated fixtures and benchmark inputs need an index and no samtools exists on the box.
bool bai_build(const std::string &bam_path, std::string *err);

bool file_exists(const std::string &p);
uint64_t file_size(const std::string &p);

} // namespace pdh
#endif
