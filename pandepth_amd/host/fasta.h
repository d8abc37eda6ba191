// fasta.h — the reference sequence behind `-c -r ref.fa[.gz]`: the GC(%) column of every table.
// The reference loads the whole file with klib's kseq reader into one string per contig id
// (PD:3506-3529, PD:2067-2090 for lists) and counts C/c/G/g (PD:3536-3538).  Restated here with the
// reader's observable rules: records start at '>' or '@', the name ends at the first white space, sequence
// lines are concatenated (one trailing '\r' per line dropped) until a line starts with '>', '@' or '+';
// '+' starts a FASTQ quality block of the same length.  Two quirks of the caller are kept:
//   * a sequence name the alignment header does not know is looked up with `map::operator[]`, which
//     ENTERS it into the name table as contig 0 — target rows naming it then land on contig 0;
//   * the first sequence that claims a contig id wins.
// Positions a sequence does not cover count as "not G/C" (the reference reads past its string there).
#ifndef PD_FASTA_H_
#define PD_FASTA_H_
#include <stdint.h>
#include <functional>
#include <map>
#include <string>

namespace pdh {

// All sequences live in ONE arena (a 3 Gb genome is one 3 GB allocation, sized from the file for plain FASTA: no
// per-contig regrowth, no copies between buffers, one set of page faults); a contig is an (offset, length) span of it.
struct RefSeqs {
    bool loaded = false;                    // RefIn
    // number of C/c/G/g among the 1-based inclusive positions [first, last] of contig tid
    uint64_t gc(int32_t tid, int64_t first, int64_t last) const;
    // the reader appends a record's bases to arena(); claim() then binds [off, off+n) to tid, or — when tid already has a
    // sequence (the first claim wins) — drops them again and returns false
    std::string *arena() { return &arena_; }
    bool claim(int32_t tid, size_t off, size_t n);
    size_t n_seqs() const { return span_.size(); }
    void clear() { std::string().swap(arena_); span_.clear(); }
private:
    std::string arena_;
    std::map<int32_t, std::pair<size_t, size_t>> span_;
};

// every record of a FASTA/FASTQ file (plain or gzip) in file order: the bases are APPENDED to *dst (reserved from the file
// size for plain files) and rec(name, offset, length) is told where; rec may shrink *dst back to `offset`.
// false when the file cannot be opened.
bool read_fasta_records(const std::string &path, std::string *dst, const std::function<void(const std::string &, size_t, size_t)> &rec);

// false when the file cannot be opened (the reference never returns from that: its reader spins on a
// NULL gzFile); chr2tid gains the unknown names (-> 0)
bool load_reference(const std::string &path, std::map<std::string, int32_t> *chr2tid, RefSeqs *out);

} // namespace pdh
#endif
