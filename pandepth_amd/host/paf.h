// paf.h — PAF input (`-i x.paf[.gz]` or a #.list of them): the reference's second front end to the same depth
// arrays and tables (paf_main, PD:852-2024).  What differs from the BAM path:
//   * targets come from `-r` (FASTA records in file order; with -r the GC(%) column is always on) or else from
//     columns 6/7 of the FIRST file, in order of first appearance (PD:866-943);
//   * a record covers target cells [tstart-1, tend) — or, when it carries a cg:Z: tag, the M/=/X runs of that CIGAR
//     walked from tstart (PD:1560-1612); start/end are swapped when reversed; column 12 is tested against -q, and
//     with bit 0x100 of -x set (the default) lines containing "tp:A:S" are skipped (PD:1549-1555);
//   * a target name the table does not know is looked up with map::operator[] and so lands on target 0;
//   * cells are the 18-bit SiteInfo type, statistics are StatChrDepthLowMEM for every target with regions.
// The product of this file is the target table and the run stream; depth, statistics and tables are the shared path.
#ifndef PD_PAF_H_
#define PD_PAF_H_
#include <stdint.h>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include "bam.h"
#include "fasta.h"
#include "options.h"

namespace pdh {

// PD:3466-3474: extension "paf"/"PAF", looked up below a trailing ".gz"
bool is_paf_path(const std::string &path);

// Builds the target table.  Returns false when the -r file cannot be opened.
bool paf_targets(const Options &o, AlnHeader *hdr, std::map<std::string, int32_t> *chr2tid, RefSeqs *ref);

struct RunEmitter { virtual void emit(int32_t tid, int32_t beg, int32_t end) = 0; virtual ~RunEmitter() {} };

// Streams one PAF file (plain or gzip): the reader cuts it into blocks of whole lines, up to `threads` parser threads
// (each with an emitter of its own from make_emitter) turn them into runs; n_records counts the lines that were walked.
// Lines with fewer than 12 columns, or a cg:Z: value that does not parse, are skipped (the reference indexes past its
// vector / dies in std::stoi there).
bool read_paf(const std::string &path, const Options &o, const std::map<std::string, int32_t> &chr2tid,
              const std::function<std::unique_ptr<RunEmitter>()> &make_emitter, int threads, uint64_t *n_records);

} // namespace pdh
#endif
