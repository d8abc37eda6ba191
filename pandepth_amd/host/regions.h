// regions.h — the region model: GFF / GTF / BED targets or synthetic bins, grouped per contig and
// per id, plus the merged spans that decide which reads the reference would have fetched.
// Semantics follow PD:3547-4051 (1-based inclusive coordinates everywhere; BED is NOT converted
// from 0-based; ids are keys of an ordered map; overlapping entries of one id double count).
#ifndef PD_REGIONS_H_
#define PD_REGIONS_H_
#include <stdint.h>
#include <map>
#include <string>
#include <utility>
#include <vector>
#include "bam.h"
#include "options.h"
#include "fasta.h"

namespace pdh {

struct Gene {
    int32_t start = 0, end = 0;                       // min first / max second over the entries
    uint64_t length = 0;                              // sum of (second - first + 1)
    std::vector<std::pair<int32_t, int32_t>> cds;     // entries in file order
    int32_t cover = 0;                                // filled by the pipeline
    uint64_t depth = 0;
    int32_t gc = 0;                                   // -c: G/C bases of the entry that CREATED the id (PD:3605-3611: later
                                                      // entries of the same id add to length but never to GeneGCGC)
};

struct Bin { int32_t start, end; int32_t cover = 0; uint64_t depth = 0; int32_t gc = 0; };   // one synthetic bin (1-based inclusive)

struct RegionModel {
    std::map<int32_t, std::map<std::string, Gene>> genes;           // tid -> id -> Gene (GFF/GTF/BED targets)
    // whole-contig bins of modes 0/5/6, in position order.  The reference keeps them in the same
    // id-keyed map as genes (PD:4009) but only ever reports them by start (PD:5101-5116), which
    // for bins is this order; a vector avoids millions of map nodes for -w 1000 on a 3 Gb genome.
    std::map<int32_t, std::vector<Bin>> bins;
    std::map<int32_t, std::vector<std::pair<int32_t, int32_t>>> merged;   // tid -> sorted disjoint spans
    bool has(int32_t tid) const { return merged.find(tid) != merged.end(); }
};

// Parses o->region_file according to o->mode (1..4); on return, if no region survived, builds the
// synthetic bins and sets o->mode to 0 / 5 / 6 (PD:3974-4051).  With `ref` (-c -r) the sequences are loaded first
// (their names join the contig-name table, see fasta.h) and every gene / bin gets its G/C count.
// Returns false on an unreadable reference file.  `names` (PAF input): a ready name table; `ref` is then already loaded.
bool build_regions(Options *o, const AlnHeader &hdr, RegionModel *rm, RefSeqs *ref = nullptr, int threads = 1,
                   const std::map<std::string, int32_t> *names = nullptr);

} // namespace pdh
#endif
