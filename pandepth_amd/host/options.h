// options.h — the command-line surface of `pandepth` (-i -o -g -f -b -w -a -q -d -x -t -s -c -r -h);
// -i takes SAM / BAM / CRAM 2.1 / 3.0 / 3.1 / PAF files or a #.list of them.
// Behaviour follows the reference's parser (PD:84-293) including its quirks: every '-' is
// stripped from a flag, `*.list`/`*.List` expands to one input per non-empty line, -w < 1 and
// -d < 1 clamp to 1, -o gets ".gz" appended and a trailing ".stat"/".bed" dropped later.
#ifndef PD_OPTIONS_H_
#define PD_OPTIONS_H_
#include <stdint.h>
#include <string>
#include <vector>

namespace pdh {

struct Options {
    std::vector<std::string> inputs;     // InStr1List
    std::string input;                   // InStr1
    std::string region_file;             // InStr2 (gff/gtf or bed)
    std::string out;                     // InStr3 (with ".gz")
    std::string feature = "CDS";
    int mode = 0;                        // InInt2: 0 chr, 1 gff, 2 gtf, 3 bed3, 4 bed4, 5 window >= 150, 6 window < 150
    int min_mapq = -1;                   // main() presets -1 (PD:3444): nothing is filtered unless -q is given
    int min_dep = 1;
    uint32_t flag_mask = 1796;
    int threads = 3;
    int decode_readers = 0;                                // device decode: reader threads per context (0: min(threads, 6)); a `#.list` run sets it so that a context's readers do not shrink with -t / #GPUs — they wait for the device most of the time
    std::string reference;            // -r: FASTA behind -c's GC(%) column (host/fasta.h)
    int win = 0;
    bool site_out = false;               // -a
    bool use_index = true;               // hidden -s clears it
    bool gc = false;
};

// Returns the number of input files (0 => nothing to do / message already printed), like
// bamCov_help01.  Messages go to stdout/stderr exactly as the reference prints them.
int parse_options(int argc, char **argv, Options *o);
void print_help();

// Development switches of the executable — which of its equivalent paths runs (device or host decode, where the gzip streams are
// parsed, how many decode threads / GPUs ...), never what it prints — come in through ONE door: the hidden option `-X key=value[,key=value]`
// or the environment variable PANDEPTH_TUNE with the same syntax (tests, benchmarks).  tune("key") returns the value or NULL.
//   device_decode=0  device_deflate=0  site_resident=0  table_resident=0  table_resident_min=N  site_overlap=0  site_identical=0|1
//   site_parallel_min=N  pgz_min=N  rccl=0|force  rccl_verbose=1  gpus=N  dd_threads=N  dd_depth=N  dd_batch_mb=N  inflate_waves=N  lz_group=N  decode_only=1
//   decode_fast=0 (the host confirms every batch's record chain, as before round 5)  decode_max_redo=N  decode_spoil=K (test hook: plants wrong guesses)
//   round 6: transport=peer|rccl (the list mode's collective: in-process peer copies — the default — or RCCL made ahead of the contexts)
//   comm=0|force (no communicator / one even for a single context; `rccl=0|force` are the same switches under their older names)
//   comm_early=0 (the communicator and its buffers in line before the first collective instead of beside the decode)
//   h2d_kernel=1|2|3 (a batch's bytes fetched by a copy kernel / + its tables / read in place by the inflate kernel: all measured slower)
//   h2d_fifo=0 (every batch's copy on its own stream, as until round 6; default: first come, first served on the main stream, one copy per batch)
//   sync_event=0 (collect waits for the batch's stream instead of its last event)  h2d_lanes=2 (two copies on the link at a time)
//   dd_trace=1 (with PANDEPTH_TIMING: a [trace] line per batch, tools/feeder_trace.py)  dd_inflight=N (at most N batches queued at a time)
//   dd_pin_ahead=N (buffers page-locked before the readers start)  decode_warm=1 (the slots made ready by a helper thread)  lz_calls=4 (provider calls of a gzip round in flight)
//   environment: PANDEPTH_DEVTRACE=1 (a [devtrace] line per batch: host-clock times of its stage events; first batches: what pd_decode_queue's own time went to)
// -X is not part of the reference's command line (which answers an unknown flag with "Error UnKnow argument"): it is accepted only in this
// spelling, is not listed by -h, and a PANDEPTH_* variable of the earlier rounds that is still set gets a note on stderr.
const char *tune(const char *key);
long long tune_int(const char *key, long long dflt);
void tune_add(const std::string &kv_list);

// text-file helper shared with the region parser: whole file (plain or gzip) split into lines
// the way `while(!in.eof()) getline(in, line)` sees them
bool read_lines(const std::string &path, std::vector<std::string> *lines);

} // namespace pdh
#endif
