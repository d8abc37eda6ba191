/* pandepth_amd_dev.h — development, tuning and measurement entry points of libpandepth_amd.so.
 *
 * NOT part of the drop-in boundary (include/pandepth_amd.h): nothing here is needed to replace the reference's depth path,
 * nothing here changes a result or a contract.  The library exports them for this repository's tests, tuning scripts
 * (tools/) and kernel micro-benchmarks. */
#ifndef PANDEPTH_AMD_DEV_H_
#define PANDEPTH_AMD_DEV_H_
#include "pandepth_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Tuning knobs (never change results): "lmax" (owner-tile look-back in cells, also the bucket width of compact samples; longer
 * runs take the overflow path), "sample" / "direct_sample" (sparse-index stride in runs), "grid_tiles" (persistent grid of the
 * tile kernels), "scatter_tile" (4096 | 8192), "accumulate_packed" (pd_accumulate_from's transport, default 1), "direct_un"
 * (which compiled variant of the direct kernels runs), "decode_crc" (0: the device decoder skips the members' CRC-32 — kernel
 * timing only), "decode_near_span" (split of the decoder's later-run stream), "inflate_waves" (one-wave inflate workgroups per CU and
 * launch, 1..23: what the kernel's 7 KB of LDS a wave allow; default 20), "lz_group" (consecutive chunks per workgroup of pd_deflate_parse's LDS parse, 0..16; 0: every chunk reads its
 * text from memory), "lz_slots" (2 | 4 parse calls in flight).  Round 6, the decode pipeline: "decode_h2d_fifo" (default 1: the batches' host-to-device copies
 * first come, first served on the context's main stream, one copy per batch; 0: on the batch's own stream), "decode_h2d_lanes" (1 | 2 copies on the link at a
 * time), "decode_sync_event" (default 1: pd_decode_collect waits for the batch's last event; 0: for its stream), "decode_h2d_kernel" (1..3: a copy kernel /
 * + the tables / the members read in place: all measured slower), "decode_warm" (1: the session's first slots made ready by a helper thread). */
int pd_set_param(pd_ctx *ctx, const char *name, uint64_t value);

/* ---- GPU-side BAM decode, the one-call synchronous form (round 1's entry point, kept on top of
 * the pd_decode_* path below): replaces htslib's BGZF inflate + bam_read1 on the host
 * (the producer side of PD:434) for whole-contig modes.  The caller hands over raw BGZF bytes of
 * record-aligned file ranges ("units", e.g. cut at index offsets); the device inflates every block
 * (one wave per block), walks the records of every unit, filters (flag & flag_mask, mapq <
 * min_mapq, contigs shorter than 2) and scatters the M/=/X runs exactly as pd_push_intervals would.
 *   blob            n_bytes of BGZF data (whole blocks, any order)
 *   blocks[k]       deflate payload [in_off, in_off+in_len) of block k inside blob, and where its
 *                   out_len inflated bytes go inside the batch's inflated buffer (out_off)
 *   units[u]        records START in [start, stop) of the inflated buffer; bytes up to `avail`
 *                   belong to the unit's blocks; first_block / n_blocks index `blocks`.  Units
 *                   must be given in file order (the first-run array is then position sorted).
 *   unit_status[u]  OUT: 0 = counted on the device; 1 = not counted, decode this unit on the host
 *                   (a record runs past `avail`, or a CIGAR lives in the CG tag); 2 = corrupt data
 * Synchronous: returns when the batch has been scattered.  The executable uses pd_decode_* (below). */
typedef struct pd_bgzf_unit { uint64_t start, stop, avail; uint32_t first_block, n_blocks; } pd_bgzf_unit;
int pd_push_bgzf_units(pd_ctx *ctx, const void *blob, size_t n_bytes, const pd_bgzf_block *blocks, uint32_t n_blocks,
                       const pd_bgzf_unit *units, uint32_t n_units, uint64_t inflated_bytes, uint32_t flag_mask,
                       int32_t min_mapq, int32_t *unit_status, uint64_t *n_records);

/* ---- experimental measuring entry for the inflate kernel alone -------------------------------
 * Inflates every block of a BGZF image held in host memory on the GPU (one lane per block) and
 * copies the result back; variant 0 keeps the per-block Huffman tables in LDS, 1 in global
 * memory.  kernel_ms = average kernel time over `reps` launches.  A measuring / validation entry:
 * the production form will keep the inflated records on the device. */
int pd_x_bgzf_inflate(int device, const void *host_bgzf, size_t n_bytes, void *host_out, size_t out_cap,
                      size_t *out_len, int variant, int reps, double *kernel_ms, uint32_t *n_blocks);

/* ---- guarded device allocations (a debugging aid for the library's own kernels) -----------------
 * With PANDEPTH_GUARD=1 in the environment every device buffer the library allocates is surrounded by two 256-byte canaries (the
 * one behind it starting exactly at the buffer's last byte + 1).  pd_guard_check waits for the device, compares every live
 * buffer's canaries and returns the number of buffers found overwritten since the process started (0 when the guard is off); the
 * text of the last finding — size and allocating source line of the buffer, how many bytes at which offsets — goes to `msg` (may be
 * NULL) and to stderr.  pd_reset checks too, and every buffer is checked when it is freed.  The GPU test suite runs with the guard
 * on (tests/conftest.py) and fails a test that leaves a finding. */
int pd_guard_check(char *msg, size_t cap);
/* Proves the guard sees what it is there for: allocates a guarded buffer, writes ONE byte just behind it (then just in front of it) and
 * expects the check to report exactly that.  0 = both found; 1 = the guard is off; -1 = the guard is on and missed a write.  The findings it
 * provokes are not counted by pd_guard_check. */
int pd_guard_selftest(void);

#ifdef __cplusplus
}
#endif
#endif /* PANDEPTH_AMD_DEV_H_ */
