/* pandepth_amd.h — C-ABI of the MI355X-native per-base depth engine.
 *
 * The reference (HuiyangYu/PanDepth, src/PanDepth.cpp = "PD") has no plugin / FFI interface:
 * its depth engine is a set of C++ functions operating on host arrays inside one translation
 * unit.  This header cuts the boundary exactly along that seam.  Each entry point names the
 * reference code it replaces; a maintainer's call-site patch is shown in INTEGRATION.md.
 *
 *   producer (stays on the host)     : the htslib record loop + CIGAR walk, PD:434-460
 *   consumer (this library, on HBM)  : depth storage            PD:4129-4145, PD:715-721, PD:2687-2699
 *                                      per-base increment       PD:449-452 (and its 6 copies)
 *                                      interval statistics      PD:295-327, PD:329-348
 *                                      small-window sweep       PD:4366-4389
 *                                      per-site read-back       PD:4278-4281
 *                                      multi-BAM accumulation   PD:2704-3014
 *
 * Conventions: plain C, caller owns every host buffer, the context owns all device memory and
 * one HIP stream; every function returns 0 on success or a negative PD_E* code, and
 * pd_strerror(ctx) gives the text of the last failure.  One context per GPU.
 * pd_push_intervals* may be called from several host threads (serialised inside); everything
 * else is single-caller.  All arithmetic is integer and bit-exact with the reference.
 *
 * The library REFUSES to run without a gfx950 device: pd_create fails with PD_ENODEV.  There
 * is no CPU fallback inside it.
 */
#ifndef PANDEPTH_AMD_H_
#define PANDEPTH_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PD_ABI_VERSION 3

enum {
    PD_OK = 0,
    PD_EINVAL = -1,   /* bad argument                                  */
    PD_ENODEV = -2,   /* no usable gfx950 device / HIP runtime failure  */
    PD_ENOMEM = -3,   /* device or pinned-host allocation failed       */
    PD_ESTATE = -4,   /* call not valid in the context's current state */
    PD_EHIP   = -5,   /* a HIP call failed; see pd_strerror            */
    PD_ERANGE = -6    /* the data does not fit this path's packed form; nothing was lost: take the general path (see pd_sliced_sum_finish) */
};

typedef struct pd_ctx pd_ctx;

/* One M/=/X CIGAR run: cells [beg, end) of contig tid, 0-based — exactly the range the
 * reference's inner `for (; StartRead<endTmp; StartRead++) depth[tid][StartRead]++` visits
 * (PD:449-452).  Runs are clipped to [0, contig_len] on the device (the reference lets reads
 * overhang into its +500 padding, PD:4132; those cells are never reported). */
typedef struct pd_iv { int32_t tid, beg, end; } pd_iv;

/* One CDSList entry: 1-based inclusive (first, second) on contig tid, visited by the reference
 * as cells [first-1, second) (PD:336-337, PD:313-314). */
typedef struct pd_region { int32_t tid, first, second; } pd_region;

/* flags for pd_push_intervals / pd_push_intervals_device / pd_stage_submit */
#define PD_PUSH_DEFAULT 0u   /* any order: two device atomics per run                             */
#define PD_PUSH_SORTED  1u   /* caller promises the batch is sorted by (tid, beg), as the runs of
                              * a coordinate-sorted BAM are when only each read's first run is
                              * taken: owner-tile scatter (LDS accumulate, plain 16-byte RMW flush,
                              * no global atomics).  A broken promise is detected on the device
                              * and reported by pd_scan / pd_scan_reduce_windows (PD_EINVAL). */
/* Nearly sorted batches (the 2nd..nth runs of reads, which trail their read's start by at most
 * the read's reference span): OR in PD_PUSH_DISORDER(D) where D bounds how far back a run may
 * start relative to any EARLIER run of the batch (beg_j >= beg_i - D for i < j, same contig
 * order).  The owner tiles then widen their candidate search by D; results stay exact. */
#define PD_PUSH_DISORDER(cells) ((((unsigned)(cells) + 255u) / 256u) << 8)
/* With PD_PUSH_SORTED: another sorted batch covering the SAME genome range follows at once (a
 * sample's first-run stream and its second-run stream); the engine defers the owner-tile pass
 * and serves up to 4 such batches in one pass over the tiles.  The batch memory must stay valid
 * until the next call without PD_PUSH_MORE (or pd_scan* / pd_synchronize) returns. */
#define PD_PUSH_MORE    2u

/* Context: replaces `new SiteInfo[len+500]` per contig + zero loop (PD:4129-4145,
 * PD:4553-4581, PD:2687-2699) and `new unsigned int[window]` (PD:715-721).  Allocates ONE
 * int32 buffer in HBM: every contig's difference array, each slot rounded up to an 8192-cell
 * tile, followed by one int32 tile-sum per tile; zero-filled. */
int pd_create(int device, int32_t n_contigs, const uint32_t *contig_len, pd_ctx **out);
int pd_destroy(pd_ctx *ctx);
const char *pd_strerror(const pd_ctx *ctx);      /* ctx may be NULL: last pd_create failure */
int pd_abi_version(void);

/* Forget every cell and return to the accumulating state (a new sample / a new step).  Nothing is
 * filled: cells count as zero until first written (the owner-tile kernel stores into them, the
 * sweep skips them); pd_device_buffer and the atomic path materialise the zeros on demand. */
int pd_reset(pd_ctx *ctx);

/* Replaces the increment loop PD:449-452: +1 at beg, -1 at end in the difference arrays.
 * Host form copies through pinned staging buffers (async H2D overlapped with the scatter
 * kernel); device form takes a pointer that is already resident in HBM on ctx's device.
 * Valid only before pd_scan. */
int pd_push_intervals(pd_ctx *ctx, const pd_iv *iv, size_t n, unsigned flags);
int pd_push_intervals_device(pd_ctx *ctx, const pd_iv *dev_iv, size_t n, unsigned flags);

/* A whole sample kept in the engine's COMPACT form (what the GPU decoder leaves for the whole-contig modes): 8 bytes per
 * run — begin inside its contig, already clipped to [0, len] as PD:449-452's cells are, and length — grouped by bucket of
 * 512 cells (the "lmax" look-back bound) with the exact index of every bucket's first run, the contig implied by the run's
 * place.  The direct window path (pd_scan_reduce_windows after pd_keep_deferred, windows of >= 8192 cells; pd_export_i4)
 * then reads a third fewer bytes, tests no contig ids and no bounds, and visits only a tile's own runs and those of the one
 * bucket before it.  pd_runs_create takes the sample as the decoder has it — `dev_sorted`, sorted by (tid, beg) (every read's
 * first run; the order is CHECKED: PD_EINVAL if it does not hold or a contig id is out of range — push such runs the ordinary
 * way), and `dev_other` in any order (the later runs of reads with deletions / skips; may be NULL / 0) — and makes it in a
 * few passes over the runs; both arrays may be freed afterwards.  The object belongs to ctx, stays valid across pd_reset, and
 * is destroyed before ctx.  pd_push_runs defers it like pd_push_intervals_device(.., PD_PUSH_SORTED | PD_PUSH_MORE) (flags:
 * 0 or PD_PUSH_MORE); whatever cannot use the compact form (narrow windows, pd_scan, other batches pushed beside it, a sample
 * with runs longer than a bucket) expands it to 12-byte runs first — same results, never an error. */
typedef struct pd_runs pd_runs;
int pd_runs_create(pd_ctx *ctx, const pd_iv *dev_sorted, size_t n_sorted, const pd_iv *dev_other, size_t n_other, pd_runs **out);
int pd_runs_destroy(pd_runs *runs);
int pd_push_runs(pd_ctx *ctx, const pd_runs *runs, unsigned flags);

/* Zero-copy variant of the host form for reader threads: acquire a pinned staging slot, write
 * up to *capacity runs into it, submit.  A slot is reusable once its scatter has completed;
 * acquire blocks on the oldest one when all are in flight. */
int pd_stage_acquire(pd_ctx *ctx, pd_iv **host_buf, size_t *capacity);
int pd_stage_submit(pd_ctx *ctx, pd_iv *host_buf, size_t n, unsigned flags);

/* A whole sample that was pushed DEFERRED (every batch with PD_PUSH_SORTED | PD_PUSH_MORE, or pd_push_runs, or what pd_decode_end
 * leaves) into a context that holds nothing else may be served without ever materialising the difference arrays: wide-window
 * statistics (pd_scan_reduce_windows, w >= 64) and the multi-GPU image (pd_export_i4) are then computed straight from the runs
 * (the direct kernels), and pd_synchronize leaves the sample deferred instead of scattering it.  Off by default because it
 * changes ONE thing for the caller: with it on, the memory of device batches pushed with PD_PUSH_MORE must stay valid until
 * pd_reset or until a call that materialises the arrays (pd_scan, pd_accumulate_from, narrow windows ...) has returned, not
 * just until the next pd_synchronize.  Results are identical either way, and the direct calls READ the sample: it stays
 * deferred and every other call works on it afterwards.  (Tuning knobs that never change results or contracts: pd_set_param,
 * include/pandepth_amd_dev.h.) */
int pd_keep_deferred(pd_ctx *ctx, int enable);

/* Difference arrays -> per-base depth, in place (the wavefront prefix-sum sweep).
 * wrap_bits = 18 reproduces the `unsigned Depth:18` cell (DataClass.h:85-88) used by the -a,
 * -w<150, no-index and #.list paths; wrap_bits = 0 the `unsigned int` window cell (PD:717). */
int pd_scan(pd_ctx *ctx, unsigned wrap_bits);

/* Replaces StatChrDepthLowMEM / StatChrDepthWin (PD:329-348 / PD:295-327) for a list of
 * CDSList entries: cover[i] = #{cells with depth >= min_dep}, sum[i] = sum of those depths.
 * Requires pd_scan first.  cover is `int` in the reference (DataClass.h:63). */
int pd_reduce_intervals(pd_ctx *ctx, const pd_region *regs, size_t n, uint32_t min_dep,
                        int32_t *cover, uint64_t *sum);

/* Fixed windows of `w` cells from cell 0 of every contig (the last one clipped to the contig
 * end): replaces the synthetic 10 Mb / -w bins + StatChrDepth* (PD:3995-4049 + PD:295-348) and
 * the -w<150 sweep (PD:4366-4389).  Window k of contig t is written at index
 * win_off[t] + k, where win_off comes from pd_window_layout (n_contigs+1 entries, last = total).
 *   pd_scan_reduce_windows : fused — reads the DIFFERENCE arrays once, never writes depth
 *                            (4 B/base of HBM traffic); state stays "accumulating".
 *                            After pd_keep_deferred(ctx, 1), with w >= 64, a context that holds nothing since
 *                            pd_reset and a sample that is entirely DEFERRED (see pd_keep_deferred), the
 *                            difference arrays are never materialised at all: one pass over the runs builds
 *                            each tile's window in LDS, counts its carry-in from the same runs, prefix-sums
 *                            and reduces it (8 or 12 B/run of HBM traffic, results identical).  The call READS
 *                            the sample: it stays deferred and every other call works on it afterwards.  Runs
 *                            longer than the look-back ("lmax") or a batch that is not sorted make the call
 *                            fall back to the materialising path on its own.
 *   pd_reduce_windows      : from the depth arrays left by pd_scan. */
int pd_window_layout(const pd_ctx *ctx, uint32_t w, uint64_t *win_off);
int pd_scan_reduce_windows(pd_ctx *ctx, uint32_t w, uint32_t min_dep, unsigned wrap_bits,
                           uint32_t *cover, uint64_t *sum);
int pd_reduce_windows(pd_ctx *ctx, uint32_t w, uint32_t min_dep, uint32_t *cover, uint64_t *sum);

/* Replaces the per-site read loop PD:4278-4281: copies depth cells [beg, beg+n) of contig tid
 * to the host.  Requires pd_scan first. */
int pd_read_depth(pd_ctx *ctx, int32_t tid, uint32_t beg, size_t n, uint32_t *out);
/* Per-site rows formatted on the device (PD:4278-4281, 4750-4753, 3031-3034): after pd_scan, the text
 * "<name>\t<index>\t<depth>\n" for cells [beg, beg + n) of contig tid (index = the 0-based cell number the reference prints),
 * written to `text` (a host buffer of `cap` bytes; n * (name_len + 23) always suffices), *n_bytes = its length.  What
 * crosses PCIe is the text the gzip stage consumes, not 4-byte cells to be formatted by host threads.  n <= 2^27 per call. */
int pd_format_sites(pd_ctx *ctx, int32_t tid, uint32_t beg, size_t n, const char *name, size_t name_len, char *text, size_t cap, size_t *n_bytes);

/* Stage 1 of the reference's gzip streams on the GPU.  The reference writes its tables and its per-site file through ONE zlib
 * stream (gzstream, PD:4264-4284 / PD:4366-4389): level 6, whose cost is the LZ77 parse (deflate_slow's hash chains and lazy
 * matching).  pd_deflate_parse produces exactly zlib's parse — literal = byte value, match = len << 16 | dist — of every chunk
 * [start, end) of `text` (a host buffer of n_text < 2^32 - 64 bytes), each parsed the way zlib parses a stream primed with
 * deflateSetDictionary(text + origin, start - origin): start - origin <= 32768 bytes of history, the parse beginning afresh at
 * start.  The symbols of chunk k land at syms[sym_off[k] .. sym_off[k + 1]) (syms: a host buffer of syms_cap entries; sym_off:
 * n_chunks + 1 entries; PD_ERANGE if syms_cap is too small).  Within MIN_LOOKAHEAD (262 bytes) of a chunk's end the parse
 * continues as if more text followed, where zlib would see the end of its input: overlap the chunks and stitch them before that
 * (host/pgzip.cpp: chunks meet where both parses end a match at the same position), and leave a stream's true end to zlib.
 * Blocks, Huffman codes and bits (stage 2) stay with the caller.  One wave per chunk; chunks of 32-256 KiB fill the device. */
typedef struct pd_lz_chunk { uint64_t start, end, origin; } pd_lz_chunk;
int pd_deflate_parse(pd_ctx *ctx, const void *text, size_t n_text, const pd_lz_chunk *chunks, uint32_t n_chunks,
                     uint32_t *syms, size_t syms_cap, uint64_t *sym_off);
/* The per-site file without its text ever leaving HBM (PD:4264-4284 written through gzstream): a text stream that lives on the
 * device.  pd_text_append_sites formats the rows of cells [beg, beg + n) of contig tid (as pd_format_sites does) at the end of the
 * stream; pd_text_parse is pd_deflate_parse on the stretch [off, off + n_text) of the stream (chunk positions relative to `off`), and
 * also returns the CRC-32 of the first crc_span bytes of every chunk, [start, min(start + crc_span, end)) — the chunk's own bytes, without
 * the overlap into its successor (crc: n_chunks entries, may be NULL) — with the symbols and the
 * CRCs the host has all that stage 2 of the gzip stream needs; pd_text_read copies a stretch to the host (the stream's last chunk,
 * which zlib parses itself because it sees the end of the input); pd_text_release tells the stream that nothing before `off` will
 * be asked for again.  The stream is a ring of `capacity` bytes: an append that finds no room returns PD_ERANGE and can be
 * repeated after a release.  Appends and parses may come from different threads.  What crosses PCIe: 4 bytes per symbol, about
 * 0.6 bytes per byte of text, instead of the text twice. */
typedef struct pd_text pd_text;
int pd_text_open(pd_ctx *ctx, size_t capacity, pd_text **out);
int pd_text_close(pd_text *t);
int pd_text_append_sites(pd_text *t, int32_t tid, uint32_t beg, size_t n, const char *name, size_t name_len, uint64_t *n_bytes);
/* The rows of the `-w` table (PD:4381-4388: "<name>\t<start>\t<end>\t<length>\t<covered>\t<depth>\t<coverage>\t<mean>\n", depth being the
 * reference's `int`, the last two columns what printf("%.2f") prints for covered * 100.0 / length and depth * 1.0 / length) for windows
 * [row_first, row_first + n_rows) of contig tid, formatted from the statistics that the last pd_scan_reduce_windows / pd_reduce_windows
 * call with this `w` left on the device (they stay there until the next window call or pd_reset; PD_ESTATE otherwise).  The reference
 * drops a contig's final window when it holds one base only: the caller does not ask for that row.  pd_text_append_bytes puts host
 * bytes (the table's header and total lines) into the stream. */
int pd_text_append_window_rows(pd_text *t, int32_t tid, uint32_t w, uint64_t row_first, size_t n_rows, const char *name, size_t name_len,
                               uint64_t *n_bytes);
int pd_text_append_bytes(pd_text *t, const void *bytes, size_t n);
int pd_text_parse(pd_text *t, uint64_t off, size_t n_text, const pd_lz_chunk *chunks, uint32_t n_chunks,
                  uint32_t *syms, size_t syms_cap, uint64_t *sym_off, uint32_t *crc, uint64_t crc_span);
int pd_text_read(pd_text *t, uint64_t off, size_t n, void *out);
int pd_text_release(pd_text *t, uint64_t off);

/* Optional: page-locks a host buffer the caller owns (text handed to pd_deflate_parse round after round, blobs for the decoder),
 * so that copies from it run at the link's rate (57 GB/s against 20-33 GB/s from pageable memory on the measured host).
 * The caller unregisters before freeing or reallocating the buffer.  Registering costs about 35 us per MB. */
int pd_host_register(pd_ctx *ctx, void *ptr, size_t bytes);
int pd_host_unregister(pd_ctx *ctx, void *ptr);


/* Test / interop access to the accumulating buffer (difference arrays + tile sums are ONE
 * contiguous int32 allocation so that a multi-BAM sum, PD:2704-3014, is a single RCCL
 * all-reduce / reduce-scatter of n_words int32 over xGMI, issued by the caller on the stream
 * returned by pd_stream).  contig_off (n_contigs entries, in cells) may be NULL. */
int pd_device_buffer(pd_ctx *ctx, void **dev_ptr, uint64_t *n_words, uint64_t *contig_off);
int pd_device_count(int *n);                      /* usable gfx950 devices */
/* Single-process form of the same sum (the CLI's `#.list` over several GPUs, one context per GPU):
 * dst += src, difference arrays and tile sums, chunk by chunk through a peer copy over xGMI and an
 * add kernel on dst's GPU.  Both contexts must describe the same contigs and be accumulating.
 * Transport: the source packs its cells to nibbles (d + 8, pd_export_i4's image) + an exception list on
 * its own GPU, so 1/8 of the int32 bytes cross the link; more than 2^20 cells outside [-8, 7], or
 * pd_set_param(dst, "accumulate_packed", 0), selects plain int32 chunks. */
int pd_accumulate_from(pd_ctx *dst, pd_ctx *src);

/* Compact transport of the difference arrays for that sum (xGMI is per-link bound; this moves
 * 1 B/cell instead of 4).  pd_export_i8 writes one byte per cell into dev_i8 (n_cells bytes, from
 * pd_device_layout), BIASED: d + threshold, an unsigned value in [0, 2*threshold]; cells with
 * |d| > threshold are written as 0 (+bias) and appended to dev_exc (at most exc_cap entries;
 * *dev_count receives the number produced — more than exc_cap is an error the caller must check).
 * With threshold = 127 / n_ranks every byte of the sum over the ranks stays below 256, so the
 * images may be added as int32 WORDS by an ordinary int32 sum collective (n_cells / 4 elements; no
 * carry crosses a byte).  pd_import_i8 replaces the context's difference arrays by
 * (byte - bias) + the given exceptions, bias = n_ranks * threshold (the tile sums are NOT touched:
 * reduce them as int32 through pd_device_buffer's tail).  Both run on the context's stream; all
 * pointers are device pointers on the context's GPU. */
typedef struct pd_exc { uint64_t cell; int32_t value; int32_t pad; } pd_exc;
int pd_device_layout(pd_ctx *ctx, uint64_t *n_cells, uint64_t *n_tile_sums);
int pd_export_i8(pd_ctx *ctx, int threshold, void *dev_i8, pd_exc *dev_exc, uint32_t exc_cap, uint32_t *dev_count);
int pd_import_i8(pd_ctx *ctx, const void *dev_i8, int bias, const pd_exc *dev_exc, uint64_t n_exc);

/* Sliced form of the multi-BAM sum (PD:2704-3014) for point-to-point xGMI: instead of funnelling
 * every rank's image into one GPU, every rank keeps ONE contiguous range of tiles ("slice") of the
 * summed arrays.  The images travel once, directly between the pairs (an all-to-all: every link of
 * the GPU carries 1/n_ranks of the image at the same time), as 4-bit cells; nothing is added on
 * the wire, so the value range does not shrink with the number of ranks.
 *   pd_export_i4      one nibble per cell into dev_i4 (n_cells / 2 bytes; cell 2k in the low
 *                     nibble of byte k), BIASED: d + 8 in [0, 15]; cells outside [-8, 7] are
 *                     written as 0 (+bias) and appended to dev_exc as in pd_export_i8.  After
 *                     pd_keep_deferred(ctx, 1), with a pristine context and an entirely deferred sample
 *                     (see pd_scan_reduce_windows) the image, the exceptions and the tile sums
 *                     come straight from the tile windows in LDS — same bytes, the difference
 *                     arrays are never written.  An export READS the sample: it stays deferred (a later
 *                     pd_scan / pd_accumulate_from materialises it as usual; pd_reset forgets it).  An
 *                     export that produces more than exc_cap exceptions always leaves the sample in the
 *                     difference arrays (never "half exported").
 *   pd_slice_sweep_i4 the receiving side, for the tiles [tile_first, tile_first + tile_count) of the
 *                     buffer (8192 cells each), in ONE fused kernel: adds the n_parts images of that
 *                     range (part j at dev_parts + j * part_stride, tile_count * 4096 bytes each,
 *                     16-byte aligned) and the exceptions (block j at dev_exc + j * exc_stride entries,
 *                     dev_exc_counts[j] of them used; entries outside the range are ignored;
 *                     dev_exc_counts may be NULL), prefix-sums them with the carries derived from
 *                     dev_tile_sums (ALL tiles, already summed over the ranks), applies the wrap and
 *                     reduces windows of w >= 8192 cells to per-tile partials: PD_TILE_PARTIAL_BYTES
 *                     per tile at dev_partials, entry 0 = tile_first.  The summed arrays are never
 *                     materialised; none of the context's accumulating state is touched.
 *   pd_gather_windows on the root: adds the partials of all tiles (gathered from the ranks, in tile
 *                     order) to the windows of pd_window_layout(w); same results as
 *                     pd_scan_reduce_windows on a context holding the sum of all samples.
 * All pointers except cover/sum are device pointers on the context's GPU; the kernels run on the
 * context's stream. */
#define PD_TILE_CELLS 8192
#define PD_TILE_PARTIAL_BYTES 24
int pd_export_i4(pd_ctx *ctx, void *dev_i4, pd_exc *dev_exc, uint32_t exc_cap, uint32_t *dev_count);
int pd_slice_sweep_i4(pd_ctx *ctx, const void *dev_parts, uint32_t n_parts, uint64_t part_stride,
                      uint64_t tile_first, uint64_t tile_count, const int32_t *dev_tile_sums,
                      const pd_exc *dev_exc, uint64_t exc_stride, const int32_t *dev_exc_counts,
                      uint32_t w, uint32_t min_dep, unsigned wrap_bits, void *dev_partials);
int pd_gather_windows(pd_ctx *ctx, const void *dev_partials, uint32_t w, uint32_t *cover, uint64_t *sum);

/* ---- the same sliced sum with RCCL called from inside the library (replaces SURVEY §8(b)'s pd_allreduce_diff(ctx, ncclComm_t,
 * root): no GPU ever receives everybody's arrays).  One pd_comm per rank = per context = per GPU; the ranks may be threads of one
 * process (pd_comm_init_all: ncclCommInitAll over the contexts' devices — the CLI's `#.list` mode) or processes
 * (pd_comm_unique_id on one of them, the 128 bytes handed to the others by whatever launcher there is, pd_comm_init on each).
 * pd_sliced_window_sum is COLLECTIVE — every rank calls it, in one process from one thread per rank — and does, on each context's
 * stream: pd_export_i4 -> grouped ncclSend / ncclRecv of the 1/n slices between all pairs (messages of <= 256 MiB) ->
 * ncclAllReduce of the int32 tile sums + ncclAllGather of the exception blocks -> pd_slice_sweep_i4 on the rank's slice ->
 * 24 B per tile to `root` -> pd_gather_windows there (windows of w >= 8192 cells: whole-chromosome mode, PD:2704-3014 + PD:3978; narrower
 * windows: see below). */
typedef struct pd_comm pd_comm;          /* belongs to its context: pd_comm_destroy before pd_destroy */
#define PD_UNIQUE_ID_BYTES 128
int pd_comm_unique_id(void *id128);
int pd_comm_init(pd_ctx *ctx, const void *id128, int rank, int n_ranks, pd_comm **out);
int pd_comm_init_all(pd_ctx **ctxs, int n, pd_comm **comms);
/* The same communicator WITHOUT RCCL, for ranks that are threads of one process (the executable's `#.list` mode, its default since
 * round 6): a rank pulls its slices out of its peers' buffers with xGMI peer copies (hipMemcpyPeerAsync on its own stream, ordered by
 * events), int32 sums with the engine's add kernel; nothing is loaded or bootstrapped (librccl's load costs a short-lived process
 * 1.1-5 s and slows every kernel launch while it runs).  Every pd_sliced_* call works on such a communicator as on an RCCL one.
 * Contexts may share a GPU.  Fails (PD_EHIP) when two of the GPUs have no peer access to each other. */
int pd_comm_init_local(pd_ctx **ctxs, int n, pd_comm **comms);
/* librccl loaded and an n-rank communicator made over `devices` AHEAD of the contexts (ncclCommInitAll needs device numbers only);
 * the next pd_comm_init_all over contexts on exactly these devices adopts it.  Meant to be called at process entry, before
 * pd_create: the load registers its code objects under the lock kernel launches need, so behind a running decode it slows the
 * decode down 3x, ahead of it nothing.  devices == NULL: load the library only.  PD_ENODEV without librccl. */
int pd_comm_preinit(const int *devices, int n);
int pd_comm_destroy(pd_comm *comm);
/* A slot's exchange buffers (twice the 4-bit image: 3 GB per rank for a 3 Gb genome) are allocated by the first pd_sliced_sum_start
 * that uses the slot; pd_comm_prepare makes them NOW — not collective, callable from any thread while the context is being filled
 * (the executable does, beside the decode of the rank's file: the allocation costs tenths of a second). */
int pd_comm_prepare(pd_comm *comm, int slot);
const char *pd_comm_strerror(const pd_comm *comm);
int pd_sliced_window_sum(pd_comm *comm, uint32_t w, uint32_t min_dep, unsigned wrap_bits, int root, uint32_t *cover, uint64_t *sum);
/* Windows narrower than 8192 cells and annotation intervals need the summed CELLS, not per-tile partials: for those every rank turns its
 * slice of the exchanged images into int32 depth cells (1/n of the genome per GPU — still no GPU holds everybody's arrays) and reduces
 * the windows / the stretches of the regions that lie in its slice.  pd_sliced_window_sum does that for w < 8192 (the window arrays are
 * merged with an all-reduce — every window has one writer — and the windows across tile edges put together from the tiles' shares);
 * pd_sliced_interval_sum is pd_reduce_intervals (PD:329-348 over the merged CDS / BED regions, PD:3002-3410 for `#.list` inputs) on
 * the sum of all ranks' samples: per-region partial results are gathered and added on `root`.  Both are COLLECTIVE like
 * pd_sliced_window_sum and return PD_ERANGE on every rank, with every sample intact, when a sample does not fit the 4-bit image. */
int pd_sliced_interval_sum(pd_comm *comm, const pd_region *regs, size_t n, uint32_t min_dep, unsigned wrap_bits, int root,
                           int32_t *cover, uint64_t *sum);
/* The same in two halves, with two slots of exchange buffers (slot 0 or 1), for a caller with several samples per rank:
 * pd_sliced_sum_start only enqueues (pack on the context's stream, the collectives on the communicator's own stream, ordered
 * by events), so the context can be reset and sample k+1 scattered while sample k's image is on the xGMI links;
 * pd_sliced_sum_finish completes a started slot and blocks until this rank's part is done.  All ranks make the same calls in
 * the same order.  pd_sliced_window_sum = start(0) + finish(0).
 * PD_ERANGE (from finish, on EVERY rank alike — the counts are all-reduced): some rank's sample has more cells outside the
 * 4-bit range than the exception block holds (2^18: amplicon or very deep data).  No sample was consumed; sum the contexts
 * with pd_accumulate_from (one process) or reduce pd_device_buffer (several) instead — the reference handles such data
 * (PD:2704-3014), so must the caller. */
int pd_sliced_sum_start(pd_comm *comm, int slot);
int pd_sliced_sum_finish(pd_comm *comm, int slot, uint32_t w, uint32_t min_dep, unsigned wrap_bits, int root, uint32_t *cover, uint64_t *sum);

void *pd_stream(pd_ctx *ctx);                     /* hipStream_t */
/* Waits for everything queued.  Deferred batches (PD_PUSH_MORE) are scattered first — except that with pd_keep_deferred
 * a whole deferred sample in an otherwise empty context stays deferred (its memory must then stay valid until pd_reset or
 * a call that materialises the arrays). */
int pd_synchronize(pd_ctx *ctx);

/* Per-kernel timing with HIP events recorded on the context's stream around every launch.
 * pd_profile_get returns the accumulated milliseconds and launch count of kernel `name`
 * ("reset", "fill", "scatter_index", "scatter_tiles", "scatter_finish", "scatter_atomic", "tile_carry",
 * "scan", "scan_reduce_windows", "reduce_intervals", "reduce_windows", "direct_tiles", "direct_export",
 * "export_i4", "export_i8", "import_i8", "slice_sweep", "gather_windows", "accumulate_from");
 * pd_profile(ctx, 0/1) switches it (and clears the accumulators). */
int pd_profile(pd_ctx *ctx, int enable);
int pd_profile_get(pd_ctx *ctx, const char *name, double *ms, uint64_t *launches);

typedef struct pd_bgzf_block { uint64_t in_off, out_off; uint32_t in_len, out_len; } pd_bgzf_block;   /* one BGZF member of a batch: deflate payload
                                   [in_off, in_off + in_len) inside the batch's bytes, out_len inflated bytes at out_off of its inflated buffer */

/* ---- GPU-side BAM decode, asynchronous batches (the CLI's default input path for BAM files) --------------------
 * Replaces the producer side of the seam as well — htslib's bgzf_read / bam_read1 and the filter + CIGAR walk of
 * PD:434-460 — so that the host only reads compressed bytes.  A BATCH is a set of whole BGZF members plus UNITS:
 * stretches [start, stop) of the batch's inflated bytes whose records are to be counted; a unit's first record
 * starts at `start` (an index offset), or, with PD_UNIT_GUESS, at the first record boundary at or after `start`,
 * which the device finds itself (no-index streams; the caller checks res->first_start against the previous batch's
 * res->next_start).  Per batch, on one of the context's decode streams: H2D of the members (from pinned memory
 * handed out by pd_decode_acquire) -> inflate, one WAVE per member (speculative parallel Huffman decoding, see
 * csrc/pd_inflate_wave.h) -> record boundaries, one wave per 64 KiB (csrc/pd_bamwalk.h) -> filter (flag, mapq,
 * contigs with targets, optionally htslib's region test against `spans`) + CIGAR walk -> the batch's runs, dense and
 * in file order, in HBM.  pd_decode_end concatenates the batches by `order` and leaves the sample DEFERRED exactly
 * as pd_push_intervals_device(.., PD_PUSH_SORTED | PD_PUSH_MORE) would: the statistics calls then take the direct
 * path (pd_keep_deferred) or materialise the arrays.  Units the device does not finish are handed back:
 *   unit_status[u] 0 counted; 1 decode it on the host (a record runs past the unit's bytes, a CIGAR lives in the CG
 *   tag, a Huffman code this decoder leaves to zlib); 2 a member read for the unit does not inflate or fails
 *   its CRC-32; 3 the record chain could not be followed: in both cases decode the unit on the host, which reads only what the
 *   unit needs and reports the corruption if it is one.
 * pd_decode_acquire / submit may be called from several threads (one batch each); submit returns when the batch's
 * runs are in HBM — batches of different threads overlap on the device. */
#define PD_UNIT_GUESS 1u
typedef struct pd_decode_cfg {
    uint32_t flag_mask; int32_t min_mapq;
    const uint8_t *contig_on;       /* n_contigs flags: count reads of this contig (NULL: contigs of >= 2 bases, PD:4000) */
    const uint32_t *span_off;       /* -g / -b: spans of contig t are [span_off[t], span_off[t+1]) ...                   */
    const int32_t *spans;           /* ... pairs (begin0, end), sorted, disjoint (PD:419-434); NULL: every read          */
    int32_t sorted;                 /* the file is coordinate sorted (first runs form a position-sorted stream)          */
    uint64_t bytes_hint;            /* about how many compressed bytes will be submitted (sizes the run arena; 0 = unknown) */
    uint64_t batch_bytes;           /* largest pd_decode_acquire the caller will make and how many batches it will have in  */
    uint32_t batches_in_flight, flags;/* flight (0 = unknown): pd_decode_begin then pins that many buffers up front, from ONE */
                                    /* thread — six threads pinning at once took 75-100 ms each, alone 4-5 ms              */
                                    /* flags: PD_DECODE_COMPACT — the caller will ask for whole-contig statistics                */
                                    /* (pd_scan_reduce_windows, w >= 8192): pd_decode_end leaves a sorted file's runs as ONE        */
                                    /* compact sample (pd_runs) instead of 12-byte arrays                                          */
    uint64_t n_batches;             /* 0 = unknown.  Otherwise the caller promises to submit exactly this many batches, once each, with */
                                    /* pd_decode_batch::order = 0 .. n_batches - 1 in file order (a batch with nothing to decode is still   */
                                    /* submitted, empty; orders need not ARRIVE in order, but a thread must hold its buffer — pd_decode_-   */
                                    /* acquire — BEFORE it takes the next order, so that the lowest outstanding order always has one).      */
                                    /* With PD_DECODE_COMPACT on a sorted file this lets every batch write its runs straight to their      */
                                    /* final places in the compact sample (places are handed out in batch order): pd_decode_end then only  */
                                    /* sorts the later runs of multi-run reads (about a tenth of all runs) by bucket.                       */
} pd_decode_cfg;
#define PD_DECODE_COMPACT 1u
typedef struct pd_decode_unit { uint64_t start, stop, avail; uint32_t first_block, n_blocks, flags, pad; } pd_decode_unit;
typedef struct pd_decode_batch {
    void *host_buf; size_t n_bytes;                              /* from pd_decode_acquire: whole BGZF members           */
    const pd_bgzf_block *blocks; uint32_t n_blocks; uint32_t pad;
    uint64_t inflated_bytes;
    const pd_decode_unit *units; uint32_t n_units; uint32_t pad2;
    uint64_t order;                                              /* file position of the batch (see pd_decode_cfg::n_batches) */
} pd_decode_batch;
typedef struct pd_decode_result {
    uint64_t n_reads, n_first, n_other;    /* records seen in the units that were counted; first runs; other runs       */
    uint64_t first_start, next_start;      /* unit 0: first record it owns / first record start >= its stop (~0: none)   */
    double ms_h2d, ms_inflate, ms_walk, ms_emit;
    /* order of the batch's first runs (key = a value that orders like (tid, begin); file order): unsorted = 1 when a key is smaller than the
     * one before it; first_key / last_key for the check across batches (valid when n_first > 0).  A file whose header says
     * SO:coordinate but whose records are not in that order must not go on as a sorted stream: the caller abandons the
     * device pass (pd_decode_abort) and reads such a file the way the reference does; pd_decode_end, for its part, pushes
     * the first runs as PD_PUSH_DEFAULT when it sees a violation. */
    uint64_t first_key, last_key;
    uint32_t unsorted, pad;
} pd_decode_result;
int pd_decode_begin(pd_ctx *ctx, const pd_decode_cfg *cfg);
int pd_decode_acquire(pd_ctx *ctx, size_t bytes, void **host_buf);
int pd_decode_submit(pd_ctx *ctx, const pd_decode_batch *batch, int32_t *unit_status, pd_decode_result *res);   /* (a batch without units only hands its buffer back) */
/* The same batch in two halves, so that the thread that read it can go and read the next one while the device works (round 5; what
 * PD:726-760 gets from htslib's reader threads).  pd_decode_queue puts EVERYTHING the batch needs on its stream and returns: the
 * copy, the inflate, the record passes — and, where the session allows it (a PD_DECODE_COMPACT session, no PD_UNIT_GUESS unit), the
 * confirmation of the record chain between the two passes on the device as well (csrc/pd_bamwalk.h: chain_device), so that no host
 * round trip is left inside a batch.  `batch`'s arrays are copied by the call; `batch->host_buf` stays the engine's until the batch is
 * collected.  pd_decode_collect(ticket) waits for that batch and reports exactly what pd_decode_submit would have (which IS these two
 * calls back to back); a batch the device found out of the ordinary (a member it leaves to zlib, a record that runs past its
 * unit, more wrong guesses than it repeats itself) is finished there the way pd_decode_submit always did.  Every queued batch must be
 * collected; pd_decode_end / pd_decode_abort collect and drop what the caller left behind.  A thread may hold several queued batches
 * (two per reader keeps a GPU busy); the rule of pd_decode_cfg::n_batches — buffer first, then the order — holds for each of them.
 * Round 6: the batches' host-to-device copies are issued first come, first served on the context's main stream — ONE copy per batch, the
 * batch's tables behind its members in the buffer pd_decode_acquire handed out (which has the room: the call owns the bytes behind
 * `bytes` as well) — and pd_decode_collect waits for an event recorded behind the batch's last command, not for its stream
 * ("the convoy"; pd_set_param "decode_h2d_fifo" / "decode_sync_event" = 0 give the older behaviour). */
int pd_decode_queue(pd_ctx *ctx, const pd_decode_batch *batch, uint64_t *ticket);
int pd_decode_collect(pd_ctx *ctx, uint64_t ticket, int32_t *unit_status /* batch->n_units entries */, pd_decode_result *res);
int pd_decode_end(pd_ctx *ctx);
int pd_decode_abort(pd_ctx *ctx);          /* forget the batches decoded since pd_decode_begin (nothing is counted) */

#ifdef __cplusplus
}
#endif
#endif /* PANDEPTH_AMD_H_ */
