// pd_kernels.h — structs and launch wrappers shared by pd_kernels.hip and pd_capi.hip.
#ifndef PD_KERNELS_H_
#define PD_KERNELS_H_
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/pandepth_amd.h"

#define PD_TILE 8192          /* cells per tile: contig slots, scatter windows and sweep tiles */

namespace pdk {

struct ContigTab {            // device pointers
    const uint64_t *off;      // first cell of each contig slot
    const uint32_t *len;
    int32_t n;
};

struct TileMap {              // device pointers, for the window reduction
    const uint32_t *tile_contig;   // contig of each tile
    const uint64_t *contig_off;
    const uint32_t *contig_len;
    const uint64_t *win_off;       // first output window of each contig
};

#define PD_CNT_SLOTS 256       /* hashed counters: same-address device atomics serialise at ~12 ns */
struct BatchDesc {            // written by k_index, consumed by the tile kernels
    uint64_t handled[PD_CNT_SLOTS];   // runs that found the owner tile of their begin
    uint64_t has[PD_CNT_SLOTS];       // runs with cells ...
    uint64_t ends[PD_CNT_SLOTS];      // ... and ends that found their owner (tile or overflow list)
    uint32_t t_first, n_active;
    uint32_t ovf_count;
    uint32_t err;             // 1 invalid tid in a sample, 2 samples out of order, 4 overflow list full
};

struct CheckWords {           // context-wide, read back at pd_scan / pd_synchronize
    uint64_t unsorted_batches;
    uint32_t err;
    uint32_t ovf_count;       // entries on the overflow list of the current tile pass
    uint32_t all_valid;       // every 4096-cell half-tile has been written since the last reset
    uint32_t pad;
};

#define PD_MAXPEND 4          /* sorted batches merged into one owner-tile pass */
#define PD_HALF 4096          /* granularity of the "written since reset" flags */

// A whole sample in the engine's COMPACT form (pd_runs_create; what the GPU decoder leaves for the whole-contig modes): 8 bytes per run —
// the low 32 bits of its FLAT begin (cell index in the context's buffer; the begin is already clamped to [0, len] of its contig) and its
// clamped length (0: a run without cells) — in TWO streams, each grouped by bucket of (8192 >> bshift) cells of the flat cell space:
//   r8[0 ..) / b1        the file's sorted stream (every read's first run) in file order, exactly as the decoder's emit kernel wrote it;
//                        bucket k's runs are r8[b1[k] .. b1[k + 1])
//   r8[o_base ..) / o1   the other runs (later runs of reads with deletions / skips), counting-sorted by bucket:
//                        r8[o_base + o1[k] .. o_base + o1[k + 1]), any order inside
// (both streams in ONE array, so that a kernel walking a tile's candidates of both selects a 32-bit index, not a pointer)
// Contig slots start on tile boundaries, so a tile's own runs are those of its 1 << bshift buckets and all belong to the tile's contig
// (which is why 32 bits of the begin are enough: a consumer only ever needs a begin relative to the tile it is working on); no run is
// longer than a bucket, so the only other runs that can reach into tile t are those of the bucket right before it.
struct Run8 { uint32_t b; uint32_t len; };
struct C8Sample { const Run8 *r8; const uint32_t *b1, *o1; uint32_t o_base, bshift; };

struct PendBatch {            // one sorted batch of a tile pass (device pointers)
    const pd_iv *iv;
    const uint32_t *ub_a, *cand_lo;
    BatchDesc *desc;
    uint32_t n, pad;
};
struct PendSet { PendBatch b[PD_MAXPEND]; int nb; uint32_t lmax; };

struct Piece { uint64_t start; uint32_t count; uint32_t region; };
struct TilePart { uint32_t c0, c1; unsigned long long s0, s1; };   // a tile's share of windows k0, k0+1

void launch_fill(hipStream_t st, void *p, size_t bytes);
void launch_scatter_atomic(hipStream_t st, const pd_iv *iv, size_t n, ContigTab tab, int *diff, int *sums);
void launch_scatter_index(hipStream_t st, const pd_iv *iv, uint32_t n, ContigTab tab, uint32_t lmax,
                          uint32_t disorder, uint32_t sample, uint32_t *ub_a, uint32_t *cand_lo,
                          uint32_t n_stiles, int stile, BatchDesc *desc);
void launch_scatter_tiles(hipStream_t st, const PendSet &ps, ContigTab tab, const uint32_t *tile_contig,
                          uint32_t n_stiles, int stile, int *diff, int *sums, uint8_t *hstate,
                          uint64_t *ovf, uint32_t ovf_cap, CheckWords *chk, unsigned grid_tiles);
void launch_direct_tiles(hipStream_t st, const PendSet &ps, ContigTab tab, const uint32_t *tile_contig, uint32_t n_tiles,
                         uint32_t wrap_mask, uint32_t w, uint32_t min_dep, TilePart *part, const uint64_t *win_off,
                         uint32_t *cover, unsigned long long *sum, uint32_t *n_long, uint32_t *fail,
                         uint32_t *heavy_list, uint32_t *heavy_count, unsigned grid_tiles, int un);
// compact samples: the passes that make one (words: [0] not sorted / invalid contig, [1] runs longer than a bucket), the reverse
// (12-byte runs, bucket by bucket), and the direct kernels that read them
void launch_c8_from_sorted(hipStream_t st, const pd_iv *iv, uint32_t n, ContigTab tab, uint32_t bshift, Run8 *out, uint32_t *b1 /* pre-set to 0xFF */, uint32_t *words);
void launch_c8_hist(hipStream_t st, const pd_iv *iv, uint32_t n, ContigTab tab, uint32_t bshift, uint32_t *hist, uint32_t *words);
void launch_excl_scan_u32(hipStream_t st, const uint32_t *in, uint32_t *out, uint32_t n, uint32_t *block_sums /* n / 1024 + 2 words */);
void launch_c8_fill_starts(hipStream_t st, uint32_t *b1, uint32_t n_buckets, uint32_t n_runs, uint32_t *tmp /* n_buckets / 1024 + 2 words */);
void launch_c8_marks_to_index(hipStream_t st, const unsigned long long *marks, uint32_t n_buckets, const uint32_t *base, uint32_t *b1);
void launch_c8_place_other(hipStream_t st, const pd_iv *iv, uint32_t n, ContigTab tab, uint32_t bshift, const uint32_t *o1, uint32_t *cursor, Run8 *out);
void launch_c8_expand(hipStream_t st, C8Sample cs, const uint32_t *tile_contig, const uint64_t *contig_off, uint32_t n_tiles, pd_iv *out);
void launch_r8_to_iv(hipStream_t st, const Run8 *r8, uint64_t n, ContigTab tab, pd_iv *out);
void launch_copy_words(hipStream_t st, void *dst, const void *src, uint64_t n_words);      // device-to-device, 4-byte words (both pointers 4-byte aligned)
void launch_direct_c8(hipStream_t st, C8Sample cs, ContigTab tab, const uint32_t *tile_contig, uint32_t n_tiles, uint32_t wrap_mask, uint32_t w,
                      uint32_t min_dep, TilePart *part, uint32_t *heavy_list, uint32_t *heavy_count, unsigned grid_tiles, int un);
void launch_direct_c8_export(hipStream_t st, C8Sample cs, ContigTab tab, const uint32_t *tile_contig, uint32_t n_tiles, void *img, pd_exc *exc,
                             uint32_t cap, uint32_t *count, int *sums, uint32_t *heavy_list, uint32_t *heavy_count, unsigned grid_tiles);
void launch_direct_export(hipStream_t st, const PendSet &ps, ContigTab tab, const uint32_t *tile_contig, uint32_t n_tiles,
                          void *img, pd_exc *exc, uint32_t cap, uint32_t *count, int *sums, uint32_t *n_long, uint32_t *fail,
                          uint32_t *heavy_list, uint32_t *heavy_count, unsigned grid_tiles);
void launch_fill_invalid(hipStream_t st, int *diff, uint8_t *hstate, uint32_t n_half, CheckWords *chk,
                         bool only_if_overflow, unsigned grid);
void launch_mark_all_valid(hipStream_t st, CheckWords *chk);
void launch_scatter_finish(hipStream_t st, const PendSet &ps, int *diff, int *sums, const uint64_t *ovf,
                           uint32_t ovf_cap, CheckWords *chk);
void launch_tile_carry(hipStream_t st, const int *sums, int *bsum, int *carry, uint32_t n_tiles);
void launch_scan_write(hipStream_t st, int *buf, const int *carry, uint32_t n_tiles, uint32_t wrap_mask,
                       const uint8_t *hstate);
int launch_sweep_windows(hipStream_t st, int *buf, const int *carry, uint32_t n_tiles, uint32_t wrap_mask,
                         TileMap tm, uint32_t w, uint32_t min_dep, uint32_t *cover, unsigned long long *sum,
                         TilePart *part, uint64_t n_windows, int32_t n_contigs, bool from_depth,
                         const uint8_t *hstate);
void launch_add_i32(hipStream_t st, int *dst, const int *src, size_t n_words);
void launch_export_i8(hipStream_t st, const int *diff, const uint8_t *hstate, void *out, uint64_t n_cells, int thr,
                      pd_exc *exc, uint32_t cap, uint32_t *count);
void launch_import_i8(hipStream_t st, const void *in, int *diff, uint64_t n_cells, int bias, const pd_exc *exc,
                      uint64_t n_exc);
void launch_export_i4(hipStream_t st, const int *diff, const uint8_t *hstate, void *out, uint64_t n_cells,
                      pd_exc *exc, uint32_t cap, uint32_t *count);
void launch_add_i4(hipStream_t st, int *dst, const void *img, uint32_t n_tiles, const pd_exc *exc, uint64_t n_exc,
                   uint64_t n_cells_total, int *dst_base);
void set_sweep_i4_fast(bool on);               // process-wide A/B switch of the packed statistics kernel of the sliced sum's sweep
void launch_sweep_i4(hipStream_t st, const void *parts, uint32_t n_parts, uint64_t stride, uint32_t tile_first,
                     uint32_t tile_count, const pd_exc *exc, uint64_t exc_stride, const int32_t *exc_counts, uint8_t *flags, size_t flags_bytes /* bytes of flags; behind them: 16 + 4 * tiles bytes for the list */,
                     const int *carry, uint32_t wrap_mask, TileMap tm, uint32_t w, uint32_t min_dep, TilePart *part,
                     int *depth_out /* non-null: no statistics, the slice's summed depth as int32 cells instead */);
int launch_sweep_windows_slice(hipStream_t st, int *depth_slice, uint32_t tile_first, uint32_t tile_count, TileMap tm, uint32_t w, uint32_t min_dep,
                               uint32_t *cover, unsigned long long *sum, TilePart *part);
void launch_window_edges(hipStream_t st, const TilePart *part, TileMap tm, uint32_t n_tiles, uint32_t w, uint32_t *cover, unsigned long long *sum);
void launch_window_gather(hipStream_t st, const TilePart *part, TileMap tm, int32_t n_contigs, uint32_t w,
                          uint64_t n_windows, uint32_t *cover, unsigned long long *sum);
void launch_reduce_pieces(hipStream_t st, const int *depth, const Piece *pieces, uint32_t n_pieces,
                          uint32_t min_dep, int *cover, unsigned long long *sum);

// GPU-side BAM decode (pd_bgzf.hip)
void launch_bgzf_inflate(hipStream_t st, const uint8_t *comp, const pd_bgzf_block *blk, uint32_t n_blk, uint8_t *out,
                         int *status, void *scratch);
size_t bgzf_scratch_bytes(uint32_t n_blk);
void launch_bgzf_inflate_wave(hipStream_t st, const uint8_t *comp, const pd_bgzf_block *blk, uint32_t n_blk, uint8_t *out,
                              int *status, void *scratch, unsigned n_wg, bool check_crc, uint32_t *next /* device word for the member counter, or null: static split */,
                              bool zero_next /* false: the caller has zeroed the counter on this stream already */);
size_t bgzf_wave_scratch_bytes(unsigned n_wg);
} // namespace pdk

// zlib's level-6 LZ77 parse on the device (pd_deflate.hip, pd_lz77.h)
namespace pdk {
void launch_lz_sort(hipStream_t st, const uint8_t *text, uint32_t np, uint64_t *keys_a, uint64_t *keys_b, uint32_t *hist, uint32_t *scan_tmp,
                    uint32_t *S, uint32_t *R, uint32_t *bucket);
// The parse takes consecutive chunks in GROUPS whose text — the first chunk's history up to the last chunk's end plus LZ_LDS_SLACK — fits the
// LDS of a CU (one wave per chunk, at most LZ_GROUP_MAX); the chunks of no group are listed and parsed with the text in memory.
struct LzGroup { uint32_t first, count; uint64_t base, len; };     // chunks [first, first + count); text [base, base + len) goes to LDS
enum : uint32_t { LZ_LDS_MAX = 160u * 1024u, LZ_LDS_SLACK = 512u, LZ_GROUP_MAX = 16u };
bool lz_parse_lds_ready(size_t lds_bytes);                         // the device grants a workgroup that much LDS
void launch_lz_parse(hipStream_t st, const uint8_t *text, uint64_t n_text, const uint32_t *S, const uint32_t *R, const uint32_t *bucket,
                     const uint64_t *chunks, const LzGroup *groups, uint32_t n_groups, uint32_t group_waves, size_t lds_bytes,
                     const uint32_t *list, uint32_t n_list, uint32_t *syms, uint64_t stride, uint32_t *counts);
void launch_lz_gather(hipStream_t st, const uint32_t *syms, uint64_t stride, const uint64_t *off, uint32_t n_chunks, uint32_t *out);
// CRC-32 (zlib's) of text[start, min(start + span, end)) of every chunk (start, end, origin triples)
void launch_lz_crc(hipStream_t st, const uint8_t *text, const uint64_t *chunks, uint32_t n_chunks, uint64_t span, uint32_t *crc);
}

// per-site text rows (pd_format.hip)
namespace pdk {
uint32_t site_rows_blocks(uint64_t n);
uint32_t window_rows_blocks(uint64_t n);
// rows of the `-w` table for windows [row_first, row_first + n_rows) of one contig; cover / sum point at the first of them
void launch_window_rows(hipStream_t st, const uint32_t *cover, const unsigned long long *sum, uint64_t row_first, uint64_t n_rows, uint32_t w, uint32_t clen,
                        uint32_t name_len, const char *dev_name, uint32_t *blk_bytes, uint64_t *blk_off, char *text, bool write);
void launch_site_rows(hipStream_t st, const uint32_t *depth, uint32_t first_index, uint64_t n, uint32_t name_len, const char *dev_name,
                      uint32_t *blk_bytes, uint64_t *blk_off, char *text, bool write);
}

// the record walk of the device decode path (pd_bamwalk.h)
namespace pdb2 { struct Cfg; struct Seg; struct LaneOut; struct ChainOut; }
namespace pdk {
void launch_walk_segments(hipStream_t st, const pdb2::Cfg &cfg, pdb2::Seg *segs, uint32_t n_seg, pdb2::LaneOut *lanes,
                          const uint32_t *only, uint32_t n_only);
void launch_emit_segments(hipStream_t st, const pdb2::Cfg &cfg, const pdb2::Seg *segs, uint32_t n_seg, const pdb2::LaneOut *lanes,
                          pd_iv *first, pd_iv *other, pd_iv *far, const pdb2::ChainOut *gate /* k_chain_segments' verdict, or null */);
void launch_spoil_segments(hipStream_t st, const pdb2::Cfg &cfg, pdb2::Seg *segs, uint32_t n_seg, pdb2::LaneOut *lanes, uint32_t k /* test hook: plants wrong guesses */);
void launch_chain_segments(hipStream_t st, const pdb2::Cfg &cfg, pdb2::Seg *segs, uint32_t n_seg, pdb2::LaneOut *lanes, const int *member_status,
                           uint32_t n_members, uint64_t cap_first, uint64_t cap_other, uint32_t max_redo, pdb2::ChainOut *out);
void launch_runs_sorted(hipStream_t st, const pd_iv *runs, uint64_t n, uint32_t *out);
}
#endif
