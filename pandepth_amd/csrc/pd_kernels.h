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

struct BatchDesc {            // written by k_index, consumed by the tile kernels
    uint64_t handled;         // runs that found the owner tile of their begin
    uint64_t has, ends;       // runs with cells / ends that found their owner (tile or overflow list)
    uint32_t t_first, n_active;
    uint32_t ovf_count;
    uint32_t err;             // 1 invalid tid in a sample, 2 samples out of order, 4 overflow list full
};

struct CheckWords {           // context-wide, read back at pd_scan / pd_synchronize
    uint64_t unsorted_batches;
    uint32_t err;
    uint32_t pad;
};

struct Piece { uint64_t start; uint32_t count; uint32_t region; };
struct TilePart { uint32_t c0, c1; unsigned long long s0, s1; };   // a tile's share of windows k0, k0+1

void launch_fill(hipStream_t st, void *p, size_t bytes);
void launch_scatter_atomic(hipStream_t st, const pd_iv *iv, size_t n, ContigTab tab, int *diff, int *sums);
void launch_scatter_index(hipStream_t st, const pd_iv *iv, uint32_t n, ContigTab tab, uint32_t lmax,
                          uint32_t disorder, uint32_t sample, uint32_t *ub_a, uint32_t *cand_lo,
                          uint32_t n_stiles, int stile, BatchDesc *desc);
void launch_scatter_tiles(hipStream_t st, const pd_iv *iv, uint32_t n, ContigTab tab, uint32_t lmax,
                          const uint32_t *ub_a, const uint32_t *cand_lo, const uint32_t *tile_contig,
                          uint32_t n_stiles, int stile, BatchDesc *desc,
                          int *diff, int *sums, uint64_t *ovf, uint32_t ovf_cap, unsigned grid_tiles);
void launch_scatter_finish(hipStream_t st, uint32_t n, BatchDesc *desc, int *diff, int *sums,
                           const uint64_t *ovf, uint32_t ovf_cap, CheckWords *chk);
void launch_tile_carry(hipStream_t st, const int *sums, int *carry, uint32_t n_tiles);
void launch_scan_write(hipStream_t st, int *buf, const int *carry, uint32_t n_tiles, uint32_t wrap_mask);
int launch_sweep_windows(hipStream_t st, int *buf, const int *carry, uint32_t n_tiles, uint32_t wrap_mask,
                         TileMap tm, uint32_t w, uint32_t min_dep, uint32_t *cover, unsigned long long *sum,
                         TilePart *part, uint64_t n_windows, int32_t n_contigs, bool from_depth);
void launch_reduce_pieces(hipStream_t st, const int *depth, const Piece *pieces, uint32_t n_pieces,
                          uint32_t min_dep, int *cover, unsigned long long *sum);

} // namespace pdk
#endif
