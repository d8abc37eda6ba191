// pd_kernels.hip — hand-written gfx950 kernels of the per-base depth engine.
//
// Everything here is HBM-bound integer work (no MFMA): a zero fill, the +1/-1 difference-array
// scatter (owner-tile LDS accumulation with plain int4 read-modify-write flush for sorted
// batches; device atomics for unsorted ones), the prefix-sum sweep (diff -> depth) with an
// optional fused fixed-window reduction, and a segmented interval reduction.
//
// Data layout in HBM (one allocation, int32 cells):
//   [ contig 0 slot | contig 1 slot | ... | tile sums ]
// Every contig slot is rounded up to PD_TILE cells, so a tile belongs to exactly one contig.
// Slot t holds the difference array d[p] (+1 at run begin, -1 at run end, runs clipped to
// [0, len]); a contig's slot sums to zero, so ONE flat prefix sum over all slots yields every
// contig's depth with no segment resets.  tile_sum[k] = sum of d over tile k is maintained by
// the scatter kernels; an exclusive scan of it gives each tile's carry-in, which makes the
// sweep a single embarrassingly parallel pass (no look-back, no inter-workgroup hand-off).
//
// Reference semantics restated (PD = /root/reference/src/PanDepth.cpp):
//   scatter        == `for (; StartRead<endTmp; StartRead++) depth[..][StartRead]++`  PD:449-452
//   scan (+wrap18) == the value a SiteInfo{unsigned Depth:18} / unsigned int cell holds PD:717, DataClass.h:85
//   reductions     == StatChrDepthLowMEM / StatChrDepthWin / mode-6 sweep              PD:329-348, 295-327, 4366-4389
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pd_kernels.h"

namespace pdk {

static constexpr int WG = 256;
static constexpr int TILE = PD_TILE;            // cells per tile (8192 -> 32 KiB of LDS)
static_assert(TILE % (WG * 4) == 0, "tile must be a multiple of one int4 row per workgroup");

// ------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------
struct Ev { uint64_t gb, ge; bool valid, has; uint32_t len; };

__device__ __forceinline__ Ev expand(const pd_iv v, const ContigTab tab)
{
    Ev e; e.valid = (v.tid >= 0 && v.tid < tab.n); e.has = false; e.gb = e.ge = 0; e.len = 0;
    if (e.valid) {
        const uint32_t len = tab.len[v.tid];
        const uint64_t off = tab.off[v.tid];
        uint32_t b = v.beg < 0 ? 0u : (uint32_t)v.beg; if (b > len) b = len;
        uint32_t x = v.end < 0 ? 0u : (uint32_t)v.end; if (x > len) x = len;
        e.gb = off + b; e.ge = off + x; e.has = b < x; e.len = e.has ? x - b : 0u;
    }
    return e;
}

__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// inclusive prefix sum across the 64 lanes of a wave with DPP row shifts / row broadcasts (pure
// VALU, no LDS crossbar traffic): Hillis-Steele inside each 16-lane row, then row 0->1 and
// 2->3 (row_bcast:15), then lanes 0-31 -> 32-63 (row_bcast:31).
__device__ __forceinline__ int wave_incl_scan(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);   // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);   // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1,3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2,3
    return x;
}

// ------------------------------------------------------------------------------------------
// fill: 16 B/lane stores, grid-stride
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WG) void k_fill(int4 *p, size_t n16)
{
    size_t i = blockIdx.x * (size_t)WG + threadIdx.x;
    const size_t st = (size_t)gridDim.x * WG;
    const int4 z = make_int4(0, 0, 0, 0);
    for (; i < n16; i += st) p[i] = z;
}

// ------------------------------------------------------------------------------------------
// scatter, unsorted fallback: two device atomics per run (+ tile-sum atomics only when the
// run crosses a tile boundary; inside one tile +1 and -1 cancel in the tile sum)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WG) void k_scatter_atomic(const pd_iv *iv, size_t n, ContigTab tab,
                                                       int *diff, int *sums)
{
    size_t i = blockIdx.x * (size_t)WG + threadIdx.x;
    const size_t st = (size_t)gridDim.x * WG;
    for (; i < n; i += st) {
        const Ev e = expand(iv[i], tab);
        if (!e.has) continue;
        atomicAdd(&diff[e.gb], 1);
        atomicAdd(&diff[e.ge], -1);
        const uint64_t tb = e.gb / TILE, te = e.ge / TILE;
        if (tb != te) { atomicAdd(&sums[tb], 1); atomicAdd(&sums[te], -1); }
    }
}

// ------------------------------------------------------------------------------------------
// scatter, sorted batches.  Step 1: sparse index.  One thread per sample (every S-th run):
// because the batch is sorted by flat begin, sample values bracket where each tile's
// candidates start and stop.  For tile t = [a, a+TILE):
//   cand_lo[t] <= first run with gb >= a - LMAX   (look-back: short runs ending in the tile)
//   ub_a[t+1]  >= first run with gb >= a + TILE
// ------------------------------------------------------------------------------------------
template <int ST>
__global__ __launch_bounds__(WG) void k_index(const pd_iv *iv, uint32_t n, uint32_t S, ContigTab tab,
                                              uint32_t lmax, uint32_t dis, uint32_t *ub_a, uint32_t *cand_lo,
                                              uint32_t n_tiles, BatchDesc *desc)
{
    // `dis` = disorder bound D: for runs i < j of the batch, gb_j >= gb_i - D (0 = sorted).  Then
    //   every run at index >= p_k has gb >= s_k - D,   every run at index < p_k has gb <= s_k + D,
    // so all thresholds below just move by D; the samples themselves need not be monotone (any
    // sample pair that brackets a threshold gives a valid, possibly looser, bound).
    const uint32_t K = (n + S - 1) / S + 1;
    const uint32_t k = blockIdx.x * WG + threadIdx.x;
    if (k >= K) return;
    const uint64_t pk64 = (uint64_t)k * S;
    const uint32_t pk = pk64 < n ? (uint32_t)pk64 : n;
    const Ev last = expand(iv[n - 1], tab);
    const int64_t D = dis, T = ST;
    const int64_t sentinel = (int64_t)last.gb + 2 * D + lmax + 1;
    int64_t sk;
    if (pk < n) { const Ev e = expand(iv[pk], tab); if (!e.valid) atomicOr(&desc->err, 1u); sk = (int64_t)e.gb; }
    else sk = sentinel;
    int64_t t_last = (sentinel - D) / T; if (t_last >= (int64_t)n_tiles) t_last = (int64_t)n_tiles - 1;
    const Ev first = expand(iv[0], tab);
    const int64_t s0 = (int64_t)first.gb;
    const int64_t t_first = (s0 - D) > 0 ? (s0 - D) / T : 0;
    const int64_t sk_left = __shfl_up(sk, 1);               // every lane of the wave is still here
    if (k == 0) {
        desc->t_first = (uint32_t)t_first;
        if (t_first > t_last) { desc->n_active = 0; atomicOr(&desc->err, 2u); return; }   // not sorted
        desc->n_active = (uint32_t)(t_last - t_first + 1);
        int64_t t1 = (s0 - D) >= 0 ? (s0 - D) / T : -1; if (t1 > t_last + 1) t1 = t_last + 1;
        for (int64_t t = t_first; t <= t1; ++t) ub_a[t] = 0;
        int64_t t2 = (s0 + lmax + D) / T; if (t2 > t_last) t2 = t_last;
        for (int64_t t = t_first; t <= t2; ++t) cand_lo[t] = 0;
        ub_a[t_last + 1] = n;
        return;
    }
    const uint32_t pkm1 = (k - 1) * S;      // < n for every k >= 1
    int64_t sp;
    {
        // the previous sample is the left neighbour's value: one scattered 12-byte load per thread
        // instead of two (lane 0 of a wave and the thread right after the k == 0 exit load their own)
        const bool own = (threadIdx.x & 63) == 0;
        sp = own ? (int64_t)expand(iv[pkm1], tab).gb : sk_left;
    }
    {   // tiles with a_t + D in (sp, sk]: every run at index >= p_k starts at or after a_t
        int64_t lo = (sp - D) >= 0 ? (sp - D) / T + 1 : 0, hi = (sk - D) >= 0 ? (sk - D) / T : -1;
        if (lo < t_first) lo = t_first;
        if (hi > t_last + 1) hi = t_last + 1;
        for (int64_t t = lo; t <= hi; ++t) ub_a[t] = pk;
    }
    {   // tiles with a_t - lmax - D in (sp, sk]: every run at index < p_{k-1} ends before a_t
        int64_t lo = (sp + lmax + D) / T + 1, hi = (sk + lmax + D) / T;
        if (lo < t_first) lo = t_first;
        if (hi > t_last) hi = t_last;
        for (int64_t t = lo; t <= hi; ++t) cand_lo[t] = pkm1;
    }
}

// Step 2: owner tiles.  A persistent grid walks the tiles the pending batches touch (up to
// PD_MAXPEND sorted batches share ONE pass, e.g. a sample's first-run stream and its nearly sorted
// second-run stream); each workgroup zeroes an ST-cell LDS window, pulls every event that lands in
// ITS tile from each batch's candidate range (LDS atomics), then writes the window to HBM with
// plain 16-byte accesses: a half-tile that has not been written since the last reset is STORED
// (no read, no prior zero fill needed), otherwise read-modify-written with all-zero 16-byte
// groups skipped.  No global atomics on the hot path.  Ends of runs longer than LMAX go to the
// overflow list (applied by k_apply_overflow after this kernel).
template <int ST>
__global__ __launch_bounds__(WG) void k_scatter_tiles(const PendSet ps, uint32_t n_tiles, ContigTab tab,
                                                      const uint32_t *tile_contig, int *diff, int *sums,
                                                      uint8_t *hstate, uint64_t *ovf, uint32_t ovf_cap,
                                                      CheckWords *chk)
{
    __shared__ __attribute__((aligned(16))) int win[ST];
    __shared__ int s_sum;
    uint32_t tf[PD_MAXPEND], na[PD_MAXPEND];
    uint32_t n_beg[PD_MAXPEND], n_has[PD_MAXPEND], n_end[PD_MAXPEND];
    uint32_t t_lo = 0xFFFFFFFFu, t_hi = 0;
#pragma unroll
    for (int b = 0; b < PD_MAXPEND; ++b) {
        n_beg[b] = n_has[b] = n_end[b] = 0; tf[b] = 0; na[b] = 0;
        if (b < ps.nb) {
            tf[b] = ps.b[b].desc->t_first; na[b] = ps.b[b].desc->n_active;
            if (na[b]) { if (tf[b] < t_lo) t_lo = tf[b]; if (tf[b] + na[b] > t_hi) t_hi = tf[b] + na[b]; }
        }
    }
    if (t_hi > n_tiles) t_hi = n_tiles;
    const uint32_t lmax = ps.lmax;
    for (uint64_t t = (uint64_t)t_lo + blockIdx.x; t < t_hi; t += gridDim.x) {
        const uint64_t a = t * ST;
        int4 *w4 = reinterpret_cast<int4 *>(win);
        for (int j = threadIdx.x; j < ST / 4; j += WG) w4[j] = make_int4(0, 0, 0, 0);
        if (threadIdx.x == 0) s_sum = 0;
        // a scatter tile lies inside ONE contig slot, so only runs of that contig can land in it
        const int32_t ctg = (int32_t)tile_contig[a / TILE];
        const uint32_t clen = tab.len[ctg];
        const int64_t rel = (int64_t)(tab.off[ctg] - a);          // slot start relative to the tile (<= 0)
        const bool valid0 = hstate[a / PD_HALF] != 0;
        const bool valid1 = ST > PD_HALF ? hstate[a / PD_HALF + 1] != 0 : valid0;
        __syncthreads();
        int net = 0;
#pragma unroll
        for (int b = 0; b < PD_MAXPEND; ++b) {
            if (b >= ps.nb || t < tf[b] || t >= (uint64_t)tf[b] + na[b]) continue;
            // clamped: on a batch that was NOT sorted the index holds garbage, and the only
            // promise then is "reported, no out-of-bounds access"
            const uint32_t n = ps.b[b].n;
            uint32_t hi = ps.b[b].ub_a[t + 1]; if (hi > n) hi = n;
            uint32_t lo = ps.b[b].cand_lo[t]; if (lo > hi) lo = hi;
            const pd_iv *__restrict__ iv = ps.b[b].iv;
            // four candidates per thread in flight: the loop is bound by load latency, not by ALU
            constexpr int UN = 4;
            for (uint32_t i0 = lo + threadIdx.x; i0 < hi; i0 += UN * WG) {
                pd_iv vv[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const uint32_t i = i0 + u * WG;
                    vv[u] = iv[i < hi ? i : hi - 1];
                    if (i >= hi) vv[u].tid = -1;                  // never equals a contig id
                }
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const pd_iv v = vv[u];
                    if (v.tid != ctg) continue;
                    uint32_t bb = v.beg < 0 ? 0u : (uint32_t)v.beg; if (bb > clen) bb = clen;
                    uint32_t x = v.end < 0 ? 0u : (uint32_t)v.end; if (x > clen) x = clen;
                    const bool has = bb < x;
                    const uint32_t len = has ? x - bb : 0u;
                    const uint64_t rb = (uint64_t)(rel + bb), re = (uint64_t)(rel + x);   // below-tile wraps high
                    if (rb < (uint64_t)ST) {
                        ++n_beg[b];
                        if (has) {
                            ++n_has[b];
                            atomicAdd(&win[rb], 1); ++net;
                            if (len > lmax) {
                                const uint32_t slot = atomicAdd(&chk->ovf_count, 1u);
                                if (slot < ovf_cap) { ovf[slot] = a + re; ++n_end[b]; } else atomicOr(&chk->err, 4u);
                            }
                        }
                    }
                    if (has && len <= lmax && re < (uint64_t)ST) { atomicAdd(&win[re], -1); --net; ++n_end[b]; }
                }
            }
        }
        net = wave_sum(net);
        if ((threadIdx.x & 63) == 0 && net != 0) atomicAdd(&s_sum, net);
        __syncthreads();
        int4 *o4 = reinterpret_cast<int4 *>(diff + a);
        constexpr int NG = ST / 4 / WG;                          // 16-byte groups per thread (4 or 8)
        int4 w[NG], o[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int j = threadIdx.x + g * WG;
            w[g] = w4[j];
            const bool valid = (ST > PD_HALF && j >= PD_HALF / 4) ? valid1 : valid0;
            // read-modify-write only what was written before AND changes now; issue all loads first
            if (valid && (w[g].x | w[g].y | w[g].z | w[g].w)) o[g] = o4[j]; else o[g] = make_int4(0, 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int j = threadIdx.x + g * WG;
            const bool valid = (ST > PD_HALF && j >= PD_HALF / 4) ? valid1 : valid0;
            if (!valid || (w[g].x | w[g].y | w[g].z | w[g].w)) {
                o[g].x += w[g].x; o[g].y += w[g].y; o[g].z += w[g].z; o[g].w += w[g].w;
                o4[j] = o[g];                                    // first touch since reset: plain store of the window
            }
        }
        if (threadIdx.x == 0) {
            if (!valid0) hstate[a / PD_HALF] = 1;
            if (ST > PD_HALF && !valid1) hstate[a / PD_HALF + 1] = 1;
            if (s_sum != 0) {
                if (ST == TILE) sums[t] += s_sum;                // sole owner of this sum
                else atomicAdd(&sums[a / TILE], s_sum);          // two scatter tiles share one sum
            }
        }
        __syncthreads();
    }
    // every run must have found the owner of its begin, and every run with cells the owner of its
    // end (in a tile or on the overflow list); anything else means the batch broke its promise.
    // One atomic per workgroup and counter, spread over PD_CNT_SLOTS addresses.
    __shared__ unsigned s_cnt[PD_MAXPEND][3];
    if (threadIdx.x < PD_MAXPEND * 3) s_cnt[threadIdx.x / 3][threadIdx.x % 3] = 0;
    __syncthreads();
#pragma unroll
    for (int b = 0; b < PD_MAXPEND; ++b) {
        if (b >= ps.nb) continue;
        const int hb = wave_sum((int)n_beg[b]), hh = wave_sum((int)n_has[b]), he = wave_sum((int)n_end[b]);
        if ((threadIdx.x & 63) == 0) {
            if (hb) atomicAdd(&s_cnt[b][0], (unsigned)hb);
            if (hh) atomicAdd(&s_cnt[b][1], (unsigned)hh);
            if (he) atomicAdd(&s_cnt[b][2], (unsigned)he);
        }
    }
    __syncthreads();
    if (threadIdx.x < ps.nb * 3) {
        const int b = threadIdx.x / 3, k = threadIdx.x % 3;
        const unsigned v = s_cnt[b][k];
        if (v) {
            BatchDesc *desc = ps.b[b].desc;
            unsigned long long *dst = (unsigned long long *)(k == 0 ? desc->handled : k == 1 ? desc->has : desc->ends);
            atomicAdd(dst + (blockIdx.x & (PD_CNT_SLOTS - 1)), (unsigned long long)v);
        }
    }
}

// The DIRECT whole-sample path (windows of >= TILE cells, i.e. whole-chromosome mode): when every
// run of the sample is still pending as sorted batches and the context holds nothing else, the
// difference array never needs to exist in HBM.  One workgroup per tile builds the tile's
// difference window in LDS exactly as k_scatter_tiles does, but instead of flushing it:
//   * the carry into the tile — the depth just before its first cell — is counted from the same
//     candidates: runs that begin before the tile and end at or after its first cell (every such
//     run is among the candidates as long as no run is longer than the look-back `lmax`; longer
//     runs are counted in *n_long and the caller falls back to the materialising path);
//   * the window is prefix-summed from LDS (same row layout / wave scans as k_sweep), wrapped, and
//     reduced to the tile's share of the (at most two) windows it touches.
// HBM traffic: the runs, once (+ the look-back overlap), and 24 bytes per tile.  No inter-workgroup
// dependency, no atomics outside LDS.
struct WinArgs {
    uint32_t w;              // window width in cells
    uint32_t min_dep;
    float inv_w;
    uint32_t *cover;         // per window (global index = win_off[contig] + k)
    unsigned long long *sum;
    TilePart *part;          // w >= TILE: per-tile partials for the (at most two) windows it touches
};

// ---- narrow windows (4 <= w < TILE): one row of a wave (64 lanes x 4 consecutive cells) into the tile's LDS accumulators ----
// A window's share of a row is a difference of two row-prefix values, so only the lanes that hold a window's last cell touch LDS:
// the (cover << 48 | depth sum) words of the lanes are prefix-summed over the wave, the lane holding the cell before a window
// start adds the prefix there to the window that ends and subtracts it from the one that begins (the fields borrow from each other
// in between; the finished word is exact), and lane 63 adds the row's total to the window of the row's last cell.  About 2 w / 256 + 1
// LDS atomics per row on distinct addresses, where one atomic per lane and window piece (64-128 per row, on 3-4 addresses) was
// what bound these sweeps.
__device__ __forceinline__ unsigned long long wave_incl_scan_u64(unsigned long long x)
{
#define PD_SCAN64_STEP(ctrl, rmask)                                                                                       \
    {                                                                                                                    \
        const uint32_t lo_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)x, ctrl, rmask, 0xf, false);          \
        const uint32_t hi_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(x >> 32), ctrl, rmask, 0xf, false);  \
        x += ((unsigned long long)hi_ << 32) | lo_;                                                                       \
    }
    PD_SCAN64_STEP(0x111, 0xf) PD_SCAN64_STEP(0x112, 0xf) PD_SCAN64_STEP(0x114, 0xf) PD_SCAN64_STEP(0x118, 0xf)
    PD_SCAN64_STEP(0x142, 0xa) PD_SCAN64_STEP(0x143, 0xc)
#undef PD_SCAN64_STEP
    return x;
}

// (x / w by a multiplication: x = cell + phase < 2^14 and w < 2^13, so floor(x * ceil(2^32 / w) / 2^32) is exact — one instruction per
// row where a float reciprocal with its two corrections took a dozen; these sweeps are bound by their vector instructions, a
// wave64 instruction occupying its SIMD for four cycles, not by HBM)
__device__ __forceinline__ uint32_t narrow_magic(uint32_t w) { return (uint32_t)((0x100000000ull + w - 1) / w); }

__device__ __forceinline__ void narrow_window_row(const uint32_t (&d)[4], uint32_t pos, uint32_t phase, uint32_t w, uint32_t magic, uint32_t min_dep,
                                                  uint32_t cells_left /* cells of the contig from the tile's first, at most TILE */, unsigned long long *acc, int lane)
{
    const uint32_t x = pos + phase;
    const uint32_t q = __umulhi(x, magic);
    const uint32_t k = (q + 1u) * w - x;                          // window q + 1 starts k cells behind this lane's first (k >= 1)
    unsigned long long v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (pos + e < cells_left && d[e] >= min_dep) ? ((1ull << 48) | d[e]) : 0ull;
    const unsigned long long T = v[0] + v[1] + v[2] + v[3];
    const unsigned long long L = wave_incl_scan_u64(T);
    if (k <= 4u && !(lane == 63 && k == 4u)) {
        const unsigned long long I = L - T + v[0] + (k >= 2u ? v[1] : 0ull) + (k >= 3u ? v[2] : 0ull) + (k >= 4u ? v[3] : 0ull);
        if (I) { atomicAdd(&acc[q], I); atomicAdd(&acc[q + 1], 0ull - I); }
    }
    if (lane == 63 && L) atomicAdd(&acc[k <= 3u ? q + 1 : q], L);
}

// The tile's finished accumulators go out with plain stores: windows inside the tile to the window arrays, the share of a window
// that began in the previous tile or goes on in the next one to the tile's TilePart (c0 / s0: first window, c1 / s1: last);
// k_window_edges adds the two halves.  (Global atomics for those two windows kept every workgroup resident for the atomics' round trip
// at its very end: 2.16 ms against 1.49 ms for widths that divide the tile, 2.0e9 cells.)
__device__ __forceinline__ void narrow_write_out(const unsigned long long *acc, uint32_t nacc, uint64_t k0, uint32_t w, uint64_t local0, uint32_t clen,
                                                 uint64_t wbase, const WinArgs &wa, TilePart *pt)
{
    for (uint32_t j = threadIdx.x; j < nacc; j += WG) {
        const unsigned long long a = acc[j];
        const uint32_t c = (uint32_t)(a >> 48);
        const unsigned long long sm = a & 0xFFFFFFFFFFFFull;
        const uint64_t k = k0 + j;
        const bool before = k * w < local0;                                                       // began in the previous tile
        const bool after = (k + 1) * (uint64_t)w > local0 + TILE && local0 + TILE < (uint64_t)clen;   // goes on in the next tile
        if (before) { pt->c0 = c; pt->s0 = sm; }
        else if (after) { pt->c1 = c; pt->s1 = sm; }
        else if (c) { wa.cover[wbase + k] = c; wa.sum[wbase + k] = sm; }
    }
}

// Narrow windows (64 <= w < TILE) for the direct kernels: k_sweep's LDS accumulators (one packed
// 64-bit word per window overlapping the tile: cover in the top 16 bits, depth sum below), fed from
// the registers that hold the tile's local prefix sums.  Called by the whole workgroup.
template <int ROWS>
__device__ __forceinline__ void direct_small_windows(const int4 (&v)[ROWS], const int (&rowex)[ROWS], int base,
                                                     uint32_t wrap_mask, uint32_t local0, uint32_t clen, const WinArgs &wa,
                                                     uint64_t wbase, unsigned long long *acc, TilePart *pt)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t w = wa.w;
    const uint32_t k0 = local0 / w;                              // first window touching the tile
    const uint32_t nacc = (uint32_t)(((uint64_t)local0 + TILE - 1) / w - k0 + 1);
    for (uint32_t j = threadIdx.x; j < nacc; j += WG) acc[j] = 0;
    __syncthreads();
    const uint32_t phase = local0 - k0 * w;                      // offset of the tile inside window k0
    const uint32_t magic = narrow_magic(w);
    const uint32_t left = clen > local0 ? (clen - local0 < (uint32_t)TILE ? clen - local0 : (uint32_t)TILE) : 0u;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int bsum = base + rowex[r];
        const uint32_t pos = (uint32_t)(wv * (ROWS * 256) + r * 256 + lane * 4);
        const uint32_t d[4] = {(uint32_t)(v[r].x + bsum) & wrap_mask, (uint32_t)(v[r].y + bsum) & wrap_mask,
                               (uint32_t)(v[r].z + bsum) & wrap_mask, (uint32_t)(v[r].w + bsum) & wrap_mask};
        narrow_window_row(d, pos, phase, w, magic, wa.min_dep, left, acc, lane);
    }
    __syncthreads();
    narrow_write_out(acc, nacc, k0, w, local0, clen, wbase, wa, pt);
}

// One candidate run of the direct pass; per-lane counters (summed over the wave by the caller).
struct DirectCnt { uint32_t n_beg; int open; int carry; };

__device__ __forceinline__ void direct_candidate(const pd_iv v, int32_t ctg, uint32_t clen, uint32_t p0, unsigned *win,
                                                 DirectCnt &c)
{
    constexpr uint32_t ST = TILE;
    uint32_t bb = v.beg < 0 ? 0u : (uint32_t)v.beg; if (bb > clen) bb = clen;
    uint32_t x = v.end < 0 ? 0u : (uint32_t)v.end; if (x > clen) x = clen;
    // tile-relative begin / end in 32-bit arithmetic: a position before the tile wraps to a huge value
    // (candidates lie within lmax + D cells of the tile, far from 2^32)
    const uint32_t sb = bb - p0, se = x - p0;
    const bool mine = v.tid == ctg, nonempty = bb < x;
    const bool pb = mine && sb < ST;                              // owns the begin (an empty run still has an owner)
    const bool eb = pb && nonempty, ee = mine && nonempty && se < ST;
    if (eb) atomicAdd(&win[sb >> 1], 1u + (sb & 1u) * 0xFFFFu);
    if (ee) atomicSub(&win[se >> 1], 1u + (se & 1u) * 0xFFFFu);
    // selects of values, not increments under the conditions: conditional increments of the three counters were
    // compiled to ONE indexed read-modify-write of a scratch slot (12 bytes of private memory in every direct kernel)
    c.n_beg += pb ? 1u : 0u;
    c.open += (eb ? 1 : 0) - (ee ? 1 : 0);
    c.carry += (mine && nonempty && bb < p0 && x >= p0) ? 1 : 0;  // covers the cell just before the tile
}

struct DirectWide { static constexpr bool narrow = false, exporting = false; uint32_t w, min_dep; TilePart *part; };
struct DirectNarrow { static constexpr bool narrow = true, exporting = false; WinArgs wa; const uint64_t *win_off; };
struct DirectExport {                  // the multi-GPU sum's 4-bit image straight from the tile windows (pd_export_i4)
    static constexpr bool narrow = false, exporting = true;
    unsigned short *img; pd_exc *exc; uint32_t cap; uint32_t *count; int *sums;
};

// A = DirectWide: windows >= TILE, per-tile partials (the bench's instantiation);
// A = DirectNarrow: 64 <= w < TILE, LDS accumulators, results straight into the window arrays;
// A = DirectExport: no statistics — the tile's difference window leaves as nibbles (k_export_i4's image), its cells
//                   outside [-8, 7] as exceptions, its sum as the tile sum: what a rank puts on the links.
template <int UN, int WPE, class A>
__global__ __launch_bounds__(WG, WPE) void k_direct_tiles(const PendSet ps, uint32_t n_tiles, ContigTab tab,
                                                        const uint32_t *tile_contig, uint32_t wrap_mask, const A args,
                                                        uint32_t *n_long, uint32_t *heavy_list, uint32_t *heavy_count)
{
    constexpr bool NARROW = A::narrow, EXPORT = A::exporting;
    uint32_t w = (uint32_t)TILE, min_dep = 1; TilePart *part = nullptr;
    if constexpr (NARROW) { w = args.wa.w; min_dep = args.wa.min_dep; }
    else if constexpr (!EXPORT) { w = args.w; min_dep = args.min_dep; part = args.part; }
    __shared__ unsigned long long acc[NARROW ? TILE / 64 + 2 : 1];
    constexpr int ST = TILE;
    constexpr int ROWS = TILE / (WG * 4);                        // 8
    // two signed 16-bit counters per word, kept as ONE 32-bit sum 65536 * H + L: half the LDS of an
    // int window, twice the resident workgroups.  L = sign-extended low half and H = (word - L) >> 16
    // are exact while both stay within +-32767, which holds when the tile has fewer candidates than
    // that; heavier tiles (pile-ups) go on the heavy list and are done by k_direct_tiles_heavy with
    // an int window.
    __shared__ __attribute__((aligned(16))) unsigned win[ST / 2];
    __shared__ int s_carry;
    __shared__ int wtot[4];
    __shared__ unsigned long long red_s[4][2];
    __shared__ int red_c[4][2];
    __shared__ uint32_t s_lo[PD_MAXPEND], s_hi[PD_MAXPEND];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // summed over the batches: begins that found their owner tile; runs with cells whose begin was owned
    // minus ends that were owned (must cancel: a run that reaches into a tile without being one of its
    // candidates — longer than the look-back — leaves its end unowned, so this also catches long runs)
    uint32_t n_beg = 0; int open = 0;
    for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint64_t a = t * ST;
        if (threadIdx.x < PD_MAXPEND) {
            const int b = threadIdx.x;
            uint32_t lo = 0, hi = 0;
            if (b < ps.nb) {
                const uint32_t tf = ps.b[b].desc->t_first, na = ps.b[b].desc->n_active;
                if (t >= tf && t < (uint64_t)tf + na) {
                    const uint32_t n = ps.b[b].n;
                    hi = ps.b[b].ub_a[t + 1]; if (hi > n) hi = n;
                    lo = ps.b[b].cand_lo[t]; if (lo > hi) lo = hi;
                }
            }
            s_lo[b] = lo; s_hi[b] = hi;
        }
        uint4 *w4 = reinterpret_cast<uint4 *>(win);
        for (int j = threadIdx.x; j < ST / 8; j += WG) w4[j] = make_uint4(0u, 0u, 0u, 0u);
        if (threadIdx.x == 0) s_carry = 0;
        const int32_t ctg = (int32_t)tile_contig[t];
        const uint32_t clen = tab.len[ctg];
        const uint32_t p0 = (uint32_t)(a - tab.off[ctg]);         // tile start inside the contig (slots are < 2^32 cells)
        __syncthreads();
        uint32_t cand = 0;
        for (int b = 0; b < ps.nb; ++b) cand += s_hi[b] - s_lo[b];
        if (cand > 32000u) {                                      // workgroup-uniform
            if (threadIdx.x == 0) heavy_list[atomicAdd(heavy_count, 1u)] = (uint32_t)t;
            __syncthreads();
            continue;
        }
        DirectCnt cnt{0u, 0, 0};
#pragma unroll 1
        for (int b = 0; b < ps.nb; ++b) {
            const uint32_t lo = s_lo[b], hi = s_hi[b];
            if (lo >= hi) continue;
            const pd_iv *__restrict__ iv = ps.b[b].iv;
            uint32_t i = lo;                                      // uniform
            // full chunks: scalar base + constant per-lane offsets, no bounds tests
            for (; i + UN * WG <= hi; i += UN * WG) {
                const pd_iv *__restrict__ p = iv + i;
                pd_iv vv[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) vv[u] = p[threadIdx.x + u * WG];
#pragma unroll
                for (int u = 0; u < UN; ++u) direct_candidate(vv[u], ctg, clen, p0, win, cnt);
            }
            if (i < hi) {                                         // the tail chunk
                const pd_iv *__restrict__ p = iv + i;
                const uint32_t left = hi - i;
                pd_iv vv[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const uint32_t j = threadIdx.x + u * WG;
                    vv[u] = p[j < left ? j : left - 1];
                    if (j >= left) vv[u].tid = -1;                // never equals a contig id
                }
#pragma unroll
                for (int u = 0; u < UN; ++u) direct_candidate(vv[u], ctg, clen, p0, win, cnt);
            }
        }
        n_beg += cnt.n_beg; open += cnt.open;
        const int carry = wave_sum(cnt.carry);
        if (lane == 0 && carry != 0) atomicAdd(&s_carry, carry);
        __syncthreads();
        if constexpr (EXPORT) {
            // k_export_i4's layout: one ushort (4 cells, nibble d + 8) per lane and row, 128 contiguous bytes per wave store
            const uint64_t cw = a + (uint64_t)wv * (ROWS * 256);
            unsigned short *o = args.img + cw / 4 + lane;
            int tsum = 0;
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                const uint2 q = reinterpret_cast<const uint2 *>(win)[wv * (ROWS * 64) + r * 64 + lane];
                const int l0 = (int)(short)(q.x & 0xffffu), l1 = (int)(short)(q.y & 0xffffu);
                int x[4] = {l0, ((int)q.x - l0) >> 16, l1, ((int)q.y - l1) >> 16};
                unsigned wb = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    tsum += x[k];
                    if (x[k] > 7 || x[k] < -8) {
                        const uint32_t slot = atomicAdd(args.count, 1u);
                        if (slot < args.cap) { args.exc[slot].cell = cw + (uint64_t)(r * 256 + lane * 4 + k); args.exc[slot].value = x[k]; args.exc[slot].pad = 0; }
                        x[k] = 0;
                    }
                    wb |= (unsigned)((x[k] + 8) & 0xf) << (4 * k);
                }
                o[r * 64] = (unsigned short)wb;
            }
            tsum = wave_sum(tsum);
            if (lane == 0) wtot[wv] = tsum;
            __syncthreads();
            if (threadIdx.x == 0) args.sums[t] = wtot[0] + wtot[1] + wtot[2] + wtot[3];
            __syncthreads();
            continue;
        }
        // ---- prefix sum of the window, straight from LDS (k_sweep's layout) ----
        int4 v[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const uint2 q = reinterpret_cast<const uint2 *>(win)[wv * (ROWS * 64) + r * 64 + lane];
            const int l0 = (int)(short)(q.x & 0xffffu), l1 = (int)(short)(q.y & 0xffffu);
            v[r] = make_int4(l0, ((int)q.x - l0) >> 16, l1, ((int)q.y - l1) >> 16);
        }
        int tot[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            v[r].y += v[r].x; v[r].z += v[r].y; v[r].w += v[r].z;
            tot[r] = v[r].w;
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) tot[r] = wave_incl_scan(tot[r]);
        int run = 0;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int e = run + tot[r] - v[r].w;                  // exclusive prefix of this lane's group in the wave
            run += __builtin_amdgcn_readlane(tot[r], 63);
            tot[r] = e;
        }
        if (lane == 0) wtot[wv] = run;
        __syncthreads();
        int base = s_carry;
        for (int k = 0; k < wv; ++k) base += wtot[k];
        if constexpr (NARROW) {                                   // narrow windows: LDS accumulators
            if (p0 < clen) direct_small_windows<ROWS>(v, tot, base, wrap_mask, p0, clen, args.wa, args.win_off[ctg], acc, args.wa.part + t);
            __syncthreads();
            continue;
        }
        // ---- the tile's share of windows k0 and k0 + 1 ----
        const uint64_t local0 = p0;
        int c0 = 0, c1 = 0; unsigned long long s0 = 0, s1 = 0;
        if (local0 < clen) {
            const uint64_t k0 = local0 / w;
            const uint64_t nb64 = (k0 + 1) * (uint64_t)w - local0;   // tile-local start of window k0+1
            const uint32_t nb = nb64 > (uint64_t)ST ? (uint32_t)ST : (uint32_t)nb64;
            const uint64_t left = (uint64_t)clen - local0;           // cells of the contig from the tile start on
            const uint32_t lim = left > (uint64_t)ST ? (uint32_t)ST : (uint32_t)left;
            const uint64_t dmax = (uint64_t)(uint32_t)s_carry + cand;     // no depth in this tile exceeds carry + begins
            if (nb >= (uint32_t)ST && lim >= (uint32_t)ST && min_dep <= 1u && dmax < (1u << 27) && dmax <= wrap_mask) {
                // the common tile — inside one window, inside the contig, no wrap possible, threshold <= 1:
                // the depths need not be formed: sum = sum of the local prefixes + 4 x base per row,
                // covered cells = those whose local prefix differs from -base (all of them for threshold 0)
                uint32_t s32 = 0;
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    const int bsum = base + tot[r];
                    s32 += (uint32_t)(v[r].x + v[r].y + v[r].z + v[r].w) + 4u * (uint32_t)bsum;
                    if (min_dep) {
                        const int z = -bsum;
                        c0 += (v[r].x != z ? 1 : 0) + (v[r].y != z ? 1 : 0) + (v[r].z != z ? 1 : 0) + (v[r].w != z ? 1 : 0);
                    } else c0 += 4;
                }
                s0 = s32;
            } else if (nb >= (uint32_t)ST && lim >= (uint32_t)ST && dmax < (1u << 27)) {
                // the common tile: inside one window and inside the contig — no position tests; no depth
                // exceeds carry + begins in the tile < 2^27, so a lane's 32 cells add up in 32 bits
                uint32_t s32 = 0;
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    const int bsum = base + tot[r];
                    const uint32_t d[4] = {(uint32_t)(v[r].x + bsum) & wrap_mask, (uint32_t)(v[r].y + bsum) & wrap_mask,
                                           (uint32_t)(v[r].z + bsum) & wrap_mask, (uint32_t)(v[r].w + bsum) & wrap_mask};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const bool ok = d[q] >= min_dep;
                        c0 += ok ? 1 : 0;
                        s32 += ok ? d[q] : 0u;
                    }
                }
                s0 = s32;
            } else if (nb >= (uint32_t)ST && lim >= (uint32_t)ST) {
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    const int bsum = base + tot[r];
                    const uint32_t d[4] = {(uint32_t)(v[r].x + bsum) & wrap_mask, (uint32_t)(v[r].y + bsum) & wrap_mask,
                                           (uint32_t)(v[r].z + bsum) & wrap_mask, (uint32_t)(v[r].w + bsum) & wrap_mask};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const bool ok = d[q] >= min_dep;
                        c0 += ok ? 1 : 0;
                        s0 += ok ? d[q] : 0u;
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    const int bsum = base + tot[r];
                    const uint32_t pos = (uint32_t)(wv * (ROWS * 256) + r * 256 + lane * 4);
                    const uint32_t d[4] = {(uint32_t)(v[r].x + bsum) & wrap_mask, (uint32_t)(v[r].y + bsum) & wrap_mask,
                                           (uint32_t)(v[r].z + bsum) & wrap_mask, (uint32_t)(v[r].w + bsum) & wrap_mask};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const bool ok = pos + q < lim && d[q] >= min_dep;
                        if (ok) { if (pos + q < nb) { ++c0; s0 += d[q]; } else { ++c1; s1 += d[q]; } }
                    }
                }
            }
        }
        c0 = wave_sum(c0); c1 = wave_sum(c1);
#pragma unroll
        for (int o = 32; o; o >>= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); }
        if (lane == 0) { red_c[wv][0] = c0; red_c[wv][1] = c1; red_s[wv][0] = s0; red_s[wv][1] = s1; }
        __syncthreads();
        if (threadIdx.x == 0) {
            TilePart tp;
            tp.c0 = (uint32_t)(red_c[0][0] + red_c[1][0] + red_c[2][0] + red_c[3][0]);
            tp.c1 = (uint32_t)(red_c[0][1] + red_c[1][1] + red_c[2][1] + red_c[3][1]);
            tp.s0 = red_s[0][0] + red_s[1][0] + red_s[2][0] + red_s[3][0];
            tp.s1 = red_s[0][1] + red_s[1][1] + red_s[2][1] + red_s[3][1];
            part[t] = tp;
        }
        __syncthreads();
    }
    // every begin and every end must have found its owner tile: totals over all batches go to batch 0's
    // counters (k_finish_direct compares sums, which is as strict: no run can be counted twice)
    __shared__ unsigned s_cnt[2];
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    {
        const int hb = wave_sum((int)n_beg), ho = wave_sum(open);
        if (lane == 0) {
            if (hb) atomicAdd(&s_cnt[0], (unsigned)hb);
            if (ho) atomicAdd(&s_cnt[1], (unsigned)ho);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        BatchDesc *desc = ps.b[0].desc;
        if (s_cnt[0]) atomicAdd((unsigned long long *)desc->handled + (blockIdx.x & (PD_CNT_SLOTS - 1)), (unsigned long long)s_cnt[0]);
        // signed total of (owned begins with cells) - (owned ends), as a 64-bit two's complement sum in `has`
        if (s_cnt[1]) atomicAdd((unsigned long long *)desc->has + (blockIdx.x & (PD_CNT_SLOTS - 1)), (unsigned long long)(long long)(int)s_cnt[1]);
    }
    (void)n_long;
}

// wave totals through the DPP scan (no LDS crossbar): the last lane holds the sum
__device__ __forceinline__ int wave_total(int v) { return __builtin_amdgcn_readlane(wave_incl_scan(v), 63); }

// The 4-bit image of one tile straight from its packed window (pd_export_i4's layout: one nibble d + 8 per cell, cell 2k in the
// low half of byte k; cells outside [-8, 7] leave as exceptions and are 0 in the image), and the tile's sum.  A lane packs the 4
// cells of each of its 4 rows to 16 bits per half-tile; the halfwords go through a KiB of LDS per wave and come back as ONE
// 16-byte piece per lane — lanes 0..31 the wave's 512 bytes of the low half-tile's image, lanes 32..63 those of the high
// half-tile's: a wave stores its KiB with one 16-byte store per lane (it was eight 2-byte stores per lane).  Exceptions are
// rare: the hot loop only notes which rows have one, a second look at those words emits them.
template <int ROWS>
__device__ __forceinline__ void export_packed_tile(const unsigned *win, const uint64_t a, const uint64_t t, const DirectExport &ex, int *wtot)
{
    constexpr uint32_t HT = TILE / 2;
    __shared__ __attribute__((aligned(16))) unsigned short stage[4][2 * ROWS * 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint4 *w4 = reinterpret_cast<const uint4 *>(win);
    unsigned short *stg = stage[wv];
    int tsum = 0;                                             // packed: sum over the lane's words
    unsigned odd = 0;                                         // bit r: row r of this lane has a cell outside [-8, 7]
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const uint4 q = w4[wv * (ROWS * 64) + r * 64 + lane];
        const unsigned wq[4] = {q.x, q.y, q.z, q.w};
        unsigned wl = 0, wh = 0, bad = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            tsum += (int)wq[k];
            const int xl = (int)(short)(wq[k] & 0xffffu), xh = ((int)wq[k] - xl) >> 16;
            const unsigned nl = (unsigned)(xl + 8), nh = (unsigned)(xh + 8);
            bad |= (nl | nh) >> 4;                            // nonzero: out of the nibble's range
            wl |= (nl < 16u ? nl : 8u) << (4 * k);
            wh |= (nh < 16u ? nh : 8u) << (4 * k);
        }
        odd |= (bad ? 1u : 0u) << r;
        stg[r * 64 + lane] = (unsigned short)wl;
        stg[ROWS * 64 + r * 64 + lane] = (unsigned short)wh;
    }
    if (__builtin_amdgcn_ballot_w64(odd != 0)) {              // rare
#pragma unroll 1
        for (int idx = 0; idx < ROWS * 4; ++idx) {
            const int r = idx >> 2, k = idx & 3;
            if (!((odd >> r) & 1u)) continue;
            const unsigned wd = win[wv * (ROWS * 256) + r * 256 + lane * 4 + k];
            const int xl = (int)(short)(wd & 0xffffu), xh = ((int)wd - xl) >> 16;
            const uint64_t cell = a + (uint64_t)(wv * (ROWS * 256) + r * 256 + lane * 4 + k);
            if ((unsigned)(xl + 8) > 15u) {
                const uint32_t slot = atomicAdd(ex.count, 1u);
                if (slot < ex.cap) { ex.exc[slot].cell = cell; ex.exc[slot].value = xl; ex.exc[slot].pad = 0; }
            }
            if ((unsigned)(xh + 8) > 15u) {
                const uint32_t slot = atomicAdd(ex.count, 1u);
                if (slot < ex.cap) { ex.exc[slot].cell = cell + HT; ex.exc[slot].value = xh; ex.exc[slot].pad = 0; }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint4 piece = reinterpret_cast<const uint4 *>(stg)[lane];
    unsigned char *img = reinterpret_cast<unsigned char *>(ex.img);
    const uint64_t at = a / 2 + (uint64_t)wv * (ROWS * 128) + (lane < 32 ? (uint64_t)lane * 16 : (uint64_t)(HT / 2) + (uint64_t)(lane - 32) * 16);
    *reinterpret_cast<uint4 *>(img + at) = piece;
    tsum = wave_total(tsum);                                  // packed 65536 * H + L over the wave (|H|, |L| <= 32 000)
    if (lane == 0) wtot[wv] = tsum;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int tp = wtot[0] + wtot[1] + wtot[2] + wtot[3];
        const int tl2 = (int)(short)(tp & 0xffff);
        ex.sums[t] = tl2 + ((tp - tl2) >> 16);
    }
    __syncthreads();
}

// k_direct_wide3 — the wide-window (w >= TILE) direct kernel, second form.  Same results as k_direct_tiles<.., DirectWide>
// (TilePart per tile, owner / open counters for k_finish_direct, heavy list); the SIMDs were 65 % busy with vector ALU work
// in that kernel (SQ_ACTIVE_INST_VALU), so this one does the same job in fewer instructions:
//   * SPLIT-HALF window: word j of the 16 KB window holds cell j in its low and cell j + TILE/2 in its high 16 bits (the
//     same "one 32-bit sum 65536 * H + L" arithmetic).  Both half-tiles go through the prefix sum in the SAME registers —
//     additions are linear in that packing — so the scan is over 4096 words, not 8192 cells: 4 row scans per lane, not 8;
//     the halves are only taken apart where depths are formed.  The event address is a mask and a shift.
//   * covered cells are counted with wave ballots of the compare masks (scalar popcounts), not per-lane adds + a reduction;
//     the remaining wave totals go through the DPP scan instead of the LDS crossbar.
//   * tiles that are not at a contig's first cell and lie entirely inside it skip the clamps of run begin / end — they cannot
//     change which cells of the tile a run touches; owner / open / carry counts are ballots accumulated in scalar registers.
//   * the tail chunk of a candidate range issues its loads together, as before, and skips the empty load slots.
#ifdef PD_WIDE3_TICKS                     /* development build only (tools/ubench/wide3_ticks.sh): shader-clock cycles per phase of a tile, summed over waves */
__device__ unsigned long long g_wide3_ticks[16];
#define W3_TICK(k) do { const long long now_ = (long long)clock64(); tk[k] += now_ - t_prev; t_prev = now_; } while (0)
#else
#define W3_TICK(k) do { } while (0)
#endif
template <int UN, int WPE, bool EXPORT>
__global__ __launch_bounds__(WG, WPE) void k_direct_wide3(const PendSet ps, uint32_t n_tiles, ContigTab tab,
                                                        const uint32_t *tile_contig, uint32_t wrap_mask, const DirectWide args,
                                                        uint32_t *heavy_list, uint32_t *heavy_count, const DirectExport ex)
{
    const uint32_t w = args.w, min_dep = args.min_dep; TilePart *const part = args.part;
    constexpr uint32_t ST = TILE, HT = TILE / 2;                 // cells per tile, per half-tile (= words of the window)
    constexpr int ROWS = (int)(HT / (WG * 4));                   // 4 rows of 4 words per lane
    __shared__ __attribute__((aligned(16))) unsigned win[HT];
    __shared__ int s_carry;
    __shared__ int wtot[4];
    __shared__ unsigned long long red_s[4][2];
    __shared__ int red_c[4][2];
    __shared__ uint32_t s_lo[PD_MAXPEND], s_hi[PD_MAXPEND];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t n_beg_s = 0; int open_s = 0;                        // owner / open counts: wave-uniform (scalar popcounts of compare masks)
    // the batches' active tile ranges and run arrays do not change from tile to tile; the candidate bounds of the NEXT tile
    // are fetched while this one is worked on (two dependent loads off the critical path)
    __shared__ uint32_t s_tf[PD_MAXPEND], s_te[PD_MAXPEND], s_n[PD_MAXPEND];
    __shared__ const pd_iv *s_iv[PD_MAXPEND];
    if (threadIdx.x < PD_MAXPEND) {
        const int b = threadIdx.x;
        uint32_t tf = 0, na = 0, n = 0; const pd_iv *ivp = nullptr;
        if (b < ps.nb) { tf = ps.b[b].desc->t_first; na = ps.b[b].desc->n_active; n = ps.b[b].n; ivp = ps.b[b].iv; }
        s_tf[b] = tf; s_te[b] = tf + na; s_n[b] = n; s_iv[b] = ivp;
    }
    __syncthreads();
    auto bounds = [&](const uint64_t t, uint32_t &lo, uint32_t &hi) {       // threads < PD_MAXPEND: batch threadIdx.x, tile t
        lo = 0; hi = 0;
        const int b = threadIdx.x;
        if (b < ps.nb && t < n_tiles && t >= s_tf[b] && t < s_te[b]) {
            const uint32_t n = s_n[b];
            hi = ps.b[b].ub_a[t + 1]; if (hi > n) hi = n;
            lo = ps.b[b].cand_lo[t]; if (lo > hi) lo = hi;
        }
    };
    uint32_t nlo = 0, nhi = 0;
    if (threadIdx.x < PD_MAXPEND) bounds(blockIdx.x, nlo, nhi);
#ifdef PD_WIDE3_TICKS
    long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = (long long)clock64();
#endif
    for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint64_t a = t * ST;
        if (threadIdx.x < PD_MAXPEND) { s_lo[threadIdx.x] = nlo; s_hi[threadIdx.x] = nhi; }
        uint4 *w4 = reinterpret_cast<uint4 *>(win);
        for (uint32_t j = threadIdx.x; j < HT / 4; j += WG) w4[j] = make_uint4(0u, 0u, 0u, 0u);
        if (threadIdx.x == 0) s_carry = 0;
        const int32_t ctg = (int32_t)tile_contig[t];
        const uint32_t clen = tab.len[ctg];
        const uint32_t p0 = (uint32_t)(a - tab.off[ctg]);         // tile start inside the contig (slots are < 2^32 cells)
        W3_TICK(0);                                               // bounds to LDS, window zeroed
        __syncthreads();
        W3_TICK(1);                                               // barrier 1
        if (threadIdx.x < PD_MAXPEND) bounds(t + gridDim.x, nlo, nhi);
        uint32_t cand = 0;
        for (int b = 0; b < ps.nb; ++b) cand += s_hi[b] - s_lo[b];
        if (cand > 32000u) {                                      // workgroup-uniform: the int-window kernel does this tile
            if (threadIdx.x == 0) heavy_list[atomicAdd(heavy_count, 1u)] = (uint32_t)t;
            __syncthreads();
            continue;
        }
        const bool interior = p0 > 0u && clen < 0x7FFFFFFFu && clen > p0 && clen - p0 >= ST;
        int carry_s = 0;                                          // wave-uniform
        // one candidate.  The predicates are single compares whose wave masks (ballots of the compares themselves: no
        // extra vector work) are combined and counted in scalar registers; only the two LDS events are per-lane work.
#define PD_B(x) __builtin_amdgcn_ballot_w64(x)
        auto one = [&](const pd_iv v, const bool fast) {
            uint32_t sb, se;
            unsigned long long m_ne, m_cov;
            bool nonempty;
            if (fast) {
                sb = (uint32_t)v.beg - p0; se = (uint32_t)v.end - p0;
                nonempty = v.beg < v.end;
                m_ne = PD_B(v.beg < v.end);
                m_cov = PD_B(v.beg < (int32_t)p0) & PD_B(v.end >= (int32_t)p0);   // begins before the tile, reaches its first cell or further
            } else {
                uint32_t bb = v.beg < 0 ? 0u : (uint32_t)v.beg; if (bb > clen) bb = clen;
                uint32_t x = v.end < 0 ? 0u : (uint32_t)v.end; if (x > clen) x = clen;
                sb = bb - p0; se = x - p0;
                nonempty = bb < x;
                m_ne = PD_B(bb < x);
                m_cov = PD_B(bb < p0) & PD_B(x >= p0);
            }
            const unsigned long long m_mine = PD_B(v.tid == ctg), m_sb = PD_B(sb < ST), m_se = PD_B(se < ST);
            const unsigned long long m_pb = m_mine & m_sb;        // owns the begin (an empty run still has an owner)
            const unsigned long long m_eb = m_pb & m_ne, m_ee = m_mine & m_ne & m_se;
            if (v.tid == ctg && nonempty) {
                if (sb < ST) atomicAdd(&win[sb & (HT - 1u)], 1u + (sb >> 12) * 0xFFFFu);
                if (se < ST) atomicSub(&win[se & (HT - 1u)], 1u + (se >> 12) * 0xFFFFu);
            }
            n_beg_s += (uint32_t)__builtin_popcountll(m_pb);
            open_s += __builtin_popcountll(m_eb) - __builtin_popcountll(m_ee);
            carry_s += __builtin_popcountll(m_mine & m_ne & m_cov);
        };
#undef PD_B
        // chunks of UN x WG candidates, stream after stream; the loads of chunk k + 1 (same stream or the next one) are in
        // flight while chunk k is worked on
        {
            auto load_chunk = [&](pd_iv (&dst)[UN], const int b, const uint32_t i) {
                const pd_iv *__restrict__ p = s_iv[b];
                const uint32_t last = s_hi[b] - 1u;
#pragma unroll
                for (int k = 0; k < UN; ++k) { const uint32_t j = i + threadIdx.x + k * WG; dst[k] = p[j < last ? j : last]; }
            };
            int b = 0;
            while (b < ps.nb && s_lo[b] >= s_hi[b]) ++b;
            uint32_t i = b < ps.nb ? s_lo[b] : 0u;
            pd_iv cur[UN];
            if (b < ps.nb) load_chunk(cur, b, i);
#pragma unroll 1
            while (b < ps.nb) {
                int nb2 = b; uint32_t ni = i + UN * WG;
                if (ni >= s_hi[b]) { nb2 = b + 1; while (nb2 < ps.nb && s_lo[nb2] >= s_hi[nb2]) ++nb2; ni = nb2 < ps.nb ? s_lo[nb2] : 0u; }
                pd_iv nx[UN];
                if (nb2 < ps.nb) load_chunk(nx, nb2, ni);
                const uint32_t left = s_hi[b] - i;
                if (left >= UN * WG) {
                    if (interior) {
#pragma unroll
                        for (int k = 0; k < UN; ++k) one(cur[k], true);
                    } else {
#pragma unroll
                        for (int k = 0; k < UN; ++k) one(cur[k], false);
                    }
                } else {                                          // the tail chunk of a stream: empty slots skipped
                    const int nu = (int)((left + WG - 1) / WG);   // uniform, 1 .. UN
#pragma unroll
                    for (int k = 0; k < UN; ++k) if (k < nu) {
                        pd_iv v = cur[k];
                        if (threadIdx.x + k * WG >= left) v.tid = -1;       // never equals a contig id
                        one(v, interior);
                    }
                }
                if (nb2 < ps.nb) {
#pragma unroll
                    for (int k = 0; k < UN; ++k) cur[k] = nx[k];
                }
                b = nb2; i = ni;
            }
        }
        if (lane == 0 && carry_s != 0) atomicAdd(&s_carry, carry_s);
        W3_TICK(2);                                               // candidates
        __syncthreads();
        W3_TICK(3);                                               // barrier 2
        if constexpr (EXPORT) {                                   // the multi-GPU sum's 4-bit image (pd_export_i4)
            export_packed_tile<ROWS>(win, a, t, ex, wtot);
            continue;
        }
        // ---- prefix sum of the packed window: both half-tiles at once ----
        int4 v[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const uint4 q = w4[wv * (ROWS * 64) + r * 64 + lane];
            v[r] = make_int4((int)q.x, (int)q.x + (int)q.y, 0, 0);
            v[r].z = v[r].y + (int)q.z; v[r].w = v[r].z + (int)q.w;
        }
        int ex[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) ex[r] = wave_incl_scan(v[r].w);
        int run = 0;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int e = run + ex[r] - v[r].w;                   // exclusive prefix of this lane's group in the wave
            run += __builtin_amdgcn_readlane(ex[r], 63);
            ex[r] = e;
        }
        if (lane == 0) wtot[wv] = run;
        W3_TICK(4);                                               // window read + scans
        __syncthreads();
        W3_TICK(5);                                               // barrier 3
        int basep = 0;                                            // packed: words of the waves before this one
        for (int k = 0; k < wv; ++k) basep += wtot[k];
        const int totp = wtot[0] + wtot[1] + wtot[2] + wtot[3];
        const int tl = (int)(short)(totp & 0xffff);               // begins - ends over the low half-tile
        const int carry_l = s_carry, carry_h = carry_l + tl;      // depth just before cell 0 / cell HT of the tile
        // ---- the tile's share of windows k0 and k0 + 1 ----
        const uint64_t local0 = p0;
        int c0 = 0, c1 = 0; unsigned long long s0 = 0, s1 = 0;
        bool uniform_counts = false;                              // c0 is a wave count (ballots), s0 a 32-bit lane sum
        if (local0 < clen) {
            const uint64_t k0 = local0 / w;
            const uint64_t nb64 = (k0 + 1) * (uint64_t)w - local0;   // tile-local start of window k0+1
            const uint32_t nb = nb64 > (uint64_t)ST ? ST : (uint32_t)nb64;
            const uint64_t left = (uint64_t)clen - local0;
            const uint32_t lim = left > (uint64_t)ST ? ST : (uint32_t)left;
            const uint64_t dmax = (uint64_t)(uint32_t)carry_l + cand;     // no depth in this tile exceeds carry + begins
            if (nb >= ST && lim >= ST && min_dep <= 1u && dmax < (1u << 27) && dmax <= wrap_mask) {
                // the common tile — inside one window, inside the contig, no wrap possible, threshold <= 1: the sum is the
                // sum of the local prefixes + 16 x (carry_l + carry_h) per lane; a cell is covered unless its prefix = -carry
                int sl = 0; int cnt = 0;
                const int zl = -carry_l, zh = -carry_h;
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    const int add = basep + ex[r];
                    const int pw[4] = {v[r].x + add, v[r].y + add, v[r].z + add, v[r].w + add};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int L = (int)(short)(pw[q] & 0xffff), H = (pw[q] - L) >> 16;
                        sl += L + H;
                        if (min_dep) cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(L != zl)) + __builtin_popcountll(__builtin_amdgcn_ballot_w64(H != zh));
                    }
                }
                c0 = min_dep ? cnt : (int)(ROWS * 8 * 64);
                s0 = (uint32_t)sl + (uint32_t)(ROWS * 4) * ((uint32_t)carry_l + (uint32_t)carry_h);
                uniform_counts = true;
            } else if (nb >= ST && lim >= ST && dmax < (1u << 27)) {
                uint32_t s32 = 0; int cnt = 0;
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    const int add = basep + ex[r];
                    const int pw[4] = {v[r].x + add, v[r].y + add, v[r].z + add, v[r].w + add};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int L = (int)(short)(pw[q] & 0xffff), H = (pw[q] - L) >> 16;
                        const uint32_t dl = (uint32_t)(L + carry_l) & wrap_mask, dh = (uint32_t)(H + carry_h) & wrap_mask;
                        const bool okl = dl >= min_dep, okh = dh >= min_dep;
                        cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(okl)) + __builtin_popcountll(__builtin_amdgcn_ballot_w64(okh));
                        s32 += (okl ? dl : 0u) + (okh ? dh : 0u);
                    }
                }
                c0 = cnt; s0 = s32; uniform_counts = true;
            } else {
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    const int add = basep + ex[r];
                    const uint32_t pos = (uint32_t)(wv * (ROWS * 256) + r * 256 + lane * 4);
                    const int pw[4] = {v[r].x + add, v[r].y + add, v[r].z + add, v[r].w + add};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int L = (int)(short)(pw[q] & 0xffff), H = (pw[q] - L) >> 16;
                        const uint32_t dl = (uint32_t)(L + carry_l) & wrap_mask, dh = (uint32_t)(H + carry_h) & wrap_mask;
                        const uint32_t pl = pos + q, ph = pl + HT;
                        if (pl < lim && dl >= min_dep) { if (pl < nb) { ++c0; s0 += dl; } else { ++c1; s1 += dl; } }
                        if (ph < lim && dh >= min_dep) { if (ph < nb) { ++c0; s0 += dh; } else { ++c1; s1 += dh; } }
                    }
                }
            }
        }
        if (uniform_counts) {                                     // workgroup-uniform
            const uint32_t s32 = (uint32_t)s0;
            const unsigned long long lo16 = (unsigned long long)(uint32_t)wave_total((int)(s32 & 0xffffu));
            const unsigned long long hi16 = (unsigned long long)(uint32_t)wave_total((int)(s32 >> 16));
            s0 = (hi16 << 16) + lo16;
        } else {
            c0 = wave_sum(c0); c1 = wave_sum(c1);
#pragma unroll
            for (int o = 32; o; o >>= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); }
        }
        if (lane == 0) { red_c[wv][0] = c0; red_c[wv][1] = c1; red_s[wv][0] = s0; red_s[wv][1] = s1; }
        W3_TICK(6);                                               // statistics
        __syncthreads();
        W3_TICK(7);                                               // barrier 4
        if (threadIdx.x == 0) {
            TilePart tp;
            tp.c0 = (uint32_t)(red_c[0][0] + red_c[1][0] + red_c[2][0] + red_c[3][0]);
            tp.c1 = (uint32_t)(red_c[0][1] + red_c[1][1] + red_c[2][1] + red_c[3][1]);
            tp.s0 = red_s[0][0] + red_s[1][0] + red_s[2][0] + red_s[3][0];
            tp.s1 = red_s[0][1] + red_s[1][1] + red_s[2][1] + red_s[3][1];
            part[t] = tp;
        }
        // (no barrier here: everything the next tile overwrites — bounds, window, carry, wave totals, partials — is last read
        // before one of the three barriers that precede the overwriting store)
    }
#ifdef PD_WIDE3_TICKS
    if (lane == 0) for (int k = 0; k < 8; ++k) atomicAdd(&g_wide3_ticks[k], (unsigned long long)tk[k]);
#endif
    // every begin and every end must have found its owner tile (k_finish_direct compares the sums over all batches)
    __shared__ unsigned s_cnt[2];
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    if (lane == 0) {
        if (n_beg_s) atomicAdd(&s_cnt[0], n_beg_s);
        if (open_s) atomicAdd(&s_cnt[1], (unsigned)open_s);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        BatchDesc *desc = ps.b[0].desc;
        if (s_cnt[0]) atomicAdd((unsigned long long *)desc->handled + (blockIdx.x & (PD_CNT_SLOTS - 1)), (unsigned long long)s_cnt[0]);
        if (s_cnt[1]) atomicAdd((unsigned long long *)desc->has + (blockIdx.x & (PD_CNT_SLOTS - 1)), (unsigned long long)(long long)(int)s_cnt[1]);
    }
}

__global__ __launch_bounds__(WG) void k_direct_tiles_heavy(const PendSet ps, uint32_t n_tiles, ContigTab tab,
                                                     const uint32_t *tile_contig, uint32_t wrap_mask, const WinArgs wa,
                                                     const uint64_t *win_off, uint32_t *n_long,
                                                     const uint32_t *heavy_list, const uint32_t *heavy_count,
                                                     const DirectExport ex,          // ex.img != null: export instead of statistics
                                                     const C8Sample cs)              // cs.r8 != null: a compact sample (then ps is empty)
{
    const uint32_t w = wa.w, min_dep = wa.min_dep;
    TilePart *const part = wa.part;
    __shared__ unsigned long long acc[TILE / 64 + 2];
    constexpr int ST = TILE;
    constexpr int ROWS = TILE / (WG * 4);                        // 8
    __shared__ __attribute__((aligned(16))) int win[ST];
    __shared__ int s_carry;
    __shared__ int wtot[4];
    __shared__ unsigned long long red_s[4][2];
    __shared__ int red_c[4][2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t tf[PD_MAXPEND], na[PD_MAXPEND];
    uint32_t n_beg[PD_MAXPEND], n_has[PD_MAXPEND], n_end[PD_MAXPEND];
    uint32_t longs = 0;
#pragma unroll
    for (int b = 0; b < PD_MAXPEND; ++b) {
        n_beg[b] = n_has[b] = n_end[b] = 0; tf[b] = 0; na[b] = 0;
        if (b < ps.nb) { tf[b] = ps.b[b].desc->t_first; na[b] = ps.b[b].desc->n_active; }
    }
    const uint32_t lmax = ps.lmax;
    const uint32_t n_heavy = *heavy_count < n_tiles ? *heavy_count : n_tiles;
    for (uint32_t hi_ = blockIdx.x; hi_ < n_heavy; hi_ += gridDim.x) {
        const uint64_t t = heavy_list[hi_];
        const uint64_t a = t * ST;
        int4 *w4 = reinterpret_cast<int4 *>(win);
        for (int j = threadIdx.x; j < ST / 4; j += WG) w4[j] = make_int4(0, 0, 0, 0);
        if (threadIdx.x == 0) s_carry = 0;
        const int32_t ctg = (int32_t)tile_contig[t];
        const uint32_t clen = tab.len[ctg];
        const int64_t rel = (int64_t)(tab.off[ctg] - a);          // slot start relative to the tile (<= 0)
        __syncthreads();
        int carry = 0;
#pragma unroll
        for (int b = 0; b < PD_MAXPEND; ++b) {
            if (b >= ps.nb || t < tf[b] || t >= (uint64_t)tf[b] + na[b]) continue;
            const uint32_t n = ps.b[b].n;
            uint32_t hi = ps.b[b].ub_a[t + 1]; if (hi > n) hi = n;
            uint32_t lo = ps.b[b].cand_lo[t]; if (lo > hi) lo = hi;
            const pd_iv *__restrict__ iv = ps.b[b].iv;
            constexpr int UN = 4;
            for (uint32_t i0 = lo + threadIdx.x; i0 < hi; i0 += UN * WG) {
                pd_iv vv[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const uint32_t i = i0 + u * WG;
                    vv[u] = iv[i < hi ? i : hi - 1];
                    if (i >= hi) vv[u].tid = -1;
                }
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const pd_iv v = vv[u];
                    if (v.tid != ctg) continue;
                    uint32_t bb = v.beg < 0 ? 0u : (uint32_t)v.beg; if (bb > clen) bb = clen;
                    uint32_t x = v.end < 0 ? 0u : (uint32_t)v.end; if (x > clen) x = clen;
                    const bool has = bb < x;
                    const int64_t sb = rel + (int64_t)bb, se = rel + (int64_t)x;     // tile-relative begin / end
                    if (sb >= 0 && sb < ST) {
                        ++n_beg[b];
                        if (has) { ++n_has[b]; atomicAdd(&win[sb], 1); if (x - bb > lmax) ++longs; }
                    }
                    if (has) {
                        if (se >= 0 && se < ST) { atomicAdd(&win[se], -1); ++n_end[b]; }
                        if (sb < 0 && se >= 0) ++carry;           // covers the cell just before the tile
                    }
                }
            }
        }
        if (cs.r8) {                                              // a compact sample: the tile's own buckets and the one before them, in both streams
            const uint32_t k0 = (uint32_t)t << cs.bshift, kl = rel < 0 ? k0 - 1 : k0;           // (a contig's first tile has nothing before it)
            const uint32_t a32 = (uint32_t)a;
            for (int strm = 0; strm < 2; ++strm) {
                const uint32_t *bs = strm ? cs.o1 : cs.b1;
                const Run8 *r8 = strm ? cs.r8 + cs.o_base : cs.r8;
                const uint32_t lo = bs[kl], hi = bs[k0 + (1u << cs.bshift)];
                for (uint32_t i = lo + threadIdx.x; i < hi; i += WG) {
                    const Run8 r = r8[i];
                    if (!r.len) continue;
                    const int64_t sb = (int64_t)(int32_t)(r.b - a32), se = sb + (int64_t)r.len;   // tile-relative (the begins are flat, mod 2^32)
                    if (sb >= 0 && sb < ST) atomicAdd(&win[sb], 1);
                    if (se >= 0 && se < ST) atomicAdd(&win[se], -1);
                    if (sb < 0 && se >= 0) ++carry;
                }
            }
        }
        carry = wave_sum(carry);
        if (lane == 0 && carry != 0) atomicAdd(&s_carry, carry);
        __syncthreads();
        if (ex.img) {                                             // workgroup-uniform: the window leaves as nibbles (k_export_i4's layout)
            const uint64_t cw = a + (uint64_t)wv * (ROWS * 256);
            unsigned short *o = ex.img + cw / 4 + lane;
            int tsum = 0;
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                const int4 q = w4[wv * (ROWS * 64) + r * 64 + lane];
                int x[4] = {q.x, q.y, q.z, q.w};
                unsigned wb = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    tsum += x[k];
                    if (x[k] > 7 || x[k] < -8) {
                        const uint32_t slot = atomicAdd(ex.count, 1u);
                        if (slot < ex.cap) { ex.exc[slot].cell = cw + (uint64_t)(r * 256 + lane * 4 + k); ex.exc[slot].value = x[k]; ex.exc[slot].pad = 0; }
                        x[k] = 0;
                    }
                    wb |= (unsigned)((x[k] + 8) & 0xf) << (4 * k);
                }
                o[r * 64] = (unsigned short)wb;
            }
            tsum = wave_sum(tsum);
            if (lane == 0) wtot[wv] = tsum;
            __syncthreads();
            if (threadIdx.x == 0) ex.sums[t] = wtot[0] + wtot[1] + wtot[2] + wtot[3];
            __syncthreads();
            continue;
        }
        // ---- prefix sum of the window, straight from LDS (k_sweep's layout) ----
        int4 v[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) v[r] = w4[wv * (ROWS * 64) + r * 64 + lane];
        int tot[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            v[r].y += v[r].x; v[r].z += v[r].y; v[r].w += v[r].z;
            tot[r] = v[r].w;
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) tot[r] = wave_incl_scan(tot[r]);
        int run = 0;
        int excl[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            excl[r] = run + tot[r] - v[r].w;
            run += __builtin_amdgcn_readlane(tot[r], 63);
        }
        if (lane == 0) wtot[wv] = run;
        __syncthreads();
        int base = s_carry;
        for (int k = 0; k < wv; ++k) base += wtot[k];
        if (w < (uint32_t)ST) {                                   // narrow windows: LDS accumulators (uniform branch)
            const uint32_t p0 = (uint32_t)(a - tab.off[ctg]);
            if (p0 < clen) direct_small_windows<ROWS>(v, excl, base, wrap_mask, p0, clen, wa, win_off[ctg], acc, wa.part + t);
            __syncthreads();
            continue;
        }
        // ---- the tile's share of windows k0 and k0 + 1 ----
        const uint64_t local0 = a - tab.off[ctg];
        int c0 = 0, c1 = 0; unsigned long long s0 = 0, s1 = 0;
        if (local0 < clen) {
            const uint64_t k0 = local0 / w;
            const uint64_t nb = (k0 + 1) * (uint64_t)w - local0;     // tile-local start of window k0+1
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                const int bsum = base + excl[r];
                const uint32_t pos = (uint32_t)(wv * (ROWS * 256) + r * 256 + lane * 4);
                const uint32_t d[4] = {(uint32_t)(v[r].x + bsum) & wrap_mask, (uint32_t)(v[r].y + bsum) & wrap_mask,
                                       (uint32_t)(v[r].z + bsum) & wrap_mask, (uint32_t)(v[r].w + bsum) & wrap_mask};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool ok = local0 + pos + q < clen && d[q] >= min_dep;
                    if (ok) { if (pos + q < nb) { ++c0; s0 += d[q]; } else { ++c1; s1 += d[q]; } }
                }
            }
        }
        c0 = wave_sum(c0); c1 = wave_sum(c1);
#pragma unroll
        for (int o = 32; o; o >>= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); }
        if (lane == 0) { red_c[wv][0] = c0; red_c[wv][1] = c1; red_s[wv][0] = s0; red_s[wv][1] = s1; }
        __syncthreads();
        if (threadIdx.x == 0) {
            TilePart tp;
            tp.c0 = (uint32_t)(red_c[0][0] + red_c[1][0] + red_c[2][0] + red_c[3][0]);
            tp.c1 = (uint32_t)(red_c[0][1] + red_c[1][1] + red_c[2][1] + red_c[3][1]);
            tp.s0 = red_s[0][0] + red_s[1][0] + red_s[2][0] + red_s[3][0];
            tp.s1 = red_s[0][1] + red_s[1][1] + red_s[2][1] + red_s[3][1];
            part[t] = tp;
        }
        __syncthreads();
    }
    // the same accounting as k_scatter_tiles: every begin and every end must have found its owner tile
    __shared__ unsigned s_cnt[PD_MAXPEND][3];
    __shared__ unsigned s_long;
    if (threadIdx.x < PD_MAXPEND * 3) s_cnt[threadIdx.x / 3][threadIdx.x % 3] = 0;
    if (threadIdx.x == 0) s_long = 0;
    __syncthreads();
#pragma unroll
    for (int b = 0; b < PD_MAXPEND; ++b) {
        if (b >= ps.nb) continue;
        const int hb = wave_sum((int)n_beg[b]), hh = wave_sum((int)n_has[b]), he = wave_sum((int)n_end[b]);
        if (lane == 0) {
            if (hb) atomicAdd(&s_cnt[b][0], (unsigned)hb);
            if (hh) atomicAdd(&s_cnt[b][1], (unsigned)hh);
            if (he) atomicAdd(&s_cnt[b][2], (unsigned)he);
        }
    }
    const int hl = wave_sum((int)longs);
    if (lane == 0 && hl) atomicAdd(&s_long, (unsigned)hl);
    __syncthreads();
    if (threadIdx.x < ps.nb * 3) {
        const int b = threadIdx.x / 3, k = threadIdx.x % 3;
        const unsigned v = s_cnt[b][k];
        if (v) {
            BatchDesc *desc = ps.b[b].desc;
            unsigned long long *dst = (unsigned long long *)(k == 0 ? desc->handled : k == 1 ? desc->has : desc->ends);
            atomicAdd(dst + (blockIdx.x & (PD_CNT_SLOTS - 1)), (unsigned long long)v);
        }
    }
    if (threadIdx.x == 0 && s_long) atomicAdd(n_long, s_long);
}

// ------------------------------------------------------------------------------------------
// compact samples (pd_runs_create / pd_push_runs, and what the GPU decoder leaves behind): see C8Sample in pd_kernels.h
// ------------------------------------------------------------------------------------------
// flat begin (clamped to the contig) and clamped length of a run; valid = its contig id is one
__device__ __forceinline__ uint64_t c8_flat(const pd_iv v, const ContigTab tab, uint32_t &len, bool &valid)
{
    valid = v.tid >= 0 && v.tid < tab.n;
    len = 0;
    if (!valid) return 0;
    const uint32_t clen = tab.len[v.tid];
    uint32_t b = v.beg < 0 ? 0u : (uint32_t)v.beg; if (b > clen) b = clen;
    uint32_t x = v.end < 0 ? 0u : (uint32_t)v.end; if (x > clen) x = clen;
    len = x > b ? x - b : 0u;
    return tab.off[v.tid] + b;             // (b <= clen < the slot's cells: always inside the contig's own slot, so below n_buckets << cshift)
}

// The sorted stream from 12-byte runs (pd_runs_create; the decoder's emit kernel, pd_bamwalk.h, does the same while it writes its runs):
// every run becomes 8 bytes — the low 32 bits of its flat begin, its clamped length —, its order is CHECKED against its predecessor
// (words[0]: a contig id out of range, or a run that begins before the one in front of it), runs longer than a bucket are counted
// (words[1]: such a sample is not used in this form), and the first run of every bucket leaves its index in b1[bucket] (pre-set to
// 0xFFFFFFFF; the buckets nobody begins in are filled by launch_c8_fill_starts).
__global__ __launch_bounds__(WG) void k_c8_from_sorted(const pd_iv *iv, uint32_t n, ContigTab tab, uint32_t bshift, Run8 *out, uint32_t *b1, uint32_t *words)
{
    const uint64_t i64 = (uint64_t)blockIdx.x * WG + threadIdx.x;
    const uint32_t i = i64 < n ? (uint32_t)i64 : n - 1;
    const uint32_t cshift = 13u - bshift;                        // log2(cells per bucket)
    uint32_t len; bool valid;
    const uint64_t gb = c8_flat(iv[i], tab, len, valid);
    uint64_t prev = ((uint64_t)(uint32_t)__shfl_up((int)(uint32_t)(gb >> 32), 1) << 32) | (uint32_t)__shfl_up((int)(uint32_t)gb, 1);
    bool pvalid = __shfl_up((int)valid, 1) != 0;
    if ((threadIdx.x & 63) == 0 && i > 0) { uint32_t pl; prev = c8_flat(iv[i - 1], tab, pl, pvalid); }
    if (i64 >= n) return;
    if (!valid || (i > 0 && (!pvalid || gb < prev))) atomicOr(&words[0], 1u);
    if (len > (1u << cshift)) atomicAdd(&words[1], 1u);
    out[i] = Run8{(uint32_t)gb, len};
    if (valid && (i == 0 || !pvalid || (prev >> cshift) != (gb >> cshift))) atomicMin(&b1[gb >> cshift], i);
}

// The other streams (any order, but in practice the later runs of a file's reads in file order: nearly sorted).  Neighbouring lanes
// whose runs begin in the same bucket act ONCE: the first of them adds their number to the bucket's counter (a device atomic per
// run was 4.4 ms for the 1.1e8 later runs of the 1e9-record sample; per group of equal neighbours it is a tenth of that).
// group(key): the lane's group of equal neighbours as (first lane, size); every lane of the wave must call it
__device__ __forceinline__ void c8_group(const uint32_t key, int &head_lane, uint32_t &size)
{
    const int lane = threadIdx.x & 63;
    const uint32_t pk = (uint32_t)__shfl_up((int)key, 1);
    const unsigned long long heads = __ballot(lane == 0 || pk != key);
    const unsigned long long le = heads & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
    head_lane = 63 - __builtin_clzll(le);                         // (bit 0 is always set)
    // size of the group = distance from its first lane to the next group's first lane
    const unsigned long long above_h = head_lane == 63 ? 0ull : heads & ~((2ull << head_lane) - 1ull);
    size = (uint32_t)((above_h ? __builtin_ctzll(above_h) : 64) - head_lane);
}
__global__ __launch_bounds__(WG) void k_c8_hist(const pd_iv *iv, uint32_t n, ContigTab tab, uint32_t bshift, uint32_t *hist, uint32_t *words)
{
    const uint32_t cshift = 13u - bshift;
    const int lane = threadIdx.x & 63;
    for (uint64_t base = (uint64_t)blockIdx.x * WG; base < n; base += (uint64_t)gridDim.x * WG) {      // (workgroup-uniform trip count)
        const uint64_t i = base + threadIdx.x;
        uint32_t key = 0xFFFFFFFFu;
        if (i < n) {
            uint32_t len; bool valid;
            const uint64_t gb = c8_flat(iv[i], tab, len, valid);
            if (!valid) atomicOr(&words[0], 1u);
            else { key = (uint32_t)(gb >> cshift); if (len > (1u << cshift)) atomicAdd(&words[1], 1u); }
        }
        int hl; uint32_t sz;
        c8_group(key, hl, sz);
        if (hl == lane && key != 0xFFFFFFFFu) atomicAdd(&hist[key], sz);
    }
}

// exclusive prefix sum of n 32-bit counts (three small kernels: sums of blocks of 1024, their scan, the blocks)
__global__ __launch_bounds__(WG) void k_scan_block_sums(const uint32_t *in, uint32_t n, uint32_t *bs)
{
    __shared__ uint32_t ws[4];
    const uint64_t base = (uint64_t)blockIdx.x * 1024 + threadIdx.x * 4;
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) if (base + k < n) v += in[base + k];
    v = (uint32_t)wave_total((int)v);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) bs[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
__global__ __launch_bounds__(1024) void k_scan_of_sums(uint32_t *bs, uint32_t nb)
{
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t run_s;
    if (threadIdx.x == 0) run_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nb; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < nb ? bs[i] : 0u;
        const uint32_t inc = (uint32_t)wave_incl_scan((int)v);
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
        __syncthreads();
        uint32_t off = run_s;
        for (int k = 0; k < (int)(threadIdx.x >> 6); ++k) off += wsum[k];
        if (i < nb) bs[i] = off + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) run_s = off + inc;
        __syncthreads();
    }
}
__global__ __launch_bounds__(WG) void k_scan_blocks(const uint32_t *in, uint32_t *out, uint32_t n, const uint32_t *bs)
{
    __shared__ uint32_t ws[4];
    const uint64_t base = (uint64_t)blockIdx.x * 1024 + threadIdx.x * 4;
    uint32_t x[4], v = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { x[k] = base + k < n ? in[base + k] : 0u; v += x[k]; }
    const uint32_t inc = (uint32_t)wave_incl_scan((int)v);
    if ((threadIdx.x & 63) == 63) ws[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t off = bs[blockIdx.x] + inc - v;
    for (int k = 0; k < (int)(threadIdx.x >> 6); ++k) off += ws[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) { if (base + k < n) out[base + k] = off; off += x[k]; }
}

// Bucket starts from the marks the first runs of the buckets left: a[k] = min(a[k], a[k + 1], ..., a[n - 1]) in place (a SUFFIX minimum;
// the caller has put the number of runs into a[n - 1]).  Same three-kernel shape as the prefix sum above, walked from the far end.
__device__ __forceinline__ uint32_t wave_suffix_min(uint32_t m)      // lane l: min over lanes l .. 63
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = (uint32_t)__shfl_down((int)m, d); if (lane + d < 64 && y < m) m = y; }
    return m;
}
__global__ __launch_bounds__(WG) void k_sfx_block_min(const uint32_t *a, uint32_t n, uint32_t *bm)
{
    __shared__ uint32_t ws[4];
    const uint64_t base = (uint64_t)blockIdx.x * 1024 + threadIdx.x * 4;
    uint32_t v = 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < 4; ++k) if (base + k < n && a[base + k] < v) v = a[base + k];
    v = wave_suffix_min(v);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t m = ws[0]; for (int k = 1; k < 4; ++k) if (ws[k] < m) m = ws[k]; bm[blockIdx.x] = m; }
}
// bm[j] := min(bm[j + 1 ..]) (what lies strictly behind block j), one workgroup
__global__ __launch_bounds__(1024) void k_sfx_of_mins(uint32_t *bm, uint32_t nb)
{
    __shared__ uint32_t wmin[16];
    __shared__ uint32_t run_s;
    if (threadIdx.x == 0) run_s = 0xFFFFFFFFu;
    __syncthreads();
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (uint32_t top = (nb + 1023u) / 1024u * 1024u; top > 0; top -= 1024) {
        const uint32_t i = top - 1024 + threadIdx.x;
        const uint32_t v = i < nb ? bm[i] : 0xFFFFFFFFu;
        const uint32_t inc = wave_suffix_min(v);                 // min over this wave's lanes lane .. 63
        if (lane == 0) wmin[wv] = inc;
        __syncthreads();
        uint32_t behind = run_s;                                  // later chunks
        for (int k = wv + 1; k < 16; ++k) if (wmin[k] < behind) behind = wmin[k];
        const uint32_t nxt = (uint32_t)__shfl_down((int)inc, 1);  // lanes lane + 1 .. 63 of this wave
        uint32_t ex = behind;
        if (lane < 63 && nxt < ex) ex = nxt;
        uint32_t all = behind; if (inc < all) all = inc;          // (thread 0: everything from this chunk on)
        __syncthreads();
        if (i < nb) bm[i] = ex;
        if (threadIdx.x == 0) run_s = all;
        __syncthreads();
    }
}
__global__ __launch_bounds__(WG) void k_sfx_blocks(uint32_t *a, uint32_t n, const uint32_t *bmx)
{
    __shared__ uint32_t ws[4];
    const uint64_t base = (uint64_t)blockIdx.x * 1024 + threadIdx.x * 4;
    uint32_t x[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = base + k < n ? a[base + k] : 0xFFFFFFFFu;
#pragma unroll
    for (int k = 2; k >= 0; --k) if (x[k + 1] < x[k]) x[k] = x[k + 1];       // suffix minimum inside the thread's four
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t inc = wave_suffix_min(x[0]);
    if (lane == 0) ws[wv] = inc;
    __syncthreads();
    uint32_t behind = bmx[blockIdx.x];
    for (int k = wv + 1; k < 4; ++k) if (ws[k] < behind) behind = ws[k];
    const uint32_t nxt = (uint32_t)__shfl_down((int)inc, 1);
    if (lane < 63 && nxt < behind) behind = nxt;                   // what lies behind this thread's four
#pragma unroll
    for (int k = 0; k < 4; ++k) if (base + k < n) a[base + k] = x[k] < behind ? x[k] : behind;
}
__global__ void k_set_u32(uint32_t *p, uint32_t v) { *p = v; }

// the decoder's marks — per bucket the smallest (batch << 32 | index in the batch) of a run that begins there, all ones where none does —
// become indices into the sample's sorted stream: base[batch] + index (0xFFFFFFFF stays "nobody")
__global__ __launch_bounds__(WG) void k_c8_marks_to_index(const unsigned long long *marks, uint32_t n, const uint32_t *base, uint32_t *b1)
{
    const uint64_t k = (uint64_t)blockIdx.x * WG + threadIdx.x;
    if (k >= n) return;
    const unsigned long long m = marks[k];
    b1[k] = m == ~0ull ? 0xFFFFFFFFu : base[(uint32_t)(m >> 32)] + (uint32_t)m;
}

// the other streams' runs to their buckets: o1 = exclusive prefix sum of the histogram; the runs of a group of equal neighbours take
// consecutive places from ONE atomic on the bucket's cursor
__global__ __launch_bounds__(WG) void k_c8_place_other(const pd_iv *iv, uint32_t n, ContigTab tab, uint32_t bshift, const uint32_t *o1,
                                                       uint32_t *cursor, Run8 *out)
{
    const uint32_t cshift = 13u - bshift;
    const int lane = threadIdx.x & 63;
    for (uint64_t base = (uint64_t)blockIdx.x * WG; base < n; base += (uint64_t)gridDim.x * WG) {
        const uint64_t i = base + threadIdx.x;
        uint32_t key = 0xFFFFFFFFu, len = 0; uint64_t gb = 0;
        if (i < n) { bool valid; gb = c8_flat(iv[i], tab, len, valid); if (valid) key = (uint32_t)(gb >> cshift); }
        int hl; uint32_t sz;
        c8_group(key, hl, sz);
        uint32_t at = 0;
        if (hl == lane && key != 0xFFFFFFFFu) at = o1[key] + atomicAdd(&cursor[key], sz);
        at = (uint32_t)__shfl((int)at, hl);
        if (key != 0xFFFFFFFFu) out[at + (uint32_t)(lane - hl)] = Run8{(uint32_t)gb, len};
    }
}

// the reverse (a compact sample that has to take a path that reads 12-byte runs): bucket by bucket — a bucket's sorted-stream runs,
// then its other runs —, so the result is sorted up to one bucket's cells of disorder
__global__ __launch_bounds__(WG) void k_c8_expand(const C8Sample cs, const uint32_t *tile_contig, const uint64_t *contig_off, uint32_t n_tiles, pd_iv *out)
{
    const uint32_t cshift = 13u - cs.bshift, nbk = 1u << cs.bshift;
    for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint32_t k0 = t << cs.bshift;
        const int32_t ctg = (int32_t)tile_contig[t];
        const uint32_t a32 = (uint32_t)((uint64_t)t * TILE), c32 = (uint32_t)contig_off[ctg];
        for (uint32_t i = cs.b1[k0] + threadIdx.x; i < cs.b1[k0 + nbk]; i += WG) {
            const Run8 r = cs.r8[i];
            const uint32_t kb = k0 + ((r.b - a32) >> cshift), beg = r.b - c32;
            out[i + cs.o1[kb]] = pd_iv{ctg, (int32_t)beg, (int32_t)(beg + r.len)};
        }
        for (uint32_t j = cs.o1[k0] + threadIdx.x; j < cs.o1[k0 + nbk]; j += WG) {
            const Run8 r = cs.r8[cs.o_base + j];
            const uint32_t kb = k0 + ((r.b - a32) >> cshift), beg = r.b - c32;
            out[j + cs.b1[kb + 1]] = pd_iv{ctg, (int32_t)beg, (int32_t)(beg + r.len)};
        }
    }
}

// A batch's runs on their way to their places (a few MB, twice per decode batch): the runtime's device-to-device copy is a blit kernel that
// took 0.15 ms per call among the decode kernels (13 % of the decode phase's kernel time, profiles/r05_decode_timeline.txt); this one moves
// 4-byte words with every lane on its own 16 bytes where source and destination allow it.
__global__ __launch_bounds__(256) void k_copy_words(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, uint64_t n)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    if ((((uintptr_t)dst | (uintptr_t)src) & 15) == 0) {
        const uint64_t n4 = n / 4;
        uint4 *d4 = reinterpret_cast<uint4 *>(dst); const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
        for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) d4[i] = s4[i];
        for (uint64_t i = n4 * 4 + (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = src[i];
        return;
    }
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

// a sorted stream of compact runs whose sample turned out not to be usable as one (the file's order does not hold after all): back to
// 12-byte runs; the contig of a flat begin by bisection over the slots (genomes below 2^32 cells only: the caller has made sure)
__global__ __launch_bounds__(WG) void k_r8_to_iv(const Run8 *r8, uint64_t n, ContigTab tab, pd_iv *out)
{
    for (uint64_t i = (uint64_t)blockIdx.x * WG + threadIdx.x; i < n; i += (uint64_t)gridDim.x * WG) {
        const Run8 r = r8[i];
        int lo = 0, hi = tab.n - 1;                               // last contig whose slot starts at or before r.b
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (tab.off[mid] <= (uint64_t)r.b) lo = mid; else hi = mid - 1; }
        const uint32_t beg = r.b - (uint32_t)tab.off[lo];
        out[i] = pd_iv{lo, (int32_t)beg, (int32_t)(beg + r.len)};
    }
}

// k_direct_c8 — the wide-window direct kernel on a compact sample: exact bounds, nothing to test but where the two
// events of a run fall.  Per run (13 vector instructions; k_direct_wide3: 26, on 12-byte runs with a contig compare and clamps):
//   sb = b - p0 (mod 2^32; b = the low 32 bits of the run's flat begin, p0 those of the tile's first cell): < TILE exactly for the tile's
//   own runs;  se = sb + len: < TILE when the end lies in the tile;  se < len exactly when the run begins before the tile and reaches
//   its first cell or further (the carry-in).
// A run without cells adds and subtracts at the same cell.  The tile's candidates are its own buckets and the one before them, in
// BOTH streams of the sample (the file's sorted first runs as the decoder wrote them, and the later runs counting-sorted by bucket):
// one loop over the two ranges laid end to end.  Same window arithmetic, prefix sum and statistics as k_direct_wide3; tiles with more
// than 32 000 candidates go to the int-window kernel through the same list.
// V4 (round 6, JOIN only, UN8 even): the full chunks of the sorted stream are fetched 16 bytes per lane — two runs per load, a kilobyte per wave and
// instruction instead of 512 bytes; which thread works on which run changes, nothing else (the window's updates commute).
template <int WPE, int UN8, bool EXPORT, bool JOIN = false, bool V4 = false>
__global__ __launch_bounds__(WG, WPE) void k_direct_c8(const C8Sample cs, uint32_t n_tiles, ContigTab tab, const uint32_t *tile_contig,
                                                     uint32_t wrap_mask, const DirectWide args, uint32_t *heavy_list, uint32_t *heavy_count,
                                                     const DirectExport ex)
{
    const uint32_t w = args.w, min_dep = args.min_dep; TilePart *const part = args.part;
    constexpr uint32_t ST = TILE, HT = TILE / 2;
    constexpr int ROWS = (int)(HT / (WG * 4));
    __shared__ __attribute__((aligned(16))) unsigned win[HT];
    __shared__ int s_carry;
    __shared__ int wtot[4];
    __shared__ unsigned long long red_s[4][2];
    __shared__ int red_c[4][2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t bsh = cs.bshift;
    // the bounds of a tile are three entries of each stream's bucket starts (scalar loads); the sorted stream's are fetched a tile ahead,
    // the other stream's at the top of the tile (they are not needed before the sorted stream's runs are through)
    auto bounds = [&](const uint64_t t, uint32_t &slo, uint32_t &shi) {
        slo = shi = 0;
        if (t < n_tiles) {
            const uint32_t k0 = (uint32_t)t << bsh;
            shi = cs.b1[k0 + (1u << bsh)];
            const uint32_t ctg = tile_contig[t];
            slo = cs.b1[(uint64_t)t * ST > tab.off[ctg] ? k0 - 1 : k0];            // a contig's first tile has nothing before it
        }
    };
    uint32_t nslo, nshi;
    bounds(blockIdx.x, nslo, nshi);
    for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint64_t a = t * ST;
        const uint32_t ns = nshi - nslo, s_at = nslo;             // the tile's candidates in the sorted stream
        uint4 *w4 = reinterpret_cast<uint4 *>(win);
        for (uint32_t j = threadIdx.x; j < HT / 4; j += WG) w4[j] = make_uint4(0u, 0u, 0u, 0u);
        if (threadIdx.x == 0) s_carry = 0;
        const int32_t ctg = (int32_t)tile_contig[t];
        const uint32_t clen = tab.len[ctg];
        const uint32_t p0 = (uint32_t)a;                          // the runs' begins are flat (mod 2^32), like this
        const uint32_t pc = (uint32_t)(a - tab.off[ctg]);         // the tile's first cell inside its contig
        const uint32_t k0 = (uint32_t)t << bsh;
        // ... and in the other stream (both possible lower bounds are fetched, so that the loads do not wait for the contig's offset)
        const uint32_t o_prev = cs.o1[k0 ? k0 - 1 : 0], o_own = cs.o1[k0], ohi = cs.o1[k0 + (1u << bsh)];
        bounds(t + gridDim.x, nslo, nshi);
        __syncthreads();
        const uint32_t olo = pc ? o_prev : o_own, no = ohi - olo;
        const uint32_t cand = ns + no;
        if (cand > 32000u) {                                      // workgroup-uniform: the int-window kernel does this tile
            if (threadIdx.x == 0) heavy_list[atomicAdd(heavy_count, 1u)] = (uint32_t)t;
            __syncthreads();
            continue;
        }
        int carry_s = 0;
        {
            // ONE sequence of chunks: those of the sorted stream's candidates, then those of the other stream's — which stream a chunk
            // belongs to is a scalar decision (base index, position, end), so the double-buffered loop runs through both without a restart
            constexpr uint32_t C = UN8 * WG;
            const Run8 *__restrict__ const p = cs.r8;
            // JOIN: the sorted stream's full chunks, then its remainder and the other stream's runs as ONE sequence (a lane picks its array by
            // its index there) — four chunks for the typical tile's 2 730 + 300 candidates instead of four + one
            const uint32_t c1 = JOIN ? ns / C : (ns + C - 1) / C;
            const uint32_t rem = JOIN ? ns - c1 * C : 0u, tail = rem + no;
            const uint32_t nq = c1 + ((JOIN ? tail : no) + C - 1) / C;
            const uint32_t o_at = cs.o_base + olo;
            auto load8 = [&](uint2 (&dst)[UN8], const uint32_t q) {
                const bool second = q >= c1;
                if constexpr (JOIN) {
                    if (!second) {
                        if constexpr (V4) {
#pragma unroll
                            for (int k2 = 0; k2 < UN8 / 2; ++k2) {
                                const uint4 u = *reinterpret_cast<const uint4 *>(p + (s_at + q * C + 2u * (threadIdx.x + k2 * WG)));
                                dst[2 * k2] = make_uint2(u.x, u.y); dst[2 * k2 + 1] = make_uint2(u.z, u.w);
                            }
                        } else {
#pragma unroll
                        for (int k = 0; k < UN8; ++k) dst[k] = *reinterpret_cast<const uint2 *>(p + (s_at + q * C + threadIdx.x + k * WG));
                        }
                    } else {
                        const uint32_t i = (q - c1) * C, s_rem = s_at + c1 * C, o_rem = o_at - rem, last = tail - 1u;
#pragma unroll
                        for (int k = 0; k < UN8; ++k) {
                            uint32_t j = i + (V4 ? 2u * (threadIdx.x + (k >> 1) * WG) + (k & 1) : threadIdx.x + k * WG); j = j < last ? j : last;
                            dst[k] = *reinterpret_cast<const uint2 *>(p + ((j < rem ? s_rem : o_rem) + j));
                        }
                    }
                    return;
                }
                const uint32_t at = second ? o_at : s_at, i = (second ? q - c1 : q) * C, last = (second ? no : ns) - 1u;
#pragma unroll
                for (int k = 0; k < UN8; ++k) { const uint32_t j = i + threadIdx.x + k * WG; dst[k] = *reinterpret_cast<const uint2 *>(p + (at + (j < last ? j : last))); }
            };
            auto ev8 = [&](const uint32_t b, const uint32_t len) {
                const uint32_t sb = b - p0, se = sb + len;
                const unsigned long long m_c = __builtin_amdgcn_ballot_w64(se < len);
                if (sb < ST) atomicAdd(&win[sb & (HT - 1u)], 1u + (sb >> 12) * 0xFFFFu);
                if (se < ST) atomicSub(&win[se & (HT - 1u)], 1u + (se >> 12) * 0xFFFFu);
                carry_s += __builtin_popcountll(m_c);
            };
            auto work8 = [&](const uint2 (&c)[UN8], const uint32_t q) {
                const bool second = q >= c1;
                const uint32_t left = JOIN ? (second ? tail - (q - c1) * C : C) : (second ? no : ns) - (second ? q - c1 : q) * C;
                if (left >= C) {
#pragma unroll
                    for (int k = 0; k < UN8; ++k) ev8(c[k].x, c[k].y);
                } else {                                          // a stream's tail chunk: slots past the end become runs outside the tile
                    const int nu = V4 ? 2 * (int)((left + 2 * WG - 1) / (2 * WG)) : (int)((left + WG - 1) / WG);   // uniform, 1 .. UN8
#pragma unroll
                    for (int k = 0; k < UN8; ++k) if (k < nu) {
                        const bool in = (V4 ? 2u * (threadIdx.x + (k >> 1) * WG) + (k & 1) : threadIdx.x + k * WG) < left;
                        ev8(in ? c[k].x : p0 + ST, in ? c[k].y : 0u);
                    }
                }
            };
            if (nq) {
                uint2 A[UN8], B[UN8];
                uint32_t q = 0;
                load8(A, q);
#pragma unroll 1
                for (;;) {                                        // two buffers, no register copies: B is in flight while A is worked on
                    if (q + 1 < nq) load8(B, q + 1);
                    work8(A, q);
                    if (++q >= nq) break;
                    if (q + 1 < nq) load8(A, q + 1);
                    work8(B, q);
                    if (++q >= nq) break;
                }
            }
        }
        if (lane == 0 && carry_s != 0) atomicAdd(&s_carry, carry_s);
        __syncthreads();
        if constexpr (EXPORT) {
            export_packed_tile<ROWS>(win, a, t, ex, wtot);
            continue;
        }
        // ---- prefix sum of the packed window: both half-tiles at once ----
        int4 v[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const uint4 q = w4[wv * (ROWS * 64) + r * 64 + lane];
            v[r] = make_int4((int)q.x, (int)q.x + (int)q.y, 0, 0);
            v[r].z = v[r].y + (int)q.z; v[r].w = v[r].z + (int)q.w;
        }
        int ex[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) ex[r] = wave_incl_scan(v[r].w);
        int run = 0;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int e = run + ex[r] - v[r].w;                   // exclusive prefix of this lane's group in the wave
            run += __builtin_amdgcn_readlane(ex[r], 63);
            ex[r] = e;
        }
        if (lane == 0) wtot[wv] = run;
        __syncthreads();
        int basep = 0;                                            // packed: words of the waves before this one
        for (int k = 0; k < wv; ++k) basep += wtot[k];
        const int totp = wtot[0] + wtot[1] + wtot[2] + wtot[3];
        const int tl = (int)(short)(totp & 0xffff);               // begins - ends over the low half-tile
        const int carry_l = s_carry, carry_h = carry_l + tl;      // depth just before cell 0 / cell HT of the tile
        // ---- the tile's share of windows k0 and k0 + 1 ----
        const uint64_t local0 = pc;
        int c0 = 0, c1 = 0; unsigned long long s0 = 0, s1 = 0;
        bool uniform_counts = false;                              // c0 is a wave count (ballots), s0 a 32-bit lane sum
        if (local0 < clen) {
            const uint64_t k0 = local0 / w;
            const uint64_t nb64 = (k0 + 1) * (uint64_t)w - local0;   // tile-local start of window k0+1
            const uint32_t nb = nb64 > (uint64_t)ST ? ST : (uint32_t)nb64;
            const uint64_t left = (uint64_t)clen - local0;
            const uint32_t lim = left > (uint64_t)ST ? ST : (uint32_t)left;
            const uint64_t dmax = (uint64_t)(uint32_t)carry_l + cand;     // no depth in this tile exceeds carry + begins
            if (nb >= ST && lim >= ST && min_dep <= 1u && dmax < (1u << 27) && dmax <= wrap_mask) {
                // the common tile — inside one window, inside the contig, no wrap possible, threshold <= 1: the sum is the
                // sum of the local prefixes + 16 x (carry_l + carry_h) per lane; a cell is covered unless its prefix = -carry
                int sl = 0; int cnt = 0;
                const int zl = -carry_l, zh = -carry_h;
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    const int add = basep + ex[r];
                    const int pw[4] = {v[r].x + add, v[r].y + add, v[r].z + add, v[r].w + add};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int L = (int)(short)(pw[q] & 0xffff), H = (pw[q] - L) >> 16;
                        sl += L + H;
                        if (min_dep) cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(L != zl)) + __builtin_popcountll(__builtin_amdgcn_ballot_w64(H != zh));
                    }
                }
                c0 = min_dep ? cnt : (int)(ROWS * 8 * 64);
                s0 = (uint32_t)sl + (uint32_t)(ROWS * 4) * ((uint32_t)carry_l + (uint32_t)carry_h);
                uniform_counts = true;
            } else if (nb >= ST && lim >= ST && dmax < (1u << 27)) {
                uint32_t s32 = 0; int cnt = 0;
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    const int add = basep + ex[r];
                    const int pw[4] = {v[r].x + add, v[r].y + add, v[r].z + add, v[r].w + add};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int L = (int)(short)(pw[q] & 0xffff), H = (pw[q] - L) >> 16;
                        const uint32_t dl = (uint32_t)(L + carry_l) & wrap_mask, dh = (uint32_t)(H + carry_h) & wrap_mask;
                        const bool okl = dl >= min_dep, okh = dh >= min_dep;
                        cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(okl)) + __builtin_popcountll(__builtin_amdgcn_ballot_w64(okh));
                        s32 += (okl ? dl : 0u) + (okh ? dh : 0u);
                    }
                }
                c0 = cnt; s0 = s32; uniform_counts = true;
            } else {
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    const int add = basep + ex[r];
                    const uint32_t pos = (uint32_t)(wv * (ROWS * 256) + r * 256 + lane * 4);
                    const int pw[4] = {v[r].x + add, v[r].y + add, v[r].z + add, v[r].w + add};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int L = (int)(short)(pw[q] & 0xffff), H = (pw[q] - L) >> 16;
                        const uint32_t dl = (uint32_t)(L + carry_l) & wrap_mask, dh = (uint32_t)(H + carry_h) & wrap_mask;
                        const uint32_t pl = pos + q, ph = pl + HT;
                        if (pl < lim && dl >= min_dep) { if (pl < nb) { ++c0; s0 += dl; } else { ++c1; s1 += dl; } }
                        if (ph < lim && dh >= min_dep) { if (ph < nb) { ++c0; s0 += dh; } else { ++c1; s1 += dh; } }
                    }
                }
            }
        }
        if (uniform_counts) {                                     // workgroup-uniform
            const uint32_t s32 = (uint32_t)s0;
            const unsigned long long lo16 = (unsigned long long)(uint32_t)wave_total((int)(s32 & 0xffffu));
            const unsigned long long hi16 = (unsigned long long)(uint32_t)wave_total((int)(s32 >> 16));
            s0 = (hi16 << 16) + lo16;
        } else {
            c0 = wave_sum(c0); c1 = wave_sum(c1);
#pragma unroll
            for (int o = 32; o; o >>= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); }
        }
        if (lane == 0) { red_c[wv][0] = c0; red_c[wv][1] = c1; red_s[wv][0] = s0; red_s[wv][1] = s1; }
        __syncthreads();
        if (threadIdx.x == 0) {
            TilePart tp;
            tp.c0 = (uint32_t)(red_c[0][0] + red_c[1][0] + red_c[2][0] + red_c[3][0]);
            tp.c1 = (uint32_t)(red_c[0][1] + red_c[1][1] + red_c[2][1] + red_c[3][1]);
            tp.s0 = red_s[0][0] + red_s[1][0] + red_s[2][0] + red_s[3][0];
            tp.s1 = red_s[0][1] + red_s[1][1] + red_s[2][1] + red_s[3][1];
            part[t] = tp;
        }
    }
}

// Outcome of a direct pass: *fail = 1 unless every batch was complete and sorted and no run was
// longer than the look-back; re-arms the descriptors (the batches stay pending for the fallback).
__global__ void k_finish_direct(const PendSet ps, const uint32_t *n_long, uint32_t *fail)
{
    uint32_t bad = *n_long != 0, errs = 0;
    uint64_t handled = 0, has = 0, ends = 0, n = 0;
    for (int b = 0; b < ps.nb; ++b) {
        BatchDesc *desc = ps.b[b].desc;
        for (int k = 0; k < PD_CNT_SLOTS; ++k) {
            handled += desc->handled[k]; has += desc->has[k]; ends += desc->ends[k];
            desc->handled[k] = 0; desc->has[k] = 0; desc->ends[k] = 0;
        }
        n += ps.b[b].n;
        if (desc->err) bad = 1;
        errs |= desc->err;
        desc->ovf_count = 0; desc->err = 0; desc->t_first = 0; desc->n_active = 0;
    }
    if (handled != n || has != ends) bad = 1;
    *fail = bad;
    if (bad) {                                   // why (read by the host for its diagnostics): fail + 2 .. fail + 9
        const uint32_t err = errs;
        fail[2] = (uint32_t)handled; fail[3] = (uint32_t)(handled >> 32); fail[4] = (uint32_t)n; fail[5] = (uint32_t)(n >> 32);
        fail[6] = (uint32_t)has; fail[7] = (uint32_t)ends; fail[8] = err; fail[9] = *n_long;
    }
}

// Zero-fills every half-tile that has not been written since the last reset (and marks it), so
// that the atomic kernels may add into any cell.  With only_if_overflow it returns at once unless
// the tile pass just put something on the overflow list.
__global__ __launch_bounds__(WG) void k_fill_invalid(int4 *buf, uint8_t *hstate, uint32_t n_half,
                                                     const CheckWords *chk, int only_if_overflow)
{
    if (chk->all_valid) return;
    if (only_if_overflow && chk->ovf_count == 0) return;
    const int4 z = make_int4(0, 0, 0, 0);
    for (uint32_t h = blockIdx.x; h < n_half; h += gridDim.x) {
        const bool written = hstate[h] != 0;
        __syncthreads();                 // every wave has read the flag before thread 0 may set it
        if (written) continue;                                   // uniform per workgroup
        int4 *p = buf + (size_t)h * (PD_HALF / 4);
        for (int j = threadIdx.x; j < PD_HALF / 4; j += WG) p[j] = z;
        if (threadIdx.x == 0) hstate[h] = 1;
    }
}

__global__ __launch_bounds__(WG) void k_apply_overflow(const uint64_t *ovf, CheckWords *chk, uint32_t ovf_cap,
                                                       int *diff, int *sums, int mark_valid)
{
    uint32_t n = chk->ovf_count; if (n > ovf_cap) n = ovf_cap;
    for (uint32_t i = blockIdx.x * WG + threadIdx.x; i < n; i += gridDim.x * WG) {
        const uint64_t ge = ovf[i];
        atomicAdd(&diff[ge], -1);
        atomicAdd(&sums[ge / TILE], -1);
    }
    // the fill that preceded this kernel ran iff the list is not empty (or unconditionally)
    if (blockIdx.x == 0 && threadIdx.x == 0 && (n > 0 || mark_valid)) chk->all_valid = 1;
}

__global__ void k_mark_all_valid(CheckWords *chk) { chk->all_valid = 1; }

// Folds one batch's outcome into the context-wide check words and re-arms the descriptor.
__global__ void k_finish_batch(const PendSet ps, CheckWords *chk)
{
    for (int b = 0; b < ps.nb; ++b) {
        BatchDesc *desc = ps.b[b].desc;
        uint64_t handled = 0, has = 0, ends = 0;
        for (int k = 0; k < PD_CNT_SLOTS; ++k) {
            handled += desc->handled[k]; has += desc->has[k]; ends += desc->ends[k];
            desc->handled[k] = 0; desc->has[k] = 0; desc->ends[k] = 0;
        }
        if (handled != (uint64_t)ps.b[b].n || has != ends) chk->unsorted_batches += 1;
        if (desc->err) chk->err |= desc->err;
        desc->ovf_count = 0; desc->err = 0; desc->t_first = 0; desc->n_active = 0;
    }
    chk->ovf_count = 0;
}

// ------------------------------------------------------------------------------------------
// tile carries: exclusive prefix sum of the tile sums (a few hundred thousand ints), one
// workgroup of 1024 threads, chunked.
// ------------------------------------------------------------------------------------------
// Two small kernels: (1) one workgroup per 1024 tile sums reduces them to a block sum; (2) every
// workgroup re-derives its block's offset from the <= few hundred block sums and scans its block.
__global__ __launch_bounds__(1024) void k_carry_block_sums(const int *sums, int *bsum, uint32_t n_tiles)
{
    __shared__ int wsum[16];
    const uint32_t i = blockIdx.x * 1024 + threadIdx.x;
    int v = i < n_tiles ? sums[i] : 0;
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) { int t = 0; for (int k = 0; k < 16; ++k) t += wsum[k]; bsum[blockIdx.x] = t; }
}

__global__ __launch_bounds__(1024) void k_tile_carry(const int *sums, const int *bsum, int *carry, uint32_t n_tiles)
{
    __shared__ int wsum[16];
    __shared__ int s_base;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // offset of this block = sum of the block sums before it
    int part = 0;
    for (uint32_t k = threadIdx.x; k < blockIdx.x; k += 1024) part += bsum[k];
    part = wave_sum(part);
    if (lane == 0) wsum[wv] = part;
    __syncthreads();
    if (threadIdx.x == 0) { int t = 0; for (int k = 0; k < 16; ++k) t += wsum[k]; s_base = t; }
    __syncthreads();
    const uint32_t i = blockIdx.x * 1024 + threadIdx.x;
    const int v = i < n_tiles ? sums[i] : 0;
    const int x = wave_incl_scan(v);
    if (lane == 63) wsum[wv] = x;
    __syncthreads();
    int pre = s_base;
    for (int k = 0; k < wv; ++k) pre += wsum[k];
    if (i < n_tiles) carry[i] = pre + x - v;
}

// ------------------------------------------------------------------------------------------
// the sweep.  One workgroup per tile; 4 waves x 8 rows x (64 lanes x int4): every load/store is
// a full 1 KiB wave transaction.  Per row: in-lane scan of 4, wave scan of the 64 lane totals
// (all 8 rows' shuffle chains are independent -> ILP), running carry across rows, then the
// wave bases through LDS, plus the tile carry-in.
//   MODE_WRITE : depth written back in place (8 B/base)
//   MODE_WIN   : fused fixed-window reduction, nothing written back (4 B/base)
//   FROM_DEPTH : input already holds depth (reduction only)
// ------------------------------------------------------------------------------------------

typedef int v4i_nt __attribute__((ext_vector_type(4)));

template <bool WRITE, bool WIN, bool FROM_DEPTH>
__global__ __launch_bounds__(WG) void k_sweep(int *buf, const int *carry, uint32_t wrap_mask,
                                              const TileMap tmap, WinArgs wa, const uint8_t *hstate, uint32_t tile0)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int wtot[4];
    constexpr int ROWS = TILE / (WG * 4);                        // 8
    // tile0: the sweep of a slice (a rank's share of the summed depth in the sharded list mode): `buf` holds tiles tile0, tile0 + 1, ...
    const uint64_t t = (uint64_t)blockIdx.x + tile0;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int4 *p4 = reinterpret_cast<int4 *>(buf + (uint64_t)blockIdx.x * TILE) + wv * (ROWS * 64) + lane;
    int4 v[ROWS];
    // a half-tile nobody has written since the reset holds stale bytes and counts as zeros; waves
    // 0-1 cover the first 4096 cells of the tile, waves 2-3 the second (wave-uniform branch)
    const bool live = FROM_DEPTH || hstate[t * (TILE / PD_HALF) + (wv >> 1)] != 0;
    if (live) {
        // every cell is read once (and, written back, written once): non-temporal, so that the lines do not wait in L2 for a reuse that never comes
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const v4i_nt x = __builtin_nontemporal_load(reinterpret_cast<const v4i_nt *>(p4 + r * 64));
            v[r] = make_int4(x.x, x.y, x.z, x.w);
        }
    } else {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) v[r] = make_int4(0, 0, 0, 0);
    }

    if (!FROM_DEPTH) {
        int tot[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            v[r].y += v[r].x; v[r].z += v[r].y; v[r].w += v[r].z;
            tot[r] = v[r].w;
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) tot[r] = wave_incl_scan(tot[r]);
        int run = 0;                                             // carry across this wave's rows
        int excl[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            excl[r] = run + tot[r] - v[r].w;
            run += __builtin_amdgcn_readlane(tot[r], 63);
        }
        if (lane == 0) wtot[wv] = run;
        __syncthreads();
        int base = carry[t];
        for (int k = 0; k < wv; ++k) base += wtot[k];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int b = base + excl[r];
            v[r].x = (int)((uint32_t)(v[r].x + b) & wrap_mask);
            v[r].y = (int)((uint32_t)(v[r].y + b) & wrap_mask);
            v[r].z = (int)((uint32_t)(v[r].z + b) & wrap_mask);
            v[r].w = (int)((uint32_t)(v[r].w + b) & wrap_mask);
        }
    }
    if (WRITE) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            v4i_nt x; x.x = v[r].x; x.y = v[r].y; x.z = v[r].z; x.w = v[r].w;
            __builtin_nontemporal_store(x, reinterpret_cast<v4i_nt *>(p4 + r * 64));
        }
    }
    if (WIN) {
        // contig-local position of this tile's first cell and the contig's length
        const uint32_t ctg = tmap.tile_contig[t];
        const uint64_t local0 = t * TILE - tmap.contig_off[ctg];
        const uint32_t clen = tmap.contig_len[ctg];
        const uint32_t w = wa.w;
        if (w >= (uint32_t)TILE) {
            // Large windows (the 10 Mb bins of whole-chromosome mode): a tile touches at most two
            // windows, k0 and k0+1.  Write the tile's two partial (cover, sum) pairs with plain
            // stores; k_window_gather adds them up per window.  No atomics: with ~1200 tiles per
            // window, same-address device atomics would serialise the whole sweep.
            __shared__ unsigned long long red_s[4][2];
            __shared__ int red_c[4][2];
            int c0 = 0, c1 = 0; unsigned long long s0 = 0, s1 = 0;
            if (local0 < clen) {
                const uint64_t k0 = local0 / w;
                const uint64_t nb = (k0 + 1) * (uint64_t)w - local0;     // tile-local start of window k0+1
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    const uint32_t pos = (uint32_t)(wv * (ROWS * 256) + r * 256 + lane * 4);
                    const uint32_t d[4] = {(uint32_t)v[r].x, (uint32_t)v[r].y, (uint32_t)v[r].z, (uint32_t)v[r].w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const bool ok = local0 + pos + q < clen && d[q] >= wa.min_dep;
                        if (ok) { if (pos + q < nb) { ++c0; s0 += d[q]; } else { ++c1; s1 += d[q]; } }
                    }
                }
            }
            c0 = wave_sum(c0); c1 = wave_sum(c1);
#pragma unroll
            for (int o = 32; o; o >>= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); }
            if (lane == 0) { red_c[wv][0] = c0; red_c[wv][1] = c1; red_s[wv][0] = s0; red_s[wv][1] = s1; }
            __syncthreads();
            if (threadIdx.x == 0) {
                TilePart tp;
                tp.c0 = (uint32_t)(red_c[0][0] + red_c[1][0] + red_c[2][0] + red_c[3][0]);
                tp.c1 = (uint32_t)(red_c[0][1] + red_c[1][1] + red_c[2][1] + red_c[3][1]);
                tp.s0 = red_s[0][0] + red_s[1][0] + red_s[2][0] + red_s[3][0];
                tp.s1 = red_s[0][1] + red_s[1][1] + red_s[2][1] + red_s[3][1];
                wa.part[t] = tp;
            }
            return;
        }
        if (local0 >= clen) return;                              // pure padding tile (uniform)
        const uint64_t wbase = tmap.win_off[ctg];
        const uint64_t k0 = local0 / w;                          // first window touching the tile
        // general case: LDS accumulators for the windows overlapping this tile
        // one 64-bit accumulator per window: cover (<= 8192, 16 bits) in the top, depth sum
        // (<= 8192 * 2^32 = 2^45) in the low 48 bits -> a single ds_add_u64 per lane segment
        const uint32_t nacc = (uint32_t)((local0 + TILE - 1) / w - k0 + 1);
        unsigned long long *acc = reinterpret_cast<unsigned long long *>(smem);
        for (uint32_t j = threadIdx.x; j < nacc; j += WG) acc[j] = 0;
        __syncthreads();
        const uint32_t phase = (uint32_t)(local0 - k0 * w);      // offset of the tile inside window k0
        const uint32_t magic = w >= 2u ? narrow_magic(w) : 0u;
        const uint32_t left = (uint64_t)clen - local0 < (uint64_t)TILE ? (uint32_t)(clen - local0) : (uint32_t)TILE;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const uint32_t pos = (uint32_t)(wv * (ROWS * 256) + r * 256 + lane * 4);
            const uint32_t d[4] = {(uint32_t)v[r].x, (uint32_t)v[r].y, (uint32_t)v[r].z, (uint32_t)v[r].w};
            if (w >= 4u) { narrow_window_row(d, pos, phase, w, magic, wa.min_dep, left, acc, lane); continue; }
            // windows of 1-3 cells: several begin inside a lane's four cells; one atomic per window piece
            const uint32_t x = pos + phase;
            uint32_t q = (uint32_t)((float)x * wa.inv_w);
            if ((uint64_t)q * w > x) --q;
            if ((uint64_t)(q + 1) * w <= x) ++q;
            uint64_t nb = (uint64_t)(q + 1) * w - phase;         // tile-local cell where window q+1 starts
            uint32_t c = 0; unsigned long long s = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t pe = pos + e;
                if (pe == nb) {
                    if (c) atomicAdd(&acc[q], ((unsigned long long)c << 48) | s);
                    c = 0; s = 0; ++q; nb += w;
                }
                if (local0 + pe < clen && d[e] >= wa.min_dep) { ++c; s += d[e]; }
            }
            if (c) atomicAdd(&acc[q], ((unsigned long long)c << 48) | s);
        }
        __syncthreads();
        narrow_write_out(acc, nacc, k0, w, local0, clen, wbase, wa, wa.part + t);
    }
}

// w < TILE: the windows that lie across a tile boundary = the last share of the tile before it + the first share of the tile behind it
__global__ __launch_bounds__(WG) void k_window_edges(const TilePart *part, const TileMap tmap, uint32_t n_tiles, uint32_t w, uint32_t *cover,
                                                     unsigned long long *sum)
{
    const uint64_t t = (uint64_t)blockIdx.x * WG + threadIdx.x;
    if (t + 1 >= n_tiles) return;
    const uint32_t ctg = tmap.tile_contig[t];
    const uint64_t local0 = t * TILE - tmap.contig_off[ctg];
    const uint64_t clen = tmap.contig_len[ctg];
    if (local0 + TILE >= clen) return;                           // the contig's last tile (or padding): nothing goes on behind it
    const uint64_t k = (local0 + TILE - 1) / w;                  // the window of the tile's last cell
    if ((k + 1) * (uint64_t)w <= local0 + TILE) return;          // ends with the tile
    const TilePart a = part[t], b = part[t + 1];
    const uint32_t c = a.c1 + b.c0;
    if (!c) return;
    cover[tmap.win_off[ctg] + k] = c;
    sum[tmap.win_off[ctg] + k] = a.s1 + b.s0;
}

// w >= TILE: one wave per window adds up the partials of the tiles it spans (plain loads/stores).
__global__ __launch_bounds__(WG) void k_window_gather(const TilePart *part, const TileMap tmap, int32_t n_contigs,
                                                      uint32_t w, uint64_t n_windows, uint32_t *cover,
                                                      unsigned long long *sum)
{
    const uint64_t g = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= n_windows) return;
    const int lane = threadIdx.x & 63;
    int lo = 0, hi = n_contigs;                                  // contig c with win_off[c] <= g < win_off[c+1]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (tmap.win_off[mid] <= g) lo = mid; else hi = mid; }
    const uint64_t k = g - tmap.win_off[lo];
    const uint64_t coff = tmap.contig_off[lo];
    const uint64_t clen = tmap.contig_len[lo];
    const uint64_t b = k * w;
    uint64_t e = b + w; if (e > clen) e = clen;
    const uint64_t t0 = (coff + b) / TILE, t1 = (coff + e - 1) / TILE;
    uint32_t c = 0; unsigned long long s = 0;
    for (uint64_t t = t0 + lane; t <= t1; t += 64) {
        const TilePart tp = part[t];
        const uint64_t k0 = (t * TILE - coff) / w;
        if (k0 == k) { c += tp.c0; s += tp.s0; } else { c += tp.c1; s += tp.s1; }
    }
    c = (uint32_t)wave_sum((int)c);
#pragma unroll
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) { cover[g] = c; sum[g] = s; }
}

// ------------------------------------------------------------------------------------------
// segmented interval reduction over the depth array: one wave per piece (a region, or a
// <= 16384-cell slice of a long region); partial results are added into the region's slot.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WG) void k_reduce_pieces(const int *depth, const Piece *pieces, uint32_t n_pieces,
                                                      uint32_t min_dep, int *cover, unsigned long long *sum)
{
    const uint32_t pi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pi >= n_pieces) return;
    const int lane = threadIdx.x & 63;
    const Piece pc = pieces[pi];
    const uint32_t *d = reinterpret_cast<const uint32_t *>(depth) + pc.start;
    int c = 0; unsigned long long s = 0;
    for (uint32_t i = lane; i < pc.count; i += 64) {
        const uint32_t x = d[i];
        if (x >= min_dep) { ++c; s += x; }
    }
    c = wave_sum(c);
#pragma unroll
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0 && c) { atomicAdd(&cover[pc.region], c); atomicAdd(&sum[pc.region], s); }
}

// ------------------------------------------------------------------------------------------
// int8 transport of the difference arrays for the multi-sample sum (xGMI is the bottleneck there:
// 4x fewer bytes on the links).  Cells with |d| <= thr travel as int8 (thr = 127 / world, so the
// sum over all ranks cannot overflow); the rare others (pile-ups) travel as (cell, value)
// exceptions and are zero in the image.  Bytes are stored BIASED (d + thr, an unsigned value in
// [0, 2 thr]): the sum over the ranks stays below 256 in every byte, so the collective can add the
// image as int32 words (four cells per element, no carry between bytes) at full int32 reduce
// speed; the importer subtracts n_ranks * thr.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WG) void k_export_i8(const int *diff, const uint8_t *hstate, int4 *out, uint64_t n_cells,
                                                  int thr, pd_exc *exc, uint32_t cap, uint32_t *count)
{
    const uint64_t n16 = n_cells / 16;
    for (uint64_t i = blockIdx.x * (uint64_t)WG + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * WG) {
        const uint64_t c0 = i * 16;
        const unsigned zb = (unsigned)thr * 0x01010101u;          // four cells of value 0
        int4 o = make_int4((int)zb, (int)zb, (int)zb, (int)zb);
        if (hstate[c0 / PD_HALF]) {                              // never-written half-tiles are zeros
            const int4 *p = reinterpret_cast<const int4 *>(diff + c0);
            int v[16];
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int4 q = p[k]; v[4 * k] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w; }
            unsigned w[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                int x = v[k];
                if (x > thr || x < -thr) {
                    const uint32_t slot = atomicAdd(count, 1u);
                    if (slot < cap) { exc[slot].cell = c0 + k; exc[slot].value = x; exc[slot].pad = 0; }
                    x = 0;
                }
                w[k >> 2] |= (unsigned)((x + thr) & 0xff) << (8 * (k & 3));
            }
            o = make_int4((int)w[0], (int)w[1], (int)w[2], (int)w[3]);
        }
        out[i] = o;
    }
}

__global__ __launch_bounds__(WG) void k_import_i8(const int4 *in, int *diff, uint64_t n_cells, int bias)
{
    const uint64_t n16 = n_cells / 16;
    for (uint64_t i = blockIdx.x * (uint64_t)WG + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * WG) {
        const int4 q = in[i];
        const unsigned w[4] = {(unsigned)q.x, (unsigned)q.y, (unsigned)q.z, (unsigned)q.w};
        int4 *p = reinterpret_cast<int4 *>(diff + i * 16);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            p[k] = make_int4((int)(w[k] & 0xff) - bias, (int)((w[k] >> 8) & 0xff) - bias,
                             (int)((w[k] >> 16) & 0xff) - bias, (int)(w[k] >> 24) - bias);
    }
}

__global__ __launch_bounds__(WG) void k_apply_exceptions(const pd_exc *exc, uint64_t n, int *diff, uint64_t n_cells)
{
    for (uint64_t i = blockIdx.x * (uint64_t)WG + threadIdx.x; i < n; i += (uint64_t)gridDim.x * WG)
        if (exc[i].cell < n_cells) atomicAdd(&diff[exc[i].cell], exc[i].value);
}

// ------------------------------------------------------------------------------------------
// 4-bit transport for the SLICED sum: the images are exchanged pairwise (all-to-all) and added on
// the receiving GPU in int32, so a nibble (d + 8) per cell is enough whatever the number of ranks.
// ------------------------------------------------------------------------------------------
// one workgroup per tile and step; every load is a full 1 KiB wave transaction (lane-contiguous
// int4), every store 128 contiguous bytes (one ushort = 4 cells per lane)
__global__ __launch_bounds__(WG) void k_export_i4(const int *diff, const uint8_t *hstate, unsigned short *out,
                                                       uint32_t n_tiles, pd_exc *exc, uint32_t cap, uint32_t *count)
{
    constexpr int ROWS = TILE / (WG * 4);                        // 8
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint64_t cw = (uint64_t)t * TILE + (uint64_t)wv * (ROWS * 256);       // first cell of this wave
        unsigned short *o = out + cw / 4 + lane;
        if (!hstate[(uint64_t)t * (TILE / PD_HALF) + (wv >> 1)]) {                   // wave-uniform
#pragma unroll
            for (int r = 0; r < ROWS; ++r) o[r * 64] = 0x8888;
            continue;
        }
        const int4 *p4 = reinterpret_cast<const int4 *>(diff + cw) + lane;
        int4 v[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) v[r] = p4[r * 64];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            int x[4] = {v[r].x, v[r].y, v[r].z, v[r].w};
            unsigned w = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (x[k] > 7 || x[k] < -8) {
                    const uint32_t slot = atomicAdd(count, 1u);
                    if (slot < cap) { exc[slot].cell = cw + r * 256 + lane * 4 + k; exc[slot].value = x[k]; exc[slot].pad = 0; }
                    x[k] = 0;
                }
                w |= (unsigned)((x[k] + 8) & 0xf) << (4 * k);
            }
            o[r * 64] = (unsigned short)w;
        }
    }
}

// dst += (nibble - 8) for the tiles [0, n_tiles) of a chunk: the single-process form of the sum
// (pd_accumulate_from).  Same row layout as k_export_i4: lane-contiguous int4 RMW of the cells, one
// ushort (4 cells) of the image per lane and row.
__global__ __launch_bounds__(WG) void k_add_i4(int *dst, const unsigned short *img, uint32_t n_tiles)
{
    constexpr int ROWS = TILE / (WG * 4);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint64_t cw = (uint64_t)t * TILE + (uint64_t)wv * (ROWS * 256);
        const unsigned short *q = img + cw / 4 + lane;
        int4 *p4 = reinterpret_cast<int4 *>(dst + cw) + lane;
        unsigned short h[ROWS];
        int4 v[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) { h[r] = q[r * 64]; v[r] = p4[r * 64]; }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const unsigned w = h[r];
            v[r].x += (int)(w & 0xf) - 8; v[r].y += (int)((w >> 4) & 0xf) - 8;
            v[r].z += (int)((w >> 8) & 0xf) - 8; v[r].w += (int)(w >> 12) - 8;
            p4[r * 64] = v[r];
        }
    }
}

// marks the tiles of the slice [tile0, tile0 + n_tiles) that own an exception of any part (blockIdx.y)
__global__ __launch_bounds__(WG) void k_flag_exception_tiles(const pd_exc *exc, uint64_t exc_stride, const int32_t *counts,
                                                             uint64_t tile0, uint64_t n_tiles, uint8_t *flags)
{
    const uint32_t j = blockIdx.y;
    uint64_t n = counts[j] < 0 ? 0 : (uint64_t)counts[j];
    if (n > exc_stride) n = exc_stride;
    const pd_exc *e = exc + (uint64_t)j * exc_stride;
    for (uint64_t i = blockIdx.x * (uint64_t)WG + threadIdx.x; i < n; i += (uint64_t)gridDim.x * WG) {
        const uint64_t t = e[i].cell / TILE;
        if (t >= tile0 && t - tile0 < n_tiles) flags[t - tile0] = 1;
    }
}

// The receiving side of the sliced sum, fused: one workgroup per tile of the slice reads the tile's
// 4 KiB of every part (one 16-byte load per lane and part = the lane's 32 consecutive cells), adds
// them in registers, prefix-sums (32 cells in the lane, one wave scan of the lane totals, wave bases
// through LDS, tile carry), wraps, and reduces the tile's share of the (at most two) windows of
// w >= TILE cells it touches.  No int32 copy of the summed arrays ever exists.  Tiles that own
// exceptions (rare: pile-ups) patch them in through LDS.
struct I4Src {
    const uint8_t *parts; uint64_t stride; uint32_t n_parts;
    const uint8_t *flags;                     // per tile of the slice, or null
    const pd_exc *exc; uint64_t exc_stride; const int32_t *counts;
};

// PATCH = false: the tiles without exceptions (nearly all) — no 32 KB patch window in LDS, so eight workgroups fit a CU instead of
// five (the kernel waits for its one load per lane and part: residency is its throughput); PATCH = true: only the flagged tiles.
template <bool PATCH>
__global__ __launch_bounds__(WG) void k_sweep_i4(const I4Src src, const int *carry, uint32_t wrap_mask, const TileMap tmap,
                                                 uint32_t w, uint32_t min_dep, TilePart *part, uint32_t tile0, const uint32_t *list,
                                                 const uint32_t *n_list, int *depth_out)
{
    __shared__ int wtot[4];
    __shared__ int patch[PATCH ? TILE : 1];
    __shared__ unsigned long long red_s[4][2];
    __shared__ int red_c[4][2];
    // PATCH: a few workgroups walk the list of the tiles that own exceptions (k_list_flagged); else one workgroup per tile
    const uint32_t n_it = PATCH ? *n_list : blockIdx.x + 1;
    for (uint32_t it = blockIdx.x; it < n_it; it += gridDim.x) {
    const uint32_t i = PATCH ? list[it] : blockIdx.x;
    if (!PATCH && src.flags && src.flags[i] != 0) return;        // workgroup-uniform
    const uint64_t t = (uint64_t)i + tile0;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int a[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) a[k] = -8 * (int)src.n_parts;
    const uint8_t *p = src.parts + (uint64_t)i * (TILE / 2) + (uint64_t)tid * 16;
    // the parts are added as packed bytes (even / odd nibbles of each word widened to bytes: up to
    // 16 parts fit, 15 * 16 < 256) and only then spread into the 32 int accumulators
    for (uint32_t j0 = 0; j0 < src.n_parts; j0 += 16) {
        unsigned lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
        const uint32_t j1 = j0 + 16 < src.n_parts ? j0 + 16 : src.n_parts;
        for (uint32_t j = j0; j < j1; ++j) {
            const uint4 q = *reinterpret_cast<const uint4 *>(p + (uint64_t)j * src.stride);
            const unsigned wd[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int m = 0; m < 4; ++m) { lo[m] += wd[m] & 0x0F0F0F0Fu; hi[m] += (wd[m] >> 4) & 0x0F0F0F0Fu; }
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                a[8 * m + 2 * b] += (int)((lo[m] >> (8 * b)) & 0xff);
                a[8 * m + 2 * b + 1] += (int)((hi[m] >> (8 * b)) & 0xff);
            }
    }
    if constexpr (PATCH) {
        for (int k = tid; k < TILE; k += WG) patch[k] = 0;
        __syncthreads();
        for (uint32_t j = 0; j < src.n_parts; ++j) {
            uint64_t n = src.counts[j] < 0 ? 0 : (uint64_t)src.counts[j];
            if (n > src.exc_stride) n = src.exc_stride;
            const pd_exc *e = src.exc + (uint64_t)j * src.exc_stride;
            for (uint64_t x = tid; x < n; x += WG)
                if (e[x].cell / TILE == t) atomicAdd(&patch[e[x].cell % TILE], e[x].value);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 32; ++k) a[k] += patch[tid * 32 + k];
    }
    int run = 0;
#pragma unroll
    for (int k = 0; k < 32; ++k) { run += a[k]; a[k] = run; }
    const int incl = wave_incl_scan(run);
    if (lane == 63) wtot[wv] = incl;
    __syncthreads();
    int base = carry[t] + incl - run;
    for (int k = 0; k < wv; ++k) base += wtot[k];

    if (depth_out) {                                              // (uniform) the slice's depth instead of statistics
#pragma unroll
        for (int k = 0; k < 32; ++k) depth_out[(uint64_t)i * TILE + (uint32_t)(tid * 32 + k)] = (int)((uint32_t)(a[k] + base) & wrap_mask);
        __syncthreads();
        continue;
    }
    const uint32_t ctg = tmap.tile_contig[t];
    const uint64_t local0 = t * TILE - tmap.contig_off[ctg];
    const uint32_t clen = tmap.contig_len[ctg];
    int c0 = 0, c1 = 0; unsigned long long s0 = 0, s1 = 0;
    if (local0 < clen) {
        const uint64_t k0 = local0 / w;
        const uint64_t nb = (k0 + 1) * (uint64_t)w - local0;     // tile-local start of window k0+1
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const uint32_t pos = (uint32_t)(tid * 32 + k);
            const uint32_t d = (uint32_t)(a[k] + base) & wrap_mask;
            if (local0 + pos < clen && d >= min_dep) { if (pos < nb) { ++c0; s0 += d; } else { ++c1; s1 += d; } }
        }
    }
    c0 = wave_sum(c0); c1 = wave_sum(c1);
#pragma unroll
    for (int o = 32; o; o >>= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); }
    if (lane == 0) { red_c[wv][0] = c0; red_c[wv][1] = c1; red_s[wv][0] = s0; red_s[wv][1] = s1; }
    __syncthreads();
    if (tid == 0) {
        TilePart tp;
        tp.c0 = (uint32_t)(red_c[0][0] + red_c[1][0] + red_c[2][0] + red_c[3][0]);
        tp.c1 = (uint32_t)(red_c[0][1] + red_c[1][1] + red_c[2][1] + red_c[3][1]);
        tp.s0 = red_s[0][0] + red_s[1][0] + red_s[2][0] + red_s[3][0];
        tp.s1 = red_s[0][1] + red_s[1][1] + red_s[2][1] + red_s[3][1];
        part[i] = tp;
    }
    __syncthreads();
    }
}

// the tiles of the slice whose flag is set, as a list
__global__ __launch_bounds__(WG) void k_list_flagged(const uint8_t *flags, uint32_t n_tiles, uint32_t *list, uint32_t *count)
{
    for (uint32_t i = blockIdx.x * WG + threadIdx.x; i < n_tiles; i += gridDim.x * WG)
        if (flags[i]) list[atomicAdd(count, 1u)] = i;
}

// The same sweep with ONE WAVE PER TILE (tiles without exceptions, at most 16 parts): a lane owns 4 x 32 consecutive cells (the
// 32 cells at lane * 32 of each of the tile's four 2048-cell quarters = one 16-byte load per quarter and part), the quarters are
// swept in order with the running depth carried in a register — no barriers, no LDS, and four loads per part in flight per lane,
// where the workgroup-per-tile form lived for one load's latency and three barriers per 4 KiB of image.
// DEPTH: no statistics — the summed, prefix-summed and wrapped cells of the slice are written out as int32 depth (depth_out: the slice's
// first cell), for the statistics that need the cells themselves (narrow windows, annotation intervals) on the rank that owns the slice.
template <bool DEPTH>
__device__ __forceinline__ void sweep_i4_tile_wave(const I4Src &src, const int *carry, uint32_t wrap_mask, const TileMap &tmap,
                                                   uint32_t w, uint32_t min_dep, TilePart *part, uint32_t tile0, uint32_t i, int *depth_out)
{
    const int lane = threadIdx.x & 63;
    if (src.flags && src.flags[i] != 0) return;                  // a tile with exceptions: k_sweep_i4<true>
    const uint64_t t = (uint64_t)i + tile0;
    const uint8_t *p = src.parts + (uint64_t)i * (TILE / 2) + (uint64_t)lane * 16;
    // the parts are added as packed bytes (even / odd nibbles of each word widened to bytes: 16 parts fit, 15 * 16 < 256)
    unsigned lo[4][4], hi[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int m = 0; m < 4; ++m) { lo[q][m] = 0; hi[q][m] = 0; }
    for (uint32_t j = 0; j < src.n_parts; ++j) {
        uint4 x[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) x[q] = *reinterpret_cast<const uint4 *>(p + (uint64_t)j * src.stride + q * 1024);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned wd[4] = {x[q].x, x[q].y, x[q].z, x[q].w};
#pragma unroll
            for (int m = 0; m < 4; ++m) { lo[q][m] += wd[m] & 0x0F0F0F0Fu; hi[q][m] += (wd[m] >> 4) & 0x0F0F0F0Fu; }
        }
    }
    const uint32_t ctg = tmap.tile_contig[t];
    const uint64_t local0 = t * TILE - tmap.contig_off[ctg];
    const uint32_t clen = tmap.contig_len[ctg];
    int base = carry[t];                                          // depth just before the quarter's first cell
    int c0 = 0, c1 = 0; unsigned long long s0 = 0, s1 = 0;
    const bool any = local0 < clen;
    const uint64_t k0 = any ? local0 / w : 0;
    const uint64_t nb64 = (k0 + 1) * (uint64_t)w - local0;        // tile-local start of window k0 + 1
    const uint32_t nb = nb64 < (uint64_t)TILE ? (uint32_t)nb64 : (uint32_t)TILE;
    const uint32_t left = !any ? 0u : ((uint64_t)clen - local0 < (uint64_t)TILE ? (uint32_t)(clen - local0) : (uint32_t)TILE);   // cells of the contig in the tile
    const int bias = -8 * (int)src.n_parts;
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    const uint32_t bN = 8u * src.n_parts;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t pos0 = (uint32_t)(q * 2048 + lane * 32);
        if constexpr (!DEPTH) {
            // The common quarter — every lane's 32 cells on one side of the window boundary and inside the contig, threshold <= 1, depths
            // below 2^16 (so neither the 18-bit wrap nor a 16-bit counter can bite) — never spreads the packed nibble sums into 32 ints:
            //   lane total and TotalDepth are dot products of the packed bytes (v_dot4_u32_u8: sum of v, and sum of (32 - c) v_c),
            //   CoveredSite counts the cells whose depth is not zero with packed 16-bit arithmetic, cells c and c + 16 side by side:
            //   prefix of v (v_pk_add), minus the value it has where the depth is zero (v_pk_sub), min 1 (v_pk_min), summed.
            // About 140 vector instructions per lane and quarter instead of 400.
            const uint32_t ones = 0x01010101u;
            const uint32_t t16 = __builtin_amdgcn_udot4(lo[q][0], ones, 0u, false) + __builtin_amdgcn_udot4(lo[q][1], ones, 0u, false) +
                                 __builtin_amdgcn_udot4(hi[q][0], ones, 0u, false) + __builtin_amdgcn_udot4(hi[q][1], ones, 0u, false);
            const uint32_t t32 = t16 + __builtin_amdgcn_udot4(lo[q][2], ones, 0u, false) + __builtin_amdgcn_udot4(lo[q][3], ones, 0u, false) +
                                 __builtin_amdgcn_udot4(hi[q][2], ones, 0u, false) + __builtin_amdgcn_udot4(hi[q][3], ones, 0u, false);
            const int run = (int)t32 - (int)(32u * bN);
            const int incl = wave_incl_scan(run);
            const int b0 = base + incl - run;
            const bool plain = pos0 + 32u <= left && (pos0 + 32u <= nb || pos0 >= nb);
            const bool small = (uint32_t)b0 < 60000u;             // (+ at most 32 x 7 x 16 inside the lane: below 2^16, and below any wrap)
            if (__builtin_expect(min_dep <= 1u && wrap_mask >= 0xFFFFu && __ballot(!plain || !small) == 0ull, 1)) {
                uint32_t w = 0;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const uint32_t wl = (32u - (8u * m)) | ((30u - 8u * m) << 8) | ((28u - 8u * m) << 16) | ((26u - 8u * m) << 24);
                    const uint32_t wh = (31u - (8u * m)) | ((29u - 8u * m) << 8) | ((27u - 8u * m) << 16) | ((25u - 8u * m) << 24);
                    w = __builtin_amdgcn_udot4(lo[q][m], wl, w, false);
                    w = __builtin_amdgcn_udot4(hi[q][m], wh, w, false);
                }
                const uint32_t sum32 = 32u * (uint32_t)b0 + w - bN * 528u;
                uint32_t cnt = 32u;
                if (min_dep) {
                    const uint32_t tl = (bN - (uint32_t)b0) & 0xffffu, th = (bN - ((uint32_t)b0 + t16 - 16u * bN)) & 0xffffu;
                    us2 tgt = __builtin_bit_cast(us2, tl | (th << 16)), r = __builtin_bit_cast(us2, 0u), acc = __builtin_bit_cast(us2, 0u);
                    const us2 step = __builtin_bit_cast(us2, bN | (bN << 16)), one2 = __builtin_bit_cast(us2, 0x00010001u);
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const int m = k >> 3, b = (k & 7) >> 1;
                        const uint32_t sel = (uint32_t)b | (0x0cu << 8) | ((4u + (uint32_t)b) << 16) | (0x0cu << 24);
                        const uint32_t pair = (k & 1) ? __builtin_amdgcn_perm(hi[q][m + 2], hi[q][m], sel) : __builtin_amdgcn_perm(lo[q][m + 2], lo[q][m], sel);
                        r += __builtin_bit_cast(us2, pair);
                        acc += __builtin_elementwise_min((us2)(r - tgt), one2);
                        tgt += step;
                    }
                    const uint32_t a32 = __builtin_bit_cast(uint32_t, acc);
                    cnt = (a32 & 0xffffu) + (a32 >> 16);
                }
                const bool first = pos0 < nb;
                c0 += first ? (int)cnt : 0; s0 += first ? (unsigned long long)sum32 : 0ull;
                c1 += first ? 0 : (int)cnt; s1 += first ? 0ull : (unsigned long long)sum32;
                base += __builtin_amdgcn_readlane(incl, 63);
                continue;
            }
        }
        int a[32];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                a[8 * m + 2 * b] = bias + (int)((lo[q][m] >> (8 * b)) & 0xff);
                a[8 * m + 2 * b + 1] = bias + (int)((hi[q][m] >> (8 * b)) & 0xff);
            }
        int run = 0;
#pragma unroll
        for (int k = 0; k < 32; ++k) { run += a[k]; a[k] = run; }
        const int incl = wave_incl_scan(run);
        const int b0 = base + incl - run;
        if constexpr (DEPTH) {
            int4 *o = reinterpret_cast<int4 *>(depth_out + (uint64_t)i * TILE + pos0);
#pragma unroll
            for (int k = 0; k < 32; k += 4)
                o[k >> 2] = make_int4((int)((uint32_t)(a[k] + b0) & wrap_mask), (int)((uint32_t)(a[k + 1] + b0) & wrap_mask),
                                      (int)((uint32_t)(a[k + 2] + b0) & wrap_mask), (int)((uint32_t)(a[k + 3] + b0) & wrap_mask));
            base += __builtin_amdgcn_readlane(incl, 63);
            continue;
        }
        // a lane's 32 cells lie on one side of the window boundary and inside the contig, except in the one quarter of a tile that
        // holds the boundary or the contig's end: there every cell is tested (wave-uniform branch), elsewhere none is
        const bool plain = pos0 + 32u <= left && (pos0 + 32u <= nb || pos0 >= nb);
        if (__builtin_expect(__ballot(!plain) == 0ull, 1)) {
            uint32_t cnt = 0; unsigned long long sm = 0;
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const uint32_t d = (uint32_t)(a[k] + b0) & wrap_mask;
                const bool ok = d >= min_dep;
                cnt += ok ? 1u : 0u; sm += ok ? d : 0u;
            }
            const bool first = pos0 < nb;
            c0 += first ? (int)cnt : 0; s0 += first ? sm : 0ull;
            c1 += first ? 0 : (int)cnt; s1 += first ? 0ull : sm;
        } else {
#pragma unroll 4
            for (int k = 0; k < 32; ++k) {
                const uint32_t pos = pos0 + k;
                const uint32_t d = (uint32_t)(a[k] + b0) & wrap_mask;
                if (pos < left && d >= min_dep) { if (pos < nb) { ++c0; s0 += d; } else { ++c1; s1 += d; } }
            }
        }
        base += __builtin_amdgcn_readlane(incl, 63);
    }
    c0 = wave_sum(c0); c1 = wave_sum(c1);
#pragma unroll
    for (int o = 32; o; o >>= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); }
    if (DEPTH) return;
    if (lane == 0) { TilePart tp; tp.c0 = (uint32_t)c0; tp.c1 = (uint32_t)c1; tp.s0 = s0; tp.s1 = s1; part[i] = tp; }
}

// one wave per tile of the slice, or (only != null) the waves of a fixed grid over a list of tiles (the ones k_sweep_i4_fast left)
template <bool DEPTH>
__global__ __launch_bounds__(WG) void k_sweep_i4_wave(const I4Src src, const int *carry, uint32_t wrap_mask, const TileMap tmap,
                                                      uint32_t w, uint32_t min_dep, TilePart *part, uint32_t tile0, uint32_t tile_count, int *depth_out,
                                                      const uint32_t *only, const uint32_t *n_only)
{
    const uint32_t wave = blockIdx.x * (WG / 64) + (threadIdx.x >> 6);
    if (!only) {
        if (wave < tile_count) sweep_i4_tile_wave<DEPTH>(src, carry, wrap_mask, tmap, w, min_dep, part, tile0, wave, depth_out);
        return;
    }
    const uint32_t n = *n_only;
    for (uint32_t k = wave; k < n; k += gridDim.x * (WG / 64)) sweep_i4_tile_wave<DEPTH>(src, carry, wrap_mask, tmap, w, min_dep, part, tile0, only[k], depth_out);
}

// The statistics sweep of the tiles that are nothing but common quarters — inside one window, inside the contig, no exceptions, threshold <= 1,
// depths below 2^16 — in a kernel of their own: the packed path of sweep_i4_tile_wave alone, taken quarter by quarter, needs a third of
// its registers, so eight waves share a SIMD instead of four.  A tile that turns out not to qualify
// (a window boundary or a contig's end inside it, a depth of 60 000 or more at some lane's first cell) goes to `slow`, the list
// k_sweep_i4_wave works through afterwards.
__global__ __launch_bounds__(WG, 8) void k_sweep_i4_fast(const I4Src src, const int *carry, const TileMap tmap, uint32_t w, uint32_t min_dep, TilePart *part,
                                                        uint32_t tile0, uint32_t tile_count, uint32_t *slow, uint32_t *n_slow)
{
    const int lane = threadIdx.x & 63;
    const uint32_t i = blockIdx.x * (WG / 64) + (threadIdx.x >> 6);
    if (i >= tile_count) return;
    if (src.flags && src.flags[i] != 0) return;                  // a tile with exceptions: k_sweep_i4<true>
    const uint64_t t = (uint64_t)i + tile0;
    const uint32_t ctg = tmap.tile_contig[t];
    const uint64_t local0 = t * TILE - tmap.contig_off[ctg];
    const uint32_t clen = tmap.contig_len[ctg];
    bool fast = local0 < clen && (uint64_t)clen - local0 >= (uint64_t)TILE;
    if (fast) { const uint64_t k0 = local0 / w; fast = (k0 + 1) * (uint64_t)w - local0 >= (uint64_t)TILE; }
    if (!fast) { if (lane == 0) slow[atomicAdd(n_slow, 1u)] = i; return; }
    const uint8_t *p = src.parts + (uint64_t)i * (TILE / 2) + (uint64_t)lane * 16;
    int base = carry[t];
    uint32_t c0 = 0; unsigned long long s0 = 0;
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    const uint32_t bN = 8u * src.n_parts, ones = 0x01010101u, bN4 = bN * 0x01010101u;
    // quarter by quarter (the running depth makes them sequential anyway), one 16-byte load per part with the next one — the next part's,
    // or the next quarter's first — in flight: eight accumulators instead of thirty-two
    uint4 x = *reinterpret_cast<const uint4 *>(p);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned lo[4] = {0u, 0u, 0u, 0u}, hi[4] = {0u, 0u, 0u, 0u};
        for (uint32_t j = 0; j < src.n_parts; ++j) {
            const bool more = j + 1 < src.n_parts;
            uint4 nx = x;
            if (more) nx = *reinterpret_cast<const uint4 *>(p + (uint64_t)(j + 1) * src.stride + q * 1024);
            else if (q < 3) nx = *reinterpret_cast<const uint4 *>(p + (q + 1) * 1024);
            const unsigned wd[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int m = 0; m < 4; ++m) { lo[m] += wd[m] & 0x0F0F0F0Fu; hi[m] += (wd[m] >> 4) & 0x0F0F0F0Fu; }
            x = nx;
        }
        // (the arithmetic of sweep_i4_tile_wave's common quarter, see there)
        const uint32_t t32 = __builtin_amdgcn_udot4(lo[0], ones, 0u, false) + __builtin_amdgcn_udot4(lo[1], ones, 0u, false) +
                             __builtin_amdgcn_udot4(hi[0], ones, 0u, false) + __builtin_amdgcn_udot4(hi[1], ones, 0u, false) +
                             __builtin_amdgcn_udot4(lo[2], ones, 0u, false) + __builtin_amdgcn_udot4(lo[3], ones, 0u, false) +
                             __builtin_amdgcn_udot4(hi[2], ones, 0u, false) + __builtin_amdgcn_udot4(hi[3], ones, 0u, false);
        const int run = (int)t32 - (int)(32u * bN);
        const int incl = wave_incl_scan(run);
        const int b0 = base + incl - run;
        if (__ballot((uint32_t)b0 >= 60000u) != 0ull) { if (lane == 0) slow[atomicAdd(n_slow, 1u)] = i; return; }
        uint32_t wsum = 0;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const uint32_t wl = (32u - (8u * m)) | ((30u - 8u * m) << 8) | ((28u - 8u * m) << 16) | ((26u - 8u * m) << 24);
            const uint32_t wh = (31u - (8u * m)) | ((29u - 8u * m) << 8) | ((27u - 8u * m) << 16) | ((25u - 8u * m) << 24);
            wsum = __builtin_amdgcn_udot4(lo[m], wl, wsum, false);
            wsum = __builtin_amdgcn_udot4(hi[m], wh, wsum, false);
        }
        s0 += 32u * (uint32_t)b0 + wsum - bN * 528u;
        uint32_t cnt = 32u;
        if (min_dep) {
            // CoveredSite.  A depth is never negative, so a cell is uncovered only where the running depth has come DOWN to zero: if the depth
            // before the lane's first cell exceeds the sum of the NEGATIVE differences among its 32 cells, no cell of the lane can be zero
            // and all 32 count — eight v_sad_u8 (sum |v_c| = positives + negatives; their difference is `run`) instead of the 96
            // instructions of the packed prefix walk below, which only the wave-quarters with a lane near depth zero still take
            // (at 50x about one in 10^4; round 5's kernel walked every quarter: 0.27 of the HBM roofline, instruction-bound).
            uint32_t sad = 0;
#pragma unroll
            for (int m = 0; m < 4; ++m) { sad = __builtin_amdgcn_sad_u8(lo[m], bN4, sad); sad = __builtin_amdgcn_sad_u8(hi[m], bN4, sad); }
            const int neg = ((int)sad - run) >> 1;
            if (__builtin_expect(__ballot(b0 <= neg) != 0ull, 0)) {
                const uint32_t t16 = __builtin_amdgcn_udot4(lo[0], ones, 0u, false) + __builtin_amdgcn_udot4(lo[1], ones, 0u, false) +
                                     __builtin_amdgcn_udot4(hi[0], ones, 0u, false) + __builtin_amdgcn_udot4(hi[1], ones, 0u, false);
                const uint32_t tl = (bN - (uint32_t)b0) & 0xffffu, th = (bN - ((uint32_t)b0 + t16 - 16u * bN)) & 0xffffu;
                us2 tgt = __builtin_bit_cast(us2, tl | (th << 16)), r = __builtin_bit_cast(us2, 0u), acc = __builtin_bit_cast(us2, 0u);
                const us2 step = __builtin_bit_cast(us2, bN | (bN << 16)), one2 = __builtin_bit_cast(us2, 0x00010001u);
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int m = k >> 3, b = (k & 7) >> 1;
                    const uint32_t sel = (uint32_t)b | (0x0cu << 8) | ((4u + (uint32_t)b) << 16) | (0x0cu << 24);
                    const uint32_t pair = (k & 1) ? __builtin_amdgcn_perm(hi[m + 2], hi[m], sel) : __builtin_amdgcn_perm(lo[m + 2], lo[m], sel);
                    r += __builtin_bit_cast(us2, pair);
                    acc += __builtin_elementwise_min((us2)(r - tgt), one2);
                    tgt += step;
                }
                const uint32_t a32 = __builtin_bit_cast(uint32_t, acc);
                cnt = (a32 & 0xffffu) + (a32 >> 16);
            }
        }
        c0 += cnt;
        base += __builtin_amdgcn_readlane(incl, 63);
    }
    c0 = (uint32_t)wave_sum((int)c0);
#pragma unroll
    for (int o = 32; o; o >>= 1) s0 += __shfl_xor(s0, o);
    if (lane == 0) { TilePart tp; tp.c0 = c0; tp.c1 = 0; tp.s0 = s0; tp.s1 = 0; part[i] = tp; }
}

__global__ __launch_bounds__(WG) void k_add_i32(int4 *dst, const int4 *src, size_t n16)
{
    size_t i = blockIdx.x * (size_t)WG + threadIdx.x;
    const size_t st = (size_t)gridDim.x * WG;
    for (; i < n16; i += st) {
        int4 a = dst[i]; const int4 b = src[i];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        dst[i] = a;
    }
}

void launch_add_i32(hipStream_t st, int *dst, const int *src, size_t n_words)
{
    const size_t n16 = n_words / 4;
    size_t g = (n16 + WG - 1) / WG; if (g > 8192) g = 8192; if (!g) return;
    hipLaunchKernelGGL(k_add_i32, dim3((unsigned)g), dim3(WG), 0, st, (int4 *)dst, (const int4 *)src, n16);
}

void launch_export_i8(hipStream_t st, const int *diff, const uint8_t *hstate, void *out, uint64_t n_cells, int thr,
                      pd_exc *exc, uint32_t cap, uint32_t *count)
{
    hipLaunchKernelGGL(k_export_i8, dim3(8192), dim3(WG), 0, st, diff, hstate, (int4 *)out, n_cells, thr, exc, cap, count);
}

void launch_import_i8(hipStream_t st, const void *in, int *diff, uint64_t n_cells, int bias, const pd_exc *exc,
                      uint64_t n_exc)
{
    hipLaunchKernelGGL(k_import_i8, dim3(8192), dim3(WG), 0, st, (const int4 *)in, diff, n_cells, bias);
    if (n_exc) hipLaunchKernelGGL(k_apply_exceptions, dim3(256), dim3(WG), 0, st, exc, n_exc, diff, n_cells);
}

void launch_export_i4(hipStream_t st, const int *diff, const uint8_t *hstate, void *out, uint64_t n_cells,
                      pd_exc *exc, uint32_t cap, uint32_t *count)
{
    hipLaunchKernelGGL(k_export_i4, dim3(16384), dim3(WG), 0, st, diff, hstate, (unsigned short *)out,
                       (uint32_t)(n_cells / TILE), exc, cap, count);
}

void launch_add_i4(hipStream_t st, int *dst, const void *img, uint32_t n_tiles, const pd_exc *exc, uint64_t n_exc,
                   uint64_t n_cells_total, int *dst_base)
{
    if (n_tiles) {
        unsigned g = n_tiles < 16384 ? n_tiles : 16384;
        hipLaunchKernelGGL(k_add_i4, dim3(g), dim3(WG), 0, st, dst, (const unsigned short *)img, n_tiles);
    }
    if (n_exc) hipLaunchKernelGGL(k_apply_exceptions, dim3(256), dim3(WG), 0, st, exc, n_exc, dst_base, n_cells_total);
}

// (A/B switch of the packed statistics kernel: pd_set_param "sweep_i4_fast")
static bool g_sweep_i4_fast = true;
void set_sweep_i4_fast(bool on) { g_sweep_i4_fast = on; }
static bool sweep_i4_fast_on() { return g_sweep_i4_fast; }

void launch_sweep_i4(hipStream_t st, const void *parts, uint32_t n_parts, uint64_t stride, uint32_t tile_first,
                     uint32_t tile_count, const pd_exc *exc, uint64_t exc_stride, const int32_t *exc_counts, uint8_t *flags, size_t flags_bytes,
                     const int *carry, uint32_t wrap_mask, TileMap tm, uint32_t w, uint32_t min_dep, TilePart *part, int *depth_out)
{
    if (!tile_count) return;
    const bool with_exc = exc && exc_counts && exc_stride && flags;
    // behind the flags (one byte per tile of the buffer): a counter and the list of the flagged tiles of this slice
    uint32_t *n_list = with_exc ? reinterpret_cast<uint32_t *>(flags + flags_bytes) : nullptr, *list = with_exc ? n_list + 4 : nullptr;
    if (with_exc) {
        (void)hipMemsetAsync(flags, 0, tile_count, st);
        (void)hipMemsetAsync(n_list, 0, 16, st);
        hipLaunchKernelGGL(k_flag_exception_tiles, dim3(64, n_parts), dim3(WG), 0, st, exc, exc_stride, exc_counts,
                           (uint64_t)tile_first, (uint64_t)tile_count, flags);
        hipLaunchKernelGGL(k_list_flagged, dim3(256), dim3(WG), 0, st, (const uint8_t *)flags, tile_count, list, n_list);
    }
    I4Src src{(const uint8_t *)parts, stride, n_parts, with_exc ? flags : nullptr, exc, exc_stride, exc_counts};
    if (n_parts <= 16) {
        const dim3 g((tile_count + WG / 64 - 1) / (WG / 64));
        const uint32_t *none = nullptr;
        // behind the flags and the list of the flagged tiles: a second counter and list, for the tiles the packed kernel leaves
        uint32_t *n_slow = flags ? reinterpret_cast<uint32_t *>(flags + flags_bytes) + 4 + ((tile_count + 3u) & ~3u) : nullptr, *slow = n_slow ? n_slow + 4 : nullptr;
        if (depth_out) hipLaunchKernelGGL(k_sweep_i4_wave<true>, g, dim3(WG), 0, st, src, carry, wrap_mask, tm, w, min_dep, part, tile_first, tile_count, depth_out, none, none);
        else if (n_slow && min_dep <= 1u && wrap_mask >= 0xFFFFu && sweep_i4_fast_on()) {
            (void)hipMemsetAsync(n_slow, 0, 16, st);
            hipLaunchKernelGGL(k_sweep_i4_fast, g, dim3(WG), 0, st, src, carry, tm, w, min_dep, part, tile_first, tile_count, slow, n_slow);
            hipLaunchKernelGGL(k_sweep_i4_wave<false>, dim3(g.x < 1024u ? g.x : 1024u), dim3(WG), 0, st, src, carry, wrap_mask, tm, w, min_dep, part, tile_first, tile_count,
                               depth_out, (const uint32_t *)slow, (const uint32_t *)n_slow);
        } else hipLaunchKernelGGL(k_sweep_i4_wave<false>, g, dim3(WG), 0, st, src, carry, wrap_mask, tm, w, min_dep, part, tile_first, tile_count, depth_out, none, none);
    } else if (depth_out) {                                      // (more than 16 parts: the workgroup form, every tile through the list-less PATCH-free kernel's depth branch)
        hipLaunchKernelGGL(k_sweep_i4<false>, dim3(tile_count), dim3(WG), 0, st, src, carry, wrap_mask, tm, w, min_dep, part, tile_first,
                           (const uint32_t *)nullptr, (const uint32_t *)nullptr, depth_out);
    } else
        hipLaunchKernelGGL(k_sweep_i4<false>, dim3(tile_count), dim3(WG), 0, st, src, carry, wrap_mask, tm, w, min_dep, part, tile_first,
                           (const uint32_t *)nullptr, (const uint32_t *)nullptr, (int *)nullptr);
    if (with_exc)
        hipLaunchKernelGGL(k_sweep_i4<true>, dim3(512), dim3(WG), 0, st, src, carry, wrap_mask, tm, w, min_dep, part, tile_first, (const uint32_t *)list,
                           (const uint32_t *)n_list, depth_out);
}

void launch_window_gather(hipStream_t st, const TilePart *part, TileMap tm, int32_t n_contigs, uint32_t w,
                          uint64_t n_windows, uint32_t *cover, unsigned long long *sum)
{
    if (!n_windows) return;
    hipLaunchKernelGGL(k_window_gather, dim3((unsigned)((n_windows + 3) / 4)), dim3(WG), 0, st, part, tm, n_contigs,
                       w, n_windows, cover, sum);
}

// ------------------------------------------------------------------------------------------
// launch wrappers (called from pd_capi.hip)
// ------------------------------------------------------------------------------------------
void launch_fill(hipStream_t st, void *p, size_t bytes)
{
    const size_t n16 = bytes / 16;
    size_t g = (n16 + WG - 1) / WG; if (g > 4096) g = 4096; if (g == 0) g = 1;
    hipLaunchKernelGGL(k_fill, dim3((unsigned)g), dim3(WG), 0, st, (int4 *)p, n16);
}

void launch_scatter_atomic(hipStream_t st, const pd_iv *iv, size_t n, ContigTab tab, int *diff, int *sums)
{
    size_t g = (n + WG - 1) / WG; if (g > 8192) g = 8192; if (g == 0) return;
    hipLaunchKernelGGL(k_scatter_atomic, dim3((unsigned)g), dim3(WG), 0, st, iv, n, tab, diff, sums);
}

void launch_scatter_index(hipStream_t st, const pd_iv *iv, uint32_t n, ContigTab tab, uint32_t lmax,
                          uint32_t disorder, uint32_t sample, uint32_t *ub_a, uint32_t *cand_lo,
                          uint32_t n_stiles, int stile, BatchDesc *desc)
{
    const uint32_t K = (n + sample - 1) / sample + 1;
    const dim3 g((K + WG - 1) / WG), b(WG);
    if (stile == 4096)
        hipLaunchKernelGGL(k_index<4096>, g, b, 0, st, iv, n, sample, tab, lmax, disorder, ub_a, cand_lo, n_stiles, desc);
    else
        hipLaunchKernelGGL(k_index<8192>, g, b, 0, st, iv, n, sample, tab, lmax, disorder, ub_a, cand_lo, n_stiles, desc);
}

void launch_scatter_tiles(hipStream_t st, const PendSet &ps, ContigTab tab, const uint32_t *tile_contig,
                          uint32_t n_stiles, int stile, int *diff, int *sums, uint8_t *hstate,
                          uint64_t *ovf, uint32_t ovf_cap, CheckWords *chk, unsigned grid_tiles)
{
    if (stile == 4096)
        hipLaunchKernelGGL(k_scatter_tiles<4096>, dim3(grid_tiles), dim3(WG), 0, st, ps, n_stiles, tab, tile_contig,
                           diff, sums, hstate, ovf, ovf_cap, chk);
    else
        hipLaunchKernelGGL(k_scatter_tiles<8192>, dim3(grid_tiles), dim3(WG), 0, st, ps, n_stiles, tab, tile_contig,
                           diff, sums, hstate, ovf, ovf_cap, chk);
}

void launch_direct_tiles(hipStream_t st, const PendSet &ps, ContigTab tab, const uint32_t *tile_contig, uint32_t n_tiles,
                         uint32_t wrap_mask, uint32_t w, uint32_t min_dep, TilePart *part, const uint64_t *win_off,
                         uint32_t *cover, unsigned long long *sum, uint32_t *n_long, uint32_t *fail,
                         uint32_t *heavy_list, uint32_t *heavy_count, unsigned grid_tiles, int un)
{
    WinArgs wa; wa.w = w; wa.min_dep = min_dep; wa.inv_w = 1.0f / (float)w; wa.cover = cover; wa.sum = sum; wa.part = part;
    const DirectWide dw{w, min_dep, part};
    const DirectNarrow dn{wa, win_off};
#define PD_DIRECT(UN_, WPE_) hipLaunchKernelGGL((k_direct_tiles<UN_, WPE_, DirectWide>), dim3(grid_tiles), dim3(WG), 0, st, ps, \
                                                n_tiles, tab, tile_contig, wrap_mask, dw, n_long, heavy_list, heavy_count)
    if (w < (uint32_t)TILE)
        hipLaunchKernelGGL((k_direct_tiles<4, 4, DirectNarrow>), dim3(grid_tiles), dim3(WG), 0, st, ps, n_tiles, tab, tile_contig,
                           wrap_mask, dn, n_long, heavy_list, heavy_count);
    else if (un == 0 || un >= 3000) {        // second form (k_direct_wide3): 3000 + 100 x waves-per-SIMD target + loads in flight per thread
#define PD_DIRECT3(UN_, WPE_) hipLaunchKernelGGL((k_direct_wide3<UN_, WPE_, false>), dim3(grid_tiles), dim3(WG), 0, st, ps, n_tiles, tab, tile_contig, wrap_mask, dw, heavy_list, heavy_count, DirectExport{})
        // measured on the bench sample (ms; variants that were tried and are no longer compiled included): <2, 6> 3.08-3.13,
        // <4, 5> 3.10-3.28, <3, 5> 3.2, <2, 7> 3.3, <2, 5> 3.4, <4, 4> 3.6-3.8, <1, 8> 3.6, <2, 8> 4.1 (spills), <4, 6> 3.5
        // (spills); first form 3.52-3.69.  Grids of 16 K .. 262 K workgroups: within 2 %.
        switch (un) {
        case 3404: PD_DIRECT3(4, 4); break;
        case 3504: PD_DIRECT3(4, 5); break;
        default: PD_DIRECT3(2, 6); break;
        }
#undef PD_DIRECT3
    }
    else switch (un) {                       // tuning knob "direct_un": loads in flight per thread + 100 x waves-per-SIMD target
    default: PD_DIRECT(4, 5); break;         // 504 (or any other value below 3000): the first form, kept as the cross-check of the second
                                             // (measured on the bench sample, ms: <4, 5> 3.5-3.7; its other variants <8, 5> 3.3-3.6, <4, 4> 3.7, <8, 4> 3.7 are no longer compiled)
    }
#undef PD_DIRECT
    hipLaunchKernelGGL(k_direct_tiles_heavy, dim3(128), dim3(WG), 0, st, ps, n_tiles, tab, tile_contig, wrap_mask, wa, win_off,
                       n_long, (const uint32_t *)heavy_list, (const uint32_t *)heavy_count, DirectExport{}, C8Sample{});
    if (w < (uint32_t)TILE && n_tiles > 1)
        hipLaunchKernelGGL(k_window_edges, dim3((n_tiles + WG - 1) / WG), dim3(WG), 0, st, (const TilePart *)part, TileMap{tile_contig, tab.off, tab.len, win_off},
                           n_tiles, w, cover, sum);
    hipLaunchKernelGGL(k_finish_direct, dim3(1), dim3(1), 0, st, ps, n_long, fail);
}

static unsigned grid_1k(uint64_t n) { return (unsigned)((n + 1023) / 1024 ? (n + 1023) / 1024 : 1); }

void launch_c8_from_sorted(hipStream_t st, const pd_iv *iv, uint32_t n, ContigTab tab, uint32_t bshift, Run8 *out, uint32_t *b1, uint32_t *words)
{
    if (!n) return;
    hipLaunchKernelGGL(k_c8_from_sorted, dim3((unsigned)(((uint64_t)n + WG - 1) / WG)), dim3(WG), 0, st, iv, n, tab, bshift, out, b1, words);
}

void launch_c8_hist(hipStream_t st, const pd_iv *iv, uint32_t n, ContigTab tab, uint32_t bshift, uint32_t *hist, uint32_t *words)
{
    if (!n) return;
    const uint64_t g = ((uint64_t)n + WG - 1) / WG;
    hipLaunchKernelGGL(k_c8_hist, dim3((unsigned)(g > 65536 ? 65536 : g)), dim3(WG), 0, st, iv, n, tab, bshift, hist, words);
}

void launch_excl_scan_u32(hipStream_t st, const uint32_t *in, uint32_t *out, uint32_t n, uint32_t *block_sums)
{
    const unsigned nb = grid_1k(n);
    hipLaunchKernelGGL(k_scan_block_sums, dim3(nb), dim3(WG), 0, st, in, n, block_sums);
    hipLaunchKernelGGL(k_scan_of_sums, dim3(1), dim3(1024), 0, st, block_sums, nb);
    hipLaunchKernelGGL(k_scan_blocks, dim3(nb), dim3(WG), 0, st, in, out, n, (const uint32_t *)block_sums);
}

// b1[0 .. n_buckets]: the marks of the buckets' first runs (0xFFFFFFFF where nobody begins) -> the index of the first run at or behind
// every bucket; b1[n_buckets] = n_runs.  tmp: n_buckets / 1024 + 2 words.
void launch_c8_fill_starts(hipStream_t st, uint32_t *b1, uint32_t n_buckets, uint32_t n_runs, uint32_t *tmp)
{
    const uint32_t n = n_buckets + 1;
    const unsigned nb = grid_1k(n);
    hipLaunchKernelGGL(k_set_u32, dim3(1), dim3(1), 0, st, b1 + n_buckets, n_runs);
    hipLaunchKernelGGL(k_sfx_block_min, dim3(nb), dim3(WG), 0, st, (const uint32_t *)b1, n, tmp);
    hipLaunchKernelGGL(k_sfx_of_mins, dim3(1), dim3(1024), 0, st, tmp, nb);
    hipLaunchKernelGGL(k_sfx_blocks, dim3(nb), dim3(WG), 0, st, b1, n, (const uint32_t *)tmp);
}

void launch_c8_marks_to_index(hipStream_t st, const unsigned long long *marks, uint32_t n_buckets, const uint32_t *base, uint32_t *b1)
{
    if (!n_buckets) return;
    hipLaunchKernelGGL(k_c8_marks_to_index, dim3((unsigned)(((uint64_t)n_buckets + WG - 1) / WG)), dim3(WG), 0, st, marks, n_buckets, base, b1);
}

void launch_c8_place_other(hipStream_t st, const pd_iv *iv, uint32_t n, ContigTab tab, uint32_t bshift, const uint32_t *o1, uint32_t *cursor, Run8 *out)
{
    if (!n) return;
    const uint64_t g = ((uint64_t)n + WG - 1) / WG;
    hipLaunchKernelGGL(k_c8_place_other, dim3((unsigned)(g > 65536 ? 65536 : g)), dim3(WG), 0, st, iv, n, tab, bshift, o1, cursor, out);
}

void launch_c8_expand(hipStream_t st, C8Sample cs, const uint32_t *tile_contig, const uint64_t *contig_off, uint32_t n_tiles, pd_iv *out)
{
    hipLaunchKernelGGL(k_c8_expand, dim3(n_tiles < 16384u ? (n_tiles ? n_tiles : 1u) : 16384u), dim3(WG), 0, st, cs, tile_contig, contig_off, n_tiles, out);
}

void launch_copy_words(hipStream_t st, void *dst, const void *src, uint64_t n_words)
{
    if (!n_words) return;
    const uint64_t g = (n_words / 4 + 255) / 256 + 1;
    hipLaunchKernelGGL(k_copy_words, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, st, (uint32_t *)dst, (const uint32_t *)src, n_words);
}

void launch_r8_to_iv(hipStream_t st, const Run8 *r8, uint64_t n, ContigTab tab, pd_iv *out)
{
    if (!n) return;
    const uint64_t g = (n + WG - 1) / WG;
    hipLaunchKernelGGL(k_r8_to_iv, dim3((unsigned)(g > 65536 ? 65536 : g)), dim3(WG), 0, st, r8, n, tab, out);
}

void launch_direct_c8(hipStream_t st, C8Sample cs, ContigTab tab, const uint32_t *tile_contig, uint32_t n_tiles, uint32_t wrap_mask, uint32_t w,
                      uint32_t min_dep, TilePart *part, uint32_t *heavy_list, uint32_t *heavy_count, unsigned grid_tiles, int un)
{
    const DirectWide dw{w, min_dep, part};
    // "direct_un" for a compact sample: 100 x waves-per-SIMD target + loads in flight per thread (0 = default)
#define PD_C8(WPE_, UN8_) hipLaunchKernelGGL((k_direct_c8<WPE_, UN8_, false>), dim3(grid_tiles), dim3(WG), 0, st, cs, n_tiles, tab, tile_contig, wrap_mask, dw, heavy_list, heavy_count, DirectExport{})
    switch (un) {
    case 502: PD_C8(5, 2); break;
    case 504: PD_C8(5, 4); break;
    case 508: PD_C8(5, 8); break;
    case 602: PD_C8(6, 2); break;
    case 604: PD_C8(6, 4); break;
    case 702: PD_C8(7, 2); break;
    case 802: PD_C8(8, 2); break;
    case 804: PD_C8(8, 4); break;
    case 801: PD_C8(8, 1); break;
    case 708: PD_C8(7, 8); break;
    case 803: PD_C8(8, 3); break;
    case 703: PD_C8(7, 3); break;
    // measured on the bench sample (ms): <8, 3> 1.89, <7, 4> 1.99, <7, 3> 2.01, <8, 2> 2.01, <6, 4> 2.16, <7, 2> 2.17, <5, 8> 2.25, <6, 2> 2.40, <5, 4> 2.41, <8, 4> 2.41 (spills),
    // <8, 1> 2.42, <5, 2> 2.70 — k_direct_wide3 on the same sample as 12-byte streams: 3.11
    case 704: PD_C8(7, 4); break;
    // the joined tail (JOIN): 1000 + the numbers above
    case 1803: hipLaunchKernelGGL((k_direct_c8<8, 3, false, true>), dim3(grid_tiles), dim3(WG), 0, st, cs, n_tiles, tab, tile_contig, wrap_mask, dw, heavy_list, heavy_count, DirectExport{}); break;
    case 1703: hipLaunchKernelGGL((k_direct_c8<7, 3, false, true>), dim3(grid_tiles), dim3(WG), 0, st, cs, n_tiles, tab, tile_contig, wrap_mask, dw, heavy_list, heavy_count, DirectExport{}); break;
    case 1704: hipLaunchKernelGGL((k_direct_c8<7, 4, false, true>), dim3(grid_tiles), dim3(WG), 0, st, cs, n_tiles, tab, tile_contig, wrap_mask, dw, heavy_list, heavy_count, DirectExport{}); break;
    case 1802: hipLaunchKernelGGL((k_direct_c8<8, 2, false, true>), dim3(grid_tiles), dim3(WG), 0, st, cs, n_tiles, tab, tile_contig, wrap_mask, dw, heavy_list, heavy_count, DirectExport{}); break;
    // 5000 +: 16-byte loads (two runs per lane and load) for the sorted stream's full chunks
    case 5704: hipLaunchKernelGGL((k_direct_c8<7, 4, false, true, true>), dim3(grid_tiles), dim3(WG), 0, st, cs, n_tiles, tab, tile_contig, wrap_mask, dw, heavy_list, heavy_count, DirectExport{}); break;
    case 5702: hipLaunchKernelGGL((k_direct_c8<7, 2, false, true, true>), dim3(grid_tiles), dim3(WG), 0, st, cs, n_tiles, tab, tile_contig, wrap_mask, dw, heavy_list, heavy_count, DirectExport{}); break;
    case 5802: hipLaunchKernelGGL((k_direct_c8<8, 2, false, true, true>), dim3(grid_tiles), dim3(WG), 0, st, cs, n_tiles, tab, tile_contig, wrap_mask, dw, heavy_list, heavy_count, DirectExport{}); break;
    case 5706: hipLaunchKernelGGL((k_direct_c8<7, 6, false, true, true>), dim3(grid_tiles), dim3(WG), 0, st, cs, n_tiles, tab, tile_contig, wrap_mask, dw, heavy_list, heavy_count, DirectExport{}); break;
    // round 4, two streams (ms by the context's events, which also bracket the pile-up and finish launches): <8, 3> 2.35; joined tail: <8, 3> 2.33, <8, 2> 2.07, <7, 3> 2.07, <7, 4> 2.04
    default: hipLaunchKernelGGL((k_direct_c8<7, 4, false, true>), dim3(grid_tiles), dim3(WG), 0, st, cs, n_tiles, tab, tile_contig, wrap_mask, dw, heavy_list, heavy_count, DirectExport{}); break;
    }
#undef PD_C8
    WinArgs wa; wa.w = w; wa.min_dep = min_dep; wa.inv_w = 1.0f / (float)w; wa.cover = nullptr; wa.sum = nullptr; wa.part = part;
    PendSet none{}; none.nb = 0; none.lmax = 0;
    hipLaunchKernelGGL(k_direct_tiles_heavy, dim3(128), dim3(WG), 0, st, none, n_tiles, tab, tile_contig, wrap_mask, wa, (const uint64_t *)nullptr,
                       heavy_count + 1 /* n_long: unused here */, (const uint32_t *)heavy_list, (const uint32_t *)heavy_count, DirectExport{}, cs);
}

void launch_direct_c8_export(hipStream_t st, C8Sample cs, ContigTab tab, const uint32_t *tile_contig, uint32_t n_tiles, void *img, pd_exc *exc,
                             uint32_t cap, uint32_t *count, int *sums, uint32_t *heavy_list, uint32_t *heavy_count, unsigned grid_tiles)
{
    const DirectExport de{(unsigned short *)img, exc, cap, count, sums};
    hipLaunchKernelGGL((k_direct_c8<7, 4, true, true>), dim3(grid_tiles), dim3(WG), 0, st, cs, n_tiles, tab, tile_contig, 0xFFFFFFFFu,
                       DirectWide{(uint32_t)TILE, 1u, nullptr}, heavy_list, heavy_count, de);
    WinArgs wa; wa.w = (uint32_t)TILE; wa.min_dep = 1; wa.inv_w = 0.f; wa.cover = nullptr; wa.sum = nullptr; wa.part = nullptr;
    PendSet none{}; none.nb = 0; none.lmax = 0;
    hipLaunchKernelGGL(k_direct_tiles_heavy, dim3(128), dim3(WG), 0, st, none, n_tiles, tab, tile_contig, 0xFFFFFFFFu, wa, (const uint64_t *)nullptr,
                       heavy_count + 1, (const uint32_t *)heavy_list, (const uint32_t *)heavy_count, de, cs);
}

void launch_direct_export(hipStream_t st, const PendSet &ps, ContigTab tab, const uint32_t *tile_contig, uint32_t n_tiles,
                          void *img, pd_exc *exc, uint32_t cap, uint32_t *count, int *sums, uint32_t *n_long, uint32_t *fail,
                          uint32_t *heavy_list, uint32_t *heavy_count, unsigned grid_tiles)
{
    const DirectExport de{(unsigned short *)img, exc, cap, count, sums};
    // the second form of the direct kernel with its export branch (the first form's export instantiation — 128 VGPRs and
    // 160 bytes of spills — took 5.2 ms per sample)
    hipLaunchKernelGGL((k_direct_wide3<2, 5, true>), dim3(grid_tiles), dim3(WG), 0, st, ps, n_tiles, tab, tile_contig, 0xFFFFFFFFu,
                       DirectWide{(uint32_t)TILE, 1u, nullptr}, heavy_list, heavy_count, de);
    // tiles with more than 32 000 candidates: the int-window kernel exports them
    WinArgs wa; wa.w = (uint32_t)TILE; wa.min_dep = 1; wa.inv_w = 0.f; wa.cover = nullptr; wa.sum = nullptr; wa.part = nullptr;
    hipLaunchKernelGGL(k_direct_tiles_heavy, dim3(128), dim3(WG), 0, st, ps, n_tiles, tab, tile_contig, 0xFFFFFFFFu, wa,
                       (const uint64_t *)nullptr, n_long, (const uint32_t *)heavy_list, (const uint32_t *)heavy_count, de, C8Sample{});
    hipLaunchKernelGGL(k_finish_direct, dim3(1), dim3(1), 0, st, ps, n_long, fail);
}

void launch_fill_invalid(hipStream_t st, int *diff, uint8_t *hstate, uint32_t n_half, CheckWords *chk,
                         bool only_if_overflow, unsigned grid)
{
    hipLaunchKernelGGL(k_fill_invalid, dim3(grid), dim3(WG), 0, st, (int4 *)diff, hstate, n_half, chk,
                       only_if_overflow ? 1 : 0);
    if (!only_if_overflow) hipLaunchKernelGGL(k_mark_all_valid, dim3(1), dim3(1), 0, st, chk);
}

void launch_mark_all_valid(hipStream_t st, CheckWords *chk)
{
    hipLaunchKernelGGL(k_mark_all_valid, dim3(1), dim3(1), 0, st, chk);
}

void launch_scatter_finish(hipStream_t st, const PendSet &ps, int *diff, int *sums, const uint64_t *ovf,
                           uint32_t ovf_cap, CheckWords *chk)
{
    hipLaunchKernelGGL(k_apply_overflow, dim3(256), dim3(WG), 0, st, ovf, chk, ovf_cap, diff, sums, 0);
    hipLaunchKernelGGL(k_finish_batch, dim3(1), dim3(1), 0, st, ps, chk);
}

void launch_tile_carry(hipStream_t st, const int *sums, int *bsum, int *carry, uint32_t n_tiles)
{
    const unsigned nb = (n_tiles + 1023) / 1024;
    hipLaunchKernelGGL(k_carry_block_sums, dim3(nb), dim3(1024), 0, st, sums, bsum, n_tiles);
    hipLaunchKernelGGL(k_tile_carry, dim3(nb), dim3(1024), 0, st, sums, bsum, carry, n_tiles);
}

void launch_scan_write(hipStream_t st, int *buf, const int *carry, uint32_t n_tiles, uint32_t wrap_mask,
                       const uint8_t *hstate)
{
    TileMap tm{}; WinArgs wa{};
    hipLaunchKernelGGL((k_sweep<true, false, false>), dim3(n_tiles), dim3(WG), 0, st, buf, carry, wrap_mask, tm, wa, hstate, 0u);
}

static size_t win_lds_bytes(uint32_t w)
{
    if (w >= (uint32_t)TILE) return 16;                // large windows use per-tile partials, no LDS accumulators
    const size_t nacc = (size_t)TILE / w + 2;
    return nacc * 8 + 16;
}

int launch_sweep_windows(hipStream_t st, int *buf, const int *carry, uint32_t n_tiles, uint32_t wrap_mask,
                         TileMap tm, uint32_t w, uint32_t min_dep, uint32_t *cover, unsigned long long *sum,
                         TilePart *part, uint64_t n_windows, int32_t n_contigs, bool from_depth,
                         const uint8_t *hstate)
{
    WinArgs wa; wa.w = w; wa.min_dep = min_dep; wa.inv_w = 1.0f / (float)w; wa.cover = cover; wa.sum = sum;
    wa.part = part;
    const size_t lds = win_lds_bytes(w);
    hipError_t e = hipSuccess;
    if (from_depth) {
        if (lds > 48 * 1024) e = hipFuncSetAttribute((const void *)k_sweep<false, true, true>,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((k_sweep<false, true, true>), dim3(n_tiles), dim3(WG), lds, st, buf, carry, wrap_mask, tm, wa, hstate, 0u);
    } else {
        if (lds > 48 * 1024) e = hipFuncSetAttribute((const void *)k_sweep<false, true, false>,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((k_sweep<false, true, false>), dim3(n_tiles), dim3(WG), lds, st, buf, carry, wrap_mask, tm, wa, hstate, 0u);
    }
    if (w >= (uint32_t)TILE && n_windows)
        hipLaunchKernelGGL(k_window_gather, dim3((unsigned)((n_windows + 3) / 4)), dim3(WG), 0, st, part, tm, n_contigs,
                           w, n_windows, cover, sum);
    if (w < (uint32_t)TILE && n_tiles > 1)
        hipLaunchKernelGGL(k_window_edges, dim3((n_tiles + WG - 1) / WG), dim3(WG), 0, st, (const TilePart *)part, tm, n_tiles, w, cover, sum);
    return 0;
}

// Narrow windows over a SLICE of the summed depth (tiles [tile_first, tile_first + tile_count) of the genome, `depth_slice` holding exactly
// those): windows inside the slice's tiles into the (whole-genome) window arrays, the shares of the windows across tile edges into
// part[tile] (whole-genome indexing); the caller merges ranks and runs launch_window_edges once everything is together.
int launch_sweep_windows_slice(hipStream_t st, int *depth_slice, uint32_t tile_first, uint32_t tile_count, TileMap tm, uint32_t w, uint32_t min_dep,
                               uint32_t *cover, unsigned long long *sum, TilePart *part)
{
    if (!tile_count) return 0;
    WinArgs wa; wa.w = w; wa.min_dep = min_dep; wa.inv_w = 1.0f / (float)w; wa.cover = cover; wa.sum = sum; wa.part = part;
    const size_t lds = win_lds_bytes(w);
    if (lds > 48 * 1024) {
        const hipError_t e = hipFuncSetAttribute((const void *)k_sweep<false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL((k_sweep<false, true, true>), dim3(tile_count), dim3(WG), lds, st, depth_slice, (const int *)nullptr, 0xFFFFFFFFu, tm, wa,
                       (const uint8_t *)nullptr, tile_first);
    return 0;
}

void launch_window_edges(hipStream_t st, const TilePart *part, TileMap tm, uint32_t n_tiles, uint32_t w, uint32_t *cover, unsigned long long *sum)
{
    if (n_tiles > 1) hipLaunchKernelGGL(k_window_edges, dim3((n_tiles + WG - 1) / WG), dim3(WG), 0, st, part, tm, n_tiles, w, cover, sum);
}

void launch_reduce_pieces(hipStream_t st, const int *depth, const Piece *pieces, uint32_t n_pieces,
                          uint32_t min_dep, int *cover, unsigned long long *sum)
{
    if (!n_pieces) return;
    hipLaunchKernelGGL(k_reduce_pieces, dim3((n_pieces + 3) / 4), dim3(WG), 0, st, depth, pieces, n_pieces,
                       min_dep, cover, sum);
}

} // namespace pdk

#ifdef PD_WIDE3_TICKS
extern "C" int pd_x_wide3_ticks(unsigned long long *out16)
{
    unsigned long long z[16] = {0};
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(pdk::g_wide3_ticks), sizeof z) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(pdk::g_wide3_ticks), z, sizeof z) == hipSuccess ? 0 : -1;
}
#endif
