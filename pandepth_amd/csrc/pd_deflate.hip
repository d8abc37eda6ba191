// pd_deflate.hip — zlib's level-6 LZ77 parse on the GPU (stage 1 of the byte-identical gzip streams of the per-site and window
// files, PD:4264-4284 / PD:4366-4389 written through gzstream): csrc/pd_lz77.h holds the parse itself (one wave evaluates the
// candidates of a position at once); here are the kernels around it:
//   * the positions of a batch of text sorted by (hash of their 3 bytes, position) — a stable LSD radix sort in two passes of
//     8 + 7 bits over 64-bit (hash << 32 | position) keys: per-wave-block digit histograms, one exclusive scan, a stable scatter
//     whose in-block ranks come from ballots (the lanes with my digit below me) — then R[p] (where p stands) and the buckets' starts;
//   * the parse: ONE WAVE PER CHUNK walks zlib's lazy-evaluation state machine (wave-uniform) and calls the wave-parallel
//     longest_match; lane 0 writes the symbols into the chunk's stretch of the symbol buffer;
//   * the chunks' symbols gathered into one contiguous array for the copy back.
// All HBM-bound or latency-bound integer work; nothing here is a contraction.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "pd_kernels.h"
#include "pd_lz77.h"

namespace pdk {

namespace {

constexpr int WGZ = 256;
constexpr uint32_t RBLK = 2048;                 // elements per wave-block of the radix passes

struct DevWaveZ {                               // the hardware wavefront (pd_lz77.h's W)
    template <class T> struct Var { T v; __device__ T &operator[](int) { return v; } __device__ const T &operator[](int) const { return v; } };
    template <class F> __device__ static __forceinline__ void each(F f) { f((int)(threadIdx.x & 63)); }
    __device__ static __forceinline__ uint64_t ballot_eq(const Var<uint32_t> &x, uint32_t v) { return __ballot(x.v == v); }
    __device__ static __forceinline__ uint64_t ballot_ne(const Var<uint32_t> &x, uint32_t v) { return __ballot(x.v != v); }
    __device__ static __forceinline__ uint32_t reduce_max(const Var<uint32_t> &x)
    {
        uint32_t m = x.v;
#pragma unroll
        for (int o = 32; o; o >>= 1) { const uint32_t y = (uint32_t)__shfl_xor((int)m, o); m = y > m ? y : m; }
        return m;
    }
    // (lane is the same in every lane: a lane read through a scalar register instead of a trip through the LDS crossbar)
    __device__ static __forceinline__ uint32_t bcast(const Var<uint32_t> &x, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)x.v, __builtin_amdgcn_readfirstlane(lane)); }
    __device__ static __forceinline__ uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
    __device__ static __forceinline__ void loads_landed() { __builtin_amdgcn_s_waitcnt(0x0F70); }            // s_waitcnt vmcnt(0)
    __device__ static __forceinline__ bool lead() { return (threadIdx.x & 63) == 0; }
};

// ---- keys: hash << 32 | position, for every position with 3 bytes left ----
__global__ __launch_bounds__(WGZ) void k_lz_keys(const uint8_t *text, uint32_t np, uint64_t *keys)
{
    for (uint64_t p = (uint64_t)blockIdx.x * WGZ + threadIdx.x; p < np; p += (uint64_t)gridDim.x * WGZ)
        keys[p] = ((uint64_t)pdz::hash3(text + p) << 32) | p;
}

// ---- one LSD pass over digit (key >> shift) & mask: histogram per wave-block (bin-major layout for the scan) ----
__global__ __launch_bounds__(WGZ) void k_lz_hist(const uint64_t *keys, uint32_t n, uint32_t shift, uint32_t mask, uint32_t n_blocks, uint32_t *hist)
{
    __shared__ uint32_t cnt[4][256];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t blk = blockIdx.x * 4 + wv;
    for (int k = lane; k < 256; k += 64) cnt[wv][k] = 0;
    __syncthreads();
    if (blk < n_blocks) {
        const uint64_t lo = (uint64_t)blk * RBLK, hi = lo + RBLK < n ? lo + RBLK : n;
        for (uint64_t i = lo + lane; i < hi; i += 64) atomicAdd(&cnt[wv][(uint32_t)(keys[i] >> shift) & mask], 1u);
    }
    __syncthreads();
    if (blk < n_blocks) for (uint32_t d = lane; d <= mask; d += 64) hist[(uint64_t)d * n_blocks + blk] = cnt[wv][d];
}

// ---- the stable scatter: rank inside the block = elements of my digit before me (earlier groups: the running count; my group: ballots) ----
__global__ __launch_bounds__(WGZ) void k_lz_scatter(const uint64_t *keys, uint32_t n, uint32_t shift, uint32_t mask, uint32_t n_blocks,
                                                    const uint32_t *offs, uint64_t *out)
{
    __shared__ uint32_t cur[4][256];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t blk = blockIdx.x * 4 + wv;
    if (blk < n_blocks) for (uint32_t d = lane; d <= mask; d += 64) cur[wv][d] = offs[(uint64_t)d * n_blocks + blk];
    __syncthreads();
    if (blk >= n_blocks) return;
    const uint64_t lo = (uint64_t)blk * RBLK, hi = lo + RBLK < n ? lo + RBLK : n;
    for (uint64_t i0 = lo; i0 < hi; i0 += 64) {
        const uint64_t i = i0 + lane;
        const bool on = i < hi;
        const uint64_t key = on ? keys[i] : 0;
        const uint32_t d = on ? (uint32_t)(key >> shift) & mask : 0xFFFFFFFFu;
        // the lanes that hold my digit: one ballot per bit of the digit
        uint64_t same = __ballot(on);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint64_t bal = __ballot((d >> b) & 1u);
            same &= ((d >> b) & 1u) ? bal : ~bal;
        }
        if (on) {
            const uint32_t below = (uint32_t)__builtin_popcountll(same & ((1ull << lane) - 1ull));
            const uint32_t base = cur[wv][d];
            out[base + below] = key;
            // the last lane of my digit moves the block's cursor on (the wave's LDS accesses are in order: every lane above has read `base`)
            if ((same >> lane) == 1ull) cur[wv][d] = base + (uint32_t)__builtin_popcountll(same);
        }
    }
}

// ---- S, R and the buckets' starts from the sorted keys ----
__global__ __launch_bounds__(WGZ) void k_lz_ranks(const uint64_t *sorted, uint32_t np, uint32_t *S, uint32_t *R, uint32_t *bucket)
{
    for (uint64_t i = (uint64_t)blockIdx.x * WGZ + threadIdx.x; i < np; i += (uint64_t)gridDim.x * WGZ) {
        const uint64_t k = sorted[i];
        const uint32_t p = (uint32_t)k, h = (uint32_t)(k >> 32);
        S[i] = p; R[p] = (uint32_t)i;
        if (i == 0 || (uint32_t)(sorted[i - 1] >> 32) != h) bucket[h] = (uint32_t)i;
    }
}

// ---- the parse: one wave per chunk ----
// A chunk's candidates lie in the 32 KiB before it and every visited position compares up to 128 of them: with the text in HBM that
// was a cache line per candidate (the counters: 48 GB fetched for a 48 MB text), and a visit lasted two dependent trips to memory.
// k_lz_parse_lds: a workgroup takes a GROUP of consecutive chunks (pdk::LzGroup, made by the caller), copies the text they read —
// the first chunk's history up to the last chunk's end — into LDS once (152 KiB for seven chunks of 16 + 4 KiB) and every wave
// parses its chunk from there; S[] and R[] stay in memory and are fetched a visit ahead (pd_lz77.h: fetch, RWin).  Workgroups are
// handed to the 8 XCDs round-robin (workgroup b runs on XCD b % 8), each XCD with its own L2: XCD x takes the x-th eighth of the
// groups, so the S / R lines an XCD has in its L2 belong to one stretch of the text.
// k_lz_parse: the same parse with the text read from memory — chunks that fit no group (a history shorter than zlib's window in the
// middle of a text, a lone oversized chunk).
__global__ __launch_bounds__(64) void k_lz_parse(const pdz::Text T, const uint64_t *chunks /* start, end, origin per chunk */, const uint32_t *list, uint32_t n_list,
                                                 uint32_t *syms, uint64_t stride, uint32_t *counts)
{
    for (uint32_t j = blockIdx.x; j < n_list; j += gridDim.x) {
        const uint32_t c = list[j];
        const uint64_t start = chunks[3 * c], end = chunks[3 * c + 1], origin = chunks[3 * c + 2];
        pdz::Out o{syms + (uint64_t)c * stride, 0u, (uint32_t)stride};
        const bool ok = pdz::parse_chunk<DevWaveZ>(T, start, end, origin, o);
        if ((threadIdx.x & 63) == 0) counts[c] = ok ? o.n : 0xFFFFFFFFu;
    }
}

__global__ __launch_bounds__(1024) void k_lz_parse_lds(const pdz::Text T, const uint64_t *chunks, const LzGroup *groups, uint32_t n_groups, uint32_t per_xcd,
                                                       uint32_t *syms, uint64_t stride, uint32_t *counts)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_text[];
    const uint32_t g = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (g >= n_groups) return;
    const LzGroup G = groups[g];
    const uint64_t a16 = G.base & ~15ull;                       // (the text's buffer is 256-byte aligned and carries 64 bytes behind its end)
    const uint32_t pad = (uint32_t)(G.base - a16), bytes = pad + (uint32_t)G.len;
    for (uint32_t i = threadIdx.x * 16u; i < bytes; i += blockDim.x * 16u)
        *reinterpret_cast<uint4 *>(lds_text + i) = *reinterpret_cast<const uint4 *>(T.text + a16 + i);
    __syncthreads();
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));    // (the wave's number, in a scalar register: so is the parse's whole state)
    if (w >= G.count) return;
    pdz::Text L = T;
    L.text = lds_text + pad; L.lo = G.base;
    const uint32_t c = G.first + w;
    const uint64_t start = chunks[3 * c], end = chunks[3 * c + 1], origin = chunks[3 * c + 2];
    pdz::Out o{syms + (uint64_t)c * stride, 0u, (uint32_t)stride};
    const bool ok = pdz::parse_chunk<DevWaveZ>(L, start, end, origin, o);
    if ((threadIdx.x & 63) == 0) counts[c] = ok ? o.n : 0xFFFFFFFFu;
}

// ---- the chunks' symbols, one after the other ----
__global__ __launch_bounds__(WGZ) void k_lz_gather(const uint32_t *syms, uint64_t stride, const uint64_t *off, uint32_t n_chunks, uint32_t *out)
{
    for (uint32_t c = blockIdx.y; c < n_chunks; c += gridDim.y) {
        const uint64_t a = off[c], n = off[c + 1] - a;
        const uint32_t *src = syms + (uint64_t)c * stride;
        for (uint64_t i = (uint64_t)blockIdx.x * WGZ + threadIdx.x; i < n; i += (uint64_t)gridDim.x * WGZ) out[a + i] = src[i];
    }
}

// ---- CRC-32 of every chunk's own bytes [start, min(start + span, end)): one wave per chunk.  A lane runs the byte-wise table loop over its 1/64 of the
// chunk; the 64 pieces are then joined in GF(2)[x] mod P: crc(A B) = crc(A) x^(8 |B|) + crc(B) (what zlib's crc32_combine computes).
__device__ __forceinline__ uint32_t crc_mulmod(uint32_t a, uint32_t b)
{
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) { p ^= b; if ((a & (m - 1)) == 0) break; }
        m >>= 1;
        b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
    }
    return p;
}
__device__ __forceinline__ uint32_t crc_shift_op(uint64_t len)             // x^(8 len) mod P
{
    uint32_t sq = 1u << 23, p = 1u << 31;
    for (; len; len >>= 1) { if (len & 1) p = crc_mulmod(sq, p); sq = crc_mulmod(sq, sq); }
    return p;
}
__global__ __launch_bounds__(WGZ) void k_lz_crc(const uint8_t *text, const uint64_t *chunks, uint32_t n_chunks, uint64_t span, uint32_t *crc)
{
    __shared__ uint32_t tab[256];
    { uint32_t c = threadIdx.x; for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0xEDB88320u : c >> 1; tab[threadIdx.x] = c; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t ch = blockIdx.x * (WGZ / 64) + (threadIdx.x >> 6);
    if (ch >= n_chunks) return;
    const uint64_t start = chunks[3 * ch], end = chunks[3 * ch + 1] - start > span ? start + span : chunks[3 * ch + 1];
    const uint64_t len = end - start, piece = (len + 63) / 64;
    const uint64_t lo = start + (uint64_t)lane * piece < end ? start + (uint64_t)lane * piece : end;
    const uint64_t hi = lo + piece < end ? lo + piece : end;
    uint32_t c = 0xFFFFFFFFu;
    for (uint64_t i = lo; i < hi; ++i) c = tab[(c ^ text[i]) & 0xFFu] ^ (c >> 8);
    c ^= 0xFFFFFFFFu;                                                      // (an empty piece: 0)
    const uint32_t op_full = crc_shift_op(piece);
    uint32_t acc = 0;
    for (int l = 0; l < 64; ++l) {
        const uint32_t cl = (uint32_t)__shfl((int)c, l);
        const uint64_t l_lo = start + (uint64_t)l * piece < end ? start + (uint64_t)l * piece : end;
        const uint64_t l_len = (l_lo + piece < end ? l_lo + piece : end) - l_lo;
        if (l_len == 0) break;
        acc = crc_mulmod(l_len == piece ? op_full : crc_shift_op(l_len), acc) ^ cl;
    }
    if (lane == 0) crc[ch] = acc;
}

} // namespace

// One text -> (S, R, bucket).  keys_a / keys_b: np 64-bit words each; hist: 256 * n_blocks + 1 words; scan_tmp: see launch_excl_scan_u32.
void launch_lz_sort(hipStream_t st, const uint8_t *text, uint32_t np, uint64_t *keys_a, uint64_t *keys_b, uint32_t *hist, uint32_t *scan_tmp,
                    uint32_t *S, uint32_t *R, uint32_t *bucket)
{
    if (!np) return;
    const uint32_t n_blocks = (np + RBLK - 1) / RBLK;
    const unsigned g = (unsigned)((n_blocks + 3) / 4);
    const uint64_t ge = ((uint64_t)np + WGZ - 1) / WGZ;
    hipLaunchKernelGGL(k_lz_keys, dim3((unsigned)(ge > 65536 ? 65536 : ge)), dim3(WGZ), 0, st, text, np, keys_a);
    // pass 1: the low 8 bits of the hash; pass 2: its high 7 bits (the positions are in order to begin with: stable passes keep them so)
    const uint32_t shifts[2] = {32, 40}, masks[2] = {255u, 127u};
    uint64_t *src = keys_a, *dst = keys_b;
    for (int pass = 0; pass < 2; ++pass) {
        const uint32_t nh = (masks[pass] + 1) * n_blocks;
        hipLaunchKernelGGL(k_lz_hist, dim3(g), dim3(WGZ), 0, st, (const uint64_t *)src, np, shifts[pass], masks[pass], n_blocks, hist);
        launch_excl_scan_u32(st, hist, hist, nh, scan_tmp);
        hipLaunchKernelGGL(k_lz_scatter, dim3(g), dim3(WGZ), 0, st, (const uint64_t *)src, np, shifts[pass], masks[pass], n_blocks, (const uint32_t *)hist, dst);
        uint64_t *t = src; src = dst; dst = t;
    }
    hipLaunchKernelGGL(k_lz_ranks, dim3((unsigned)(ge > 65536 ? 65536 : ge)), dim3(WGZ), 0, st, (const uint64_t *)src, np, S, R, bucket);
}

// Chunks in groups (the text of a group fits LDS) + the list of chunks outside any group; groups / list: device copies.
bool lz_parse_lds_ready(size_t lds_bytes)
{
    static int state = 0;                                       // 0 unknown, 1 usable, -1 not
    static size_t granted = 0;
    if (state == 0 || (state == 1 && lds_bytes > granted)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lz_parse_lds), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LZ_LDS_MAX);
        if (e != hipSuccess) { (void)hipGetLastError(); state = -1; } else { state = 1; granted = LZ_LDS_MAX; }
    }
    return state == 1 && lds_bytes <= granted;
}

void launch_lz_parse(hipStream_t st, const uint8_t *text, uint64_t n_text, const uint32_t *S, const uint32_t *R, const uint32_t *bucket,
                     const uint64_t *chunks, const LzGroup *groups, uint32_t n_groups, uint32_t group_waves, size_t lds_bytes,
                     const uint32_t *list, uint32_t n_list, uint32_t *syms, uint64_t stride, uint32_t *counts)
{
    const pdz::Text T{text, S, R, bucket, n_text};
    if (n_groups) {
        const uint32_t per_xcd = (n_groups + 7) / 8;
        hipLaunchKernelGGL(k_lz_parse_lds, dim3(8 * per_xcd), dim3(64 * group_waves), lds_bytes, st, T, chunks, groups, n_groups, per_xcd, syms, stride, counts);
    }
    if (n_list) hipLaunchKernelGGL(k_lz_parse, dim3(n_list < 8192u ? n_list : 8192u), dim3(64), 0, st, T, chunks, list, n_list, syms, stride, counts);
}

void launch_lz_gather(hipStream_t st, const uint32_t *syms, uint64_t stride, const uint64_t *off, uint32_t n_chunks, uint32_t *out)
{
    if (!n_chunks) return;
    hipLaunchKernelGGL(k_lz_gather, dim3(64, n_chunks < 4096u ? n_chunks : 4096u), dim3(WGZ), 0, st, syms, stride, off, n_chunks, out);
}

void launch_lz_crc(hipStream_t st, const uint8_t *text, const uint64_t *chunks, uint32_t n_chunks, uint64_t span, uint32_t *crc)
{
    if (!n_chunks) return;
    hipLaunchKernelGGL(k_lz_crc, dim3((n_chunks + WGZ / 64 - 1) / (WGZ / 64)), dim3(WGZ), 0, st, text, chunks, n_chunks, span, crc);
}

} // namespace pdk
