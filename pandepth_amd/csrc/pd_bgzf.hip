// pd_bgzf.hip — GPU-side BGZF inflate (the end-to-end lever: on the GPU box the
// host gets 16 cores, and all of the CLI's wall time is libdeflate).  A BGZF file is a sequence of
// independent <= 64 KiB DEFLATE members, so the blocks are decoded concurrently: ONE LANE PER
// BLOCK, the per-block Huffman tables of a wave's 64 lanes in LDS (64 x 2208 B = 138 KiB, one
// wave per CU) or in a global scratch area (many waves per CU, table lookups through L2).  The
// decoder itself is pd_inflate_core.h, verified against zlib on the host.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "pd_inflate_core.h"
#include "pd_inflate_wave.h"
#include "pd_bamwalk.h"
#include "pd_kernels.h"
#include "../../include/pandepth_amd.h"

namespace {

typedef pd_bgzf_block BlkDesc;      // { in_off, out_off, in_len, out_len }
#define PD_WAVE_TOKENS ((int)pdw::TOK_SCRATCH)

// One lane per block.  The fast (one-lookup) tables of a wave's 64 lanes live in LDS (64 x 576 B =
// 36 KiB, four waves per CU); the cold canonical arrays in a global scratch area.  LDS_FAST = false
// keeps everything in global memory (more waves per CU, every lookup through L2).
template <bool LDS_FAST, int WAVES_PER_SIMD = 1>
__global__ __launch_bounds__(64, WAVES_PER_SIMD) void k_inflate_blocks(const uint8_t *comp, const BlkDesc *blk, uint32_t n_blk,
                                                                       uint8_t *out, int *status, pdi::Tables *scratch)
{
    __shared__ pdi::Fast s_fast[LDS_FAST ? 64 : 1];
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n_blk) return;
    pdi::Fast &tf = LDS_FAST ? s_fast[threadIdx.x] : scratch[i].fast;
    const BlkDesc d = blk[i];
    status[i] = d.out_len ? pdi::inflate_block(comp + d.in_off, d.in_len, out + d.out_off, d.out_len, tf, scratch[i].slow) : 0;
}

// One WAVE per BGZF member (pd_inflate_wave.h): persistent one-wave workgroups walk the members with a grid stride
// (members of a BAM are alike, so a static split balances); Huffman tables in LDS (9 KiB per wave), match tokens in a
// per-workgroup slice of global scratch.
#ifndef PD_INFLATE_MIN_WAVES
#define PD_INFLATE_MIN_WAVES 5            /* waves per SIMD the register allocation leaves room for: 96 VGPRs, no scratch.  The launcher puts 20 waves on a CU (5 per
                                             SIMD; the kernel's LDS would allow 22), so the 80-VGPR cap of "6" bought nothing and cost 112 B of spills: 294 -> 298 GB/s
                                             (profiles/r05_inflate_ticks.txt, r5c28); at 8 — 64 VGPRs — the kernel loses a quarter */
#endif
__global__ __launch_bounds__(64, PD_INFLATE_MIN_WAVES) void k_inflate_wave(const uint8_t *comp, const BlkDesc *blk, uint32_t n_blk, uint8_t *out, int *status,
                                                                           pdw::Token *tok_scratch, int check_crc, uint32_t *next)
{
    __shared__ pdw::Tables T;
    pdw::Token *tok = tok_scratch + (size_t)blockIdx.x * PD_WAVE_TOKENS;
    // members are handed out one at a time from a counter (`next`, zeroed by the launcher): members differ in cost, and a launch may hold
    // several times more of them than there are waves — a static split would leave most waves idle behind the unlucky ones
    for (uint32_t i = blockIdx.x;;) {
        if (next) { uint32_t t = 0; if (threadIdx.x == 0) t = atomicAdd(next, 1u); i = (uint32_t)__builtin_amdgcn_readfirstlane((int)t); }
        if (i >= n_blk) break;
        const BlkDesc d = blk[i];
        int rc = 0;
        // inflate + the CRC-32 of the output against the member's trailer (an empty member still has one: 0)
        rc = pdw::inflate_member<pdw::DevWave>(comp + d.in_off, d.in_len, out + d.out_off, d.out_len, T, tok, nullptr, check_crc != 0);
        if (threadIdx.x == 0) status[i] = rc;
        __syncthreads();
        if (!next) i += gridDim.x;
    }
}

// The record chain of the inflated bytes, one wave per <= 64 KiB segment (pd_bamwalk.h): pass 1 finds every lane's first
// record and counts what it will emit; `only` (or null) lists the segments to (re)do.
__global__ __launch_bounds__(64) void k_walk_segments(const pdb2::Cfg cfg, pdb2::Seg *segs, uint32_t n_seg, pdb2::LaneOut *lanes,
                                                      const uint32_t *only, uint32_t n_only)
{
    const uint32_t k = blockIdx.x;
    if (k >= (only ? n_only : n_seg)) return;
    const uint32_t j = only ? only[k] : k;
    pdb2::walk_segment<pdw::DevWave>(cfg, segs[j], lanes + (size_t)j * 64);
}

// Test hook ("decode_spoil" = k): every k-th segment behind a unit's first walks again from a start that is no record — what a wrong guess
// of the header search looks like to whoever confirms the chain, planted far more often than real data ever shows one (one in ~1e8 records),
// so that the suite exercises the repair (k_chain_segments' repeats, the host's rounds) on the device.
__global__ __launch_bounds__(64) void k_spoil_segments(const pdb2::Cfg cfg, pdb2::Seg *segs, uint32_t n_seg, pdb2::LaneOut *lanes, uint32_t k)
{
    const uint32_t j = blockIdx.x;
    if (j >= n_seg || segs[j].unit_first || j % k != 0) return;
    const uint64_t h = segs[j].begin + 1 + (j % 7);
    pdb2::walk_segment<pdw::DevWave>(cfg, segs[j], lanes + (size_t)j * 64, &h);
}

// The chain across a batch's segments, confirmed by ONE wave on the device (pdb2::chain_device): segments whose guessed start is not
// where the chain before them ends walk again right here, every segment gets the places of its runs, and `out` says how many runs
// there are — or that the batch is out of the ordinary and the host must go through it (ChainOut::slow; pass 2 then writes nothing).
__global__ __launch_bounds__(64) void k_chain_segments(const pdb2::Cfg cfg, pdb2::Seg *segs, uint32_t n_seg, pdb2::LaneOut *lanes, const int *member_status,
                                                       uint32_t n_members, uint64_t cap_first, uint64_t cap_other, uint32_t max_redo, pdb2::ChainOut *out)
{
    pdb2::chain_device<pdw::DevWave>(segs, n_seg, member_status, n_members, cap_first, cap_other, max_redo,
        [&](uint32_t j, uint64_t start) { return pdb2::walk_segment<pdw::DevWave>(cfg, segs[j], lanes + (size_t)j * 64, &start); }, out);
}

// pass 2: the runs, at the offsets every segment was given (base_first / base_other: by the host, or by k_chain_segments — `gate`
// is then its verdict, and a batch it left to the host writes nothing)
__global__ __launch_bounds__(64) void k_emit_segments(const pdb2::Cfg cfg, const pdb2::Seg *segs, uint32_t n_seg, const pdb2::LaneOut *lanes,
                                                      pd_iv *first, pd_iv *other, pd_iv *far, const pdb2::ChainOut *gate)
{
    const uint32_t j = blockIdx.x;
    if (j >= n_seg) return;
    if (gate && gate->slow) return;
    pdb2::SegOut *so = cfg.c8.seg_out ? cfg.c8.seg_out + j : nullptr;       // compact emission: the segment's keys for the host's order check
    if ((segs[j].n_first | segs[j].n_other | segs[j].n_far) == 0) {
        if (so && threadIdx.x == 0) { so->first_key = pdb2::NONE; so->last_key = 0; so->unsorted = 0; so->n_long = 0; }
        return;
    }
    pdb2::emit_segment<pdw::DevWave>(cfg, segs[j], lanes + (size_t)j * 64, first, other, far, so);
}

// are the first runs of a batch in (tid, begin) order?  out[0] = 1 if not; out[2..3] / out[4..5] = first / last key
__global__ __launch_bounds__(256) void k_runs_sorted(const pd_iv *runs, uint64_t n, uint32_t *out)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = ((uint64_t)(uint32_t)runs[i].tid << 32) | (uint32_t)runs[i].beg;
    if (i > 0) {
        const uint64_t kp = ((uint64_t)(uint32_t)runs[i - 1].tid << 32) | (uint32_t)runs[i - 1].beg;
        if (k < kp) out[0] = 1;
    } else { out[2] = (uint32_t)k; out[3] = (uint32_t)(k >> 32); }
    if (i + 1 == n) { out[4] = (uint32_t)k; out[5] = (uint32_t)(k >> 32); }
}

} // namespace

namespace pdk {

void launch_runs_sorted(hipStream_t st, const pd_iv *runs, uint64_t n, uint32_t *out)
{
    if (!n) return;
    hipLaunchKernelGGL(k_runs_sorted, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, runs, n, out);
}

// the wave-cooperative decoder: n_wg persistent one-wave workgroups; `scratch` = bgzf_wave_scratch_bytes(n_wg) bytes
void launch_bgzf_inflate_wave(hipStream_t st, const uint8_t *comp, const pd_bgzf_block *blk, uint32_t n_blk, uint8_t *out,
                              int *status, void *scratch, unsigned n_wg, bool check_crc, uint32_t *next, bool zero_next)
{
    if (!n_blk) return;
    if (n_wg > n_blk) n_wg = n_blk;
    if (next && zero_next) (void)hipMemsetAsync(next, 0, 4, st);
    hipLaunchKernelGGL(k_inflate_wave, dim3(n_wg), dim3(64), 0, st, comp, blk, n_blk, out, status, (pdw::Token *)scratch, check_crc ? 1 : 0, next);
}
size_t bgzf_wave_scratch_bytes(unsigned n_wg) { return (size_t)n_wg * PD_WAVE_TOKENS * sizeof(pdw::Token) + 64; }

void launch_walk_segments(hipStream_t st, const pdb2::Cfg &cfg, pdb2::Seg *segs, uint32_t n_seg, pdb2::LaneOut *lanes,
                          const uint32_t *only, uint32_t n_only)
{
    const uint32_t n = only ? n_only : n_seg;
    if (!n) return;
    hipLaunchKernelGGL(k_walk_segments, dim3(n), dim3(64), 0, st, cfg, segs, n_seg, lanes, only, n_only);
}
void launch_emit_segments(hipStream_t st, const pdb2::Cfg &cfg, const pdb2::Seg *segs, uint32_t n_seg, const pdb2::LaneOut *lanes,
                          pd_iv *first, pd_iv *other, pd_iv *far, const pdb2::ChainOut *gate)
{
    if (!n_seg) return;
    hipLaunchKernelGGL(k_emit_segments, dim3(n_seg), dim3(64), 0, st, cfg, segs, n_seg, lanes, first, other, far, gate);
}
void launch_spoil_segments(hipStream_t st, const pdb2::Cfg &cfg, pdb2::Seg *segs, uint32_t n_seg, pdb2::LaneOut *lanes, uint32_t k)
{
    if (!n_seg || !k) return;
    hipLaunchKernelGGL(k_spoil_segments, dim3(n_seg), dim3(64), 0, st, cfg, segs, n_seg, lanes, k);
}
void launch_chain_segments(hipStream_t st, const pdb2::Cfg &cfg, pdb2::Seg *segs, uint32_t n_seg, pdb2::LaneOut *lanes, const int *member_status,
                           uint32_t n_members, uint64_t cap_first, uint64_t cap_other, uint32_t max_redo, pdb2::ChainOut *out)
{
    hipLaunchKernelGGL(k_chain_segments, dim3(1), dim3(64), 0, st, cfg, segs, n_seg, lanes, member_status, n_members, cap_first, cap_other, max_redo, out);
}

void launch_bgzf_inflate(hipStream_t st, const uint8_t *comp, const pd_bgzf_block *blk, uint32_t n_blk, uint8_t *out,
                         int *status, void *scratch)
{
    // measured (tools/bgzf_gpu_bench.py): with everything in global memory more waves fit a CU and the
    // kernel is faster (30 vs 23 GB/s at 98 K blocks) than with the fast tables in LDS
    hipLaunchKernelGGL(k_inflate_blocks<false>, dim3((n_blk + 63) / 64), dim3(64), 0, st, comp, blk, n_blk, out, status,
                       (pdi::Tables *)scratch);
}
size_t bgzf_scratch_bytes(uint32_t n_blk) { return (size_t)n_blk * sizeof(pdi::Tables); }

} // namespace pdk

extern "C" int pd_x_bgzf_inflate(int device, const void *host_bgzf, size_t n_bytes, void *host_out, size_t out_cap,
                                 size_t *out_len, int variant, int reps, double *kernel_ms, uint32_t *n_blocks_out)
{
    if (!host_bgzf || !out_len) return PD_EINVAL;
    if (hipSetDevice(device) != hipSuccess) return PD_ENODEV;
    const uint8_t *p = (const uint8_t *)host_bgzf;
    std::vector<BlkDesc> blks;
    size_t o = 0; uint64_t uo = 0;
    while (o + 18 <= n_bytes) {
        if (p[o] != 0x1f || p[o + 1] != 0x8b || !(p[o + 3] & 4)) return PD_EINVAL;
        const uint32_t xlen = p[o + 10] | (p[o + 11] << 8);
        uint32_t bsize = 0;
        for (uint32_t x = 12; x + 4 <= 12 + xlen;) {
            const uint32_t sl = p[o + x + 2] | (p[o + x + 3] << 8);
            if (p[o + x] == 'B' && p[o + x + 1] == 'C' && sl == 2) { bsize = (p[o + x + 4] | (p[o + x + 5] << 8)) + 1; break; }
            x += 4 + sl;
        }
        if (!bsize || o + bsize > n_bytes) return PD_EINVAL;
        const uint32_t isize = p[o + bsize - 4] | (p[o + bsize - 3] << 8) | (p[o + bsize - 2] << 16) | ((uint32_t)p[o + bsize - 1] << 24);
        blks.push_back(BlkDesc{o + 12 + xlen, uo, bsize - 12 - xlen - 8, isize});
        uo += isize; o += bsize;
    }
    *out_len = uo;
    if (n_blocks_out) *n_blocks_out = (uint32_t)blks.size();
    if (host_out && out_cap < uo) return PD_EINVAL;
    uint8_t *d_in = nullptr, *d_out = nullptr; BlkDesc *d_blk = nullptr; int *d_st = nullptr; pdi::Tables *d_scr = nullptr;
    const uint32_t nb = (uint32_t)blks.size();
    int rc = PD_OK;
    hipEvent_t e0, e1;
    // variant >= 2: the wave-cooperative decoder with ((variant >> 4) & 0xff, default 16) persistent waves per CU; + 0x8000: without the CRC-32 check
    unsigned n_wg = 0;
    const bool no_crc = (variant & 0x8000) != 0;
    variant &= 0x7fff;
    if (variant >= 2) {
        hipDeviceProp_t pr;
        if (hipGetDeviceProperties(&pr, device) != hipSuccess) return PD_ENODEV;
        const unsigned per_cu = (variant >> 4) ? (unsigned)(variant >> 4) : 20u;
        n_wg = (unsigned)pr.multiProcessorCount * per_cu;
    }
    if (hipMalloc(&d_in, n_bytes + 16) != hipSuccess || hipMalloc(&d_out, uo + 16) != hipSuccess ||
        hipMalloc(&d_blk, (size_t)nb * sizeof(BlkDesc) + 16) != hipSuccess || hipMalloc(&d_st, (size_t)nb * 4 + 16) != hipSuccess) rc = PD_ENOMEM;
    if (rc == PD_OK && hipMalloc(&d_scr, variant >= 2 ? pdk::bgzf_wave_scratch_bytes(n_wg) : (size_t)nb * sizeof(pdi::Tables)) != hipSuccess) rc = PD_ENOMEM;
#define HIPV(x) do { if ((x) != hipSuccess) rc = PD_EHIP; } while (0)
    if (rc == PD_OK) {
        HIPV(hipMemcpy(d_in, p, n_bytes, hipMemcpyHostToDevice));
        HIPV(hipMemcpy(d_blk, blks.data(), (size_t)nb * sizeof(BlkDesc), hipMemcpyHostToDevice));
        HIPV(hipEventCreate(&e0)); HIPV(hipEventCreate(&e1));
        for (int r = 0; r < reps + 1; ++r) {
            if (r == 1 || reps == 0) HIPV(hipEventRecord(e0, 0));
            const dim3 g((nb + 63) / 64), b(64);
            if (variant >= 2) pdk::launch_bgzf_inflate_wave(0, d_in, d_blk, nb, d_out, d_st, d_scr, n_wg, !no_crc, (uint32_t *)(d_st + nb), true);
            else if (variant == 0) hipLaunchKernelGGL(k_inflate_blocks<true>, g, b, 0, 0, d_in, d_blk, nb, d_out, d_st, d_scr);
            else hipLaunchKernelGGL(k_inflate_blocks<false>, g, b, 0, 0, d_in, d_blk, nb, d_out, d_st, d_scr);
        }
        HIPV(hipEventRecord(e1, 0));
        if (hipEventSynchronize(e1) != hipSuccess) rc = PD_EHIP;
        float ms = 0; HIPV(hipEventElapsedTime(&ms, e0, e1));
        if (kernel_ms) *kernel_ms = ms / (reps > 0 ? reps : 1);
        std::vector<int> st(nb);
        HIPV(hipMemcpy(st.data(), d_st, (size_t)nb * 4, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < nb; ++i) if (st[i] != 0) { rc = PD_EHIP; fprintf(stderr, "pd_x_bgzf_inflate: block %u failed with %d\n", i, st[i]); break; }
        if (host_out && rc == PD_OK) HIPV(hipMemcpy(host_out, d_out, uo, hipMemcpyDeviceToHost));
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    (void)hipFree(d_in); (void)hipFree(d_out); (void)hipFree(d_blk); (void)hipFree(d_st); if (d_scr) (void)hipFree(d_scr);
    return rc;
}
