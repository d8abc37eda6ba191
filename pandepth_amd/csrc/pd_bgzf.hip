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
#include "pd_bamdev_core.h"
#include "pd_kernels.h"
#include "../../include/pandepth_amd.h"

namespace {

typedef pd_bgzf_block BlkDesc;      // { in_off, out_off, in_len, out_len }
#define PD_WAVE_TOKENS (65536 / 3 + 64)

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
__global__ __launch_bounds__(64) void k_inflate_wave(const uint8_t *comp, const BlkDesc *blk, uint32_t n_blk, uint8_t *out, int *status,
                                                     pdw::Token *tok_scratch)
{
    __shared__ pdw::Tables T;
    pdw::Token *tok = tok_scratch + (size_t)blockIdx.x * PD_WAVE_TOKENS;
    for (uint32_t i = blockIdx.x; i < n_blk; i += gridDim.x) {
        const BlkDesc d = blk[i];
        int rc = 0;
        if (d.out_len) rc = pdw::inflate_block<pdw::DevWave>(comp + d.in_off, d.in_len, out + d.out_off, d.out_len, T, tok, nullptr);
        if (threadIdx.x == 0) status[i] = rc;
        __syncthreads();
    }
}

// thread per unit: record offsets (sequential by nature: each record's length says where the next
// starts), plus the checks that send a unit back to the host (record past the inflated bytes,
// CIGAR in the CG tag)
__global__ __launch_bounds__(64) void k_walk_units(const uint8_t *buf, pdb::Unit *units, uint32_t n_units,
                                                   uint64_t *rec_off, uint64_t rec_cap, const int *blk_status,
                                                   const uint32_t *unit_first_blk, const uint32_t *unit_n_blk)
{
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n_units) return;
    pdb::Unit u = units[i];
    int bad = 0;
    for (uint32_t b = 0; b < unit_n_blk[i]; ++b) if (blk_status[unit_first_blk[i] + b] != 0) bad = 1;
    if (bad) { u.n_rec = 0; u.status = 2; units[i] = u; return; }
    pdb::walk_unit(buf, u, rec_off, rec_cap);
    if (u.status == 0) {
        for (uint32_t k = 0; k < u.n_rec; ++k) {          // long-CIGAR placeholder (SAM spec §4.2.2) -> host
            const uint8_t *rec = buf + rec_off[u.rec_base + k];
            if (pdb::ld16(rec + 16) == 2) {
                const uint8_t *cg = rec + 36 + rec[12];
                if ((pdb::ld32(cg) & 0xf) == 4 && (pdb::ld32(cg) >> 4) == pdb::ld32(rec + 20) && (pdb::ld32(cg + 4) & 0xf) == 3) { u.status = 1; break; }
            }
        }
    }
    if (u.status != 0) u.n_rec = 0;
    units[i] = u;
}

// dense record numbering over the units that stay on the device
__global__ void k_unit_bases(const pdb::Unit *units, uint32_t n_units, uint64_t *dense_base)
{
    uint64_t acc = 0;
    for (uint32_t i = 0; i < n_units; ++i) { dense_base[i] = acc; acc += units[i].n_rec; }
    dense_base[n_units] = acc;
}

// thread per record: first run into the dense, position-sorted array; the other runs appended to
// the batch's unordered list with ONE atomic per wave
__global__ __launch_bounds__(256) void k_parse_records(const uint8_t *buf, const pdb::Unit *units, uint32_t n_units,
                                                       const uint64_t *dense_base, const uint64_t *rec_off,
                                                       pdb::Filter f, const uint32_t *contig_len, pd_iv *first,
                                                       pd_iv *other, uint32_t other_cap, uint32_t *other_count, uint32_t *err)
{
    const uint64_t n = dense_base[n_units];
    const uint64_t j = blockIdx.x * (uint64_t)256 + threadIdx.x;
    const bool live = j < n;
    const uint8_t *rec = nullptr;
    if (live) {
        uint32_t lo = 0, hi = n_units;                  // last unit with dense_base <= j
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (dense_base[mid] <= j) lo = mid; else hi = mid; }
        rec = buf + rec_off[units[lo].rec_base + (j - dense_base[lo])];
    }
    uint32_t n_other = 0;
    pd_iv fr{0, 0, 0};
    if (live) pdb::parse_record(rec, f, contig_len, &fr, [&](pd_iv) { ++n_other; });
    // wave-level exclusive scan of the counts, one atomic for the wave
    uint32_t incl = n_other;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(incl, o); if ((threadIdx.x & 63) >= o) incl += y; }
    const uint32_t total = __shfl(incl, 63);
    uint32_t base = 0;
    if (total) {
        if ((threadIdx.x & 63) == 0) base = atomicAdd(other_count, total);
        base = __shfl(base, 0);
    }
    if (live) {
        first[j] = fr;
        if (n_other) {
            uint32_t w = base + incl - n_other;
            pd_iv dummy;
            pdb::parse_record(rec, f, contig_len, &dummy, [&](pd_iv v) { if (w < other_cap) other[w] = v; else atomicOr(err, 1u); ++w; });
        }
    }
}

} // namespace

namespace pdk {

// the wave-cooperative decoder: n_wg persistent one-wave workgroups; `scratch` = bgzf_wave_scratch_bytes(n_wg) bytes
void launch_bgzf_inflate_wave(hipStream_t st, const uint8_t *comp, const pd_bgzf_block *blk, uint32_t n_blk, uint8_t *out,
                              int *status, void *scratch, unsigned n_wg)
{
    if (!n_blk) return;
    if (n_wg > n_blk) n_wg = n_blk;
    hipLaunchKernelGGL(k_inflate_wave, dim3(n_wg), dim3(64), 0, st, comp, blk, n_blk, out, status, (pdw::Token *)scratch);
}
size_t bgzf_wave_scratch_bytes(unsigned n_wg) { return (size_t)n_wg * PD_WAVE_TOKENS * sizeof(pdw::Token) + 64; }

void launch_bgzf_inflate(hipStream_t st, const uint8_t *comp, const pd_bgzf_block *blk, uint32_t n_blk, uint8_t *out,
                         int *status, void *scratch)
{
    // measured (tools/bgzf_gpu_bench.py): with everything in global memory more waves fit a CU and the
    // kernel is faster (30 vs 23 GB/s at 98 K blocks) than with the fast tables in LDS
    hipLaunchKernelGGL(k_inflate_blocks<false>, dim3((n_blk + 63) / 64), dim3(64), 0, st, comp, blk, n_blk, out, status,
                       (pdi::Tables *)scratch);
}
size_t bgzf_scratch_bytes(uint32_t n_blk) { return (size_t)n_blk * sizeof(pdi::Tables); }

void launch_bam_walk(hipStream_t st, const uint8_t *buf, void *units, uint32_t n_units, uint64_t *rec_off, uint64_t rec_cap,
                     const int *blk_status, const uint32_t *unit_first_blk, const uint32_t *unit_n_blk, uint64_t *dense_base)
{
    hipLaunchKernelGGL(k_walk_units, dim3((n_units + 63) / 64), dim3(64), 0, st, buf, (pdb::Unit *)units, n_units, rec_off,
                       rec_cap, blk_status, unit_first_blk, unit_n_blk);
    hipLaunchKernelGGL(k_unit_bases, dim3(1), dim3(1), 0, st, (const pdb::Unit *)units, n_units, dense_base);
}

void launch_bam_parse(hipStream_t st, const uint8_t *buf, const void *units, uint32_t n_units, const uint64_t *dense_base,
                      uint64_t n_rec_upper, const uint64_t *rec_off, uint32_t flag_mask, int32_t min_mapq, int32_t n_contigs,
                      const uint32_t *contig_len, pd_iv *first, pd_iv *other, uint32_t other_cap, uint32_t *other_count,
                      uint32_t *err)
{
    if (!n_rec_upper) return;
    pdb::Filter f{flag_mask, min_mapq, n_contigs};
    hipLaunchKernelGGL(k_parse_records, dim3((unsigned)((n_rec_upper + 255) / 256)), dim3(256), 0, st, buf,
                       (const pdb::Unit *)units, n_units, dense_base, rec_off, f, contig_len, first, other, other_cap,
                       other_count, err);
}

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
    // variant >= 2: the wave-cooperative decoder with (variant >> 4, default 16) persistent waves per CU
    unsigned n_wg = 0;
    if (variant >= 2) {
        hipDeviceProp_t pr;
        if (hipGetDeviceProperties(&pr, device) != hipSuccess) return PD_ENODEV;
        const unsigned per_cu = (variant >> 4) ? (unsigned)(variant >> 4) : 16u;
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
            if (variant >= 2) pdk::launch_bgzf_inflate_wave(0, d_in, d_blk, nb, d_out, d_st, d_scr, n_wg);
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
