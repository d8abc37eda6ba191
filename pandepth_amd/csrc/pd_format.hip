// pd_format.hip — per-site rows on the device (include/pandepth_amd.h: pd_format_sites).  The reference writes one line per
// base, "<contig>\t<0-based index>\t<depth>\n" (PD:4278-4281); the host used to read the cells back (4 B each) and format
// them on its threads.  Here the text itself is produced in HBM — lengths, an exclusive scan of the lengths, the bytes — and
// what crosses PCIe is the text the gzip stage consumes.  HBM-bound byte work: 4 B read + ~15 B written per cell.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pd_kernels.h"

namespace pdk {
namespace {
constexpr int FWG = 256;                 // threads per workgroup
constexpr int FPER = 16;                 // consecutive cells per thread
constexpr uint32_t FBLK = FWG * FPER;    // cells per workgroup

__device__ __forceinline__ uint32_t dec_digits(uint32_t v)
{
    return 1u + (v >= 10u) + (v >= 100u) + (v >= 1000u) + (v >= 10000u) + (v >= 100000u) + (v >= 1000000u) + (v >= 10000000u) +
           (v >= 100000000u) + (v >= 1000000000u);
}
// writes v in decimal, `nd` digits, ending just before p_end
__device__ __forceinline__ void put_dec(char *p_end, uint32_t v, uint32_t nd)
{
    for (uint32_t k = 0; k < nd; ++k) { const uint32_t q = v / 10u; *--p_end = (char)('0' + (v - q * 10u)); v = q; }
}
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t x)
{
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
    return x;
}

// bytes of the rows of this thread's cells; the workgroup's exclusive offsets
template <bool WRITE>
__global__ __launch_bounds__(FWG) void k_site_rows(const uint32_t *depth, uint32_t first_index, uint64_t n, uint32_t name_len,
                                                  const char *name, uint32_t *blk_bytes, const uint64_t *blk_off, char *text)
{
    __shared__ uint32_t wsum[FWG / 64];
    const uint64_t c0 = (uint64_t)blockIdx.x * FBLK + (uint64_t)threadIdx.x * FPER;
    uint32_t d[FPER], len[FPER], mine = 0;
#pragma unroll
    for (int k = 0; k < FPER; ++k) {
        const uint64_t c = c0 + k;
        d[k] = c < n ? depth[c] : 0u;
        len[k] = c < n ? name_len + 3u + dec_digits(first_index + (uint32_t)c) + dec_digits(d[k]) : 0u;
        mine += len[k];
    }
    const uint32_t incl = wave_incl_scan_u32(mine);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    uint32_t base = incl - mine;
    for (int k = 0; k < wv; ++k) base += wsum[k];
    if (!WRITE) {
        if (threadIdx.x == FWG - 1) blk_bytes[blockIdx.x] = base + mine;
        return;
    }
    char *p = text + blk_off[blockIdx.x] + base;
#pragma unroll 1
    for (int k = 0; k < FPER; ++k) {
        if (!len[k]) break;
        for (uint32_t j = 0; j < name_len; ++j) p[j] = name[j];
        const uint32_t idx = first_index + (uint32_t)(c0 + k);
        const uint32_t n1 = dec_digits(idx), n2 = dec_digits(d[k]);
        char *q = p + name_len;
        *q++ = '\t'; q += n1; put_dec(q, idx, n1);
        *q++ = '\t'; q += n2; put_dec(q, d[k], n2);
        *q = '\n';
        p += len[k];
    }
}

// exclusive scan of the workgroups' byte counts (one workgroup; a few thousand entries at most per call)
__global__ __launch_bounds__(FWG) void k_site_scan(const uint32_t *blk_bytes, uint32_t n_blk, uint64_t *blk_off)
{
    __shared__ unsigned long long wsum[FWG / 64];
    __shared__ unsigned long long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < n_blk; b0 += FWG) {
        const uint32_t i = b0 + threadIdx.x;
        const unsigned long long v = i < n_blk ? blk_bytes[i] : 0ull;
        unsigned long long x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned long long y = __shfl_up(x, o); if ((int)(threadIdx.x & 63) >= o) x += y; }
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        if (lane == 63) wsum[wv] = x;
        __syncthreads();
        unsigned long long base = carry;
        for (int k = 0; k < wv; ++k) base += wsum[k];
        if (i < n_blk) blk_off[i] = base + x - v;
        __syncthreads();
        if (threadIdx.x == FWG - 1) carry = base + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) blk_off[n_blk] = carry;
}

// ---- rows of the `-w` table (PD:4381-4388): "<name>\t<start>\t<end>\t<length>\t<covered>\t<depth>\t<coverage %.2f>\t<mean %.2f>\n" ----
// One row per thread.  The two %.2f columns are what glibc's correctly rounded printf prints for the doubles the reference computes
// (covered * 100.0 / length and depth * 1.0 / length, depth being the reference's `int`): the double is M * 2^e exactly, so the value
// times 100 is (M * 100) >> -e with an exact remainder, rounded half to even (host/report.cpp: fmt2_to — the same arithmetic).
struct Fmt2 { uint64_t ip; uint32_t fp; bool neg; };
__device__ __forceinline__ Fmt2 fmt2_of(double v)
{
    uint64_t bits = (uint64_t)__double_as_longlong(v);
    Fmt2 r; r.neg = (bits >> 63) != 0;
    bits &= ~(1ull << 63);
    const int be = (int)((bits >> 52) & 0x7ff);
    uint64_t m = bits & ((1ull << 52) - 1);
    int e;
    if (be == 0) e = -1074; else { m |= 1ull << 52; e = be - 1075; }
    const uint64_t n = m * 100ull;                                // < 2^60 (|v| < 2^52 for every table value)
    const int sh = -e;
    uint64_t q = 0;
    if (sh < 64) {
        q = n >> sh;
        const uint64_t rem = n & ((1ull << sh) - 1ull), half = 1ull << (sh - 1);
        if (rem > half || (rem == half && (q & 1ull))) ++q;
    }
    r.ip = q / 100ull; r.fp = (uint32_t)(q % 100ull);
    return r;
}
__device__ __forceinline__ uint32_t dec_digits64(uint64_t v)
{
    uint32_t n = 1;
    while (v >= 10ull) { v /= 10ull; ++n; }
    return n;
}
__device__ __forceinline__ char *put_dec64(char *p, uint64_t v, uint32_t nd)   // writes nd digits at p, returns p + nd
{
    char *e = p + nd;
    for (uint32_t k = 0; k < nd; ++k) { const uint64_t q = v / 10ull; *--e = (char)('0' + (uint32_t)(v - q * 10ull)); v = q; }
    return p + nd;
}
__device__ __forceinline__ uint32_t fmt2_len(const Fmt2 &f) { return (f.neg ? 1u : 0u) + dec_digits64(f.ip) + 3u; }
__device__ __forceinline__ char *put_fmt2(char *p, const Fmt2 &f)
{
    if (f.neg) *p++ = '-';
    p = put_dec64(p, f.ip, dec_digits64(f.ip));
    *p++ = '.'; *p++ = (char)('0' + f.fp / 10u); *p++ = (char)('0' + f.fp % 10u);
    return p;
}

constexpr int WPER = 4;                                         // rows per thread
constexpr uint32_t WBLK = FWG * WPER;

template <bool WRITE>
__global__ __launch_bounds__(FWG) void k_window_rows(const uint32_t *cover, const unsigned long long *sum, uint64_t row_first, uint64_t n_rows, uint32_t w,
                                                    uint32_t clen, uint32_t name_len, const char *name, uint32_t *blk_bytes, const uint64_t *blk_off,
                                                    char *text)
{
    __shared__ uint32_t wsum[FWG / 64];
    const uint64_t r0 = (uint64_t)blockIdx.x * WBLK + (uint64_t)threadIdx.x * WPER;
    uint32_t len[WPER], mine = 0;
#pragma unroll
    for (int k = 0; k < WPER; ++k) {
        const uint64_t r = r0 + k;
        len[k] = 0;
        if (r < n_rows) {
            const uint64_t row = row_first + r;
            const int64_t j = 1 + (int64_t)row * w;
            int64_t end = j - 1 + w; if (end > (int64_t)clen) end = clen;
            const int64_t L = end - j + 1;
            const int32_t c = (int32_t)cover[r], d = (int32_t)sum[r];          // `int GeneDepth` (PD:4364)
            const Fmt2 f1 = fmt2_of(c * 100.0 / (double)L), f2 = fmt2_of(d * 1.0 / (double)L);
            len[k] = name_len + 8u + dec_digits64((uint64_t)j) + dec_digits64((uint64_t)end) + dec_digits64((uint64_t)L) +
                     (c < 0 ? 1u : 0u) + dec_digits64((uint64_t)(c < 0 ? -(int64_t)c : (int64_t)c)) +
                     (d < 0 ? 1u : 0u) + dec_digits64((uint64_t)(d < 0 ? -(int64_t)d : (int64_t)d)) + fmt2_len(f1) + fmt2_len(f2);
        }
        mine += len[k];
    }
    const uint32_t incl = wave_incl_scan_u32(mine);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    uint32_t base = incl - mine;
    for (int k = 0; k < wv; ++k) base += wsum[k];
    if (!WRITE) {
        if (threadIdx.x == FWG - 1) blk_bytes[blockIdx.x] = base + mine;
        return;
    }
    char *p = text + blk_off[blockIdx.x] + base;
#pragma unroll 1
    for (int k = 0; k < WPER; ++k) {
        if (!len[k]) break;
        const uint64_t r = r0 + k, row = row_first + r;
        const int64_t j = 1 + (int64_t)row * w;
        int64_t end = j - 1 + w; if (end > (int64_t)clen) end = clen;
        const int64_t L = end - j + 1;
        const int32_t c = (int32_t)cover[r], d = (int32_t)sum[r];
        const Fmt2 f1 = fmt2_of(c * 100.0 / (double)L), f2 = fmt2_of(d * 1.0 / (double)L);
        char *q = p;
        for (uint32_t x = 0; x < name_len; ++x) q[x] = name[x];
        q += name_len;
        *q++ = '\t'; q = put_dec64(q, (uint64_t)j, dec_digits64((uint64_t)j));
        *q++ = '\t'; q = put_dec64(q, (uint64_t)end, dec_digits64((uint64_t)end));
        *q++ = '\t'; q = put_dec64(q, (uint64_t)L, dec_digits64((uint64_t)L));
        *q++ = '\t'; if (c < 0) *q++ = '-';
        { const uint64_t a = (uint64_t)(c < 0 ? -(int64_t)c : (int64_t)c); q = put_dec64(q, a, dec_digits64(a)); }
        *q++ = '\t'; if (d < 0) *q++ = '-';
        { const uint64_t a = (uint64_t)(d < 0 ? -(int64_t)d : (int64_t)d); q = put_dec64(q, a, dec_digits64(a)); }
        *q++ = '\t'; q = put_fmt2(q, f1);
        *q++ = '\t'; q = put_fmt2(q, f2);
        *q = '\n';
        p += len[k];
    }
}
} // namespace

uint32_t site_rows_blocks(uint64_t n) { return (uint32_t)((n + FBLK - 1) / FBLK); }

void launch_site_rows(hipStream_t st, const uint32_t *depth, uint32_t first_index, uint64_t n, uint32_t name_len, const char *dev_name,
                      uint32_t *blk_bytes, uint64_t *blk_off, char *text, bool write)
{
    const uint32_t nb = site_rows_blocks(n);
    if (!nb) return;
    if (!write) {
        hipLaunchKernelGGL((k_site_rows<false>), dim3(nb), dim3(FWG), 0, st, depth, first_index, n, name_len, dev_name, blk_bytes, blk_off, text);
        hipLaunchKernelGGL(k_site_scan, dim3(1), dim3(FWG), 0, st, blk_bytes, nb, blk_off);
    } else {
        hipLaunchKernelGGL((k_site_rows<true>), dim3(nb), dim3(FWG), 0, st, depth, first_index, n, name_len, dev_name, blk_bytes, blk_off, text);
    }
}
uint32_t window_rows_blocks(uint64_t n) { return (uint32_t)((n + WBLK - 1) / WBLK); }

void launch_window_rows(hipStream_t st, const uint32_t *cover, const unsigned long long *sum, uint64_t row_first, uint64_t n_rows, uint32_t w, uint32_t clen,
                        uint32_t name_len, const char *dev_name, uint32_t *blk_bytes, uint64_t *blk_off, char *text, bool write)
{
    const uint32_t nb = window_rows_blocks(n_rows);
    if (!nb) return;
    if (!write) {
        hipLaunchKernelGGL((k_window_rows<false>), dim3(nb), dim3(FWG), 0, st, cover, sum, row_first, n_rows, w, clen, name_len, dev_name, blk_bytes, blk_off, text);
        hipLaunchKernelGGL(k_site_scan, dim3(1), dim3(FWG), 0, st, blk_bytes, nb, blk_off);
    } else {
        hipLaunchKernelGGL((k_window_rows<true>), dim3(nb), dim3(FWG), 0, st, cover, sum, row_first, n_rows, w, clen, name_len, dev_name, blk_bytes, blk_off, text);
    }
}

} // namespace pdk
