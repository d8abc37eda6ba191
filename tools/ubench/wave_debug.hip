// tools/ubench/wave_debug.hip — DEVELOPMENT TOOL (not product code): runs pd_inflate_wave.h on the GPU over the members of a BGZF file the way
// k_inflate_wave does (one-wave workgroups, members handed out from a counter, every member's CRC checked), compares the first CHECK members
// with zlib, and — unless built with -DNO_TICKS — says where a member's time goes: the decoder's PW_TICK(phase) hooks charge the shader clock
// since the previous tick to a phase, summed per workgroup in LDS and added up once when the workgroup ends.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/wave_debug.hip -lz -o tools/ubench/wave_debug      (tools/ubench/build_wave_variants.sh builds
//   several: -DNO_TICKS, -DPD_INFLATE_MIN_WAVES=n, -DPDW_HEADER='"/path/to/an/older/pd_inflate_wave.h"' for A/B runs on one box)
//   [CHECK=n] [MARKS=1] wave_debug file.bam [n_wg] [max_blocks] [seconds]        MARKS: host-visible progress marks + a polling watchdog (hang hunting)
// Results of round 5: profiles/r05_inflate_ticks.txt (tools/calls/r5_call16.sh ... r5_call32.sh).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <zlib.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>
#define PW_MARK(code, val) do { if (g_dbg && (threadIdx.x & 63) == 0) { g_dbg[blockIdx.x * 8 + 0] = (code); g_dbg[blockIdx.x * 8 + 1] = (val); g_dbg[blockIdx.x * 8 + 2] += 1; } } while (0)
__device__ volatile unsigned *g_dbg;
__device__ unsigned long long g_ticks[16];
// ticks are summed per workgroup in LDS and added to the totals once, when the workgroup ends (an atomic per tick on thirteen
// addresses serialised 5 120 waves: the kernel ran at 9 GB/s and the ticks measured the atomics)
__shared__ unsigned long long s_ticks[16];
__shared__ long long s_last;
#ifdef NO_TICKS
#define PW_TICK(phase) do { } while (0)
#else
#define PW_TICK(phase) do { if ((threadIdx.x & 63) == 0) { const long long now_ = wall_clock64(); s_ticks[phase] += (unsigned long long)(now_ - s_last); s_last = now_; } } while (0)
#endif
#ifndef PDW_HEADER
#define PDW_HEADER "../../pandepth_amd/csrc/pd_inflate_wave.h"      /* (-DPDW_HEADER=... : an older version of the decoder, for A/B runs) */
#endif
#include PDW_HEADER
struct Blk { unsigned long long in_off, out_off; unsigned in_len, out_len; };
#ifdef PD_TOK_CAP
#define NTOK (PD_TOK_CAP + 63)
#else
#define NTOK (65536 / 3 + 64)
#endif
#ifndef PD_INFLATE_MIN_WAVES
#define PD_INFLATE_MIN_WAVES 5
#endif
__global__ __launch_bounds__(64, PD_INFLATE_MIN_WAVES) void k_dbg(const uint8_t *comp, const Blk *blk, unsigned n_blk, uint8_t *out, int *status, pdw::Token *tok_scratch,
                                            unsigned *next, volatile unsigned *dbg)
{
    __shared__ pdw::Tables T;
    const bool getenv_crc = true;
    if (threadIdx.x < 16) s_ticks[threadIdx.x] = 0;
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 0) g_dbg = dbg;
    pdw::Token *tok = tok_scratch + (size_t)blockIdx.x * NTOK;
    // (as k_inflate_wave: members handed out from a counter, the CRC of every member checked)
    for (;;) {
        unsigned t = 0; if (threadIdx.x == 0) t = atomicAdd(next, 1u);
        const unsigned i = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
        if (i >= n_blk) break;
        if (threadIdx.x == 0 && dbg) { dbg[blockIdx.x * 8 + 3] = i; dbg[blockIdx.x * 8 + 4] = 1; }
        const Blk d = blk[i];
        int rc = 0;
        if (threadIdx.x == 0) s_last = wall_clock64();
        rc = pdw::inflate_member<pdw::DevWave>(comp + d.in_off, d.in_len, out + d.out_off, d.out_len, T, tok, nullptr, getenv_crc);
        PW_TICK(12);
        if (threadIdx.x == 0) { status[i] = rc; if (dbg) dbg[blockIdx.x * 8 + 4] = 2; }
        __syncthreads();
    }
    if (threadIdx.x < 16 && s_ticks[threadIdx.x]) atomicAdd(&g_ticks[threadIdx.x], s_ticks[threadIdx.x]);
}
int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    unsigned n_wg = argc > 2 ? atoi(argv[2]) : 1, max_blocks = argc > 3 ? atoi(argv[3]) : 1000000; int secs = argc > 4 ? atoi(argv[4]) : 10;
    FILE *f = fopen(argv[1], "rb"); if (!f) return 2;
    std::vector<unsigned char> d; unsigned char buf[1 << 16]; size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + n);
    fclose(f);
    std::vector<Blk> blks; size_t o = 0; unsigned long long uo = 0;
    while (o + 18 <= d.size() && blks.size() < max_blocks) {
        const unsigned char *p = d.data() + o;
        const unsigned xlen = p[10] | (p[11] << 8), bsize = (p[16] | (p[17] << 8)) + 1;
        const unsigned isize = p[bsize - 4] | (p[bsize - 3] << 8) | (p[bsize - 2] << 16) | ((unsigned)p[bsize - 1] << 24);
        blks.push_back(Blk{o + 12 + xlen, uo, bsize - 12 - xlen - 8, isize}); uo += isize; o += bsize;
    }
    const unsigned nb = blks.size();
    if (n_wg > nb) n_wg = nb;
    uint8_t *d_in, *d_out; Blk *d_blk; int *d_st; pdw::Token *d_tok; unsigned *d_next; unsigned *h_dbg;
    hipMalloc(&d_in, d.size() + 16); hipMalloc(&d_out, uo + 16); hipMalloc(&d_blk, nb * sizeof(Blk)); hipMalloc(&d_st, nb * 4 + 4);
    hipMalloc(&d_tok, (size_t)n_wg * NTOK * sizeof(pdw::Token)); hipMalloc(&d_next, 64);
    hipHostMalloc(&h_dbg, n_wg * 32, hipHostMallocMapped); memset(h_dbg, 0, n_wg * 32);
    hipMemcpy(d_in, d.data(), d.size(), hipMemcpyHostToDevice); hipMemcpy(d_blk, blks.data(), nb * sizeof(Blk), hipMemcpyHostToDevice);
    hipMemset(d_next, 0, 64); hipMemset(d_st, 0xff, nb * 4); hipMemset(d_out, 0xEE, uo);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const bool marks = getenv("MARKS") != nullptr;
    // STREAMS=K BATCH=n: the members in launches of n on K streams (every stream its own token scratch, every launch its own counter) — what the
    // decode pipeline's batch-sized launches on several streams can reach together, without the host, the copies and the other kernels
    const int n_streams = getenv("STREAMS") ? atoi(getenv("STREAMS")) : 0;
    const unsigned batch = getenv("BATCH") ? atoi(getenv("BATCH")) : 3200;
    bool done = false;
    if (n_streams > 0) {
        std::vector<hipStream_t> sts(n_streams); std::vector<pdw::Token *> toks(n_streams);
        for (int k = 0; k < n_streams; ++k) { hipStreamCreateWithFlags(&sts[k], hipStreamNonBlocking); hipMalloc(&toks[k], (size_t)n_wg * NTOK * sizeof(pdw::Token)); }
        const unsigned n_launch = (nb + batch - 1) / batch;
        unsigned *d_ctr; hipMalloc(&d_ctr, (size_t)n_launch * 4); hipMemset(d_ctr, 0, (size_t)n_launch * 4);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0); hipEventSynchronize(e0);
        const auto t0 = std::chrono::steady_clock::now();
        // COPY=1: beside the launches a thread copies a 32 MB page-locked buffer to the device again and again on a stream of its own: what a batch's copy takes
        // beside the decoder's kernels IN THE SAME PROCESS
        std::atomic<bool> copy_stop{false}; std::vector<float> copy_ms; std::thread copier;
        if (getenv("COPY")) {
            copier = std::thread([&]() {
                void *h = nullptr, *dv = nullptr; hipStream_t cs; hipEvent_t a, b;
                const size_t n = (size_t)32 << 20;
                hipHostMalloc(&h, n, hipHostMallocDefault); memset(h, 1, n); hipMalloc(&dv, n); hipStreamCreateWithFlags(&cs, hipStreamNonBlocking); hipEventCreate(&a); hipEventCreate(&b);
                while (!copy_stop.load()) {
                    hipEventRecord(a, cs); hipMemcpyAsync(dv, h, n, hipMemcpyHostToDevice, cs); hipEventRecord(b, cs); hipEventSynchronize(b);
                    float ms = 0; hipEventElapsedTime(&ms, a, b); copy_ms.push_back(ms);
                    usleep(300);
                }
            });
            usleep(200000);
        }
        const int repeat = getenv("REPEAT") ? atoi(getenv("REPEAT")) : 1;      // (REPEAT=n: the whole file n times over — a steady load for something measured beside it)
        for (int rp = 0; rp < repeat; ++rp) {
        if (rp) { for (int k = 0; k < n_streams; ++k) hipStreamSynchronize(sts[k]); hipMemset(d_ctr, 0, (size_t)n_launch * 4); }
        for (unsigned j = 0; j < n_launch; ++j) {
            const unsigned first = j * batch, n = nb - first < batch ? nb - first : batch;
            hipLaunchKernelGGL(k_dbg, dim3(n_wg < n ? n_wg : n), dim3(64), 0, sts[j % n_streams], d_in, d_blk + first, n, d_out, d_st + first, toks[j % n_streams], d_ctr + j, (volatile unsigned *)nullptr);
        }
        }
        for (int k = 0; k < n_streams; ++k) hipStreamSynchronize(sts[k]);
        const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("%d streams, launches of %u members: %u launches in %.3f ms = %.1f GB/s\n", n_streams, batch, n_launch, wall_ms, uo / wall_ms / 1e6);
        if (copier.joinable()) {
            copy_stop = true; copier.join();
            std::vector<float> during(copy_ms.begin() + std::min<size_t>(copy_ms.size(), 150), copy_ms.end());      // (the first 0.2 s ran before the launches)
            std::sort(during.begin(), during.end());
            std::vector<float> before(copy_ms.begin(), copy_ms.begin() + std::min<size_t>(copy_ms.size(), 150)); std::sort(before.begin(), before.end());
            if (!during.empty() && !before.empty())
                printf("32 MB host-to-device copies beside the launches: %zu, median %.3f ms (p10 %.3f, p90 %.3f); before the launches: median %.3f ms\n", during.size(), during[during.size() / 2],
                       during[during.size() / 10], during[during.size() * 9 / 10], before[before.size() / 2]);
        }
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        done = true;
    } else {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_dbg, dim3(n_wg), dim3(64), 0, 0, d_in, d_blk, nb, d_out, d_st, d_tok, d_next, marks ? (volatile unsigned *)h_dbg : (volatile unsigned *)nullptr);
    hipEventRecord(e1, 0);
    for (int t = 0; t < secs * 10; ++t) { if (hipEventQuery(e1) == hipSuccess) { done = true; break; } usleep(100000); }
    }
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const unsigned n_check = getenv("CHECK") ? atoi(getenv("CHECK")) : nb;    // (zlib on one host thread: bound it on big files)
    std::vector<int> st(nb); std::vector<unsigned char> out(uo + 1);
    hipMemcpy(st.data(), d_st, nb * 4, hipMemcpyDeviceToHost); hipMemcpy(out.data(), d_out, uo, hipMemcpyDeviceToHost);
    unsigned bad = 0, badst = 0;
    for (unsigned i = 0; i < nb; ++i) {
        if (st[i] != 0) { if (badst < 5) printf("member %u status %d\n", i, st[i]); ++badst; continue; }
        if (i >= n_check) continue;
        std::vector<unsigned char> ref(blks[i].out_len + 1);
        z_stream zs; memset(&zs, 0, sizeof zs); inflateInit2(&zs, -15);
        zs.next_in = d.data() + blks[i].in_off; zs.avail_in = blks[i].in_len; zs.next_out = ref.data(); zs.avail_out = blks[i].out_len;
        inflate(&zs, Z_FINISH); inflateEnd(&zs);
        if (memcmp(ref.data(), out.data() + blks[i].out_off, blks[i].out_len)) { if (bad < 5) { unsigned k = 0; while (ref[k] == out[blks[i].out_off + k]) ++k; printf("member %u differs at byte %u of %u\n", i, k, blks[i].out_len); } ++bad; }
    }
    { unsigned long long t[16]; hipMemcpyFromSymbol(t, HIP_SYMBOL(g_ticks), sizeof t);
      double tot = 0; for (int k = 0; k < 16; ++k) tot += (double)t[k];
      const char *nm[16] = {"code-length stream", "table build", "phase 1: later rounds + scans", "phase 2 (literals+tokens)", "phase 3: rest", "block header", "end of inflate + fence", "phase 1: first pass",
                            "phase 3: batch + dependencies", "phase 3: round head (ready, scan)", "phase 3: piece chunks", "phase 3: fence + done", "CRC-32", "-", "-", "-"};
      for (int k = 0; k < 13; ++k) printf("   %-28s %6.1f %%   %.0f ticks per member\n", nm[k], 100.0 * t[k] / (tot > 0 ? tot : 1), (double)t[k] / nb); }
    printf("%u members, %.1f MB out, kernel %.3f ms = %.1f GB/s; bad status %u, mismatching %u\n", nb, uo / 1e6, ms, uo / ms / 1e6, badst, bad);
    return bad || badst;
}
