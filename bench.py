#!/usr/bin/env python3
"""bench.py — the depth hot path on MI355X: BASELINE.json's metric on BASELINE.json's config.

A "step" is ONE whole pass of the hot path over one sample's run stream, whole-chromosome mode
(configs[1]: 3 Gb reference, 50x short-read BAM = 10^9 alignment records of 150 bp):

    pd_reset                      forget the 3.0e9 cells (per-half-tile "written" flags + tile sums; no fill:
                                  unwritten cells count as zero until the scatter stores into them)
    pd_push_intervals_device x2   the sample's two run streams, as the product's GPU decoder leaves them: the sorted
                                  first runs and the ~11 % later runs of D/I/N reads, which are only nearly sorted
                                  (PD_PUSH_DISORDER(max read span)); both deferred (PD_PUSH_MORE)
    pd_scan_reduce_windows        10 Mb-bin CoveredSite/TotalDepth, results copied back to the host.
                                  N = 1 (default, "direct_windows"): ONE pass over the runs — each tile's
                                  difference window is built in LDS (+1/-1 scatter), its carry-in counted
                                  from the same candidate runs, prefix-summed and reduced on the spot; the
                                  difference arrays never reach HBM (k_direct_wide3).
                                  PD_BENCH_PATH=arrays (and always for N > 1): the general path — owner-tile
                                  scatter into the int32 difference arrays in HBM (k_scatter_tiles), then the
                                  prefix-sum sweep fused with the bin reduction (k_sweep).  A few steps of it
                                  are also run after the timed region: their kernel figures are reported under
                                  "arrays_path" and their results must equal the direct path's.
    [N > 1, instead of the last]  the samples' difference arrays are summed SLICED (pd_comm_init + pd_sliced_sum_start / _finish:
                                  RCCL issued inside the library; PD_BENCH_SUM=sliced_torch: tools/multi_torch.py's SlicedSum):
                                  4-bit image (pd_export_i4; packed straight from the tile windows in LDS, the arrays
                                  are not written on any rank), all-to-all over RCCL so that every xGMI link of a
                                  GPU carries 1/N of it at once, every rank sums + sweeps its 1/N of the tiles
                                  (pd_slice_sweep_i4), rank 0 adds the per-tile partials up per bin
                                  (pd_gather_windows).  Steps are software-pipelined: sample k+1 is scattered
                                  while sample k's image is on the links; the timed region still contains K
                                  complete steps (fill and drain included).  PD_BENCH_SUM=int8|int32 select
                                  the older reduce-to-rank-0 forms, PD_BENCH_PIPELINE=0 the unpipelined order.

The run stream is synthetic (tools/synth.py) and is resident in HBM before the
timed region, as the contract asks; value = records / step time, on the kernel the `pandepth`
executable itself runs for this mode (its GPU decoder leaves the sample's runs resident and deferred,
then pd_scan_reduce_windows takes the direct path).  BAM decode is NOT in `value`; it IS in the
"e2e" object of the same JSON line: the executable and the reference binary on one payload BAM
generated on this box (tools/bamgen), process wall clock, outputs compared byte for byte.  The
reference's run there is also the contract's "cpu_baseline".  e2e["annotation"] is the same pair of executables
in `-g` mode on that BAM (configs[2]'s annotation density: index-driven fetch of the targets' chunks, decoded on the GPU).

Launch: python bench.py [--gpus N --steps K --warmup W]; for N > 1 under torch.distributed.run,
one rank per GPU, each rank holding its own sample ("one BAM per GPU", #.list mode).

--config gff | w100a: the same contract line for BASELINE.json's other two full-size configurations on the same sample
(they are parity-test cases, tests/test_gpu_fullsize.py; these legs give their kernels a measured roofline):
    gff    configs[2], `-g` with 33 688 transcripts / 175 274 CDS entries: owner-tile scatter INTO the difference arrays
           (k_scatter_tiles), write-back prefix sum (pd_scan: k_sweep reads and writes every cell), interval statistics
           (pd_reduce_intervals: k_reduce_pieces)
    w100a  configs[3], `-w 100 -a`: scatter, write-back prefix sum with the 18-bit wrap (the per-site file needs the cells),
           100-base windows off the depth (pd_reduce_windows: 3.0e7 windows); the read-back of the cells for the per-site
           text (pd_read_depth, PCIe) is timed separately ("site_readback") — host side, not in `value`
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
BIN = 10000000                 # whole-chromosome mode's synthetic bins (PD:3978)

# algorithmic bytes per unit (SURVEY.md §8d / DESIGN.md §4)
B_FILL_PER_CELL = 4
B_SCATTER_PER_RUN = 28         # 12 B run + 2 x (4 B read + 4 B write) — the contract's algorithmic figure for the scatter
B_SCATTER_ONCE_PER_RUN = 12    # what the owner-tile kernel must move at least: every run read once ...
B_SCATTER_ONCE_PER_CELL = 4    # ... and every cell of the difference arrays written once (after pd_reset half-tiles are stored, not read)
B_SWEEP_FUSED_PER_BASE = 4
B_RUN = 12                     # one packed (tid, beg, end) run


def cpu_quota():
    """CPUs the cgroup lets this job use (the GPU boxes cap the container below nproc)."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            return max(1, int(int(q) / int(p)))
    except (OSError, ValueError):
        pass
    return os.cpu_count() or 1


RUN_PAUSE_S = 2.5


def _best_wall(cmd, reps, env=None, want_stderr=False):
    """best wall time (exec to exit) of `reps` runs; with want_stderr also the stderr text of that run"""
    best, err = None, ""
    for k in range(reps):
        if k:
            time.sleep(1.0)       # the executable leaves without freeing its HBM (the driver reclaims it): a process started
                                  # right behind it waits for that in its own runtime start-up (0.09 -> 0.26 s of pd_create)
        t0 = time.perf_counter()
        p = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env, timeout=3600)
        if p.returncode != 0:
            raise RuntimeError("%s: exit %d: %s" % (os.path.basename(cmd[0]), p.returncode, p.stderr.decode(errors="replace")[-600:]))
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, err = dt, p.stderr.decode(errors="replace")
    return (best, err) if want_stderr else best


def _timed_runs(cmd, warmup, steps, env=None):
    """`warmup` untimed runs, then `steps` timed ones (process wall, exec to exit), back to back with the pause a process needs to find the
    HBM of the one before it released.  Returns (walls, stderr texts) of the timed runs."""
    walls, errs = [], []
    for k in range(warmup + steps):
        if k:
            # (the device memory a process held — tens of GB here — is wiped after it has left, and a process that allocates GBs meanwhile waits for
            # that: a 17 GB hipMalloc took 1.0-1.8 s in every third run one second behind another, none in runs three seconds apart
            # (profiles/r06_alloc_after_exit.txt).  The pause is between the steps, not inside one.)
            time.sleep(RUN_PAUSE_S)
        t0 = time.perf_counter()
        p = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env, timeout=3600)
        dt = time.perf_counter() - t0
        if p.returncode != 0:
            raise RuntimeError("%s: exit %d: %s" % (os.path.basename(cmd[0]), p.returncode, p.stderr.decode(errors="replace")[-600:]))
        if k >= warmup:
            walls.append(dt)
            errs.append(p.stderr.decode(errors="replace"))
    return walls, errs


def _spread(walls, digits=4):
    """median / min / max / mean of a list of wall times: the end-to-end legs scatter +-10 % from run to run and box to box, so a
    single run (or the best of two) cannot show a 10 % change"""
    w = sorted(walls)
    n = len(w)
    med = w[n // 2] if n % 2 else 0.5 * (w[n // 2 - 1] + w[n // 2])
    return {"median": round(med, digits), "min": round(w[0], digits), "max": round(w[-1], digits), "mean": round(sum(w) / n, digits), "runs": n}


def _median_run(walls):
    """index of the run whose wall time is the (upper) median"""
    order = sorted(range(len(walls)), key=lambda i: walls[i])
    return order[len(order) // 2]


def parse_timing(err):
    """PANDEPTH_TIMING=1 lines of one run -> {phase: seconds} + the device-decode totals (diagnostics printed by the executable:
    host/pipeline.cpp PhaseTimer and read_bam_device)"""
    import re
    ph, dec = {}, None
    for ln in err.splitlines():
        m = re.match(r"\[timing\] (\S.*?)\s+([0-9.]+) s   \(total ([0-9.]+) s\)", ln)
        if m:
            ph[m.group(1).strip()] = float(m.group(2))
            ph["total_in_main"] = float(m.group(3))
        m = re.match(r"\[timing\] comm init\s+([0-9.]+) s   \((.*?)\)", ln)
        if m:
            ph["comm_init"] = float(m.group(1))
            w = re.search(r"waited ([0-9.]+) s", m.group(2))
            ph["comm_init_waited_for"] = float(w.group(1)) if w else float(m.group(1))
        m = re.match(r"\[timing\] collective\s+([0-9.]+) s", ln)
        if m:
            ph["collective"] = float(m.group(1))
        m = re.search(r"device decode: (\d+) batches .*?(\d+) feeders(?: holding \d+ buffers each)?, (\d+) records on the device, (\d+) units handed back \((\d+) records on the host\).*?"
                      r"device ms summed over batches: H2D ([0-9.]+), inflate ([0-9.]+), walk ([0-9.]+), emit ([0-9.]+); bytes: compressed (\d+), inflated (\d+)", ln)
        if m:
            dec = {"batches": int(m.group(1)), "feeders": int(m.group(2)), "records_on_device": int(m.group(3)), "units_handed_back": int(m.group(4)),
                   "records_on_host": int(m.group(5)), "device_ms_summed": {"h2d": float(m.group(6)), "inflate": float(m.group(7)), "walk": float(m.group(8)),
                                                                             "emit": float(m.group(9))},
                   "compressed_bytes": int(m.group(10)), "inflated_bytes": int(m.group(11))}
    return ph, dec


def _bam_contigs(path):
    """(names, lengths) from a BAM's header (its first BGZF members)"""
    import struct
    import zlib
    data = b""
    with open(path, "rb") as f:
        for _ in range(64):
            h = f.read(12)
            if len(h) < 12:
                break
            xlen = struct.unpack_from("<H", h, 10)[0]
            extra = f.read(xlen)
            bsize = struct.unpack_from("<H", extra, 4)[0] + 1
            body = f.read(bsize - 12 - xlen)
            data += zlib.decompress(body[:-8], -15)
            if len(data) >= 12:
                l_text = struct.unpack_from("<i", data, 4)[0]
                if len(data) >= 12 + l_text:
                    n_ref = struct.unpack_from("<i", data, 8 + l_text)[0]
                    q, names, lens, ok = 12 + l_text, [], [], True
                    for _ in range(n_ref):
                        if q + 4 > len(data):
                            ok = False
                            break
                        l_name = struct.unpack_from("<i", data, q)[0]
                        if q + 8 + l_name > len(data):
                            ok = False
                            break
                        names.append(data[q + 4:q + 4 + l_name - 1].decode())
                        lens.append(struct.unpack_from("<i", data, q + 4 + l_name)[0])
                        q += 8 + l_name
                    if ok:
                        return names, lens
    raise RuntimeError("BAM header not found in the first members of " + path)


def e2e_annotation(td, bam, cli, ref, threads, records):
    """configs[2] end to end on the same BAM: a synthetic annotation of SURVEY.md 8(d) C3's shape and density (33 688
    transcripts / 175 274 CDS rows per 3.0 Gb, on the twelve largest contigs), `pandepth -g` (index-driven fetch of the targets' chunks, decoded on the
    GPU) against the reference binary, gene.stat.gz compared byte for byte."""
    names, lens = _bam_contigs(bam)
    big = sorted(range(len(lens)), key=lambda i: -lens[i])[:12]
    big = [i for i in big if lens[i] > 400000]
    rng = np.random.default_rng(3)
    # the same density as 33 688 transcripts / 175 274 CDS rows on 3.0 Gb: the generated BAM covers a scaled genome at 50x
    G = float(sum(lens))
    n_tx = max(100, int(round(33688 * G / 3.0e9)))
    n_cds = int(round(n_tx * 175274 / 33688))
    per_tx = np.full(n_tx, n_cds // n_tx)
    per_tx[:n_cds - int(per_tx.sum())] += 1
    w = np.array([lens[i] for i in big], dtype=np.float64)
    chrom = rng.choice(len(big), n_tx, p=w / w.sum())
    gff = os.path.join(td, "c3.gff")
    with open(gff, "w") as fh:
        for t in range(n_tx):
            c = big[int(chrom[t])]
            s0 = int(rng.integers(1, lens[c] - 200000))
            for e in range(int(per_tx[t])):
                el = int(min(5000, max(30, rng.lognormal(np.log(150), 0.7))))
                fh.write("%s\tsynth\tCDS\t%d\t%d\t.\t+\t0\tID=cds%d.%d;Parent=tx%05d\n" % (names[c], s0, s0 + el - 1, t, e, t))
                s0 += el + int(rng.integers(80, 3000))
    mine = os.path.join(td, "mine_g")
    w_dev = _best_wall([cli, "-i", bam, "-g", gff, "-o", mine, "-t", str(threads)], 2)
    out = {"mode": "pandepth -i s.bam -g c3.gff -o out -t N: %d transcripts / %d CDS rows on a %.2f Gb genome (the density of "
                   "33 688 / 175 274 on 3.0 Gb)" % (n_tx, n_cds, G / 1e9),
           "pandepth": {"wall_s": round(w_dev, 4), "records_in_file_per_s": records / w_dev, "threads": threads}}
    if os.access(ref, os.X_OK):
        w_ref = _best_wall([ref, "-i", bam, "-g", gff, "-o", os.path.join(td, "ref_g"), "-t", "36"], 2)
        out["reference"] = {"wall_s": round(w_ref, 4), "threads": 36}
        out["byte_identical"] = open(mine + ".gene.stat.gz", "rb").read() == open(os.path.join(td, "ref_g.gene.stat.gz"), "rb").read()
        out["speedup_vs_reference"] = round(w_ref / w_dev, 2)
    return out


PCIE_PEAK_GBS = 64.0           # PCIe Gen5 x16, one direction, raw (MI355X_MICROARCH.md: host link); ~55 GB/s is what a pinned H2D copy reaches
INFLATE_ISOLATED_GBS = 300.0      # k_inflate_wave on 102 037 members in one launch, 20 waves per CU: 299-314 on three boxes (profiles/r06_inflate_ab.txt, round 6; round 5: 277-293; round 4: 242).
INFLATE_ISOLATED_SOURCE = "profiles/r06_inflate_ab.txt (round 6, commit ca5ca02: 299-314 GB/s on three boxes, wave_debug cur / b3h16); a constant, not measured in this run"


def e2e_site_windows(td, gen, cli, ref, threads, records=20000000):
    """configs[3] end to end: `-w 100 -a` on a generated BAM (default 2e7 records / 60 Mb: 6e7 per-site lines, 0.9 GB of text),
    the product executable against the reference binary, win.stat.gz AND SiteDepth.gz compared byte for byte."""
    bam = os.path.join(td, "w.bam")
    g = subprocess.run([gen, "-o", bam, "-n", str(int(records)), "-t", str(min(32, os.cpu_count() or 1))], check=True, stderr=subprocess.PIPE, timeout=1800)
    mine = os.path.join(td, "mine_w")
    w_dev, err = _best_wall([cli, "-i", bam, "-w", "100", "-a", "-o", mine, "-t", str(threads)], 2, env=dict(os.environ, PANDEPTH_TIMING="1"), want_stderr=True)
    ph, _ = parse_timing(err)
    out = {"mode": "pandepth -i w.bam -w 100 -a -o out -t N (" + g.stderr.decode().strip().replace("bamgen: ", "") + ")", "records": int(records),
           "pandepth": {"wall_s": round(w_dev, 4), "records_per_s": records / w_dev, "threads": threads, "phases_s": ph}}
    if os.access(ref, os.X_OK):
        w_ref = _best_wall([ref, "-i", bam, "-w", "100", "-a", "-o", os.path.join(td, "ref_w"), "-t", "36"], 1)
        out["reference"] = {"wall_s": round(w_ref, 4), "records_per_s": records / w_ref, "threads": 36}
        same = {}
        for suf in ("win.stat.gz", "SiteDepth.gz"):
            same[suf] = open(mine + "." + suf, "rb").read() == open(os.path.join(td, "ref_w." + suf), "rb").read()
        out["byte_identical"] = all(same.values())
        out["byte_identical_files"] = same
        out["site_depth_gz_bytes"] = os.path.getsize(mine + ".SiteDepth.gz")
        out["speedup_vs_reference"] = round(w_ref / w_dev, 2)
    os.remove(bam)
    return out


def _sha256(path):
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def e2e_site_windows_fullsize(td, bam, cli, threads, records):
    """configs[3] at FULL size, on the configs[1] file itself (3.0 Gb genome, 1e9 records): `pandepth -w 100 -a` — 3.0e7 window rows and
    3.0e9 per-site lines (~60 GB of text through the device-side formatter, LZ77 parse and CRC, ~8 GB of gzip on disk).  The reference
    needs ~16 minutes for this run, so it is not repeated here: tools/bamgen is deterministic (the file is a function of -n and -S,
    not of the thread count), the reference was run ONCE on the identical file builder-side and the SHA-256 of its two outputs is
    committed (profiles/r04_fullsize_reference.json); this leg compares hashes.  Without that file the per-site stream is checked
    against the window table instead (line count and depth sum, tools/sitecheck)."""
    mine = os.path.join(td, "mine_wfull")
    w_dev, err = _best_wall([cli, "-i", bam, "-w", "100", "-a", "-o", mine, "-t", str(threads)], 1, env=dict(os.environ, PANDEPTH_TIMING="1"), want_stderr=True)
    ph, _ = parse_timing(err)
    out = {"mode": "pandepth -i s.bam -w 100 -a -o out -t N on the configs[1] file (full size)", "records": int(records),
           "pandepth": {"wall_s": round(w_dev, 3), "records_per_s": records / w_dev, "threads": threads, "phases_s": ph},
           "win_stat_gz_bytes": os.path.getsize(mine + ".win.stat.gz"), "site_depth_gz_bytes": os.path.getsize(mine + ".SiteDepth.gz")}
    t0 = time.perf_counter()
    got = {"win.stat.gz": _sha256(mine + ".win.stat.gz"), "SiteDepth.gz": _sha256(mine + ".SiteDepth.gz")}
    out["sha256"] = got
    out["hash_s"] = round(time.perf_counter() - t0, 1)
    ref_file = os.path.join(ROOT, "profiles", "r04_fullsize_reference.json")
    ref = None
    try:
        ref = json.load(open(ref_file))
    except (OSError, ValueError):
        ref = None
    if ref and int(ref.get("records", 0)) == int(records) and "w100a" in ref:
        # the same input?  (size of the file + hash of its index: the index names every record's virtual offset)
        out["same_input_as_reference_run"] = os.path.getsize(bam) == ref.get("bam_bytes") and _sha256(bam + ".bai") == ref.get("bai_sha256")
        same = {k: got[k] == ref["w100a"]["sha256"].get(k) for k in got}
        out["byte_identical_files"] = same
        out["byte_identical"] = all(same.values())
        out["reference"] = {"wall_s": ref["w100a"].get("wall_s"), "threads": ref["w100a"].get("threads"), "where": ref.get("where"),
                            "source": "profiles/" + os.path.basename(ref_file)}
        if ref["w100a"].get("wall_s"):
            out["speedup_vs_reference_builder_side"] = round(ref["w100a"]["wall_s"] / w_dev, 1)
    else:
        chk = os.path.join(ROOT, "tools", "sitecheck")
        if not os.access(chk, os.X_OK):
            subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tools", "sitecheck.cpp"), "-lz", "-o", chk], check=True)
        p = subprocess.run([chk, mine + ".SiteDepth.gz", mine + ".win.stat.gz"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=3000)
        out["site_stream_check"] = p.stdout.decode().strip()[-400:]
        out["site_stream_consistent_with_window_table"] = p.returncode == 0
    for suf in ("win.stat.gz", "SiteDepth.gz"):
        os.remove(mine + "." + suf)
    return out


def e2e_variant_leg(td, gen, cli, ref, threads, what, gen_args, records, runs=3):
    """The whole-chromosome run on a generated BAM of another SHAPE than configs[1]'s (tools/bamgen flags): `--long` — 10-20 kb reads with
    ~2 600 CIGAR operations each, 1 % of them in the CG tag (reference: README.md:149-150, PD:440-460) — or `-Q 40` — the short reads of
    configs[1] with unbinned qualities, about twice the compressed bytes per record.  The executable (median of `runs`) against the
    reference binary (one run, -t <quota>), chr.stat.gz compared byte for byte; what the device decoder did with the file (batches, units
    handed back to the host readers, the compact session's fallbacks) is read off PANDEPTH_TIMING's lines."""
    bam = os.path.join(td, what + ".bam")
    g = subprocess.run([gen, "-o", bam, "-n", str(int(records)), "-t", str(min(32, os.cpu_count() or 1))] + gen_args, check=True, stderr=subprocess.PIPE, timeout=3600)
    try:
        size = os.path.getsize(bam)
        mine = os.path.join(td, "mine_" + what)
        walls, errs = _timed_runs([cli, "-i", bam, "-o", mine, "-t", str(threads)], 1, runs, env=dict(os.environ, PANDEPTH_TIMING="1"))
        sp = _spread(walls)
        err = errs[_median_run(walls)]
        ph, dec = parse_timing(err)
        out = {"mode": "pandepth -i %s.bam -o out -t N (tools/bamgen %s: %s)" % (what, " ".join(gen_args), g.stderr.decode().strip().replace("bamgen: ", "")),
               "records": int(records), "bam_bytes": size, "bam_bytes_per_record": round(size / records, 1),
               "pandepth": {"wall_s": sp["median"], "wall_s_spread": sp, "records_per_s": records / sp["median"], "threads": threads, "phases_s": ph, "device_decode": dec,
                            "decoder_said": [ln.strip()[:400] for ln in err.splitlines() if "device decode" in ln or "declin" in ln or "compact" in ln or "slow path" in ln][:6]}}
        t_dec = ph.get("decode + scatter")
        if dec and t_dec:
            out["roofline"] = {"phase": "decode + scatter", "seconds": t_dec, "compressed_GBps": round(dec["compressed_bytes"] / t_dec / 1e9, 2),
                               "frac_pcie": round(dec["compressed_bytes"] / t_dec / 1e9 / PCIE_PEAK_GBS, 3), "inflated_GBps": round(dec["inflated_bytes"] / t_dec / 1e9, 1),
                               "share_of_units_handed_back": (round(dec["units_handed_back"] / max(1, dec["batches"]), 3) if dec.get("batches") else None),
                               "records_on_host": dec.get("records_on_host")}
        if os.access(ref, os.X_OK):
            q = cpu_quota()
            w_ref = _best_wall([ref, "-i", bam, "-o", os.path.join(td, "ref_" + what), "-t", str(q)], 1)
            out["reference"] = {"wall_s": round(w_ref, 4), "records_per_s": records / w_ref, "threads": q}
            out["byte_identical"] = open(mine + ".chr.stat.gz", "rb").read() == open(os.path.join(td, "ref_" + what + ".chr.stat.gz"), "rb").read()
            out["speedup_vs_reference"] = round(w_ref / sp["median"], 2)
        return out
    finally:
        for f in (bam, bam + ".bai"):
            if os.path.exists(f):
                os.remove(f)


def e2e_multi_leg(n_bams, records, steps=3, warmup=1):
    """Multi-BAM `#.list` end to end (configs[4]'s shape): n_bams payload BAMs (tools/bamgen, seeds 42 ..), `pandepth -i s.list` — ONE
    process, one context per visible GPU, the files decoded in parallel, the per-GPU samples summed in slices over RCCL / xGMI — against
    the reference binary's list mode on the same files (it reads them one after another into one array, PD:2704-3014), chr.stat.gz
    compared byte for byte.  records/s = n_bams x records / wall: THE figure on which the sharded path can scale with the number of
    GPUs (the device-only `value` is bounded by the links)."""
    gen = os.path.join(ROOT, "tools", "bamgen")
    cli = os.path.join(ROOT, "pandepth_amd", "pandepth")
    ref = os.path.join(ROOT, "oracle", "_ref", "pandepth_ref")
    quota = cpu_quota()
    td = tempfile.mkdtemp(prefix="pdmulti", dir="/tmp")
    try:
        free = os.statvfs(td).f_bavail * os.statvfs(td).f_frsize
        while records > 1e7 and n_bams * records * 56 > 0.6 * free:
            records = int(records // 2)
        t0 = time.perf_counter()
        lst = os.path.join(td, "s.list")
        total_bytes = 0
        with open(lst, "w") as fh:
            for k in range(n_bams):
                b = os.path.join(td, "s%d.bam" % k)
                subprocess.run([gen, "-o", b, "-n", str(int(records)), "-S", str(42 + k), "-t", str(min(32, os.cpu_count() or 1))], check=True,
                               stderr=subprocess.PIPE, timeout=3600)
                total_bytes += os.path.getsize(b)
                fh.write(b + "\n")
        t_gen = time.perf_counter() - t0
        threads = max(4, min(16 * n_bams, quota, 64))
        mine = os.path.join(td, "mine")
        walls, errs = _timed_runs([cli, "-i", lst, "-o", mine, "-t", str(threads)], warmup, steps, env=dict(os.environ, PANDEPTH_TIMING="1"))
        sp = _spread(walls)
        w_dev = sp["median"]
        err = errs[_median_run(walls)]
        ph, _ = parse_timing(err)
        said = lambda e: [ln.strip() for ln in e.splitlines() if "summed over" in ln or "added into" in ln or "communicator" in ln or "comm ahead" in ln][:4]
        out = {"n_bams": n_bams, "records_per_bam": int(records), "bam_bytes_total": total_bytes, "generated_in_s": round(t_gen, 1),
               "mode": "pandepth -i s.list -o out -t N (one context per visible GPU; process wall clock exec-to-exit, warm page cache, %d warm-up + %d timed runs, median)" % (warmup, steps),
               "pandepth": {"wall_s": w_dev, "wall_s_spread": sp, "records_per_s": n_bams * records / w_dev, "threads": threads, "phases_s": ph, "sum": said(err)}}
        if n_bams == 1:
            # one GPU: the list path's COLLECTIVE code with one rank (-X comm=force) over both transports — the executable's default
            # (in-process peer copies: nothing to load) and real RCCL (-X transport=rccl: librccl loaded and the communicator bootstrapped
            # at process entry, ahead of the contexts).  What a communicator costs a list run here; the table must not change.
            # RCCL: the first process on a fresh box that loads librccl reads its gigabyte from disk (4-5 s: profiles/r05_comm_init.txt) — run
            # once untimed-for-the-median and reported as first_run_on_the_box.
            for key, tune_s in (("one_rank_collective", "comm=force"), ("one_rank_rccl", "comm=force,transport=rccl")):
                try:
                    envk = dict(os.environ, PANDEPTH_TIMING="1", PANDEPTH_TUNE=tune_s)
                    cmdk = [cli, "-i", lst, "-o", mine + "_" + key, "-t", str(threads)]
                    w_c, err_c = _best_wall(cmdk, 1, env=envk, want_stderr=True)
                    time.sleep(1.0)
                    wk, ek = _timed_runs(cmdk, 0, 3, env=envk)
                    spk = _spread(wk)
                    err_f = ek[_median_run(wk)]
                    out[key] = {"wall_s": spk["median"], "wall_s_spread": spk, "phases_s": parse_timing(err_f)[0], "transport": tune_s,
                                "vs_no_communicator": round(spk["median"] / w_dev, 3),
                                "first_run_on_the_box": {"wall_s": round(w_c, 4), "phases_s": parse_timing(err_c)[0]}, "sum": said(err_f),
                                "same_table": open(mine + ".chr.stat.gz", "rb").read() == open(mine + "_" + key + ".chr.stat.gz", "rb").read()}
                except Exception as ex:                                # noqa: BLE001
                    out[key] = {"failed": repr(ex)[:300]}
        if os.access(ref, os.X_OK):
            w_ref = _best_wall([ref, "-i", lst, "-o", os.path.join(td, "ref"), "-t", "36"], 1)
            out["reference"] = {"wall_s": round(w_ref, 4), "records_per_s": n_bams * records / w_ref, "threads": 36}
            out["byte_identical"] = open(mine + ".chr.stat.gz", "rb").read() == open(os.path.join(td, "ref.chr.stat.gz"), "rb").read()
            out["speedup_vs_reference"] = round(w_ref / w_dev, 2)
        return out
    finally:
        import shutil
        shutil.rmtree(td, ignore_errors=True)


def e2e_leg(records, site_records, fullsize_site=False, steps=5, warmup=1):
    """END TO END, product path: the `pandepth` executable (GPU-side BGZF inflate + record parsing + the direct window
    kernel) and the reference binary on the SAME coordinate-sorted BAM with SEQ/QUAL/tag payload, written here by
    tools/bamgen (libdeflate level 6, BAI alongside), whole-chromosome mode, process wall clock (exec to exit, warm page
    cache, best of 2 for both), outputs compared byte for byte.  Returns (e2e object, cpu_baseline object).
    The reference run is the contract's cpu_baseline (kind "reference"): its own multithreaded CPU path on this box's
    host cores, the file being the bounded sample of configs[1]."""
    gen = os.path.join(ROOT, "tools", "bamgen")
    if not os.access(gen, os.X_OK):
        subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", os.path.join(ROOT, "tools", "bamgen.cpp"), "-lz", "-ldl", "-o", gen], check=True)
    cli = os.path.join(ROOT, "pandepth_amd", "pandepth")
    ref = os.path.join(ROOT, "oracle", "_ref", "pandepth_ref")
    quota = cpu_quota()
    td = tempfile.mkdtemp(prefix="pde2e", dir="/tmp")
    bam = os.path.join(td, "s.bam")
    try:
        # room for the file (~53 B per record) beside what the box already holds: a smaller sample rather than a failed leg
        free = os.statvfs(td).f_bavail * os.statvfs(td).f_frsize
        asked = int(records)
        while records > 2e7 and records * 56 > 0.8 * free:
            records = int(records // 2)
        t0 = time.perf_counter()
        g = subprocess.run([gen, "-o", bam, "-n", str(int(records)), "-t", str(min(32, os.cpu_count() or 1))], check=True,
                           stderr=subprocess.PIPE, timeout=3600)
        t_gen = time.perf_counter() - t0
        size = os.path.getsize(bam)
        threads = max(4, min(16, quota))
        mine = os.path.join(td, "mine")
        # THE METRIC (BASELINE.json: alignment records/sec + wall-clock on the configs[1] BAM): `warmup` untimed runs of the executable (page
        # cache, clocks), then `steps` timed ones; the line's `value` is records / the MEDIAN wall of those (min / max / mean beside it)
        walls, errs = _timed_runs([cli, "-i", bam, "-o", mine, "-t", str(threads)], warmup, steps, env=dict(os.environ, PANDEPTH_TIMING="1"))
        sp = _spread(walls)
        w_dev = sp["median"]
        parsed = [parse_timing(e) for e in errs]
        ph, dec = parsed[_median_run(walls)]
        ph_spread = {}
        for key in ("decode + scatter", "engine create", "scan + statistics"):
            vals = [q[0][key] for q in parsed if key in q[0]]
            if vals:
                ph_spread[key] = _spread(vals)
        insitu = [q[1]["inflated_bytes"] / (q[1]["device_ms_summed"]["inflate"] / 1e3) / 1e9 for q in parsed if q[1] and q[1]["device_ms_summed"]["inflate"]]
        w_host = _best_wall([cli, "-i", bam, "-o", os.path.join(td, "host"), "-t", str(threads)], 1,
                            env=dict(os.environ, PANDEPTH_TUNE="device_decode=0"))
        e2e = {
            "records": int(records), "records_asked": asked, "bam_bytes": size, "bam_bytes_per_record": round(size / records, 1),
            "bam": "tools/bamgen: coordinate-sorted, 150-base reads with names, SEQ from a synthetic reference, binned QUAL, "
                   "NM/MD/AS/XS/RG tags; BGZF by libdeflate level 6; .bai alongside (" + g.stderr.decode().strip().replace("bamgen: ", "") +
                   "; generated in %.0f s)" % t_gen,
            "mode": "whole-chromosome (pandepth -i s.bam -o out -t N), process wall clock exec-to-exit, warm page cache; pandepth: %d warm-up + %d timed runs, "
                    "MEDIAN reported (min / max / mean beside it); the reference: best of its runs" % (warmup, steps),
            "pandepth": {"wall_s": w_dev, "wall_s_spread": sp, "walls_s": [round(w, 4) for w in walls], "records_per_s": records / w_dev, "threads": threads,
                         "path": "GPU decode (k_inflate_wave, k_walk_segments, k_emit_segments) -> one compact sample (pd_runs) -> k_direct_c8; host only reads the file",
                         "phases_s": ph, "phases_s_spread": ph_spread, "device_decode": dec,
                         "inflate_kernel_GBps_in_situ_spread": (_spread(insitu, 1) if insitu else None)},
            "pandepth_host_decode": {"wall_s": round(w_host, 4), "records_per_s": records / w_host, "threads": threads,
                                     "path": "-X device_decode=0: libdeflate on the host threads + pd_push_intervals"},
            "cpu_quota": quota, "host_cpus": os.cpu_count(),
        }
        # what bounds the end-to-end run: the compressed bytes cross PCIe once (host page cache -> pinned buffer -> HBM), the
        # inflated bytes are produced by k_inflate_wave; both rates over the "decode + scatter" phase of the timed run
        t_dec = ph.get("decode + scatter")
        if dec and t_dec:
            e2e["roofline"] = {
                "phase": "decode + scatter", "seconds": t_dec,
                "compressed_GBps": round(dec["compressed_bytes"] / t_dec / 1e9, 2), "pcie_peak_GBps": PCIE_PEAK_GBS,
                "frac_pcie": round(dec["compressed_bytes"] / t_dec / 1e9 / PCIE_PEAK_GBS, 3),
                "inflated_GBps": round(dec["inflated_bytes"] / t_dec / 1e9, 1),
                "inflate_kernel_busy_s_summed_over_streams": round(dec["device_ms_summed"]["inflate"] / 1e3, 3),
                "inflate_kernel_GBps_in_situ": round(dec["inflated_bytes"] / (dec["device_ms_summed"]["inflate"] / 1e3) / 1e9, 1) if dec["device_ms_summed"]["inflate"] else None,
                # the decode phase against ITS dominant kernel's own ceiling: k_inflate_wave alone, on launches that keep the CUs full, inflates
                # 285 GB/s (profiles/r05_inflate_ab.txt: 277-293 on one box)
                "inflate_kernel_isolated_GBps": INFLATE_ISOLATED_GBS, "inflate_kernel_isolated_source": INFLATE_ISOLATED_SOURCE,
                "frac": round(dec["inflated_bytes"] / t_dec / 1e9 / INFLATE_ISOLATED_GBS, 3),
                "frac_is": "inflated bytes / decode-phase seconds / the inflate kernel's isolated rate: 1.0 = the phase runs as fast as that kernel alone could",
                "note": "inflate_kernel_GBps_in_situ = inflated bytes / sum of the batches' inflate-kernel times (batches of different feeders "
                        "overlap on the device, so the wall-clock rate inflated_GBps can exceed it); isolated kernel rate: profiles/",
            }
        cb = None
        if os.access(ref, os.X_OK):
            # the reference at -t <the job's CPU quota> (no oversubscription on these boxes).  Rounds 4-5 also timed -t 36 (12 chromosome workers x
            # (1 + 2 BGZF threads): its best shape on an unrestricted host) and kept the faster: on the 16-CPU quota of the GPU boxes that was
            # -t 16 every time (23.5 s against 27.8-28.2 s, profiles/r05_bench*.json), so the second 28 s run is no longer made
            # (PD_BENCH_REF_T36=1 brings it back).
            rthreads = quota
            w_ref = _best_wall([ref, "-i", bam, "-o", os.path.join(td, "ref"), "-t", str(rthreads)], 1)
            same = open(mine + ".chr.stat.gz", "rb").read() == open(os.path.join(td, "ref.chr.stat.gz"), "rb").read()
            by_threads = {str(rthreads): round(w_ref, 4)}
            if os.environ.get("PD_BENCH_REF_T36") == "1" and quota != 36:
                w36 = _best_wall([ref, "-i", bam, "-o", os.path.join(td, "ref36"), "-t", "36"], 1)
                by_threads["36"] = round(w36, 4)
                if w36 < w_ref:
                    w_ref, rthreads = w36, 36
            e2e["reference"] = {"wall_s": round(w_ref, 4), "records_per_s": records / w_ref, "threads": rthreads, "wall_s_by_threads": by_threads}
            e2e["byte_identical"] = same
            e2e["speedup_vs_reference"] = round(w_ref / w_dev, 2)
            cb = {"value": records / w_ref, "unit": "records/s", "cores": min(rthreads, quota), "kind": "reference",
                  "sample": "%d records of the configs[1] workload with payload (%.2f GB BAM, %.0f B/record compressed), BAM+BAI, warm "
                            "cache; pandepth_ref -t %d (= the job's CPU quota; -t 36 was slower on these boxes in rounds 4-5) on a cgroup quota of %d CPUs, "
                            "%.2f s wall" % (records, size / 1e9, size / records, rthreads, quota, w_ref)}
        if os.environ.get("PD_BENCH_E2E_ANNOTATION", "1") == "1":
            try:                                    # an extra: whatever goes wrong here must not cost the line its e2e object
                e2e["annotation"] = e2e_annotation(td, bam, cli, ref, threads, records)
            except Exception as ex:                 # noqa: BLE001
                e2e["annotation"] = {"failed": repr(ex)[:300]}
        if fullsize_site and records >= 9e8 and os.statvfs(td).f_bavail * os.statvfs(td).f_frsize > 12e9:
            try:                                    # configs[3] at full size on this very file (9.5 GB of output)
                e2e["site_windows_fullsize"] = e2e_site_windows_fullsize(td, bam, cli, threads, records)
            except Exception as ex:                 # noqa: BLE001
                e2e["site_windows_fullsize"] = {"failed": repr(ex)[:300]}
        os.remove(bam)
        # two other input shapes (round 6): long reads through the device decoder, and the configs[1] reads with unbinned qualities (a BAM of
        # about twice the compressed bytes per record: how close to the PCIe link the decode phase then runs)
        for key, gen_args, recs in (("long_reads", ["--long"], float(os.environ.get("PD_BENCH_LONG_RECORDS", "4.0e5"))),
                                    ("q40", ["-Q", "40"], float(os.environ.get("PD_BENCH_Q40_RECORDS", "1.0e8")))):
            if recs <= 0:
                continue
            try:
                e2e[key] = e2e_variant_leg(td, gen, cli, ref, threads, key, gen_args, recs)
            except Exception as ex:                 # noqa: BLE001
                e2e[key] = {"failed": repr(ex)[:300]}
        if site_records > 0:
            try:
                e2e["site_windows"] = e2e_site_windows(td, gen, cli, ref, threads, site_records)
            except Exception as ex:                 # noqa: BLE001
                e2e["site_windows"] = {"failed": repr(ex)[:300]}
        if cb is None:
            cb = {"value": None, "unit": "records/s", "cores": 0, "kind": "reference",
                  "sample": "oracle/_ref/pandepth_ref is not built on this box (oracle/Makefile needs /root/reference): the reference was not timed"}
        return e2e, cb
    finally:
        import shutil
        shutil.rmtree(td, ignore_errors=True)


def config_leg(args, which, eng, pda, synth, torch, dist, first, other, lens, rank, world, use_dist):
    """configs[2] / configs[3] on the bench sample: K timed steps of that configuration's device work, one JSON line."""
    G = int(lens.sum())
    n_first, n_other = int(first.shape[0]), int(other.shape[0])
    n_runs = n_first + n_other
    R = int(args.records)
    n_cells = eng.device_layout()[0]
    eng.keep_deferred(False)
    regs, region_bases = None, 0
    if which == "gff":
        # synthetic annotation of configs[2] (README:128): 33 688 transcripts / 175 274 CDS entries, exon length log-normal
        rng = np.random.default_rng(3)
        per_tx = np.full(33688, 175274 // 33688); per_tx[:175274 - per_tx.sum()] += 1
        chrom = rng.choice(12, 33688, p=lens[:12] / lens[:12].sum())
        rl = []
        for t in range(33688):
            c = int(chrom[t]); s0 = int(rng.integers(1, lens[c] - 200000))
            for _ in range(int(per_tx[t])):
                el = int(min(5000, max(30, rng.lognormal(np.log(150), 0.7))))
                rl.append((c, s0, s0 + el - 1)); s0 += el + int(rng.integers(80, 3000))
        regs = np.array(rl, dtype=np.int32)
        region_bases = int((regs[:, 2] - regs[:, 1] + 1).sum())

    win_out = None
    if which == "w100a":
        nw = int(eng.window_layout(100)[-1])
        win_out = (np.zeros(nw, dtype=np.uint32), np.zeros(nw, dtype=np.uint64))

    def step():
        eng.reset()
        eng.push_intervals_device(first.data_ptr(), n_first, pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE)
        eng.push_intervals_device(other.data_ptr(), n_other, pda.PD_PUSH_SORTED | pda.PD_PUSH_DISORDER(synth.MAX_SPAN))
        if which == "gff":
            eng.scan(0)                                          # indexed single BAM: uint32 cells (PD:676-786)
            return eng.reduce_intervals(regs, 1)
        eng.scan(18)                                             # -a: SiteInfo cells (PD:4127)
        return eng.reduce_windows(100, 1, out=win_out)           # 3.0e7 windows into the caller's arrays, as the executable does

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        eng.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    eng.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=first.device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    keys = ("reset", "fill", "scatter_index", "scatter_tiles", "scatter_finish", "scan", "reduce_intervals", "reduce_windows")
    prof = {k: eng.profile_get(k) for k in keys}
    eng.profile(False)
    site = None
    if which == "w100a" and rank == 0:
        # the per-site text needs every cell on the host: the largest contig's cells over PCIe (pageable destination)
        t1 = time.perf_counter()
        d = eng.read_depth(0, 0, int(lens[0]))
        t1 = time.perf_counter() - t1
        site = {"what": "pd_read_depth of Chr01 (%d cells, 4 B each) into a pageable host buffer" % int(lens[0]), "seconds": round(t1, 4),
                "GBps": round(4 * int(lens[0]) / t1 / 1e9, 2), "cells_sum_check": int(d.astype(np.uint64).sum())}
    if rank != 0:
        return None

    def ent(name, alg):
        ms, n = prof[name]
        if not n:
            return None
        avg = ms / n
        return {"avg_ms": round(avg, 4), "launches": n, "achieved": round(alg / (avg * 1e-3) / 1e9, 1), "unit": "GB/s",
                "frac": round(alg / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes": int(alg)}
    lt = max(1, prof["scatter_tiles"][1] // args.steps)
    kernels = {
        # SURVEY §8(d)'s 28 B per run; "store_once" beside it: runs read once + every cell written once (what the kernel must move)
        "scatter_tiles": ent("scatter_tiles", n_runs * B_SCATTER_PER_RUN / lt),
        # write-back prefix sum: every cell read and written once
        "scan": ent("scan", 8 * n_cells),
    }
    if kernels["scatter_tiles"]:
        once = (n_runs * B_SCATTER_ONCE_PER_RUN + n_cells * B_SCATTER_ONCE_PER_CELL) / lt
        kernels["scatter_tiles"]["store_once_bytes"] = int(once)
        kernels["scatter_tiles"]["store_once_frac"] = round(once / (kernels["scatter_tiles"]["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    if which == "gff":
        kernels["reduce_intervals"] = ent("reduce_intervals", 4 * region_bases + 24 * len(regs))
    else:
        kernels["reduce_windows"] = ent("reduce_windows", 4 * G + 12 * (G // 100))
    dom = max((k for k in kernels if kernels[k]), key=lambda k: prof[k][0])
    kd = kernels[dom]
    traffic, pmc_file = None, ""
    try:
        import glob
        pmc_file = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[-1]
        pmc = json.load(open(pmc_file))
        want = {"scatter_tiles": "k_scatter_tiles<8192", "scan": "k_sweep<true, false, false"}.get(dom)
        key = next((k for k in pmc["kernels"] if want and k.startswith(want)), None)
        if key and R == int(1e9):
            traffic = pmc["kernels"][key]["hbm_bytes_per_launch"]
    except (OSError, IndexError, KeyError, ValueError):
        traffic = None
    if which == "gff":
        cov, tot = res
        check = {"regions": int(len(regs)), "region_bases": region_bases, "depth_sum_over_regions": int(np.asarray(tot, dtype=np.uint64).sum())}
        metric = "alignment records/sec (3 Gb genome, 50x BAM, -g annotation mode: 33688 transcripts / 175274 CDS)"
        workload = "configs[2]: the configs[1] sample + a synthetic annotation of 33 688 transcripts / 175 274 CDS entries (%d bases)" % region_bases
    else:
        woff, cov, tot = res
        check = {"windows": int(len(tot)), "depth_sum_over_windows": int(np.asarray(tot, dtype=np.uint64).sum())}
        metric = "alignment records/sec (3 Gb genome, 50x BAM, -w 100 -a mode)"
        workload = "configs[3]: the configs[1] sample, 100-base windows (%d of them) + per-site cells kept for the per-site file" % len(tot)
    return {
        "metric": metric, "value": world * R / (dt / args.steps), "unit": "records/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int32", "data": "synthetic",
        "config": {"workload": workload, "records_per_gpu": R, "runs_sorted": n_first, "runs_unsorted": n_other, "cells": int(n_cells),
                   "path": "arrays (difference arrays in HBM, prefix sum written back)", "parallelism": "replicas" if world > 1 else "1 GPU",
                   "check": check},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": kd["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": kd["frac"],
                     "traffic": None,
                     "traffic_from_profile": ({"hbm_bytes_per_launch": traffic, "source": "profiles/" + os.path.basename(pmc_file)} if traffic else None),
                     "avg_launch_ms": kd["avg_ms"], "algorithmic_bytes_per_launch": kd["algorithmic_bytes"]},
        "kernels": kernels, "site_readback": site,
        "cpu_baseline": {"value": None, "unit": "records/s", "cores": 0, "kind": "reference",
                         "sample": "not timed in this leg: the reference's run is timed by the default (--config chr) invocation's e2e leg"},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--records", type=float, default=1.0e9, help="alignment records per GPU (per sample)")
    ap.add_argument("--e2e-records", type=float, default=float(os.environ.get("PD_BENCH_E2E_RECORDS", "-1")),
                    help="records of the end-to-end leg's BAM (product CLI and reference binary on the same file; 0 = skip; "
                         "1e9 = BASELINE's configs[1] in full: a 53 GB file, ~4 min to write; -1 (default) = 1e9 when /tmp has 70 GB "
                         "free, otherwise 3e8 — the line says which and why)")
    ap.add_argument("--e2e-multi-records", type=float, default=float(os.environ.get("PD_BENCH_E2E_MULTI_RECORDS", "3.0e8")),
                    help="records per BAM of the multi-BAM `#.list` end-to-end leg (one BAM per GPU of the run; 0 = skip).  3e8 (16 GB per file) so that "
                         "the run is the sharded decode and not process start-up; halved until the N files fit 60 %% of /tmp")
    ap.add_argument("--e2e-site-records", type=float, default=float(os.environ.get("PD_BENCH_E2E_SITE_RECORDS", "2.0e7")),
                    help="records of the `-w 100 -a` end-to-end pair (configs[3]; 0 = skip)")
    ap.add_argument("--config", choices=["chr", "gff", "w100a"], default="chr",
                    help="chr = configs[1] (BASELINE.json's metric; default); gff = configs[2]; w100a = configs[3]")
    args = ap.parse_args()
    e2e_size_note = "asked for"
    if args.e2e_records < 0:
        try:
            st = os.statvfs("/tmp")
            free = st.f_bavail * st.f_frsize
        except OSError:
            free = 0
        if free >= 70e9:
            args.e2e_records, e2e_size_note = 1.0e9, "BASELINE's configs[1] in full (1e9 records, 3.0 Gb genome): /tmp has %.0f GB free" % (free / 1e9)
        else:
            args.e2e_records, e2e_size_note = 3.0e8, "3e8 records instead of configs[1]'s 1e9: /tmp has only %.0f GB free (the 53 GB file and the 9.5 GB of `-w 100 -a` output want 70)" % (free / 1e9)

    # the contract: rank 0 prints ONE JSON line on stdout.  Libraries loaded below write there too (RCCL announces its version
    # when a communicator is made): file descriptor 1 is pointed at stderr for the life of the process and the line goes to
    # the real stdout through emit().
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(real_stdout, (json.dumps(obj) + "\n").encode())

    import torch
    import torch.distributed as dist
    import pandepth_amd as pda
    from tools import multi_torch as multi      # torch.distributed forms of the sum, kept for comparison (PD_BENCH_SUM=sliced_torch|int8|int32)
    from tools import synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # PD_BENCH_FORCE_DIST=1 exercises the collective code path with a 1-rank group (single-GPU boxes)
    use_dist = world > 1 or os.environ.get("PD_BENCH_FORCE_DIST") == "1"
    if use_dist:
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=dev)
    # a CPU-side group for the one long wait of an N > 1 run: while rank 0 runs the multi-BAM executable on every GPU the other ranks
    # must not sit in an RCCL barrier (a kernel spinning on their GPUs)
    cpu_group = dist.new_group(backend="gloo", timeout=__import__("datetime").timedelta(hours=2)) if use_dist and world > 1 else None

    R = int(args.records)
    names, lens = synth.genome_c2()
    G = int(lens.sum())
    eng = pda.Engine(lens.astype(np.uint32), device=local)
    if os.environ.get("PD_BENCH_SWEEP_I4_FAST") == "0":     # (A/B of the sliced sum's packed statistics kernel, N > 1 legs only)
        eng.set_param("sweep_i4_fast", 0)
    if os.environ.get("PD_BENCH_DIRECT_UN"):                # (A/B of the direct kernel's forms: "direct_un", see launch_direct_c8)
        eng.set_param("direct_un", int(os.environ["PD_BENCH_DIRECT_UN"]))
    # the two run streams the product's GPU decoder leaves in HBM (pd_decode_end): every read's first run (position sorted)
    # and its later runs (D / I / N reads), which trail the sorted order by at most the longest gap
    first, other = synth.gen_runs_torch(lens, R, dev, seed=42 + rank)
    torch.cuda.synchronize()
    n_first, n_other, n_far = int(first.shape[0]), int(other.shape[0]), 0
    if args.config != "chr":
        line = config_leg(args, args.config, eng, pda, synth, torch, dist, first, other, lens, rank, world, use_dist)
        if rank == 0:
            emit(line)
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        eng.close()
        return
    _, n_words, _ = eng.device_buffer()
    n_cells = eng.device_layout()[0]
    # N > 1: int8 transport of the difference arrays (1 B/cell on the xGMI links instead of 4);
    # PD_BENCH_SUM=int32 selects the plain int32 reduce of the whole buffer instead
    sum_mode = os.environ.get("PD_BENCH_SUM", "sliced") if use_dist else None
    # "sliced": RCCL called inside the library (pd_comm_init + pd_sliced_sum_start / _finish; torch.distributed only hands the
    # 128-byte unique id round);  "sliced_torch": the same protocol with the collectives issued from torch.distributed
    sliced = None
    if sum_mode == "sliced":
        ids = [pda.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        try:
            sliced = pda.Comm(eng, ids[0], rank, world)
        except pda.PdError as e:               # say why and let every rank take the torch.distributed form of the same protocol
            sys.stderr.write("[bench] rank %d: pd_comm_init failed (%s)\n" % (rank, e))
        ok = torch.tensor([0 if sliced is None else 1], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            if sliced is not None:
                sliced.close()
            sliced, sum_mode = multi.SlicedSum(eng, dev), "sliced_torch"
    elif sum_mode == "sliced_torch":
        sliced = multi.SlicedSum(eng, dev)
    packed = multi.PackedSum(eng, dev) if sum_mode == "int8" else None
    buf = multi.buffer_view(eng, dev) if sum_mode == "int32" else None
    pipelined = sliced is not None and os.environ.get("PD_BENCH_PIPELINE", "1") == "1"
    wrap = 18 if use_dist else 0         # #.list mode keeps 18-bit cells (PD:2687-2699); single BAM + index: uint32
    # N > 1 with the sliced sum: the same deferred pushes, and pd_export_i4 packs the tile windows straight from LDS
    # (no arrays on any rank); the reduce-to-rank-0 forms need the arrays
    direct = os.environ.get("PD_BENCH_PATH", "direct") == "direct" and (not use_dist or sliced is not None)
    eng.keep_deferred(bool(direct))
    # the sample in the engine's compact form (pd_runs_create: 8 bytes per run, bucketed, exact bounds) — what pd_decode_end
    # leaves for the whole-contig modes (PD_DECODE_COMPACT), made once before the timed region like the rest of the resident input
    runs8 = None
    if direct and os.environ.get("PD_BENCH_COMPACT", "1") == "1":
        runs8 = eng.runs_create(first.data_ptr(), n_first, other.data_ptr(), n_other)
    used_compact = runs8 is not None

    def scatter():
        eng.reset()
        if runs8 is not None and direct:
            eng.push_runs(runs8, pda.PD_PUSH_MORE)
            return
        eng.push_intervals_device(first.data_ptr(), n_first, pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE)
        eng.push_intervals_device(other.data_ptr(), n_other, pda.PD_PUSH_SORTED | pda.PD_PUSH_DISORDER(synth.MAX_SPAN)
                                  | (pda.PD_PUSH_MORE if direct else 0))

    def run_steps(k):
        """k complete steps; returns the last step's result (rank 0)"""
        res = None
        if sliced is None:
            for _ in range(k):
                res = step()
        elif not pipelined:
            for _ in range(k):
                scatter()
                res = sliced.run(BIN, 1, wrap, 0)
        else:
            for i in range(k):
                scatter()
                sliced.start(i % 2)
                if i:
                    res = sliced.finish((i - 1) % 2, BIN, 1, wrap, 0)
            if k:
                res = sliced.finish((k - 1) % 2, BIN, 1, wrap, 0)
        return res

    def sum_to_rank0():
        """the older forms: everybody's arrays into rank 0, which sweeps them"""
        if use_dist:
            if packed is not None:
                is_root = packed.run(0)
            else:
                eng.device_buffer()                # materialise zeros in half-tiles this sample never wrote
                eng.synchronize()                  # the engine's stream is not torch's: order by host sync
                is_root = multi.sum_to_root(buf, 0)
                torch.cuda.synchronize()
            if not is_root:
                return None
        return eng.scan_reduce_windows(BIN, 1, wrap)

    def step():
        scatter()
        return sum_to_rank0()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        eng.synchronize()

    selfcheck = None
    if use_dist:
        # the multi-GPU sum against what it must equal: depth sums are additive, so the bins of the summed
        # result (rank 0) = the sum over ranks of each rank's own single-GPU bins (no wrap below 2^18)
        scatter()
        own = torch.from_numpy(eng.scan_reduce_windows(BIN, 1, wrap)[2].astype(np.int64)).to(dev)
        dist.all_reduce(own, op=dist.ReduceOp.SUM)
        if direct:
            scatter()                              # (a fresh step, as in the timed loop; the direct call itself leaves the sample deferred)
        got = sliced.run(BIN, 1, wrap, 0) if sliced is not None else sum_to_rank0()
        if rank == 0:
            selfcheck = bool(np.array_equal(got[2].astype(np.int64), own.cpu().numpy()))
            if not selfcheck:
                raise SystemExit("multi-GPU sum (%s) disagrees with the sum of the per-rank results" % sum_mode)
    run_steps(args.warmup)
    barrier()
    eng.profile(True)
    t0 = time.perf_counter()
    res = run_steps(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    PROF_KEYS = ("reset", "fill", "scatter_index", "scatter_tiles", "scatter_finish", "scatter_atomic", "tile_carry",
                 "scan_reduce_windows", "export_i8", "import_i8", "export_i4", "slice_sweep", "gather_windows", "direct_tiles",
                 "direct_export")
    prof = {k: eng.profile_get(k) for k in PROF_KEYS}
    eng.profile(False)

    # What making the resident compact sample costs (it is made ONCE per sample, before the timed region): pd_runs_create in two parts —
    # "compact_runs" turns the sorted 12-byte stream into 8-byte runs and marks the buckets' first runs, which in the product is not a
    # pass at all (the GPU decoder's emit kernel writes its runs that way as it goes); "compact_finish" is what pd_decode_end really does
    # at the end of a file: bucket starts from the marks, and the counting sort of the later runs.  value_from_decoder_output charges
    # that second part to every step — the rate of ONE pass over a sample exactly as the decoder's batches leave it.
    sample_prep = None
    if used_compact and not use_dist:
        eng.synchronize()
        eng.profile(True)
        PREP = 3
        for _ in range(PREP):
            tmp_runs = eng.runs_create(first.data_ptr(), n_first, other.data_ptr(), n_other)
            eng.runs_destroy(tmp_runs)
        a_ms, a_n = eng.profile_get("compact_runs")
        b_ms, b_n = eng.profile_get("compact_finish")
        eng.profile(False)
        if a_n and b_n:
            sample_prep = {"from_12_byte_runs_ms": round(a_ms / a_n, 3), "decode_end_ms": round(b_ms / b_n, 3), "repeats": PREP,
                           "note": "from_12_byte_runs_ms: pd_runs_create's first pass (k_c8_from_sorted; the decoder's emit kernel does this work as it writes, "
                                   "no pass); decode_end_ms: bucket starts + counting sort of the later runs (k_sfx_*, k_c8_hist, k_scan_*, k_c8_place_other) = "
                                   "what pd_decode_end runs per sample"}

    # the general (materialising) path next to the direct one: a few untimed-for-`value` steps, same inputs
    arrays_path = None
    if direct and not use_dist:
        ASTEPS = 3
        direct = False
        eng.keep_deferred(False)
        step()
        eng.synchronize()
        eng.profile(True)
        ta = time.perf_counter()
        for _ in range(ASTEPS):
            res_a = step()
        eng.synchronize()
        ta = time.perf_counter() - ta
        prof_a = {k: eng.profile_get(k) for k in PROF_KEYS}
        eng.profile(False)
        same = bool(np.array_equal(res_a[1], res[1]) and np.array_equal(res_a[2], res[2]))
        if not same:
            raise SystemExit("direct path and materialising path disagree")
        arrays_path = {"ms_per_step": ta / ASTEPS * 1e3, "steps": ASTEPS, "equals_direct": same, "prof": prof_a}
        direct = True

    run_multi = world > 1 and args.e2e_multi_records > 0
    if run_multi:
        # every rank lets go of its GPU before rank 0 starts the executable, which makes its own context on every GPU
        dist.barrier()
        if isinstance(sliced, pda.Comm):
            sliced.close()
        sliced = None
        if runs8 is not None:
            eng.reset(); eng.runs_destroy(runs8); runs8 = None
        eng.close()
        first = other = None
        torch.cuda.empty_cache()
    if rank == 0:
        # sanity: the result of the last step must account for every base that was pushed
        woff, cover, tot = res
        total_depth = int(tot.sum())
        ms_step = dt / args.steps * 1e3
        value = world * R / (dt / args.steps)

        def k_entry(name, alg_bytes_per_launch):
            ms, n = prof[name]
            if n == 0:
                return None
            avg = ms / n
            ach = alg_bytes_per_launch / (avg * 1e-3) / 1e9
            return {"avg_ms": round(avg, 4), "launches": n, "achieved": round(ach, 1), "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "algorithmic_bytes": int(alg_bytes_per_launch)}

        def entries(prof, steps):
            launches_tiles = max(1, prof["scatter_tiles"][1] // steps)
            return {
                "scatter_tiles": k_entry2(prof, "scatter_tiles", (n_first + n_other + n_far) * B_SCATTER_PER_RUN / launches_tiles),
                "scan_reduce_windows": k_entry2(prof, "scan_reduce_windows", G * B_SWEEP_FUSED_PER_BASE)}

        def k_entry2(pr, name, alg):
            ms, n = pr[name]
            if n == 0:
                return None
            avg = ms / n
            ach = alg / (avg * 1e-3) / 1e9
            return {"avg_ms": round(avg, 4), "launches": n, "achieved": round(ach, 1), "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "algorithmic_bytes": int(alg)}

        if arrays_path is not None:
            arrays_path["kernels"] = entries(arrays_path.pop("prof"), arrays_path["steps"])
        launches_tiles = max(1, prof["scatter_tiles"][1] // args.steps)
        kernels = {
            # the direct path's only big kernel: reads every run once (12 B), writes 24 B per tile
            "direct_tiles": k_entry("direct_tiles", (n_first + n_other + n_far) * B_RUN + (n_cells // 8192) * 24),
            # N > 1: the runs read once, the 4-bit image (n_cells / 2 bytes) and the tile sums written
            "direct_export": k_entry("direct_export", (n_first + n_other + n_far) * B_RUN + n_cells // 2 + (n_cells // 8192) * 4),
            # on-demand zero fill of never-written half-tiles (multi-GPU reduce only): bytes depend on the sample
            "fill": ({"avg_ms": round(prof["fill"][0] / prof["fill"][1], 4), "launches": prof["fill"][1]}
                     if prof["fill"][1] else None),
            "scatter_tiles": k_entry("scatter_tiles", (n_first + n_other + n_far) * B_SCATTER_PER_RUN / launches_tiles),
            "scan_reduce_windows": k_entry("scan_reduce_windows", G * B_SWEEP_FUSED_PER_BASE),
            "export_i8": k_entry("export_i8", 5 * (n_words - (n_words - G) % 1)),     # 4 B read + 1 B written per cell
            "import_i8": k_entry("import_i8", 5 * (n_words - (n_words - G) % 1)),     # 1 B read + 4 B written per cell
            "export_i4": k_entry("export_i4", 4.5 * n_cells),                        # 4 B read + 4 bits written per cell
            # per rank: `world` nibble images of 1/world of the cells = n_cells / 2 bytes read, nothing written
            "slice_sweep": k_entry("slice_sweep", 0.5 * n_cells),
        }
        dom = max((k for k in kernels if kernels[k] and "frac" in kernels[k]), key=lambda k: prof[k][0])
        kd = kernels[dom]
        # HBM bytes per launch from the PMC counters: taken from the committed rocprofv3 --pmc passes
        # (tools/pmc_collect.sh -> profiles/*_pmc_traffic.json, FETCH_SIZE x2 / WRITE_SIZE x1 as calibrated
        # there); bench.py itself never runs under a profiler
        traffic, pmc_file = None, ""
        try:
            import glob
            pmc_file = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[-1]
            pmc = json.load(open(pmc_file))
            key = {"scatter_tiles": "k_scatter_tiles<8192>", "scan_reduce_windows": "k_sweep<false, true, false>",
                   "direct_tiles": "k_direct_c8<" if used_compact else "k_direct_wide3<",
                   "direct_export": "k_direct_c8<" if used_compact else "k_direct_wide3<"}.get(dom)
            # (the compact kernel's third template argument says which instantiation it is: true = the export, false = the statistics)
            import re
            third = "true" if dom == "direct_export" else "false"
            cands = [k for k in pmc["kernels"] if key and k.startswith(key.rstrip(">"))]
            key = next((k for k in cands if not used_compact or re.match(r"k_direct_c8<\d+, \d+, " + third + r"\b", k)), None)
            if key and R == int(1e9):
                traffic = pmc["kernels"][key]["hbm_bytes_per_launch"]
        except (OSError, IndexError, KeyError, ValueError):
            traffic = None
        # `traffic` is a live PMC measurement or null; this run is not under a profiler, so the committed figure is
        # reported beside it with its source, never as if it had been measured here
        # the same kernel's average in the committed rocprofv3 --kernel-trace --stats run of this command (profiles/): an
        # independent clock on the launch; bench.py itself never runs under a profiler
        rocprof_avg = None
        try:
            import csv
            import glob
            kf = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_kernel_stats.csv")))[-1]
            want = {"direct_tiles": "k_direct_c8<" if used_compact else "k_direct_wide3<", "scatter_tiles": "k_scatter_tiles<8192>",
                    "scan_reduce_windows": "k_sweep<false, true, false>", "direct_export": "k_direct_c8<" if used_compact else "k_direct_wide3<"}.get(dom)
            import re
            third = "true" if dom == "direct_export" else "false"
            for row in csv.DictReader(open(kf)):
                if want and want in row["Name"] and (not used_compact or not want.startswith("k_direct_c8") or re.search(r"k_direct_c8<\d+, \d+, " + third + r"\b", row["Name"])):
                    rocprof_avg = {"avg_launch_ms": round(float(row["AverageNs"]) / 1e6, 4), "calls": int(row["Calls"]), "source": "profiles/" + os.path.basename(kf)}
                    break
        except (OSError, IndexError, KeyError, ValueError):
            rocprof_avg = None
        # what the dominant kernel actually has to read when the sample is resident in the compact form: 8 B per run, a tile's own
        # buckets plus the one bucket (1/16 of a tile) before it, 12 B of bounds and 24 B of partials per tile.  `frac` stays on
        # SURVEY 8(d)'s 12 B per run (the contract's algorithmic figure, comparable across rounds); this one is reported beside it
        bytes_read_model = None
        if used_compact and dom in ("direct_tiles", "direct_export"):
            bytes_read_model = int((n_first + n_other + n_far) * 8 * (1 + 1 / 16) + (n_cells // 8192) * 36 + (n_cells // 2 if dom == "direct_export" else 0))
        # Round 5: `achieved` / `frac` are on the bytes the kernel MOVES — the PMC traffic of the committed counter pass when it is of this
        # kernel at this sample size, else the model below (within 3 % of each other) — not on SURVEY 8(d)'s 12 B per run, which this
        # kernel has not read since the compact form (8 B per run) went in; the contract-bytes figure stays beside it for comparison
        # across rounds (`frac_on_contract_bytes`: 0.81 in round 4's line, where it was the headline).
        moved = traffic if traffic else bytes_read_model
        ach_moved = (moved / (kd["avg_ms"] * 1e-3) / 1e9) if moved else kd["achieved"]
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(ach_moved, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(ach_moved / HBM_PEAK_GBS, 4),
                    "frac_basis": ("HBM bytes per launch by the PMC counters (traffic)" if traffic else "bytes the kernel must read (bytes_moved_model)" if bytes_read_model else "SURVEY 8(d) algorithmic bytes"),
                    "achieved_on_contract_bytes": kd["achieved"], "frac_on_contract_bytes": kd["frac"],
                    # HBM bytes per launch by the PMC counters (separate rocprofv3 --pmc passes of this command, committed under profiles/:
                    # bench.py itself never runs under a profiler); null when no committed pass matches this kernel and sample size
                    "traffic": traffic,
                    "bytes_moved_model": bytes_read_model,
                    "frac_on_bytes_moved": (round(bytes_read_model / (kd["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if bytes_read_model else None),
                    "traffic_from_profile": ({"hbm_bytes_per_launch": traffic, "source": "profiles/" + os.path.basename(pmc_file)}
                                             if traffic else None),
                    "avg_launch_ms": kd["avg_ms"], "avg_launch_ms_rocprof": rocprof_avg, "algorithmic_bytes_per_launch": kd["algorithmic_bytes"]}
        cb, e2e = {"value": None, "unit": "records/s", "cores": 0, "kind": "reference",
                   "sample": "not timed: the end-to-end leg runs on rank 0 of a 1-GPU invocation with --e2e-records > 0"}, None
        # configs[2] and configs[3] on the same resident sample, three steps each after the timed region (their kernels are
        # the general path's: scatter into the arrays, write-back sweep, interval / narrow-window reductions); `--config gff |
        # w100a` runs either as a contract line of its own
        configs = None
        if world == 1 and not use_dist and os.environ.get("PD_BENCH_CONFIG_LEGS", "1") == "1":
            configs = {}
            sub = argparse.Namespace(**vars(args))
            sub.steps, sub.warmup = 3, 1
            for which in ("gff", "w100a"):
                try:
                    ln = config_leg(sub, which, eng, pda, synth, torch, dist, first, other, lens, rank, world, use_dist)
                    configs[which] = {k: ln[k] for k in ("metric", "value", "unit", "steps", "ms_per_step", "roofline", "kernels", "site_readback")}
                    configs[which]["workload"] = ln["config"]["workload"]
                    configs[which]["check"] = ln["config"]["check"]
                except Exception as ex:                            # noqa: BLE001 — an extra must not cost the line
                    configs[which] = {"failed": repr(ex)[:300]}
        if world == 1 and args.e2e_records > 0:
            try:
                if runs8 is not None:
                    eng.reset(); eng.runs_destroy(runs8); runs8 = None
                eng.close()                                        # the CLI makes its own context on this GPU
                del first, other                                   # ... and the bench sample's 13 GB of runs go too
                torch.cuda.empty_cache()
                e2e, cb = e2e_leg(int(args.e2e_records), int(args.e2e_site_records), fullsize_site=os.environ.get("PD_BENCH_E2E_FULLSIZE_SITE", "1") == "1",
                                  steps=args.steps, warmup=args.warmup)
                if isinstance(e2e, dict):
                    e2e["size_note"] = e2e_size_note
            except Exception as ex:                                # never lose the GPU line over this leg
                e2e = {"failed": repr(ex)}
                cb = {"value": None, "unit": "records/s", "cores": 0, "kind": "failed", "sample": repr(ex)}
        e2e_multi = None
        if args.e2e_multi_records > 0 and (world > 1 or args.e2e_records > 0):
            try:
                if eng.h:                                          # (N > 1: this rank's context goes before the executable makes its own on every GPU)
                    if runs8 is not None:
                        eng.reset(); eng.runs_destroy(runs8); runs8 = None
                    if isinstance(sliced, pda.Comm):
                        sliced.close(); sliced = None
                    eng.close()
                    first = other = None
                    torch.cuda.empty_cache()
                # N > 1: the `#.list` run over N BAMs on N GPUs IS the step the line's value is quoted on (K timed runs); N = 1: a leg beside it
                e2e_multi = e2e_multi_leg(world, int(args.e2e_multi_records), steps=(args.steps if world > 1 else 3), warmup=(min(args.warmup, 2) if world > 1 else 1))
            except Exception as ex:                                # noqa: BLE001
                e2e_multi = {"failed": repr(ex)[:300]}
        if configs is not None and isinstance(e2e, dict):
            # the same two configurations end to end (executable vs reference binary on a generated BAM, bytes compared)
            if "gff" in configs and "annotation" in e2e:
                configs["gff"]["e2e"] = e2e["annotation"]
            if "w100a" in configs and "site_windows" in e2e:
                configs["w100a"]["e2e"] = e2e["site_windows"]
        value_dec = None
        if sample_prep is not None:
            value_dec = world * R / ((ms_step + sample_prep["decode_end_ms"]) * 1e-3)
        # ---- the headline.  BASELINE.json's metric is "alignment records/sec (whole node) + wall-clock, 3 Gb genome 50x BAM": the rate at
        # which the EXECUTABLE turns the BAM file(s) into the .stat.gz table, process wall clock.  Since round 6 that is `value` (rounds
        # 1-5 quoted the repeated device pass over a resident sample here — three orders of magnitude above what a user gets; it stays in
        # the line as `device_step`, with its roofline).  N = 1: the configs[1] file through `pandepth -i s.bam`; N > 1: N BAMs through
        # `pandepth -i s.list` on the N GPUs.  A step = one run of the executable; `warmup` untimed runs, `steps` timed ones, MEDIAN wall.
        device_step = {"records_per_s": value, "ms_per_step": ms_step, "steps": args.steps, "warmup": args.warmup,
                       "what": "pd_reset + pd_push_runs (the sample resident in HBM in the decoder's compact form) + pd_scan_reduce_windows incl. copy-back: a REPEATED pass "
                               "of the statistics kernels over one resident sample; `roofline` is this step's dominant kernel",
                       "records_per_s_from_decoder_output": value_dec}
        headline = None
        if world == 1 and isinstance(e2e, dict) and isinstance(e2e.get("pandepth"), dict):
            pe = e2e["pandepth"]
            headline = {"value": pe["records_per_s"], "ms": pe["wall_s"] * 1e3, "steps": pe["wall_s_spread"]["runs"], "spread": pe["wall_s_spread"],
                        "records": e2e["records"], "is": "the `pandepth` executable on the configs[1] BAM (%d records, %.1f GB), exec to exit, warm page cache: "
                        "file read + PCIe + GPU inflate + record walk + statistics + table; median of %d timed runs" % (e2e["records"], e2e["bam_bytes"] / 1e9, pe["wall_s_spread"]["runs"])}
        elif world > 1 and isinstance(e2e_multi, dict) and isinstance(e2e_multi.get("pandepth"), dict):
            pm = e2e_multi["pandepth"]
            headline = {"value": pm["records_per_s"], "ms": pm["wall_s"] * 1e3, "steps": pm["wall_s_spread"]["runs"], "spread": pm["wall_s_spread"],
                        "records": e2e_multi["n_bams"] * e2e_multi["records_per_bam"],
                        "is": "`pandepth -i s.list` over %d BAMs of %d records on %d GPUs (one process, a context and a rank thread per GPU), exec to exit; median of %d timed runs"
                              % (e2e_multi["n_bams"], e2e_multi["records_per_bam"], world, pm["wall_s_spread"]["runs"])}
        line = {
            "metric": "alignment records/sec (whole node) + wall-clock, 3 Gb genome 50x BAM (whole-chromosome mode, BAM file -> chr.stat.gz)",
            "value": (headline["value"] if headline else value), "unit": "records/s", "n_gpus": world,
            "steps": (headline["steps"] if headline else args.steps), "warmup": args.warmup,
            "ms_per_step": (headline["ms"] if headline else ms_step), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "wall_s_spread": (headline["spread"] if headline else None),
            # the repeated device pass over a resident sample (rounds 1-5's `value`) and the figure that adds pd_decode_end's work to every step
            "device_step": device_step,
            "device_step_records_per_s": value, "value_from_decoder_output": value_dec, "sample_preparation": sample_prep,
            # (kept under their round-5 names: the same figures as `value` / `ms_per_step` at N = 1)
            "e2e_records_per_s": (e2e.get("pandepth", {}).get("records_per_s") if isinstance(e2e, dict) else None),
            "e2e_wall_s": (e2e.get("pandepth", {}).get("wall_s") if isinstance(e2e, dict) else None),
            "config": {"workload": "configs[1]: 3 Gb ref (12 chr + 500 scaffolds, %d bp), 50x short-read BAM, "
                                   "whole-chromosome mode" % G,
                       "records_per_gpu": (headline["records"] // world if headline else R), "runs_sorted": n_first, "runs_unsorted": n_other,
                       "value_is": (headline["is"] if headline else
                                    "NO end-to-end leg in this invocation (--e2e-records 0 / no room in /tmp): `value` falls back to the device step — a REPEATED pass of the "
                                    "statistics kernels over one sample resident in HBM"),
                       "device_step_is": device_step["what"],
                       "cells": int(n_words), "path": ("direct (difference windows stay in LDS" + (", exported as 4-bit images)" if use_dist else "; the kernel path the pandepth CLI runs in this mode)") +
                                (" — sample resident in the compact form (8 B/run, grouped by 512-cell bucket with exact bounds), as pd_decode_end leaves it" if used_compact else "")) if direct
                               else "arrays (difference arrays in HBM)",
                       "parallelism": "1 BAM per GPU" + ((", " + {
                           "sliced": "device step: sliced sum behind the C-ABI (pd_sliced_sum_start / _finish: RCCL grouped send/recv of 4-bit slices issued by the library), every rank sweeps 1/N of the tiles" + (", steps pipelined" if pipelined else "") +
                                     "; the executable (`value`): one process, in-process peer-copy transport by default (-X transport=rccl: RCCL)",
                           "sliced_torch": "sliced sum: 4-bit all-to-all issued from torch.distributed, every rank sweeps 1/N of the tiles" + (", steps pipelined" if pipelined else ""),
                           "int8": "RCCL reduce to rank 0 (int8 transport)", "int32": "RCCL reduce to rank 0 (int32)"}[sum_mode]) if use_dist else ""),
                       "total_depth_check": total_depth, "multi_gpu_sum_selfcheck": selfcheck},
            "roofline": roofline,
            "kernels": kernels,
            "arrays_path": arrays_path,
            "configs": configs,
            "e2e": e2e,
            # configs[4]'s shape end to end: `world` BAMs through `pandepth -i s.list` on `world` GPUs vs the reference's list mode; the
            # figure to read weak scaling from (value stays the device-only step, whose 1 -> N efficiency the links bound: DESIGN.md 6)
            "e2e_multi": e2e_multi,
            "cpu_baseline": cb,
        }
        emit(line)
    if run_multi:
        dist.barrier(group=cpu_group)
    if use_dist:
        dist.barrier()
        if isinstance(sliced, pda.Comm):
            sliced.close()                      # the communicator goes before the context it belongs to and before the process group
        dist.destroy_process_group()
    if eng.h:
        if runs8 is not None:
            eng.reset(); eng.runs_destroy(runs8)
        eng.close()


if __name__ == "__main__":
    main()
