#!/usr/bin/env python3
"""bench.py — the depth hot path on MI355X: BASELINE.json's metric on BASELINE.json's config.

A "step" is ONE whole pass of the hot path over one sample's run stream, whole-chromosome mode
(configs[1]: 3 Gb reference, 50x short-read BAM = 10^9 alignment records of 150 bp):

    pd_reset                      forget the 3.0e9 cells (per-half-tile "written" flags + tile sums; no fill:
                                  unwritten cells count as zero until the scatter stores into them)
    pd_push_intervals_device x2   the sample's two run streams: the sorted first runs and the ~11 % second
                                  runs of D/I/N reads, which are only nearly sorted
                                  (PD_PUSH_DISORDER(max read span)); both deferred (PD_PUSH_MORE)
    pd_scan_reduce_windows        10 Mb-bin CoveredSite/TotalDepth, results copied back to the host.
                                  N = 1 (default, "direct_windows"): ONE pass over the runs — each tile's
                                  difference window is built in LDS (+1/-1 scatter), its carry-in counted
                                  from the same candidate runs, prefix-summed and reduced on the spot; the
                                  difference arrays never reach HBM (k_direct_tiles).
                                  PD_BENCH_PATH=arrays (and always for N > 1): the general path — owner-tile
                                  scatter into the int32 difference arrays in HBM (k_scatter_tiles), then the
                                  prefix-sum sweep fused with the bin reduction (k_sweep).  A few steps of it
                                  are also run after the timed region: their kernel figures are reported under
                                  "arrays_path" and their results must equal the direct path's.
    [N > 1, instead of the last]  the samples' difference arrays are summed SLICED (pandepth_amd.multi.SlicedSum):
                                  4-bit image (pd_export_i4; packed straight from the tile windows in LDS, the arrays
                                  are not written on any rank), all-to-all over RCCL so that every xGMI link of a
                                  GPU carries 1/N of it at once, every rank sums + sweeps its 1/N of the tiles
                                  (pd_slice_sweep_i4), rank 0 adds the per-tile partials up per bin
                                  (pd_gather_windows).  Steps are software-pipelined: sample k+1 is scattered
                                  while sample k's image is on the links; the timed region still contains K
                                  complete steps (fill and drain included).  PD_BENCH_SUM=int8|int32 select
                                  the older reduce-to-rank-0 forms, PD_BENCH_PIPELINE=0 the unpipelined order.

The run stream is synthetic (tools/synth.py) and is resident in HBM before the
timed region, as the contract asks; value = records / step time.  Host-side BAM decode is NOT in
this number (see DESIGN.md for the end-to-end figures).

Launch: python bench.py [--gpus N --steps K --warmup W]; for N > 1 under torch.distributed.run,
one rank per GPU, each rank holding its own sample ("one BAM per GPU", #.list mode).
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
B_SCATTER_PER_RUN = 28         # 12 B run + 2 x (4 B read + 4 B write)
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


def cpu_baseline(sample_records, log):
    """Times the reference's own CPU path (oracle/_ref/pandepth_ref, built from /root/reference in
    the dev container) on the host cores, on a bounded sample of the same workload; falls back
    to the single-threaded C restatement (oracle/libpd_oracle.so) when the binary is absent."""
    from tools import synth
    ncpu = os.cpu_count() or 1
    scale = sample_records / 1.0e9
    names, lens = synth.genome_c2(scale=scale)
    rec = synth.gen_records_numpy(lens, sample_records, seed=4242)
    ref = os.path.join(ROOT, "oracle", "_ref", "pandepth_ref")
    s2b = os.path.join(ROOT, "oracle", "_ref", "sam2bam")
    sample = "%d records, %.0f Mb scaled C2 genome (%d contigs), 50x, whole-chromosome mode" % (
        sample_records, lens.sum() / 1e6, len(lens))
    if os.access(ref, os.X_OK) and os.access(s2b, os.X_OK):
        with tempfile.TemporaryDirectory(prefix="pdbench") as td:
            raw, bam = os.path.join(td, "raw.bam"), os.path.join(td, "s.bam")
            synth.write_bam(raw, names, lens, rec, procs=min(32, ncpu))
            subprocess.run([s2b, raw, bam], check=True, stderr=subprocess.DEVNULL)
            threads = min(36, ncpu)
            best = None
            for _ in range(2):                       # second run = warm page cache
                t0 = time.perf_counter()
                subprocess.run([ref, "-i", bam, "-o", os.path.join(td, "o"), "-t", str(threads)], check=True,
                               stdout=subprocess.DEVNULL)
                best = time.perf_counter() - t0
            return {"value": sample_records / best, "unit": "records/s", "cores": min(threads, cpu_quota()),
                    "kind": "reference",
                    "sample": sample + "; pandepth_ref -t %d = 12 chromosome workers x (1 + 2 BGZF threads), cgroup "
                              "quota %d CPUs, BAM+BAI, warm cache, %.2f s" % (threads, cpu_quota(), best)}
    sys.path.insert(0, os.path.join(ROOT, "oracle"))      # the checker's directory is on the path for this leg only
    import pd_oracle as O
    first, other = synth.records_to_runs(rec)
    runs = np.concatenate([first, other])
    t0 = time.perf_counter()
    d, off = O.depth_from_intervals(list(lens), runs)
    regs = np.array([[t, 1, int(l)] for t, l in enumerate(lens)], dtype=np.int32)
    O.stat_regions(d, off, regs, 1)
    dt = time.perf_counter() - t0
    return {"value": sample_records / dt, "unit": "records/s", "cores": 1, "kind": "port",
            "sample": sample + "; oracle/pd_oracle.c increment+stat loops, no BAM decode (%.2f s)" % dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--records", type=float, default=1.0e9, help="alignment records per GPU (per sample)")
    ap.add_argument("--cpu-sample", type=float, default=4.0e7, help="records in the CPU-baseline sample (0 = skip)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import pandepth_amd as pda
    from pandepth_amd import multi
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

    R = int(args.records)
    names, lens = synth.genome_c2()
    G = int(lens.sum())
    eng = pda.Engine(lens.astype(np.uint32), device=local)
    first, other = synth.gen_runs_torch(lens, R, dev, seed=42 + rank)
    torch.cuda.synchronize()
    n_first, n_other = int(first.shape[0]), int(other.shape[0])
    _, n_words, _ = eng.device_buffer()
    n_cells = eng.device_layout()[0]
    # N > 1: int8 transport of the difference arrays (1 B/cell on the xGMI links instead of 4);
    # PD_BENCH_SUM=int32 selects the plain int32 reduce of the whole buffer instead
    sum_mode = os.environ.get("PD_BENCH_SUM", "sliced") if use_dist else None
    sliced = multi.SlicedSum(eng, dev) if sum_mode == "sliced" else None
    packed = multi.PackedSum(eng, dev) if sum_mode == "int8" else None
    buf = multi.buffer_view(eng, dev) if sum_mode == "int32" else None
    pipelined = sliced is not None and os.environ.get("PD_BENCH_PIPELINE", "1") == "1"
    wrap = 18 if use_dist else 0         # #.list mode keeps 18-bit cells (PD:2687-2699); single BAM + index: uint32
    # N > 1 with the sliced sum: the same deferred pushes, and pd_export_i4 packs the tile windows straight from LDS
    # (no arrays on any rank); the reduce-to-rank-0 forms need the arrays
    direct = os.environ.get("PD_BENCH_PATH", "direct") == "direct" and (not use_dist or sliced is not None)
    eng.set_param("direct_windows", 1 if direct else 0)

    def scatter():
        eng.reset()
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
            scatter()                              # the direct window call consumed the sample
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

    # the general (materialising) path next to the direct one: a few untimed-for-`value` steps, same inputs
    arrays_path = None
    if direct and not use_dist:
        ASTEPS = 3
        direct = False
        eng.set_param("direct_windows", 0)
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
                "scatter_tiles": k_entry2(prof, "scatter_tiles", (n_first + n_other) * B_SCATTER_PER_RUN / launches_tiles),
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
            "direct_tiles": k_entry("direct_tiles", (n_first + n_other) * B_RUN + (n_cells // 8192) * 24),
            # N > 1: the runs read once, the 4-bit image (n_cells / 2 bytes) and the tile sums written
            "direct_export": k_entry("direct_export", (n_first + n_other) * B_RUN + n_cells // 2 + (n_cells // 8192) * 4),
            # on-demand zero fill of never-written half-tiles (multi-GPU reduce only): bytes depend on the sample
            "fill": ({"avg_ms": round(prof["fill"][0] / prof["fill"][1], 4), "launches": prof["fill"][1]}
                     if prof["fill"][1] else None),
            "scatter_tiles": k_entry("scatter_tiles", (n_first + n_other) * B_SCATTER_PER_RUN / launches_tiles),
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
        traffic = None
        try:
            import glob
            pmc = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[-1]))
            key = {"scatter_tiles": "k_scatter_tiles<8192>", "scan_reduce_windows": "k_sweep<false, true, false>",
                   "direct_tiles": "k_direct_tiles<4, 5>"}.get(dom)
            key = next((k for k in pmc["kernels"] if key and k.startswith(key.rstrip(">"))), None)    # template arguments may grow
            if key and R == int(1e9):
                traffic = pmc["kernels"][key]["hbm_bytes_per_launch"]
        except (OSError, IndexError, KeyError, ValueError):
            traffic = None
        roofline = {"bound": "hbm", "kernel": dom, "achieved": kd["achieved"], "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": kd["frac"], "traffic": traffic,
                    "avg_launch_ms": kd["avg_ms"], "algorithmic_bytes_per_launch": kd["algorithmic_bytes"]}
        cb = None
        if world == 1 and args.cpu_sample > 0:
            try:
                cb = cpu_baseline(int(args.cpu_sample), None)
            except Exception as ex:                                # never lose the GPU line over the CPU leg
                cb = {"value": None, "unit": "records/s", "cores": 0, "kind": "failed", "sample": repr(ex)}
        line = {
            "metric": "alignment records/sec (3 Gb genome, 50x BAM, whole-chromosome mode)",
            "value": value, "unit": "records/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "config": {"workload": "configs[1]: 3 Gb ref (12 chr + 500 scaffolds, %d bp), 50x short-read BAM, "
                                   "whole-chromosome mode" % G,
                       "records_per_gpu": R, "runs_sorted": n_first, "runs_unsorted": n_other,
                       "cells": int(n_words), "path": ("direct (difference windows stay in LDS" + (", exported as 4-bit images)" if use_dist else ")")) if direct
                               else "arrays (difference arrays in HBM)",
                       "parallelism": "1 BAM per GPU" + ((", " + {
                           "sliced": "sliced sum: 4-bit all-to-all over RCCL, every rank sweeps 1/N of the tiles" + (", steps pipelined" if pipelined else ""),
                           "int8": "RCCL reduce to rank 0 (int8 transport)", "int32": "RCCL reduce to rank 0 (int32)"}[sum_mode]) if use_dist else ""),
                       "total_depth_check": total_depth, "multi_gpu_sum_selfcheck": selfcheck},
            "roofline": roofline,
            "kernels": kernels,
            "arrays_path": arrays_path,
            "cpu_baseline": cb,
        }
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
