#!/usr/bin/env python3
"""tools/bench_modes.py — device-side timings of the OTHER modes' kernels on the config-2 sample
(configs[2]: -g with 175 274 CDS entries; configs[3]: -w 100 + -a): write-back sweep, interval
reduction, small-window sweeps, per-site read-back.  One JSON object per line; copied to profiles/."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import pandepth_amd as pda  # noqa: E402
from tools import synth  # noqa: E402

PEAK = 8000.0


def main():
    R = int(float(sys.argv[1])) if len(sys.argv) > 1 else int(1e9)
    dev = torch.device("cuda", 0)
    names, lens = synth.genome_c2()
    G = int(lens.sum())
    eng = pda.Engine(lens.astype(np.uint32), device=0)
    first, other = synth.gen_runs_torch(lens, R, dev, seed=42)
    torch.cuda.synchronize()

    def load():
        eng.reset()
        eng.push_intervals_device(first.data_ptr(), int(first.shape[0]), pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE)
        eng.push_intervals_device(other.data_ptr(), int(other.shape[0]), pda.PD_PUSH_SORTED | pda.PD_PUSH_DISORDER(synth.MAX_SPAN))

    def emit(name, ms, alg_bytes, extra=None):
        o = {"kernel": name, "avg_ms": round(ms, 4), "algorithmic_bytes": int(alg_bytes),
             "achieved_GBps": round(alg_bytes / ms / 1e6, 1), "frac_of_8TBps": round(alg_bytes / ms / 1e6 / PEAK, 4)}
        if extra:
            o.update(extra)
        print(json.dumps(o), flush=True)

    # synthetic annotation of config 3: 33 688 transcripts / 175 274 CDS, exon length log-normal (median 150)
    rng = np.random.default_rng(3)
    regs = []
    per_tx = np.full(33688, 175274 // 33688); per_tx[:175274 - per_tx.sum()] += 1
    chrom = rng.choice(12, 33688, p=lens[:12] / lens[:12].sum())
    for t in range(33688):
        c = int(chrom[t]); s = int(rng.integers(1, lens[c] - 200000))
        for e in range(int(per_tx[t])):
            el = int(min(5000, max(30, rng.lognormal(np.log(150), 0.7))))
            regs.append((c, s, s + el - 1)); s += el + int(rng.integers(80, 3000))
    regs = np.array(regs, dtype=np.int32)
    region_bases = int((regs[:, 2] - regs[:, 1] + 1).sum())

    reps = 5
    eng.profile(True)
    t_rd = []
    for _ in range(reps):
        load()
        eng.scan(18)                                        # write-back sweep (8 B/base)
        eng.reduce_intervals(regs, 1)
        eng.reduce_windows(100, 1)
        t0 = time.perf_counter()
        d = eng.read_depth(0, 0, int(lens[0]))               # per-site read-back of the largest contig
        t_rd.append(time.perf_counter() - t0)
        load()
        eng.scan_reduce_windows(100, 1, 18)                  # fused small-window sweep (mode 6)
        load()
        eng.scan_reduce_windows(1000, 1, 0)                  # fused mode-5 sweep
    for key, name, b in (("scan", "k_sweep<write-back> (pd_scan, 18-bit wrap)", 8 * G),
                         ("reduce_intervals", "k_reduce_pieces (175274 CDS entries, %d bases)" % region_bases, 4 * region_bases + 24 * len(regs)),
                         ("reduce_windows", "k_sweep<from depth, w=100> (3.0e7 windows)", 4 * G + 12 * (G // 100)),
                         ("scan_reduce_windows", "k_sweep<fused, w=100 and w=1000> (avg)", 4 * G + 12 * (G // 100 + G // 1000) // 2)):
        ms, n = eng.profile_get(key)
        emit(name, ms / n, b, {"launches": n})
    emit("pd_read_depth D2H (Chr01, %d cells, pageable host buffer)" % int(lens[0]), np.median(t_rd) * 1e3, 4 * int(lens[0]))


if __name__ == "__main__":
    main()
