#!/usr/bin/env python3
"""Kernel + memory-copy timeline of a `pandepth` run (rocprofv3 --kernel-trace --memory-copy-trace --output-format csv): what the device was doing
during the decode phase.  Prints, for the window between the first and the last k_inflate_wave: the share of time SOME kernel runs, the
concurrency histogram, per-kernel launch counts / durations, the copy engine's busy share, and per hardware queue the idle gaps between a
batch's stages.  Usage: tools/timeline.py <dir with *_kernel_trace.csv and *_memory_copy_trace.csv>"""
import csv
import glob
import os
import sys
from collections import defaultdict


def union_len(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for a, b in iv:
        if cs is None:
            cs, ce = a, b
        elif a <= ce:
            ce = max(ce, b)
        else:
            tot += ce - cs
            cs, ce = a, b
    if cs is not None:
        tot += ce - cs
    return tot


def short(name):
    for k in ("k_inflate_wave", "k_walk_segments", "k_emit_segments", "k_chain_segments", "k_spoil_segments", "copyBuffer", "fillBuffer", "k_direct_c8", "k_c8", "k_sfx", "k_scan"):
        if k in name:
            return k
    return name.split("(")[0][:40]


def main(d):
    kf = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    cf = glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True)
    K = []
    for r in csv.DictReader(open(kf)):
        K.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), int(r["Queue_Id"]), int(r.get("Stream_Id", 0) or 0), int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))))
    C = []
    if cf:
        for r in csv.DictReader(open(cf[0])):
            C.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"].replace("MEMORY_COPY_", ""), int(r.get("Stream_Id", 0) or 0)))
    inf = [k for k in K if k[2] == "k_inflate_wave"]
    if not inf:
        print("no k_inflate_wave in the trace"); return
    t0, t1 = min(k[0] for k in inf), max(k[1] for k in K if k[2] in ("k_emit_segments", "k_inflate_wave"))
    W = (t1 - t0) / 1e6
    Kw = [k for k in K if k[1] > t0 and k[0] < t1]
    busy = union_len([(max(k[0], t0), min(k[1], t1)) for k in Kw]) / 1e6
    print("decode window %.1f ms; some kernel running %.1f ms = %.1f %%" % (W, busy, 100 * busy / W))
    # concurrency histogram
    ev = []
    for k in Kw:
        ev.append((max(k[0], t0), 1)); ev.append((min(k[1], t1), -1))
    ev.sort()
    hist, cur, last = defaultdict(int), 0, t0
    for t, dlt in ev:
        hist[cur] += t - last; last = t; cur += dlt
    print("kernels running at once: " + ", ".join("%d: %.1f %%" % (n, 100 * v / (t1 - t0)) for n, v in sorted(hist.items())))
    by = defaultdict(list)
    for k in Kw:
        by[k[2]].append((k[1] - k[0]) / 1e6)
    print("%-22s %7s %9s %9s %9s %10s" % ("kernel", "calls", "mean ms", "median", "max", "sum ms"))
    for n, v in sorted(by.items(), key=lambda x: -sum(x[1])):
        v2 = sorted(v)
        print("%-22s %7d %9.3f %9.3f %9.3f %10.1f" % (n, len(v), sum(v) / len(v), v2[len(v) // 2], v2[-1], sum(v)))
    print("sum of all kernel time %.1f ms = %.2f x the window" % (sum(sum(v) for v in by.values()), sum(sum(v) for v in by.values()) / W))
    for direction in ("HOST_TO_DEVICE", "DEVICE_TO_HOST", "DEVICE_TO_DEVICE"):
        cc = [c for c in C if c[2] == direction and c[1] > t0 and c[0] < t1]
        if not cc:
            continue
        big = [c for c in cc if c[1] - c[0] > 200000]
        b = union_len([(max(c[0], t0), min(c[1], t1)) for c in cc]) / 1e6
        print("%s copies: %d (%d over 0.2 ms, mean %.3f ms); engine busy %.1f ms = %.1f %% of the window; summed %.1f ms" % (
            direction, len(cc), len(big), (sum(c[1] - c[0] for c in big) / len(big) / 1e6) if big else 0, b, 100 * b / W, sum(c[1] - c[0] for c in cc) / 1e6))
    # per hardware queue: inflate -> walk -> (chain) -> emit latencies
    byq = defaultdict(list)
    for k in Kw:
        byq[k[3]].append(k)
    gaps = defaultdict(list)
    for q, ks in byq.items():
        ks.sort()
        for a, b in zip(ks, ks[1:]):
            gaps[(a[2], b[2])].append((b[0] - a[1]) / 1e6)
    print("gaps between consecutive kernels on one hardware queue (end -> start), ms:")
    for (a, b), v in sorted(gaps.items(), key=lambda x: -sum(x[1]))[:12]:
        v2 = sorted(v)
        print("  %-18s -> %-18s n %5d  median %7.3f  mean %7.3f  p90 %7.3f  sum %8.1f" % (a, b, len(v), v2[len(v) // 2], sum(v) / len(v), v2[int(len(v) * 0.9)], sum(v)))
    print("hardware queues carrying kernels: %d; streams: %d" % (len(byq), len(set(k[4] for k in Kw))))


if __name__ == "__main__":
    main(sys.argv[1])
