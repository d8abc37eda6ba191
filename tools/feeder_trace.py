#!/usr/bin/env python3
"""What the decode's feeder threads did, batch by batch (`pandepth -X dd_trace=1` under PANDEPTH_TIMING prints a [trace] line per batch):
per-stage means, the host-observed latency queue -> collected against the device's own stage times, batches in flight over time,
and where a thread's time went.  Usage: tools/feeder_trace.py run.log"""
import re
import sys


def main(path):
    rows = []
    pat = re.compile(r"\[trace\] batch (\d+) thread (-?\d+) acquire (\d+) read (\d+) (\d+) queued (\d+) collect (\d+) (\d+) device ms h2d ([\d.]+) inflate ([\d.]+) walk ([\d.]+) emit ([\d.]+)")
    for line in open(path, errors="replace"):
        m = pat.search(line)
        if m:
            g = m.groups()
            rows.append(dict(bi=int(g[0]), th=int(g[1]), acq=int(g[2]), rd0=int(g[3]), rd1=int(g[4]), q=int(g[5]), c0=int(g[6]), c1=int(g[7]),
                             dev=[float(x) for x in g[8:12]]))
    rows = [r for r in rows if r["th"] >= 0]
    if not rows:
        print("no [trace] lines in", path)
        return
    n = len(rows)
    t_end = max(r["c1"] for r in rows)
    t_begin = min(r["acq"] for r in rows)
    mean = lambda xs: sum(xs) / max(len(xs), 1)
    med = lambda xs: sorted(xs)[len(xs) // 2] if xs else 0
    print("%d batches, first acquire at %.1f ms, last collect at %.1f ms: %.3f ms per batch" % (n, t_begin / 1e3, t_end / 1e3, (t_end - t_begin) / 1e3 / n))
    print("per batch (mean / median ms): acquire %.2f / %.2f, read+scan %.2f / %.2f, queue call %.2f / %.2f, queued -> collect starts %.2f / %.2f, collect call (waiting) %.2f / %.2f" % (
        mean([(r["rd0"] - r["acq"]) / 1e3 for r in rows]), med([(r["rd0"] - r["acq"]) / 1e3 for r in rows]),
        mean([(r["rd1"] - r["rd0"]) / 1e3 for r in rows]), med([(r["rd1"] - r["rd0"]) / 1e3 for r in rows]),
        mean([(r["q"] - r["rd1"]) / 1e3 for r in rows]), med([(r["q"] - r["rd1"]) / 1e3 for r in rows]),
        mean([(r["c0"] - r["q"]) / 1e3 for r in rows]), med([(r["c0"] - r["q"]) / 1e3 for r in rows]),
        mean([(r["c1"] - r["c0"]) / 1e3 for r in rows]), med([(r["c1"] - r["c0"]) / 1e3 for r in rows])))
    lat = [(r["c1"] - r["rd1"]) / 1e3 for r in rows]
    dev = [sum(r["dev"]) for r in rows]
    print("read done -> collected (host clock): mean %.2f median %.2f p90 %.2f ms; the device's own stages summed: mean %.2f median %.2f (h2d %.2f inflate %.2f walk %.2f emit %.2f)" % (
        mean(lat), med(lat), sorted(lat)[int(0.9 * n)], mean(dev), med(dev), *[mean([r["dev"][k] for r in rows]) for k in range(4)]))
    # batches in flight (read done .. collected) and threads reading, sampled over the steady part
    ev = []
    for r in rows:
        ev.append((r["rd1"], 1, 0)); ev.append((r["c1"], -1, 0)); ev.append((r["rd0"], 0, 1)); ev.append((r["rd1"], 0, -1))
    ev.sort()
    lo, hi = t_begin + (t_end - t_begin) * 0.1, t_begin + (t_end - t_begin) * 0.9
    fl = rd = 0; last = None; acc = {}; accr = {}
    for t, df, dr in ev:
        if last is not None and t > lo and last < hi:
            a, b = max(last, lo), min(t, hi)
            if b > a:
                acc[fl] = acc.get(fl, 0) + (b - a); accr[rd] = accr.get(rd, 0) + (b - a)
        fl += df; rd += dr; last = t
    tot = sum(acc.values()) or 1
    print("batches in flight (middle 80 %% of the phase): mean %.2f; " % (sum(k * v for k, v in acc.items()) / tot) + ", ".join("%d: %.0f %%" % (k, 100 * v / tot) for k, v in sorted(acc.items())))
    print("threads reading at once: mean %.2f; " % (sum(k * v for k, v in accr.items()) / tot) + ", ".join("%d: %.0f %%" % (k, 100 * v / tot) for k, v in sorted(accr.items())))
    ths = sorted(set(r["th"] for r in rows))
    for th in ths[:3]:
        rr = sorted([r for r in rows if r["th"] == th], key=lambda r: r["acq"])
        print("thread %d: %d batches; its first ten: " % (th, len(rr)) + " | ".join("b%d acq %.1f rd %.1f-%.1f q %.1f col %.1f-%.1f" % (r["bi"], r["acq"] / 1e3, r["rd0"] / 1e3, r["rd1"] / 1e3, r["q"] / 1e3, r["c0"] / 1e3, r["c1"] / 1e3) for r in rr[5:11]))


if __name__ == "__main__":
    main(sys.argv[1])
