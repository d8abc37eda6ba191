#!/usr/bin/env python3
"""gpurun_out/pmc/{FETCH_SIZE,WRITE_SIZE}_pdk.csv (tools/pmc_collect.sh) -> profiles/rNN_pmc_traffic.json:
HBM bytes per launch of every pdk:: kernel.  Counter_Value is KiB; FETCH_SIZE x2 on gfx950 (calibrated with
tools/ubench/fetch_calib.hip), WRITE_SIZE x1."""
import collections
import csv
import json
import re
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r01_pmc_traffic.json"


def short(name):
    m = re.search(r"pdk::(?:\(anonymous namespace\)::)?(k_\w+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name


def load(path, scale):
    tot, n = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        tot[k] += float(r["Counter_Value"]) * 1024 * scale
        n[k].add(r["Dispatch_Id"])
    return {k: tot[k] / len(n[k]) for k in tot}, {k: len(n[k]) for k in tot}


rd, n_rd = load(src + "/FETCH_SIZE_pdk.csv", 2.0)
wr, n_wr = load(src + "/WRITE_SIZE_pdk.csv", 1.0)
out = {"method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/pmc_workload.py "
                 "(the bench step on both paths + one write-back scan), no tracing options; Counter_Value is KiB",
       "calibration": {"FETCH_SIZE": "x2 (tools/ubench/fetch_calib.hip: the guide's gfx950 half-count; cross-check: the fused k_sweep "
                                     "reads the 12.0 GB of difference arrays once)",
                       "WRITE_SIZE": "x1 (the write-back k_sweep writes 12.0 GB)"},
       "kernels": {}}
for k in sorted(set(rd) | set(wr)):
    out["kernels"][k] = {"read_bytes_per_launch": int(rd.get(k, 0)), "write_bytes_per_launch": int(wr.get(k, 0)),
                         "hbm_bytes_per_launch": int(rd.get(k, 0) + wr.get(k, 0)), "launches": n_rd.get(k, n_wr.get(k, 0))}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: v["hbm_bytes_per_launch"] for k, v in out["kernels"].items()}, indent=1))
