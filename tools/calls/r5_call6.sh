#!/bin/bash
# round 5, call 6: `python bench.py` as the driver runs it (N = 1, defaults): the contract line with every leg -> profiles/r05_bench.json
O=$GRAFT_REPO_ROOT/gpurun_out/r5c6; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( time timeout 2400 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/time.txt; echo "rc $?" >> $O/time.txt
cat $O/time.txt | grep -E 'real|rc'; tail -c 600 $O/bench.err
python - <<'PY'
import json, os
p = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r5c6/bench.json"
try:
    d = json.loads(open(p).read().strip().splitlines()[-1])
    r = d["roofline"]; e = d.get("e2e") or {}
    print("value %.3e  ms/step %.3f  roofline frac %.3f (contract %.3f) kernel %s avg %.3f ms" % (d["value"], d["ms_per_step"], r["frac"], r.get("frac_on_contract_bytes", -1), r["kernel"], r["avg_launch_ms"]))
    print("e2e:", {k: e.get(k) for k in ("records", "byte_identical", "speedup_vs_reference")}, (e.get("pandepth") or {}).get("wall_s"), (e.get("pandepth") or {}).get("phases_s"), e.get("reference"))
    print("e2e.roofline:", e.get("roofline"))
    for k in ("annotation", "site_windows", "site_windows_fullsize"):
        x = e.get(k) or {}
        print(k, {kk: x.get(kk) for kk in ("byte_identical", "sha256_identical", "speedup_vs_reference", "failed")}, (x.get("pandepth") or {}).get("wall_s"), (x.get("pandepth") or {}).get("phases_s"))
    m = d.get("e2e_multi") or {}
    print("e2e_multi:", {k: m.get(k) for k in ("n_bams", "records_per_bam", "byte_identical", "failed")}, (m.get("pandepth") or {}), m.get("one_rank_rccl"), m.get("reference"))
    print("cpu_baseline:", d.get("cpu_baseline"))
except Exception as ex:
    print("could not parse", repr(ex))
PY
