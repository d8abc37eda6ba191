#!/bin/bash
# round 6, call 42: the whole GPU suite with the round's final pipeline, then `python bench.py --gpus 1 --steps 20 --warmup 5` as the driver runs it
O=$GRAFT_REPO_ROOT/gpurun_out/r6c42; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -m gpu -q --timeout 900 ) > $O/pytest_gpu.log 2>&1; echo rc=$? >> $O/pytest_gpu.log; tail -6 $O/pytest_gpu.log; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time timeout 3000 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ); echo rc=$?
tail -c 1500 $O/bench.err; python - <<'PY'
import json,os
p=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r6c42/bench.json"
d=json.loads(open(p).readline())
print({k:d[k] for k in ("metric","value","unit","n_gpus","steps","warmup","ms_per_step","wall_s_spread")})
print("device_step", d["device_step"]["records_per_s"], d["device_step"]["ms_per_step"], "roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
e=d["e2e"]; print("e2e", e["pandepth"]["wall_s_spread"], e["pandepth"]["phases_s_spread"], e.get("byte_identical"), e.get("reference",{}).get("wall_s"), e.get("roofline"))
for k in ("annotation","site_windows_fullsize","site_windows","long_reads","q40"):
    v=e.get(k,{}); print(k, {kk:v.get(kk) for kk in ("failed","byte_identical","speedup_vs_reference")}, v.get("pandepth",{}).get("wall_s"), v.get("reference",{}).get("wall_s"))
m=d["e2e_multi"]; print("multi", m.get("failed"), m.get("pandepth",{}).get("wall_s_spread"), {k:(m[k].get("wall_s"), m[k].get("vs_no_communicator"), m[k].get("same_table")) for k in ("one_rank_collective","one_rank_rccl") if k in m}, m.get("byte_identical"))
print("cpu_baseline", d["cpu_baseline"])
PY
