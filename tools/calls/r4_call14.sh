#!/bin/bash
# round 4, call 14: the sliced sum's receiving side with the packed fast path — its tests, and the 1-rank collective bench
O=$GRAFT_REPO_ROOT/gpurun_out/r4c14; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE-OK')" ) > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python -m pytest tests/test_comm_gpu.py tests/test_comm_loopback_gpu.py tests/test_gpu_engine.py -q -x --timeout 600 -k "sliced or comm or slice or loopback or list or sum" > $O/pytest_comm.log 2>&1; tail -3 $O/pytest_comm.log
PD_BENCH_FORCE_DIST=1 timeout 600 python bench.py --e2e-records 0 --e2e-multi-records 0 > $O/bench_1rank.json 2> $O/bench_1rank.err; tail -c 300 $O/bench_1rank.err
python - <<'PY'
import json,os
d=json.loads(open(os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/r4c14/bench_1rank.json")).read().strip().splitlines()[-1])
print("1-rank step ms",d["ms_per_step"]); print({k:v for k,v in d["kernels"].items() if v})
PY
