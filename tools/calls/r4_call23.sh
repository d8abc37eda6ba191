#!/bin/bash
# round 4, call 23: the sliced sum's statistics sweep with the common tiles in a kernel of their own (k_sweep_i4_fast) — the tests that sweep images
# (loopback ranks, sliced sums, slice sweep against the oracle) and the 1-rank collective bench with and without it
O=$GRAFT_REPO_ROOT/gpurun_out/r4c23; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( timeout 500 python -m pytest tests/test_comm_gpu.py tests/test_comm_loopback_gpu.py tests/test_gpu_engine.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout 300 -k "sliced or comm or loopback or slice or sweep or export or image or list" ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for f in 1 0 1 0; do PD_BENCH_SWEEP_I4_FAST=$f PD_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 6 --warmup 2 --e2e-records 0 --e2e-multi-records 0 > $O/b$f.json 2> $O/b$f.err
  python - <<PY
import json
d = json.loads(open("$O/b$f.json").read().strip().splitlines()[-1])
print("sweep_i4_fast $f: step %.3f ms, slice_sweep %s, total_depth_check %s selfcheck %s" % (d["ms_per_step"], d["kernels"]["slice_sweep"], d["config"].get("total_depth_check"), d["config"].get("multi_gpu_sum_selfcheck")))
PY
done 2>&1 | tee $O/ab.txt
