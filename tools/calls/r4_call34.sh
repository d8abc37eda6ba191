#!/bin/bash
# round 4, call 34: the export instantiation of the direct kernel with the joined tail too — the tests that exchange and sweep images, the 1-rank collective bench
O=$GRAFT_REPO_ROOT/gpurun_out/r4c34; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( timeout 100 python -m pytest tests/test_comm_gpu.py tests/test_comm_loopback_gpu.py tests/test_gpu_engine.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout 90 -k "sliced or comm or loopback or export or image or compact" ) 2>&1 | tail -1 | tee $O/pytest.txt
PD_BENCH_FORCE_DIST=1 timeout 100 python bench.py --steps 6 --warmup 2 --e2e-records 0 --e2e-multi-records 0 > $O/b1.json 2> $O/b1.err
python - <<PY
import json
d = json.loads(open("$O/b1.json").read().strip().splitlines()[-1])
print("step %.3f ms, export %s, sweep %s, selfcheck %s" % (d["ms_per_step"], d["kernels"]["direct_export"], d["kernels"]["slice_sweep"]["avg_ms"], d["config"].get("multi_gpu_sum_selfcheck")))
PY
