#!/bin/bash
# round 4, call 32: the default direct kernel is the joined-tail <7, 4> form — compact-sample tests, smoke, and the bench's device leg
O=$GRAFT_REPO_ROOT/gpurun_out/r4c32; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( timeout 120 python -m pytest tests/test_gpu_engine.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout 100 -k "compact or runs" ) 2>&1 | tail -1 | tee $O/pytest.txt
( python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE-OK')" ) 2>&1 | tail -1 | tee -a $O/pytest.txt
PD_BENCH_CONFIG_LEGS=0 timeout 200 python bench.py --e2e-records 0 --e2e-multi-records 0 > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("value %.4g rec/s, %.3f ms/step, from decoder output %.4g; roofline %s" % (d["value"], d["ms_per_step"], d["value_from_decoder_output"], {k: d["roofline"][k] for k in ("frac", "avg_launch_ms", "frac_on_bytes_moved")}))
print("arrays path equals direct:", d["arrays_path"]["equals_direct"], "total depth", d["config"]["total_depth_check"])
PY
