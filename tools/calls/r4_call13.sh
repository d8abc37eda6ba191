#!/bin/bash
# round 4, call 13: the driver's sequence — GPU suite (guards on), smoke, the default bench line (full-size e2e legs)
O=$GRAFT_REPO_ROOT/gpurun_out/r4c13; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 ) > $O/pytest_gpu.log 2>&1; echo rc=$? >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
( python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE-OK')" ) > $O/smoke.log 2>&1; tail -2 $O/smoke.log
df -h /tmp | tail -1
( time timeout 1700 python bench.py ) > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err; wc -c $O/bench.json
