#!/bin/bash
# round 4, call 24: the driver's round-end sequence at the round's last commit — the GPU suite (guarded allocations) and smoke
O=$GRAFT_REPO_ROOT/gpurun_out/r4c24; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 ) > $O/pytest_gpu.log 2>&1; echo rc=$? >> $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log
( python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE-OK')" ) > $O/smoke.log 2>&1; tail -2 $O/smoke.log
