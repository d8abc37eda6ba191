#!/bin/bash
# round 5, call 26: the round's final code — smoke() and the whole gpu suite
O=$GRAFT_REPO_ROOT/gpurun_out/r5c26; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( time timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke.log 2>&1; tail -4 $O/smoke.log
( time timeout 1800 python -m pytest tests -m gpu -q -x --timeout 900 ) > $O/pytest_gpu.log 2>&1; echo rc=$? >> $O/pytest_gpu.log; tail -6 $O/pytest_gpu.log
df -h /tmp | tail -1
