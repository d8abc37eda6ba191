# round 3, call 25: the context's small buffers from one allocation: GPU suite, smoke, and pd_create's own timing on three runs
O=$GRAFT_REPO_ROOT/gpurun_out/r3c25; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE-OK')" ) > $O/smoke.log 2>&1
( time python -m pytest tests -q -m gpu -x ) > $O/pytest_gpu.log 2>&1
mkdir -p /tmp/e2e && cd /tmp/e2e
$GRAFT_REPO_ROOT/tools/bamgen -o w.bam -n 2000000 -t 16 2> $O/gen.log
for k in 1 2 3 4; do PANDEPTH_TIMING=1 $GRAFT_REPO_ROOT/pandepth_amd/pandepth -i w.bam -o m -t 16 2>&1 >/dev/null | grep -E "pd_create|engine create" >> $O/create.log; echo >> $O/create.log; sleep 0.5; done
rm -rf /tmp/e2e
