# round 3, call 24: what the driver runs at the end of the round: smoke(), the GPU suite, python bench.py
O=$GRAFT_REPO_ROOT/gpurun_out/r3c24; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( time python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE-OK')" ) > $O/smoke.log 2>&1
( time python -m pytest tests -q -m gpu ) > $O/pytest_gpu.log 2>&1
( time timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time
