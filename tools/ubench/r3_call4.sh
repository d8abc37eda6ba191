# round 3, call 4: the whole GPU suite, the default bench line, kernel trace of the bench command
O=$GRAFT_REPO_ROOT/gpurun_out/r3c4; mkdir -p $O; cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -15 ) > $O/tests_gpu.log 2>&1
( time timeout 1500 python bench.py ) > $O/bench.json 2> $O/bench.err
export VARIANTS=c0,c708,c803,c703,c0
( EXPORT=0 timeout 600 python tools/ubench/direct_ab.py > $O/ab.log 2>&1 )
