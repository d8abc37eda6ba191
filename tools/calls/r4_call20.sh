#!/bin/bash
# round 4, call 20: the default bench line as the driver runs it, with the per-site streams mended in place
O=$GRAFT_REPO_ROOT/gpurun_out/r4c20; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
df -h /tmp | tail -1 > $O/df.txt
( time timeout 1500 python bench.py ) > $O/bench.json 2> $O/bench.err; echo rc=$? >> $O/bench.err; tail -c 600 $O/bench.err; wc -c $O/bench.json
