#!/bin/bash
# round 4, call 15: the default bench line as the driver runs it (full-size e2e legs when /tmp has the room)
O=$GRAFT_REPO_ROOT/gpurun_out/r4c15; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
df -h /tmp | tail -1 > $O/df.txt; nproc >> $O/df.txt
( time timeout 1500 python bench.py ) > $O/bench.json 2> $O/bench.err; echo rc=$? >> $O/bench.err; tail -c 1500 $O/bench.err; wc -c $O/bench.json
