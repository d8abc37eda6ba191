#!/bin/bash
# round 4, call 25: the cost of leaving a process that holds device memory, pinned host memory and streams (tools/ubench/exit_cost.hip): wall clock around the
# process against its own clock.  usage: r4_call25.sh [pinned]  (no argument: device memory sizes; "pinned": 4 GB of device memory + pinned buffers / streams)
O=$GRAFT_REPO_ROOT/gpurun_out/r4c25; mkdir -p $O; cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O2 tools/ubench/exit_cost.hip -o /tmp/exit_cost 2> $O/build.log
TIMEFORMAT='wall %R'
run() { for r in 1 2; do echo "GB pieces touch free pinned_MB streams = $*: $( { time /tmp/exit_cost "$@" ; } 2>&1 | tr '\n' ' ')"; sleep 0.5; done; }
if [ "$1" = pinned ]; then
  { run 4 1 1 0 0 0; run 4 1 1 0 128 0; run 4 1 1 0 384 0; run 4 1 1 0 768 0; run 4 1 1 0 1536 0; run 4 1 1 0 0 16; run 4 1 1 0 384 16; run 4 12 1 0 384 16; } | tee $O/exit_cost_pinned.txt
else
  { run 0 1 0 0; run 4 1 1 0; run 12 1 1 0; run 24 1 1 0; run 24 1 0 0; run 12 12 1 0; run 12 1 1 1; run 24 1 1 1; run 0 1 0 0; run 12 1 1 0; } | tee $O/exit_cost.txt
fi
