#!/bin/bash
# round 4, call 25: the cost of leaving a process that holds device memory (tools/ubench/exit_cost.hip): wall clock around the process against its own clock
O=$GRAFT_REPO_ROOT/gpurun_out/r4c25; mkdir -p $O; cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O2 tools/ubench/exit_cost.hip -o /tmp/exit_cost 2> $O/build.log
TIMEFORMAT='wall %R'
for cfg in "0 1 0 0" "4 1 1 0" "12 1 1 0" "24 1 1 0" "24 1 0 0" "12 12 1 0" "12 1 1 1" "24 1 1 1" "0 1 0 0" "12 1 1 0"; do
  for r in 1 2; do echo "GB pieces touch free = $cfg: $( { time /tmp/exit_cost $cfg ; } 2>&1 | tr '\n' ' ')"; sleep 0.5; done
done | tee $O/exit_cost.txt
