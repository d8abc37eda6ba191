#!/bin/bash
# round 4, call 27: does a host-to-device copy read a pinned buffer slower when the CPU has just written it?  (tools/ubench/h2d_dirty.py)
O=$GRAFT_REPO_ROOT/gpurun_out/r4c27; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 200 python tools/ubench/h2d_dirty.py 2>&1 | grep -v amdgpu.ids | tee $O/h2d_dirty.txt
