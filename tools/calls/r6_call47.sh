#!/bin/bash
# round 6, call 47: host-to-device copy rate by size and from six buffers in turn (tools/ubench/h2d_size)
O=$GRAFT_REPO_ROOT/gpurun_out/r6c47; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 120 tools/ubench/h2d_size > $O/h2d_size.txt 2>&1; cat $O/h2d_size.txt
