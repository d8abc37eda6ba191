#!/bin/bash
# round 6, call 12: tools/ubench/h2d_fill — the host-to-device copy of a 32 MB page-locked buffer by how the CPU filled it just before (pread from the page cache
# as the feeders do, memcpy, non-temporal stores, pread through a bounce buffer, pread + cache-line flush), alone and beside five other reading threads
O=$GRAFT_REPO_ROOT/gpurun_out/r6c12; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 300 tools/ubench/h2d_fill 5 20 > $O/h2d_fill.txt 2>&1; cat $O/h2d_fill.txt
lscpu | grep -E "Model name|Socket|NUMA|L3|Thread|Core" >> $O/h2d_fill.txt; cat /sys/fs/cgroup/cpuset.cpus.effective >> $O/h2d_fill.txt 2>/dev/null; tail -12 $O/h2d_fill.txt
