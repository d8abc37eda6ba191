#!/bin/bash
# round 6, call 48: host-to-device copy rate by the NUMA node of the source, alone and beside five threads reading a 16 GB file from the page cache (tools/ubench/h2d_numa)
O=$GRAFT_REPO_ROOT/gpurun_out/r6c48; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tools/bamgen -o /tmp/s.bam -n 300000000 -t 32 2>> $O/gen.log
cat /tmp/s.bam > /dev/null
timeout 120 tools/ubench/h2d_numa /tmp/s.bam 5 > $O/h2d_numa.txt 2>&1; cat $O/h2d_numa.txt
numactl -H 2>/dev/null | head -8 >> $O/h2d_numa.txt; cat /sys/fs/cgroup/cpuset.cpus.effective /sys/fs/cgroup/cpuset.mems.effective 2>/dev/null | tr '\n' ' ' >> $O/h2d_numa.txt; echo >> $O/h2d_numa.txt
( cd /tmp && PANDEPTH_TIMING=1 $GRAFT_REPO_ROOT/pandepth_amd/pandepth -i /tmp/s.bam -o /tmp/o_t -t 16 2>&1 | grep "decode + scatter" ) >> $O/h2d_numa.txt
tail -4 $O/h2d_numa.txt
rm -f /tmp/s.bam* /tmp/o_t*
