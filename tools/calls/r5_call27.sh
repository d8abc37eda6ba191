#!/bin/bash
# round 5, call 27: the product's inflate kernel alone (pd_x_bgzf_inflate through tools/ubench/inflate_ab.py), 102 037 members of a 2e7-record payload BAM,
# 20 waves per CU: old = pd_inflate_wave.h of d784431 (round 5 before the kernel work), v2 = this tree (the default build), v4 = this tree with registers for
# 5 waves per SIMD and a 7-bit distance root; then one 32 MB batch's worth of members (what a launch in the pipeline holds)
O=$GRAFT_REPO_ROOT/gpurun_out/r5c27; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tools/bamgen -o /tmp/w.bam -n 20000000 -t 32 2> $O/gen.log
for rep in 1 2; do for v in old v2 v4; do
  PANDEPTH_AMD_LIB=$GRAFT_REPO_ROOT/tools/ubench/libpd_inflate_$v.so WAVES=20 REPS=5 MAX_BYTES=1.2e9 timeout 200 python tools/ubench/inflate_ab.py /tmp/w.bam >> $O/ab.log 2>&1
done; done
for v in old v2; do
  PANDEPTH_AMD_LIB=$GRAFT_REPO_ROOT/tools/ubench/libpd_inflate_$v.so WAVES=20 REPS=20 MAX_BYTES=32e6 timeout 200 python tools/ubench/inflate_ab.py /tmp/w.bam >> $O/ab.log 2>&1
done
rm -f /tmp/w.bam*
cat $O/ab.log
