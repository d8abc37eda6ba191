# round 5, call 32: the inflate kernel with phase 3's rounds in an LDS window (v4) against the kernel before (v3), tools/ubench/wave_debug:
# same file, same box, 5120 one-wave workgroups, CRC on; every member compared with zlib for the first CHECK members, all statuses checked
O=$GRAFT_REPO_ROOT/gpurun_out/r5c32; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p /tmp/e2e && cd /tmp/e2e
$GRAFT_REPO_ROOT/tools/bamgen -o w.bam -n 20000000 -t 32 2> $O/gen.log
rm -f $O/ab.log
for B in v12 v13 v12 v13 v13_ticks; do
  echo "== $B" >> $O/ab.log
  CHECK=4000 timeout 120 $GRAFT_REPO_ROOT/tools/ubench/wd_$B w.bam 5120 1000000 60 2>&1 | grep -v " 0.0 %" >> $O/ab.log
done
rm -rf /tmp/e2e
cat $O/ab.log
