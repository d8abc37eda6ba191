# round 5, call 18: A/B of inflate-kernel variants through tools/ubench/wave_debug (same file, same box, 5120 one-wave workgroups, CRC on):
# base = the committed kernel; new = scalar code-length stream + CRC by four tables + readlane broadcasts; p16 = 16-byte pieces in phase 3;
# r1 = sources redirected one level; ticks = the per-phase shader-clock breakdown of a variant
O=$GRAFT_REPO_ROOT/gpurun_out/r5c18; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p /tmp/e2e && cd /tmp/e2e
$GRAFT_REPO_ROOT/tools/bamgen -o w.bam -n 20000000 -t 32 2> $O/gen.log
rm -f $O/ab.log
for B in base v3 ald ast aboth base v3 ald ast aboth v3_ticks aboth_ticks; do
  echo "== $B" >> $O/ab.log
  CHECK=2000 timeout 120 $GRAFT_REPO_ROOT/tools/ubench/wd_$B w.bam 5120 1000000 60 2>&1 | grep -v " 0.0 %" >> $O/ab.log
done
rm -rf /tmp/e2e
cat $O/ab.log
