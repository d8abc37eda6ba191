# round 5, call 16: where a member's time goes in the inflate kernel (tools/ubench/wave_debug, instrumented build: shader-clock ticks per phase),
# at the product's occupancy (5120 one-wave workgroups, members from a counter, CRC on) on a 2e7-record payload BAM
O=$GRAFT_REPO_ROOT/gpurun_out/r5c16; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p /tmp/e2e && cd /tmp/e2e
$GRAFT_REPO_ROOT/tools/bamgen -o w.bam -n 20000000 -t 32 2> $O/gen.log
rm -f $O/ticks.log
for B in wave_debug_noticks wave_debug wave_debug wave_debug_noticks; do
  echo "== $B, 5120 workgroups" >> $O/ticks.log
  CHECK=3000 timeout 120 $GRAFT_REPO_ROOT/tools/ubench/$B w.bam 5120 1000000 60 >> $O/ticks.log 2>&1
done
rm -rf /tmp/e2e
cat $O/ticks.log
