#!/bin/bash
# round 6, call 7: the inflate kernel with phase 2 balanced over a pool of parts (three checkpoints; `bal`), with literals gathered eight to a store (`lit`),
# both (`ballit`), against round 5's kernel (`r5`): tools/ubench/wave_debug (isolated, 5120 one-wave workgroups, CRC on, the first CHECK members against zlib),
# then the same four builds of the library under the executable on the 3e8-record file (tools/ubench/var_*: pandepth + libpandepth_amd.so), interleaved;
# k_direct_c8 with 16-byte loads (direct_ab.py variants 57xx / 58xx); what lies outside main() of a run (the wall-clock stamps of PANDEPTH_TIMING)
O=$GRAFT_REPO_ROOT/gpurun_out/r6c7; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_bgzf.py tests/test_inflate_core.py -m gpu -q --timeout 600 -x ) > $O/pytest_gpu.log 2>&1; echo rc=$? >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
mkdir -p /tmp/e2e && cd /tmp/e2e
$GRAFT_REPO_ROOT/tools/bamgen -o w.bam -n 20000000 -t 32 2> $O/gen.log
rm -f $O/ab.log
for B in r5 bal ballit lit ck3 bal16 r5 bal ballit lit r5_ticks bal_ticks ballit_ticks; do
  echo "== $B" >> $O/ab.log
  CHECK=4000 timeout 120 $GRAFT_REPO_ROOT/tools/ubench/wd_$B w.bam 5120 1000000 60 2>&1 | grep -v " 0.0 %" >> $O/ab.log
done
# a batch-sized launch (3 200 members): what the pipeline's kernel sees
for B in r5 bal ballit r5 bal ballit; do
  echo "== $B, 3200 members" >> $O/ab.log
  CHECK=100 timeout 120 $GRAFT_REPO_ROOT/tools/ubench/wd_$B w.bam 5120 3200 60 2>&1 | grep -v " 0.0 %" >> $O/ab.log
done
cat $O/ab.log
rm -f w.bam
cd $GRAFT_REPO_ROOT
VARIANTS=c0,c5704,c5702,c5802,c5706,c0,c5704 EXPORT=0 timeout 600 python tools/ubench/direct_ab.py > $O/direct_ab.txt 2>&1; grep -E "variant" $O/direct_ab.txt | cut -c1-200
GEN=tools/bamgen
$GEN -o /tmp/s.bam -n 300000000 -t 32 2>> $O/gen.log
run() { # name dir-of-the-executable
  local t0=$(date +%s.%N)
  ( cd /tmp && PANDEPTH_TIMING=1 timeout 300 $2/pandepth -i /tmp/s.bam -o /tmp/o_$1 -t 16 > $O/cli_$1.log 2>&1 ); local rc=$?
  local t1=$(date +%s.%N)
  local en=$(grep 'main entered' $O/cli_$1.log | sed 's/.* at \([0-9.]*\) .*/\1/'); local lv=$(grep 'main leaving' $O/cli_$1.log | sed 's/.* at \([0-9.]*\) .*/\1/')
  echo "$1 rc $rc wall $(awk "BEGIN{print $t1-$t0}") s: exec -> main $(awk "BEGIN{print $en-$t0}"), main $(awk "BEGIN{print $lv-$en}"), main's end -> reaped $(awk "BEGIN{print $t1-$lv}") | $(grep -E 'decode \+ scatter|engine create' $O/cli_$1.log | tr -s ' ' | tr '\n' ';') | $(grep -E 'summed over' $O/cli_$1.log | sed 's/.*device ms summed over batches: \([^;]*\);.*/\1/') | $(zcat /tmp/o_$1.chr.stat.gz 2>/dev/null | md5sum | cut -c1-8)" >> $O/summary.txt
  sleep 1
}
$GRAFT_REPO_ROOT/pandepth_amd/pandepth -i /tmp/s.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 1
for rep in 1 2 3 4; do
  run r5_$rep $GRAFT_REPO_ROOT/tools/ubench/var_r5
  run bal_$rep $GRAFT_REPO_ROOT/pandepth_amd
  run ballit_$rep $GRAFT_REPO_ROOT/tools/ubench/var_ballit
  run lit_$rep $GRAFT_REPO_ROOT/tools/ubench/var_lit
done
rm -f /tmp/o_* /tmp/warm* /tmp/s.bam*
cat $O/summary.txt | cut -c1-700
