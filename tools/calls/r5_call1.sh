#!/bin/bash
# round 5, call 1: the queued decode path on the GPU for the first time — the decode-related gpu tests, then the 3e8-record file (15.9 GB BAM)
# through the executable with the readers' depth and the chain confirmation varied (same binary), every table compared with the first.
O=$GRAFT_REPO_ROOT/gpurun_out/r5c1; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
CLI=$GRAFT_REPO_ROOT/pandepth_amd/pandepth; GEN=tools/bamgen
timeout 900 python -m pytest tests/test_host_generated.py tests/test_cli_gpu.py tests/test_gpu_bgzf.py -x -q -m gpu > $O/pytest_decode.log 2>&1; echo "pytest rc $?" >> $O/pytest_decode.log
tail -5 $O/pytest_decode.log
[ -x $GEN ] || g++ -O2 -std=c++17 -pthread tools/bamgen.cpp -lz -ldl -o $GEN
$GEN -o /tmp/s.bam -n 300000000 -t 32 2> $O/gen.txt
$CLI -i /tmp/s.bam -o /tmp/warm -t 16 > /dev/null 2>&1
run() { # name tune
  for rep in 1 2; do
    ( cd /tmp && PANDEPTH_TUNE="$2" PANDEPTH_TIMING=1 timeout 300 $CLI -i /tmp/s.bam -o /tmp/o_$1 -t 16 > $O/cli_$1_$rep.log 2>&1 ); echo "$1 rep $rep rc $?" >> $O/summary.txt
    grep -E "decode \+ scatter" $O/cli_$1_$rep.log >> $O/summary.txt
  done
  cmp /tmp/o_$1.chr.stat.gz /tmp/warm.chr.stat.gz >> $O/summary.txt 2>&1 && echo "$1 identical" >> $O/summary.txt
}
run default ""
run depth1 "dd_depth=1"
run r4way "dd_depth=1,decode_fast=0"
run depth3x4 "dd_depth=3,dd_threads=4"
run depth2x4 "dd_depth=2,dd_threads=4"
run depth4x3 "dd_depth=4,dd_threads=3"
run b64 "dd_batch_mb=64"
run b16 "dd_batch_mb=16"
run waves16 "inflate_waves=16"
rm -f /tmp/s.bam /tmp/o_* /tmp/warm*
cat $O/summary.txt
grep -h "device decode\|decode entry" $O/cli_default_2.log | cut -c1-900
