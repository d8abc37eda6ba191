#!/bin/bash
# round 6, call 22: the per-site writer with four parse calls of a round in flight (-X lz_calls=4, the new default) against two, `-w 100 -a` on 2e8 records;
# the gzip / LZ77 GPU tests
O=$GRAFT_REPO_ROOT/gpurun_out/r6c22; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_pgzip.py tests/test_lz77.py tests/test_cli_gpu.py -m gpu -q --timeout 900 -x ) > $O/pytest_gpu.log 2>&1; echo rc=$? >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
tools/bamgen -o /tmp/m.bam -n 200000000 -t 32 2>> $O/gen.log
P=$GRAFT_REPO_ROOT/pandepth_amd
$P/pandepth -i /tmp/m.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 1
for rep in 1 2 3; do for tn in lz_calls=2 lz_calls=4 lz_calls=4,lz_mix=1; do
  t0=$(date +%s.%N)
  ( cd /tmp && PANDEPTH_TUNE=$tn PANDEPTH_TIMING=1 timeout 600 $P/pandepth -i /tmp/m.bam -w 100 -a -o /tmp/o_s -t 16 > $O/site_${tn//[=,]/_}_$rep.log 2>&1 )
  t1=$(date +%s.%N)
  echo "$tn run $rep: wall $(awk "BEGIN{print $t1-$t0}") | $(grep -E 'decode \+ scatter|per-site file|per-site writer' $O/site_${tn//[=,]/_}_$rep.log | tr -s ' ' | tr '\n' ' ' | cut -c1-600) | $(md5sum < /tmp/o_s.SiteDepth.gz | cut -c1-8)" >> $O/summary.txt
  sleep 1
done; done
rm -f /tmp/o_* /tmp/warm* /tmp/m.bam*
cat $O/summary.txt
