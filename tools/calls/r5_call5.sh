#!/bin/bash
# round 5, call 5: (a) the new 12-byte-session tests; (b) decode on the 3e8-record file with the placement copies as k_copy_words, and the inflate kernel compiled for
# 5 waves per SIMD (96 VGPRs, no scratch) against 6 (80 VGPRs, 76 B of scratch per lane): same binary, the library preloaded; (c) a timeline of the default.
O=$GRAFT_REPO_ROOT/gpurun_out/r5c5; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
CLI=$GRAFT_REPO_ROOT/pandepth_amd/pandepth; GEN=tools/bamgen; ALT=$GRAFT_REPO_ROOT/pandepth_amd/alt5/libpandepth_amd.so
timeout 900 python -m pytest tests/test_host_generated.py -x -q -m gpu -k "device_chain or queued" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
$GEN -o /tmp/s.bam -n 300000000 -t 32 2> $O/gen.txt
$CLI -i /tmp/s.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 1
run() { # name preload tune
  ( cd /tmp && LD_PRELOAD=$2 PANDEPTH_TUNE="$3" PANDEPTH_TIMING=1 timeout 300 $CLI -i /tmp/s.bam -o /tmp/o_$1 -t 16 > $O/cli_$1.log 2>&1 ); echo "$1 rc $? $(grep -E 'decode \+ scatter' $O/cli_$1.log) | inflate ms summed: $(grep -o 'inflate [0-9.]*' $O/cli_$1.log | head -1)" >> $O/summary.txt
  cmp /tmp/o_$1.chr.stat.gz /tmp/warm.chr.stat.gz >> $O/summary.txt 2>&1 || echo "$1 DIFFERENT" >> $O/summary.txt
  sleep 1
}
for rep in 1 2 3; do
  run w6_$rep "" ""
  run w5_$rep $ALT ""
done
run w5_x1 $ALT "dd_depth=1"
run w6_x1 "" "dd_depth=1"
run w5_iw22 $ALT "inflate_waves=22"
cat $O/summary.txt
( cd /tmp && PANDEPTH_TIMING=1 PANDEPTH_ORDERLY_EXIT=1 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tr -o cli -- $CLI -i /tmp/s.bam -o /tmp/o_tr -t 16 > $O/tr.log 2>&1 )
grep -E "decode \+ scatter" $O/tr.log
# isolated inflate kernel, both builds (pd_x_bgzf_inflate on the first 1 GB of the file)
head -c 1000000000 /tmp/s.bam > /tmp/s1g.bam
for L in pandepth_amd/libpandepth_amd.so pandepth_amd/alt5/libpandepth_amd.so; do
  PANDEPTH_AMD_LIB=$GRAFT_REPO_ROOT/$L WAVES=20,22 REPS=5 MAX_BYTES=1.0e9 timeout 300 python tools/ubench/inflate_ab.py /tmp/s1g.bam 2>&1 | sed "s|^|$L: |" >> $O/summary.txt
done
tail -4 $O/summary.txt
rm -f /tmp/s.bam* /tmp/s1g.bam /tmp/o_* /tmp/warm*
