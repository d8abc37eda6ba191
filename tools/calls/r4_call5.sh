#!/bin/bash
# round 4, call 5: inflate variants (root widths / waves per CU), the direct kernel with one chunk sequence over both streams, the
# executable on a 3e8-record file (compact session with asynchronous placement), decoder + CLI tests
O=$GRAFT_REPO_ROOT/gpurun_out/r4c5; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
$GRAFT_REPO_ROOT/tools/bamgen -o /tmp/ab.bam -n 12000000 -t 32 2> $O/gen.log
U=$GRAFT_REPO_ROOT/tools/ubench
for pass in 1 2; do
  PANDEPTH_AMD_LIB=$U/libpd_inflate_old.so WAVES=16 timeout 200 python $U/inflate_ab.py /tmp/ab.bam >> $O/ab.log 2>&1
  PANDEPTH_AMD_LIB=$U/libpd_inflate_v1.so WAVES=16 timeout 200 python $U/inflate_ab.py /tmp/ab.bam >> $O/ab.log 2>&1
  PANDEPTH_AMD_LIB=$U/libpd_inflate_v2.so WAVES=20,21 timeout 200 python $U/inflate_ab.py /tmp/ab.bam >> $O/ab.log 2>&1
  PANDEPTH_AMD_LIB=$U/libpd_inflate_v3.so WAVES=20,24 timeout 200 python $U/inflate_ab.py /tmp/ab.bam >> $O/ab.log 2>&1
  PANDEPTH_AMD_LIB=$U/libpd_inflate_v4.so WAVES=20,24 timeout 200 python $U/inflate_ab.py /tmp/ab.bam >> $O/ab.log 2>&1
done
cat $O/ab.log
cd $GRAFT_REPO_ROOT && VARIANTS=c803,c703,c704,c802 EXPORT=0 timeout 300 python tools/ubench/direct_ab.py > $O/direct_ab.log 2>&1; tail -5 $O/direct_ab.log
mkdir -p /tmp/e2e && cd /tmp/e2e && $GRAFT_REPO_ROOT/tools/bamgen -o s.bam -n 300000000 -t 32 2>> $O/gen.log
for k in 1 2 3; do ( time PANDEPTH_TIMING=1 $GRAFT_REPO_ROOT/pandepth_amd/pandepth -i s.bam -o m$k -t 16 ) >> $O/cli_3e8.log 2>&1; sleep 1; done
grep -E "decode \+ scatter|pd_decode_end|decode entry|real|device decode" $O/cli_3e8.log | cut -c1-420
$GRAFT_REPO_ROOT/oracle/_ref/pandepth_ref -i s.bam -o r -t 36 > /dev/null 2>&1; cmp m1.chr.stat.gz r.chr.stat.gz && echo E2E_IDENTICAL >> $O/cli_3e8.log; tail -1 $O/cli_3e8.log
rm -rf /tmp/e2e
cd $GRAFT_REPO_ROOT && timeout 900 python -m pytest tests/test_gpu_bgzf.py tests/test_cli_gpu.py -q -x --timeout 600 > $O/pytest_cli.log 2>&1; tail -3 $O/pytest_cli.log
