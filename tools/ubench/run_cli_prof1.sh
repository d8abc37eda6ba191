mkdir -p gpurun_out/cp1; cd /tmp && export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/cp1
mkdir -p /tmp/e2e && cd /tmp/e2e && timeout 300 $GRAFT_REPO_ROOT/tools/bamgen -o s.bam -n 40000000 -t 32 2>&1 | tail -1
(PANDEPTH_DD_THREADS=1 PANDEPTH_ORDERLY_EXIT=1 timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o cli -- $GRAFT_REPO_ROOT/pandepth_amd/pandepth -i s.bam -o mine -t 16 > $O/kt.log 2>&1)
F=$(find $O/kt -name "*kernel_stats.csv" | head -1); cut -c1-60,120-400 "$F" | head -8; rm -rf $O/kt
