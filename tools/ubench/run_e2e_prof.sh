mkdir -p gpurun_out/r2e; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=${R:-4e7}; T=${T:-16}
tools/bamgen -o /tmp/e2e.bam -n $R -t 16 2> gpurun_out/r2e/gen.log
CLI=$GRAFT_REPO_ROOT/pandepth_amd/pandepth
PANDEPTH_TIMING=1 $CLI -i /tmp/e2e.bam -o /tmp/dd -t $T > /dev/null 2> gpurun_out/r2e/warm.log
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2e/prof -o cli -- $CLI -i /tmp/e2e.bam -o /tmp/dd -t $T > $GRAFT_REPO_ROOT/gpurun_out/r2e/prof.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r2e/prof -name "*kernel_stats*" | head -3
f=$(find gpurun_out/r2e/prof -name "*kernel_stats.csv" | head -1); cat $f | cut -c1-200 | head -20
