mkdir -p gpurun_out/e2et; cd /tmp && export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/e2et
mkdir -p /tmp/e2e && cd /tmp/e2e && timeout 300 $GRAFT_REPO_ROOT/tools/bamgen -o s.bam -n 100000000 -t 32 2>&1 | tail -1
P=$GRAFT_REPO_ROOT/pandepth_amd/pandepth
for T in 6 6; do PANDEPTH_TIMING=1 $P -i s.bam -o m -t 16 2>&1 | grep -E "pd_create|pd_decode|slot|wait|decode \+|engine create" | cut -c1-600; done
