# GPU box: end-to-end CLI wall clock on a generated payload BAM (tools/bamgen), device decode vs host decode vs the reference
mkdir -p gpurun_out/r2e; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=${R:-4e7}; T=${T:-16}
timeout 120 tools/bamgen -o /tmp/e2e.bam -n $R -t 16 > gpurun_out/r2e/gen.log 2>&1
CLI=pandepth_amd/pandepth
{
export PANDEPTH_TIMING=1
for i in 1 2; do timeout 60 $CLI -i /tmp/e2e.bam -o /tmp/dd -t $T; echo "--- rc=$?"; done
unset PANDEPTH_TIMING
python3 - <<'PY'
import subprocess, time
def t(cmd, env=None):
    import os
    e=dict(os.environ); e.update(env or {})
    best=1e9
    for _ in range(3):
        a=time.perf_counter(); subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=e, timeout=120); best=min(best,time.perf_counter()-a)
    return best
R=4e7
cli="pandepth_amd/pandepth"
for name,cmd,env in (("device decode -t 16",[cli,"-i","/tmp/e2e.bam","-o","/tmp/dd","-t","16"],{}),
                     ("device decode -t 6",[cli,"-i","/tmp/e2e.bam","-o","/tmp/dd6","-t","6"],{}),
                     ("host decode -t 16",[cli,"-i","/tmp/e2e.bam","-o","/tmp/hd","-t","16"],{"PANDEPTH_DEVICE_DECODE":"0"}),
                     ("reference -t 36",["oracle/_ref/pandepth_ref","-i","/tmp/e2e.bam","-o","/tmp/ref","-t","36"],{})):
    w=t(cmd,env); print("%-22s best of 3: %.3f s wall = %.3e records/s"%(name,w,R/w), flush=True)
PY
cmp /tmp/dd.chr.stat.gz /tmp/ref.chr.stat.gz && echo "device decode == reference: IDENTICAL"
} > gpurun_out/r2e/e2e.log 2>&1
cat gpurun_out/r2e/gen.log gpurun_out/r2e/e2e.log
