mkdir -p gpurun_out/e2etab; cd /tmp && export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/e2etab
cd $GRAFT_REPO_ROOT && (timeout 600 python tests/e2e_fullsize.py 3e6 > $O/fullsize.log 2>&1); grep -E "genome|identical" $O/fullsize.log | cut -c1-200
mkdir -p /tmp/e2e && cd /tmp/e2e && timeout 300 $GRAFT_REPO_ROOT/tools/bamgen -o s.bam -n 100000000 -t 32 2>&1 | tail -1
python3 - <<'PY' | tee $O/modes.log
import subprocess, time, os
R=os.environ["GRAFT_REPO_ROOT"]; cli=R+"/pandepth_amd/pandepth"; ref=R+"/oracle/_ref/pandepth_ref"
def wall(cmd, reps):
    best=1e9
    for k in range(reps):
        time.sleep(1.0); t0=time.time(); subprocess.run(cmd,stdout=subprocess.DEVNULL,stderr=subprocess.DEVNULL); best=min(best,time.time()-t0)
    return best
for name, extra, suf in (("no index (-s)", ["-s"], "chr.stat.gz"), ("-w 1000", ["-w","1000"], "win.stat.gz")):
    a=wall([cli,"-i","s.bam","-o","m","-t","16"]+extra, 2); b=wall([ref,"-i","s.bam","-o","r","-t","36"]+extra, 1)
    same=open("m."+suf,"rb").read()==open("r."+suf,"rb").read()
    print("1e8 records, 5.31 GB BAM, %s: pandepth %.2f s (%.2e records/s), pandepth_ref %.2f s, byte-identical %s" % (name, a, 1e8/a, b, same), flush=True)
PY
