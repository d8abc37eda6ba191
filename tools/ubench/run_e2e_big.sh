mkdir -p gpurun_out/e2ebig; cd /tmp && export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/e2ebig
mkdir -p /tmp/e2e && cd /tmp/e2e && (time timeout 600 $GRAFT_REPO_ROOT/tools/bamgen -o b.bam -n 300000000 -t 32) 2>&1 | tail -4
ls -la b.bam | awk '{print $5}'
python3 - <<'PY' | tee $O/big.log
import subprocess, time, os
R=os.environ["GRAFT_REPO_ROOT"]; cli=R+"/pandepth_amd/pandepth"; ref=R+"/oracle/_ref/pandepth_ref"
best=1e9
for k in range(2):
    time.sleep(1.0); t0=time.time(); p=subprocess.run([cli,"-i","b.bam","-o","m","-t","16"],stdout=subprocess.DEVNULL,stderr=subprocess.PIPE,env=dict(os.environ,PANDEPTH_TIMING="1")); dt=time.time()-t0; best=min(best,dt)
    print("run %d wall %.3f" % (k, dt)); print("\n".join(l[:160] for l in p.stderr.decode().splitlines() if "engine create" in l or "decode + scatter" in l or "device decode" in l))
t0=time.time(); subprocess.run([ref,"-i","b.bam","-o","r","-t","36"],stdout=subprocess.DEVNULL,stderr=subprocess.DEVNULL); b=time.time()-t0
print("3e8 records: pandepth %.3f s (%.3e records/s), pandepth_ref %.2f s, byte-identical %s" % (best, 3e8/best, b, open("m.chr.stat.gz","rb").read()==open("r.chr.stat.gz","rb").read()))
PY
rm -f /tmp/e2e/b.bam*
