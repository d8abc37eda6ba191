# round 3, call 12: config 4 end to end, device parse only, with the stage timers of pd_deflate_parse (PD_LZ_DEBUG); files compared with the reference's
O=$GRAFT_REPO_ROOT/gpurun_out/r3c12; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p /tmp/e2e && cd /tmp/e2e
$GRAFT_REPO_ROOT/tools/bamgen -o w.bam -n 20000000 -t 32 2> $O/gen.log
python3 - > $O/e2e.log 2>&1 <<'PY'
import os, subprocess, time
R=os.environ["GRAFT_REPO_ROOT"]; cli=R+"/pandepth_amd/pandepth"; ref=R+"/oracle/_ref/pandepth_ref"
def run(cmd, env=None, tag=""):
    t0=time.time(); p=subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, **(env or {}))); dt=time.time()-t0
    print("%s: wall %.3f s rc %d" % (tag, dt, p.returncode))
    for l in p.stderr.decode().splitlines():
        if any(k in l for k in ("[lz]","per-site writer","per-site file","table gzip","deflate_parse","[pgz]","decode + scatter","engine create","scan + statistics")): print("   ", l[:260])
    return dt
run([cli,"-i","w.bam","-w","100","-a","-o","dev","-t","16"], {"PANDEPTH_TIMING":"1","PGZ_DEBUG":"1","PD_LZ_DEBUG":"1"}, "device parse, stage timers")
for k in range(3):
    time.sleep(1); run([cli,"-i","w.bam","-w","100","-a","-o","dev","-t","16"], {"PANDEPTH_TIMING":"1"}, "device parse #%d" % k)
if os.environ.get("WITH_REF"):
    run([ref,"-i","w.bam","-w","100","-a","-o","ref","-t","36"], None, "reference")
    for a,b in (("dev.SiteDepth.gz","ref.SiteDepth.gz"),("dev.win.stat.gz","ref.win.stat.gz")):
        print(a, "==", b, open(a,"rb").read()==open(b,"rb").read())
PY
rm -rf /tmp/e2e
