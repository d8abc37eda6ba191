# round 3, call 14: config 4 with the device parse: full timing lines for chosen geometries (GEOMS="chunkKB,tailKB,roundMB ...")
O=$GRAFT_REPO_ROOT/gpurun_out/r3c14; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p /tmp/e2e && cd /tmp/e2e
$GRAFT_REPO_ROOT/tools/bamgen -o w.bam -n 20000000 -t 32 2> $O/gen.log
python3 - > $O/e2e.log 2>&1 <<'PY'
import os, subprocess, time, hashlib
R=os.environ["GRAFT_REPO_ROOT"]; cli=R+"/pandepth_amd/pandepth"
def run(cmd, env=None, tag="", full=False):
    t0=time.time(); p=subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, **(env or {}))); dt=time.time()-t0
    print("%s: wall %.3f s rc %d" % (tag, dt, p.returncode))
    for l in p.stderr.decode().splitlines():
        if (full and any(k in l for k in ("[lz]","[pgz]"))) or any(k in l for k in ("per-site writer","per-site file","engine create","decode + scatter","scan + stat","table gzip")): print("   ", l[:300])
    return dt
sha=None
for g in os.environ.get("GEOMS","64,8,192 32,8,128").split():
    ch,tl,mb=g.split(",")
    env={"PANDEPTH_TIMING":"1","PGZ_DEV_CHUNK_KB":ch,"PGZ_DEV_TAIL_KB":tl,"PGZ_DEV_BATCH_MB":mb}
    run([cli,"-i","w.bam","-w","100","-a","-o","dev","-t","16"], dict(env, PGZ_DEBUG="1", PD_LZ_DEBUG="1"), "chunk %s KiB tail %s KiB round %s MiB, debug" % (ch,tl,mb), True)
    for k in range(3):
        time.sleep(0.5); run([cli,"-i","w.bam","-w","100","-a","-o","dev","-t","16"], env, "chunk %s KiB tail %s KiB round %s MiB #%d" % (ch,tl,mb,k))
    h=hashlib.sha256(open("dev.SiteDepth.gz","rb").read()).hexdigest()
    if sha is None: sha=h
    print("   same file as the first geometry:", h==sha)
PY
rm -rf /tmp/e2e
