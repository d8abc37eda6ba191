# round 3, call 13: the device parse's geometry (chunk / tail / round size) on config 4, and what pinned host memory costs
O=$GRAFT_REPO_ROOT/gpurun_out/r3c13; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_lz77.py -x -q -m gpu > $O/pytest_lz77.log 2>&1
mkdir -p /tmp/e2e && cd /tmp/e2e
$GRAFT_REPO_ROOT/tools/bamgen -o w.bam -n 20000000 -t 32 2> $O/gen.log
python3 - > $O/e2e.log 2>&1 <<'PY'
import os, subprocess, time, hashlib
R=os.environ["GRAFT_REPO_ROOT"]; cli=R+"/pandepth_amd/pandepth"
def run(cmd, env=None, tag="", show=()):
    t0=time.time(); p=subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, **(env or {}))); dt=time.time()-t0
    ps=[l for l in p.stderr.decode().splitlines() if "per-site file" in l or "per-site writer" in l]
    print("%s: wall %.3f s rc %d | %s" % (tag, dt, p.returncode, " | ".join(x.strip()[9:120] for x in ps)))
    shown=0
    for l in p.stderr.decode().splitlines():
        if any(k in l for k in show) and shown < 3: print("   ", l[:260]); shown+=1
    return dt
sha=None
for ch,tl,mb in ((64,8,192),(32,8,192),(32,8,128),(32,4,128),(16,4,128),(16,4,96),(32,8,96)):
    for k in range(2):
        time.sleep(0.5)
        run([cli,"-i","w.bam","-w","100","-a","-o","dev","-t","16"], {"PANDEPTH_TIMING":"1","PGZ_DEV_CHUNK_KB":str(ch),"PGZ_DEV_TAIL_KB":str(tl),"PGZ_DEV_BATCH_MB":str(mb),"PD_LZ_DEBUG":"1" if k==0 else ""} if k==0 else
            {"PANDEPTH_TIMING":"1","PGZ_DEV_CHUNK_KB":str(ch),"PGZ_DEV_TAIL_KB":str(tl),"PGZ_DEV_BATCH_MB":str(mb)}, "chunk %d KiB tail %d KiB round %d MiB #%d" % (ch,tl,mb,k), ("[lz]",) if k==0 else ())
    h=hashlib.sha256(open("dev.SiteDepth.gz","rb").read()).hexdigest()
    if sha is None: sha=h
    print("   same file as the first geometry:", h==sha)
PY
rm -rf /tmp/e2e
