#!/bin/bash
# round 5, call 8: (a) `-g` (region fetch: many units per batch) after the chain kernel's segmented scan, against decode_fast=0; (b) what a 1-rank RCCL communicator costs in a
# fresh process, under a few environments (no network on these boxes: where does the bootstrap spend its five seconds?)
O=$GRAFT_REPO_ROOT/gpurun_out/r5c8; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
CLI=$GRAFT_REPO_ROOT/pandepth_amd/pandepth; GEN=tools/bamgen
$GEN -o /tmp/s.bam -n 300000000 -t 32 2> $O/gen.txt
python - > $O/annot.txt 2>&1 <<'PY'
import os, sys, json, subprocess, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench
cli = os.path.join(bench.ROOT, "pandepth_amd", "pandepth"); ref = os.path.join(bench.ROOT, "oracle", "_ref", "pandepth_ref")
os.makedirs("/tmp/an", exist_ok=True)
print(json.dumps(bench.e2e_annotation("/tmp/an", "/tmp/s.bam", cli, ref, 16, 3e8)))
for tag, tune in (("fast", ""), ("host_chain", "decode_fast=0"), ("fast_depth2", "dd_depth=2"), ("fast", ""), ("host_chain", "decode_fast=0")):
    time.sleep(1)
    t0 = time.perf_counter()
    p = subprocess.run([cli, "-i", "/tmp/s.bam", "-g", "/tmp/an/c3.gff", "-o", "/tmp/an/x_" + tag, "-t", "16"], env=dict(os.environ, PANDEPTH_TIMING="1", PANDEPTH_TUNE=tune), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    dt = time.perf_counter() - t0
    err = p.stderr.decode()
    print(tag, "wall %.3f" % dt, [ln for ln in err.splitlines() if "decode + scatter" in ln or "device decode" in ln or "decode entry" in ln][:3])
    print(open("/tmp/an/x_%s.gene.stat.gz" % tag, "rb").read() == open("/tmp/an/mine_g.gene.stat.gz", "rb").read())
PY
cut -c1-700 $O/annot.txt
rm -rf /tmp/s.bam* /tmp/an
run() { ( cd /tmp && env "$@" timeout 120 python $GRAFT_REPO_ROOT/tools/ubench/comm_init_time.py 2>&1 | grep -v amdgpu.ids | tail -2 >> $O/comm.txt ); }
run TAG=default
run TAG=default_again
run TAG=ib_disabled NCCL_IB_DISABLE=1
run TAG=socket_lo NCCL_IB_DISABLE=1 NCCL_SOCKET_IFNAME=lo NCCL_NET=Socket
run TAG=no_plugin NCCL_IB_DISABLE=1 NCCL_SOCKET_IFNAME=lo NCCL_NET_PLUGIN=none RCCL_MSCCL_ENABLE=0 RCCL_MSCCLPP_ENABLE=0
run TAG=debug_info NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,ENV
cat $O/comm.txt | cut -c1-300 | tail -60
