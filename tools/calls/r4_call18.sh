#!/bin/bash
# round 4, call 18: the parse with the visit behind a jump fetched ahead too (slot E) — isolated A/B (text in memory vs in LDS, texts of one and two
# full rounds of LDS workgroups), the 2e8-record -w 100 -a run with both
O=$GRAFT_REPO_ROOT/gpurun_out/r4c18; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python - <<'PY'
import random
random.seed(5)
d = 30; out = []; n = 0
for c in range(6):
    for i in range(1200000):
        d = max(0, d + random.choice([-1, 0, 0, 0, 0, 1]))
        out.append("chr%d\t%d\t%d\n" % (c + 1, 1000000 + i, d))
t = "".join(out).encode()
open("/tmp/site37.txt", "wb").write(t[:2304 * 16384])
open("/tmp/site75.txt", "wb").write(t[:2 * 2304 * 16384])
open("/tmp/site60.txt", "wb").write(t[:60479617])
PY
: > $O/ab.txt
for f in site37 site75 site60; do for g in 0 16; do
  LZ_GROUP=$g PD_LZ_DEBUG=1 timeout 300 tests/harness/lz77_gpu_check /tmp/$f.txt 16384 4096 > /tmp/ab1.txt 2>&1
  echo "$f lz_group $g: $(grep groups /tmp/ab1.txt | tail -1 | cut -c1-60) | $(tail -1 /tmp/ab1.txt)" >> $O/ab.txt; done; done
cat $O/ab.txt
CLI=pandepth_amd/pandepth; GEN=tools/bamgen
[ -x $GEN ] || g++ -O2 -std=c++17 -pthread tools/bamgen.cpp -lz -ldl -o $GEN
TIMEFORMAT='wall %R s'
$GEN -o /tmp/m.bam -n 200000000 -t 32 2> $O/site.txt
run() { tag=$1; shift; for r in 1 2; do ( time env "$@" PANDEPTH_TIMING=1 $CLI -i /tmp/m.bam -w 100 -a -o /tmp/o_$tag -t 16 ) 2>&1 | grep -E "wall|per-site writer \(text" | tr '\n' ' '; echo; done | sed "s/^/$tag: /" >> $O/site.txt; sha256sum /tmp/o_$tag.SiteDepth.gz | cut -c1-16 >> $O/site.txt; }
run mem X=1
run lds PANDEPTH_TUNE=lz_group=16
run lds72 PANDEPTH_TUNE=lz_group=16 PGZ_DEV_BATCH_MB=72
run mem72 PGZ_DEV_BATCH_MB=72
cat $O/site.txt
