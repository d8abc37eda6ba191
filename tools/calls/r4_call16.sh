#!/bin/bash
# round 4, call 16: the LDS parse (k_lz_parse_lds) — zlib-identity tests, isolated A/B against the parse with its text in memory, the -w 100 -a runs
# (2e7 and 2e8 records) with both, chunk geometries; the box's host-to-device copy rates
O=$GRAFT_REPO_ROOT/gpurun_out/r4c16; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python - > /tmp/site60.txt <<'PY'
import random, sys
random.seed(5)
d = 30
w = sys.stdout.write
for c in range(3):
    for i in range(1200000):
        d = max(0, d + random.choice([-1, 0, 0, 0, 0, 1]))
        w("chr%d\t%d\t%d\n" % (c + 1, 1000000 + i, d))
PY
ls -la /tmp/site60.txt > $O/ab.txt
for v in "0 16384 4096" "16 16384 4096" "4 16384 4096" "16 8192 2048" "0 8192 2048" "16 32768 4096"; do set -- $v
  LZ_GROUP=$1 PD_LZ_DEBUG=1 timeout 300 tests/harness/lz77_gpu_check /tmp/site60.txt $2 $3 > /tmp/ab1.txt 2>&1
  echo "lz_group $1 chunk $2 tail $3: $(grep groups /tmp/ab1.txt | tail -1) $(tail -1 /tmp/ab1.txt)" >> $O/ab.txt; done
cat $O/ab.txt
( timeout 600 python -m pytest tests/test_lz77.py tests/test_pgzip.py tests/test_cli_gpu.py -m gpu -q -x --timeout 300 -k "parse or per_site or resident or geometry or (byte_identical and not host_decode)" ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
CLI=pandepth_amd/pandepth; GEN=tools/bamgen
[ -x $GEN ] || g++ -O2 -std=c++17 -pthread tools/bamgen.cpp -lz -ldl -o $GEN
TIMEFORMAT='wall %R s'
$GEN -o /tmp/w.bam -n 20000000 -t 32 2>> $O/site.txt
run() { tag=$1; shift; for r in 1 2; do ( time env "$@" PANDEPTH_TIMING=1 $CLI -i $BAM -w 100 -a -o /tmp/o_$tag -t 16 ) 2>&1 | grep -E "wall|per-site|parse calls" | tr '\n' ' ' ; echo; done | sed "s/^/$tag: /" >> $O/site.txt; sha256sum /tmp/o_$tag*.gz | awk '{print $1}' | tr '\n' ' ' >> $O/site.txt; echo >> $O/site.txt; }
BAM=/tmp/w.bam
run lds X=1
run mem PANDEPTH_TUNE=lz_group=0
run lds8k PGZ_DEV_CHUNK_KB=8 PGZ_DEV_TAIL_KB=2
run lds32k PGZ_DEV_CHUNK_KB=32 PGZ_DEV_TAIL_KB=4
rm -f /tmp/w.bam /tmp/o_*
$GEN -o /tmp/m.bam -n 200000000 -t 32 2>> $O/site.txt
BAM=/tmp/m.bam
( time env PANDEPTH_TIMING=1 PGZ_DEBUG=1 $CLI -i $BAM -w 100 -a -o /tmp/o_mid -t 16 ) > $O/mid_lds.log 2>&1; grep -E "wall|\[timing\]" $O/mid_lds.log | tail -14 >> $O/site.txt
( time env PANDEPTH_TIMING=1 PANDEPTH_TUNE=lz_group=0 $CLI -i $BAM -w 100 -a -o /tmp/o_mid0 -t 16 ) > $O/mid_mem.log 2>&1; grep -E "wall|per-site" $O/mid_mem.log | tail -4 >> $O/site.txt
cmp /tmp/o_mid.SiteDepth.gz /tmp/o_mid0.SiteDepth.gz && echo "mid: same SiteDepth.gz" >> $O/site.txt
ls -la /tmp/o_mid* >> $O/site.txt
cat $O/site.txt
python tools/ubench/h2d_bw.py > $O/h2d.txt 2>&1; cat $O/h2d.txt
