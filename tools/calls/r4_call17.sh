#!/bin/bash
# round 4, call 17: the per-site file at 2e8 records (680 Mb, 11.9 GB of rows) with pairs of parses that do not meet mended in place; LZ77 tests incl. the LDS kernel
O=$GRAFT_REPO_ROOT/gpurun_out/r4c17; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( timeout 500 python -m pytest tests/test_lz77.py tests/test_pgzip.py -m gpu -q -x --timeout 300 ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
CLI=pandepth_amd/pandepth; GEN=tools/bamgen
[ -x $GEN ] || g++ -O2 -std=c++17 -pthread tools/bamgen.cpp -lz -ldl -o $GEN
TIMEFORMAT='wall %R s'
$GEN -o /tmp/m.bam -n 200000000 -t 32 2> $O/site.txt
for r in 1 2; do ( time env PANDEPTH_TIMING=1 PGZ_DEBUG=1 $CLI -i /tmp/m.bam -w 100 -a -o /tmp/o_mid -t 16 ) > $O/mid$r.log 2>&1; grep -E "wall|\[timing\]" $O/mid$r.log | grep -v "pd_deflate_parse\|pd_create" | tail -9 >> $O/site.txt; grep -E "[1-9][0-9]* parsed again" $O/mid$r.log | head -3 >> $O/site.txt; done
sha256sum /tmp/o_mid.SiteDepth.gz /tmp/o_mid.win.stat.gz >> $O/site.txt; ls -la /tmp/o_mid* >> $O/site.txt
( time env PANDEPTH_TIMING=1 PANDEPTH_TUNE=lz_group=16 $CLI -i /tmp/m.bam -w 100 -a -o /tmp/o_lds -t 16 ) > $O/mid_lds.log 2>&1; grep -E "wall|per-site" $O/mid_lds.log | tail -3 >> $O/site.txt
cmp /tmp/o_mid.SiteDepth.gz /tmp/o_lds.SiteDepth.gz && echo "lz_group=16: same SiteDepth.gz" >> $O/site.txt
cat $O/site.txt
