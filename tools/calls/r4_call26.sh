#!/bin/bash
# round 4, call 26: the gzip streams after the last change to host/pgzip.cpp — LZ77 / pgzip GPU tests, and the 2e7-record -w 100 -a run with 2 KiB overlaps and
# 1 MiB rounds (pairs that do not meet about once per hundred chunks, also at the rounds' boundaries: all must be mended inside the resident stream)
O=$GRAFT_REPO_ROOT/gpurun_out/r4c26; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( timeout 400 python -m pytest tests/test_lz77.py tests/test_pgzip.py -m gpu -q -x --timeout 300 ) > $O/pytest.log 2>&1; tail -2 $O/pytest.log
CLI=pandepth_amd/pandepth; GEN=tools/bamgen
[ -x $GEN ] || g++ -O2 -std=c++17 -pthread tools/bamgen.cpp -lz -ldl -o $GEN
$GEN -o /tmp/w.bam -n 20000000 -t 32 2> $O/site.txt
TIMEFORMAT='wall %R s'
( time env PANDEPTH_TIMING=1 $CLI -i /tmp/w.bam -w 100 -a -o /tmp/o_a -t 16 ) > $O/a.log 2>&1
( time env PANDEPTH_TIMING=1 PGZ_DEBUG=1 PGZ_DEV_TAIL_KB=2 PGZ_DEV_BATCH_MB=1 $CLI -i /tmp/w.bam -w 100 -a -o /tmp/o_b -t 16 ) > $O/b.log 2>&1
for t in a b; do grep -E "wall|per-site writer" $O/$t.log | tr '\n' ' ' >> $O/site.txt; echo >> $O/site.txt; done
echo "rounds with mended pairs: $(grep -cE '[1-9][0-9]* parsed again' $O/b.log), pairs mended: $(grep -oE '[0-9]+ parsed again' $O/b.log | awk '{s+=$1} END {print s}')" >> $O/site.txt
cmp /tmp/o_a.SiteDepth.gz /tmp/o_b.SiteDepth.gz && cmp /tmp/o_a.win.stat.gz /tmp/o_b.win.stat.gz && echo "same files" >> $O/site.txt
cat $O/site.txt
