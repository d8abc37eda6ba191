#!/bin/bash
# round 5, call 4: the whole gpu suite; then `-w 100 -a` on 2e8 records with the host half of the gzip streams reworked (pieces, parallel splice, writer thread):
# new and old parse geometry, the rounds' stage times on this box.
O=$GRAFT_REPO_ROOT/gpurun_out/r5c4; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
CLI=$GRAFT_REPO_ROOT/pandepth_amd/pandepth; GEN=tools/bamgen
( time timeout 2400 python -m pytest tests/ -q -m gpu -x > $O/pytest_gpu.log 2>&1 ) 2> $O/pytest_time.txt; echo "pytest rc $?" >> $O/pytest_gpu.log
tail -6 $O/pytest_gpu.log | cut -c1-300; cat $O/pytest_time.txt | grep real
$GEN -o /tmp/s.bam -n 200000000 -t 32 2> $O/gen.txt
run() { # name env...
  n=$1; shift
  ( cd /tmp && env "$@" PANDEPTH_TIMING=1 timeout 600 $CLI -i /tmp/s.bam -w 100 -a -o /tmp/o_$n -t 16 > $O/site_$n.log 2>&1 ); echo "$n rc $? $(grep -E 'per-site|decode \+ scatter' $O/site_$n.log | tr '\n' ' ' | cut -c1-420)" >> $O/summary.txt
  sleep 1
}
run new
run old PANDEPTH_TUNE=lz_group=0 PGZ_DEV_CHUNK_KB=16 PGZ_DEV_TAIL_KB=4
run new2
run old2 PANDEPTH_TUNE=lz_group=0 PGZ_DEV_CHUNK_KB=16 PGZ_DEV_TAIL_KB=4
run t32 PANDEPTH_TUNE=x=1
cmp /tmp/o_new.SiteDepth.gz /tmp/o_old.SiteDepth.gz && echo "SiteDepth identical (new vs old geometry)" >> $O/summary.txt
cmp /tmp/o_new.win.stat.gz /tmp/o_old.win.stat.gz && echo "win.stat identical" >> $O/summary.txt
sha256sum /tmp/o_new.SiteDepth.gz >> $O/summary.txt
( cd /tmp && PGZ_DEBUG=1 PANDEPTH_TIMING=1 timeout 600 $CLI -i /tmp/s.bam -w 100 -a -o /tmp/o_dbg -t 16 2>&1 | grep 'pgz' | sed -n '20,31p' | cut -c1-260 >> $O/summary.txt )
rm -f /tmp/s.bam* /tmp/o_*
cat $O/summary.txt
