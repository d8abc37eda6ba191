#!/bin/bash
# round 5, call 3: above 2^32 cells (engine + executable), the dense no-index file again, the gzip-stream tests, and `-w 100 -a` on 2e8 records with the
# parse's old and new geometry (16 + 4 KiB from memory / 8 + 2 KiB out of LDS), files compared.
O=$GRAFT_REPO_ROOT/gpurun_out/r5c3; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
CLI=$GRAFT_REPO_ROOT/pandepth_amd/pandepth; GEN=tools/bamgen
timeout 1500 python -m pytest tests/test_gpu_above4g.py "tests/test_cli_gpu.py::test_compact_session_over_many_batches" tests/test_lz77.py tests/test_pgzip.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -15 $O/pytest.log | cut -c1-300
$GEN -o /tmp/s.bam -n 200000000 -t 32 2> $O/gen.txt
run() { # name env...
  n=$1; shift
  ( cd /tmp && env "$@" PANDEPTH_TIMING=1 timeout 600 $CLI -i /tmp/s.bam -w 100 -a -o /tmp/o_$n -t 16 > $O/site_$n.log 2>&1 ); echo "$n rc $? $(grep -E 'per-site|table gzip|decode \+ scatter' $O/site_$n.log | tr '\n' ' ' | cut -c1-400)" >> $O/summary.txt
  sleep 1
}
run new
run old PANDEPTH_TUNE=lz_group=0 PGZ_DEV_CHUNK_KB=16 PGZ_DEV_TAIL_KB=4
run new2
run mem82 PANDEPTH_TUNE=lz_group=0
cmp /tmp/o_new.SiteDepth.gz /tmp/o_old.SiteDepth.gz && echo "SiteDepth identical (new vs old geometry)" >> $O/summary.txt
cmp /tmp/o_new.win.stat.gz /tmp/o_old.win.stat.gz && echo "win.stat identical" >> $O/summary.txt
cmp /tmp/o_mem82.SiteDepth.gz /tmp/o_old.SiteDepth.gz && echo "SiteDepth identical (mem 8+2 vs old)" >> $O/summary.txt
ls -la /tmp/o_new.SiteDepth.gz >> $O/summary.txt
( cd /tmp && PGZ_DEBUG=1 PANDEPTH_TIMING=1 timeout 600 $CLI -i /tmp/s.bam -w 100 -a -o /tmp/o_dbg -t 16 2>&1 | grep 'pgz. round' | awk '{for(i=1;i<=NF;i++){if($i=="chunks"){c+=$(i-1)} if($i=="parsed" && $(i+1)=="again"){m+=$(i-1)}}} END{print "rounds",NR,"chunks",c,"mended",m}' >> $O/summary.txt )
rm -f /tmp/s.bam* /tmp/o_*
cat $O/summary.txt
