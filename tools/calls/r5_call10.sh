#!/bin/bash
# round 5, call 10: the two parse calls a per-site stream keeps in flight — both out of LDS (default), one out of LDS and one from memory (lz_mix=1), both from memory —
# on the 2e8-record file (`-w 100 -a`, 11.9 GB of per-site text); files compared
O=$GRAFT_REPO_ROOT/gpurun_out/r5c10; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
CLI=$GRAFT_REPO_ROOT/pandepth_amd/pandepth; GEN=tools/bamgen
$GEN -o /tmp/s.bam -n 200000000 -t 32 2> $O/gen.txt
run() { # name tune
  ( cd /tmp && PANDEPTH_TUNE="$2" PANDEPTH_TIMING=1 timeout 600 $CLI -i /tmp/s.bam -w 100 -a -o /tmp/o_$1 -t 16 > $O/site_$1.log 2>&1 ); echo "$1 rc $? $(grep -E 'per-site' $O/site_$1.log | tr '\n' ' ' | cut -c1-330)" >> $O/summary.txt
  sleep 1
}
run lds ""
run mix "lz_mix=1"
run mem "lz_group=0"
run lds2 ""
run mix2 "lz_mix=1"
cmp /tmp/o_lds.SiteDepth.gz /tmp/o_mix.SiteDepth.gz && echo "SiteDepth identical (lds vs mix)" >> $O/summary.txt
cmp /tmp/o_lds.SiteDepth.gz /tmp/o_mem.SiteDepth.gz && echo "SiteDepth identical (lds vs mem)" >> $O/summary.txt
rm -f /tmp/s.bam* /tmp/o_*
cat $O/summary.txt
