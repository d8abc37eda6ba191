# GPU box: the round's evidence — bench lines (all configs), rocprofv3 kernel stats of the bench and of the executable, PMC traffic
mkdir -p gpurun_out/r2p; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2p
(timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err)
(timeout 300 python bench.py --config gff > $O/bench_gff.json 2> $O/bench_gff.err; echo "rc=$?" >> $O/bench_gff.err)
(timeout 300 python bench.py --config w100a > $O/bench_w100a.json 2> $O/bench_w100a.err; echo "rc=$?" >> $O/bench_w100a.err)
(PD_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 6 --warmup 2 --e2e-records 0 > $O/bench_dist1.json 2> $O/bench_dist1.err; echo "rc=$?" >> $O/bench_dist1.err)
(timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o bench -- python bench.py --steps 10 --warmup 2 --e2e-records 0 > $O/kt.log 2>&1)
F=$(find $O/kt -name "*kernel_stats.csv" | head -1); if [ -n "$F" ]; then head -1 "$F" > $O/bench_kernel_stats.csv; grep "pdk::\|pdw\|k_inflate\|k_walk\|k_emit" "$F" >> $O/bench_kernel_stats.csv; fi
rm -rf $O/kt
timeout 600 bash tools/pmc_collect.sh > $O/pmc.log 2>&1
mkdir -p /tmp/e2e && timeout 200 tools/bamgen -o /tmp/e2e/s.bam -n 40000000 -t 32 > $O/bamgen.log 2>&1
(cd /tmp/e2e && PANDEPTH_ORDERLY_EXIT=1 timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/ktc -o cli -- $GRAFT_REPO_ROOT/pandepth_amd/pandepth -i s.bam -o mine -t 16 > $GRAFT_REPO_ROOT/$O/ktc.log 2>&1)
F=$(find $O/ktc -name "*kernel_stats.csv" | head -1); if [ -n "$F" ]; then cp "$F" $O/cli_kernel_stats.csv; fi
rm -rf $O/ktc
for f in bench bench_gff bench_w100a bench_dist1; do tail -2 $O/$f.err | cut -c1-200; done
python3 - <<'PY'
import json
for f in ("bench", "bench_gff", "bench_w100a", "bench_dist1"):
    try:
        d = json.load(open("gpurun_out/r2p/%s.json" % f))
        print(f, "%.4g" % d["value"], round(d["ms_per_step"], 3), json.dumps(d["roofline"])[:330])
        if d.get("e2e"): print("   e2e", json.dumps(d["e2e"])[-420:])
    except Exception as e: print(f, "unreadable", e)
PY
cat gpurun_out/r2p/bench_kernel_stats.csv | cut -c1-200 | head -12; cat gpurun_out/r2p/cli_kernel_stats.csv 2>/dev/null | cut -c1-200 | head -12; ls gpurun_out/pmc
