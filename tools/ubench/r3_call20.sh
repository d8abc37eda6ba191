# round 3, call 20: the driver's command (python bench.py) as it stands, with its kernel trace
O=$GRAFT_REPO_ROOT/gpurun_out/r3c20; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( time timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --e2e-records 0 > $O/kt_bench.json 2> $O/kt_bench.err
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && (head -1 $f; grep -E "pdk::|anonymous" $f) > $O/bench_kernel_stats.csv; rm -rf $O/kt
