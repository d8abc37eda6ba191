# round 3, call 9: BASELINE configs[1] in full, end to end: 1e9 records (a ~53 GB payload BAM written on the box), product executable vs reference binary
O=$GRAFT_REPO_ROOT/gpurun_out/r3c9; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
df -h /tmp > $O/df.log 2>&1; free -g >> $O/df.log 2>&1; nproc >> $O/df.log
( time PD_BENCH_CONFIG_LEGS=0 timeout 2400 python bench.py --e2e-records 1e9 --e2e-site-records 0 ) > $O/bench_e2e_1e9.json 2> $O/bench_e2e_1e9.err
