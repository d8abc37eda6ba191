mkdir -p gpurun_out/fin; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/fin
(timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o bench -- python bench.py --steps 10 --warmup 2 --e2e-records 0 > $O/kt.log 2>&1)
F=$(find $O/kt -name "*kernel_stats.csv" | head -1); if [ -n "$F" ]; then head -1 "$F" > $O/bench_kernel_stats.csv; grep "pdk::" "$F" >> $O/bench_kernel_stats.csv; fi; rm -rf $O/kt
head -3 $O/bench_kernel_stats.csv | cut -c1-60,180-260
(timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err); tail -1 $O/bench.err; wc -l $O/bench.json
python3 -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], json.dumps(d['roofline'])); print(json.dumps(d['e2e'])[-520:])"
