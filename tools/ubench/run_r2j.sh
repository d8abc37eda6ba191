mkdir -p gpurun_out/r2j; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(PANDEPTH_TIMING=1 timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -30) > gpurun_out/r2j/fullsize.log
(timeout 900 python bench.py > gpurun_out/r2j/bench.json 2> gpurun_out/r2j/bench.err; echo "rc=$?" >> gpurun_out/r2j/bench.err)
(PD_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 6 --warmup 2 --e2e-records 0 > gpurun_out/r2j/bench_dist1.json 2> gpurun_out/r2j/bench_dist1.err; echo "rc=$?" >> gpurun_out/r2j/bench_dist1.err)
tail -12 gpurun_out/r2j/fullsize.log | cut -c1-900; tail -3 gpurun_out/r2j/bench.err; tail -3 gpurun_out/r2j/bench_dist1.err
python3 -c "
import json; d=json.load(open('gpurun_out/r2j/bench.json')); print(d['value'], d['ms_per_step'], json.dumps(d['roofline']), json.dumps(d['e2e']), json.dumps(d['cpu_baseline']))" | cut -c1-3000
