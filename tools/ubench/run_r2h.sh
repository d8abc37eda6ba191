# GPU box: communicator tests, full-size config tests, late CLI tests, the 1-rank collective bench path, then bench.py with the e2e leg
mkdir -p gpurun_out/r2h; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(timeout 500 python -m pytest tests/test_comm_gpu.py -x -q -m gpu 2>&1 | tail -25) > gpurun_out/r2h/comm.log
(PANDEPTH_TIMING=1 timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -25) > gpurun_out/r2h/fullsize.log
(timeout 600 python -m pytest tests/test_z_cli_gpu_late.py -x -q -m gpu 2>&1 | tail -15) > gpurun_out/r2h/late.log
(PD_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 6 --warmup 2 --e2e-records 0 > gpurun_out/r2h/bench_dist1.json 2> gpurun_out/r2h/bench_dist1.err; echo "rc=$?" >> gpurun_out/r2h/bench_dist1.err)
(timeout 900 python bench.py > gpurun_out/r2h/bench.json 2> gpurun_out/r2h/bench.err; echo "rc=$?" >> gpurun_out/r2h/bench.err)
tail -6 gpurun_out/r2h/comm.log; tail -8 gpurun_out/r2h/fullsize.log; tail -4 gpurun_out/r2h/late.log; tail -3 gpurun_out/r2h/bench_dist1.err; cut -c1-600 gpurun_out/r2h/bench_dist1.json; tail -3 gpurun_out/r2h/bench.err
python3 -c "
import json; d=json.load(open('gpurun_out/r2h/bench.json')); print(d['value'], d['ms_per_step'], json.dumps(d['roofline']), json.dumps(d['e2e']), json.dumps(d['cpu_baseline']))"
