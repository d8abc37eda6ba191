# GPU box: correctness of the wave decoder against zlib (pytest) + kernel rates (tools/bgzf_gpu_bench.py) + a kernel trace
mkdir -p gpurun_out/r2c; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(timeout 120 python -m pytest tests/test_inflate_core.py -x -q -m gpu 2>&1 | tail -5) > gpurun_out/r2c/t_inflate.log
BGZF_LEVEL=6 BGZF_VARIANTS=${V6:-1,130,194,226,258,322} timeout 200 python tools/bgzf_gpu_bench.py ${R:-6e6} > gpurun_out/r2c/bench_l6.log 2>&1
BGZF_LEVEL=1 BGZF_VARIANTS=${V1:-258} timeout 200 python tools/bgzf_gpu_bench.py ${R:-6e6} > gpurun_out/r2c/bench_l1.log 2>&1
cat gpurun_out/r2c/t_inflate.log gpurun_out/r2c/bench_l6.log gpurun_out/r2c/bench_l1.log
