mkdir -p gpurun_out/r2d; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(timeout 500 python -m pytest tests/test_gpu_bgzf.py -x -q -m gpu 2>&1 | tail -15) > gpurun_out/r2d/t_bgzf.log
(timeout 900 python -m pytest tests/test_host_generated.py -x -q -m gpu 2>&1 | tail -25) > gpurun_out/r2d/t_gen.log
(timeout 900 python -m pytest tests/test_cli_gpu.py -x -q -m gpu -k "device_decode or host_decode" 2>&1 | tail -25) > gpurun_out/r2d/t_cli.log
tail -12 gpurun_out/r2d/*.log
