mkdir -p gpurun_out/ab; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export VARIANTS=504,0,3602,504
(timeout 500 python tools/ubench/direct_ab.py > gpurun_out/ab/ab.log 2>&1); tail -6 gpurun_out/ab/ab.log | cut -c1-400
(timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -5) > gpurun_out/ab/pytest.log; tail -4 gpurun_out/ab/pytest.log
