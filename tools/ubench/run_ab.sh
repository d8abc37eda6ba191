mkdir -p gpurun_out/ab; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export VARIANTS=0,4408,4316,4412,4416,0
(timeout 500 python tools/ubench/direct_ab.py > gpurun_out/ab/ab.log 2>&1); tail -6 gpurun_out/ab/ab.log | cut -c1-400
