# builds nothing on the box: tools/ubench/libpandepth_ticks.so is made in the dev container with
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DPD_WIDE3_TICKS -c -x hip pandepth_amd/csrc/pd_kernels.hip -o /tmp/w3/pd_kernels.o
#   hipcc --offload-arch=gfx950 -shared -fPIC /tmp/w3/pd_kernels.o pandepth_amd/csrc/{pd_capi,pd_bgzf,pd_format}.o -ldl -o tools/ubench/libpandepth_ticks.so
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ticks
timeout 300 python tools/ubench/wide3_ticks.py 2>&1 | tee gpurun_out/ticks/ticks.log | tail -22
