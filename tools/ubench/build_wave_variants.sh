# builds the A/B binaries of tools/ubench/wave_debug.hip in the dev container (hipcc cross-compiles); they travel to the GPU box with the snapshot
# usage: build_wave_variants.sh name "flags" [name "flags" ...]
cd "$(dirname "$0")"
while [ $# -ge 2 ]; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value $2 wave_debug.hip -lz -o wd_$1 > /tmp/wd_$1.log 2>&1 || { echo "build of $1 failed"; grep -m5 error /tmp/wd_$1.log; }
  shift 2
done
ls -la wd_*
