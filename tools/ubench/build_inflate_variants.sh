#!/bin/bash
# tools/ubench/build_inflate_variants.sh — tuning builds of the library that differ in the wave inflate kernel only (A/B on one box):
#   old   pd_inflate_wave.h of a given git revision (default HEAD~0 of the round's start: pass the revision as $1)
#   v1    the working tree's kernel with the 10-bit literal/length root and no register cap (round 3's geometry)
#   v2    9-bit literal/length root (2 KiB less LDS per wave), registers capped for 6 waves per SIMD (the default build)
#   v3    v2 + a 7-bit distance root (another 512 B), v4 the same with registers for 5 waves per SIMD
set -e
cd "$(dirname "$0")/../../pandepth_amd"
REV=${1:-0f4c6db}
OUT=../tools/ubench
HIPCC=/opt/rocm/bin/hipcc; FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result"
OBJS="csrc/pd_kernels.o csrc/pd_capi.o csrc/pd_format.o csrc/pd_deflate.o"
T=$(mktemp -d); cp csrc/*.h csrc/pd_bgzf.hip $T/; mkdir -p $T/../../include; 
build() { name=$1; shift; $HIPCC $FL "$@" -I$PWD/csrc -c $T/pd_bgzf_$name.hip -o $T/pd_bgzf_$name.o && $HIPCC --offload-arch=gfx950 -shared -fPIC $OBJS $T/pd_bgzf_$name.o -ldl -o $OUT/libpd_inflate_$name.so; }
sed "s#\"../../include/pandepth_amd.h\"#\"$PWD/../include/pandepth_amd.h\"#" csrc/pd_bgzf.hip > $T/pd_bgzf_v1.hip
sed -i "s#\"../../include/pandepth_amd.h\"#\"$PWD/../include/pandepth_amd.h\"#" $T/*.h
for v in v2 v3 v4 old; do cp $T/pd_bgzf_v1.hip $T/pd_bgzf_$v.hip; done
build v1 -DPD_LL_ROOT=10 -DPD_INFLATE_MIN_WAVES=1
build v2 -DPD_LL_ROOT=9 -DPD_INFLATE_MIN_WAVES=6
build v3 -DPD_LL_ROOT=9 -DPD_D_ROOT=7 -DPD_INFLATE_MIN_WAVES=6
build v4 -DPD_LL_ROOT=9 -DPD_D_ROOT=7 -DPD_INFLATE_MIN_WAVES=5
git show $REV:pandepth_amd/csrc/pd_inflate_wave.h | sed "s#\"../../include/pandepth_amd.h\"#\"$PWD/../include/pandepth_amd.h\"#" > $T/pd_inflate_wave.h
git show $REV:pandepth_amd/csrc/pd_bamwalk.h | sed "s#\"../../include/pandepth_amd.h\"#\"$PWD/../include/pandepth_amd.h\"#" > $T/pd_bamwalk_old.h
# (the old header with today's pd_bamwalk.h: only the inflate kernel is compared)
build old
rm -rf $T
ls -la $OUT/libpd_inflate_*.so
