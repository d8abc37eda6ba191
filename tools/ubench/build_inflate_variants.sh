#!/bin/bash
# tools/ubench/build_inflate_variants.sh — tuning builds of the library that differ in the wave inflate kernel only (A/B on one box):
#   old   pd_inflate_wave.h of a given git revision (default HEAD~0 of the round's start: pass the revision as $1)
#   v1    the working tree
#   v2    the working tree with a 9-bit literal/length root (2 KiB less LDS per wave) and registers capped for 6 waves per SIMD
#   v3    9-bit root, registers capped for 8 waves per SIMD
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
cp $T/pd_bgzf_v1.hip $T/pd_bgzf_v2.hip; cp $T/pd_bgzf_v1.hip $T/pd_bgzf_v3.hip; cp $T/pd_bgzf_v1.hip $T/pd_bgzf_old.hip
build v1
build v2 -DPD_LL_ROOT=9 -DPD_INFLATE_MIN_WAVES=6
build v3 -DPD_LL_ROOT=9 -DPD_INFLATE_MIN_WAVES=8
git show $REV:pandepth_amd/csrc/pd_inflate_wave.h | sed "s#\"../../include/pandepth_amd.h\"#\"$PWD/../include/pandepth_amd.h\"#" > $T/pd_inflate_wave.h
git show $REV:pandepth_amd/csrc/pd_bamwalk.h | sed "s#\"../../include/pandepth_amd.h\"#\"$PWD/../include/pandepth_amd.h\"#" > $T/pd_bamwalk_old.h
# (the old header with today's pd_bamwalk.h: only the inflate kernel is compared)
build old
rm -rf $T
ls -la $OUT/libpd_inflate_*.so
