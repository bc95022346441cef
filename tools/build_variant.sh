#!/bin/bash
# Builds an A/B variant of libsmirk_hip.so: the files named in $2.. get the extra flags in $1 (quoted); used through SMIRK_HIP_LIBRARY.
#   bash tools/build_variant.sh "-mllvm -amdgpu-waitcnt-forcezero=1" flame.hip render.hip   ->  smirk_amd/lib_fz/libsmirk_hip_variant.so
set -e
cd "$(dirname "$0")/.."
EXTRA="$1"; shift
OUT=smirk_amd/lib_fz
mkdir -p $OUT
objs=""
for s in smirk_amd/csrc/*.hip; do
  b=$(basename $s .hip)
  flags="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function -fno-slp-vectorize -fno-vectorize"   # = smirk_amd/build.py COMMON
  case $b in render|video) flags="$flags -ffp-contract=off";; esac
  for v in "$@"; do [ "$v" = "$b.hip" ] && flags="$flags $EXTRA"; done
  /opt/rocm/bin/hipcc $flags -c $s -o $OUT/$b.o &
  objs="$objs $OUT/$b.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libsmirk_hip_variant.so $objs
echo $OUT/libsmirk_hip_variant.so
