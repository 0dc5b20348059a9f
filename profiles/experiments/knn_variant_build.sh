#!/bin/bash
# scratch/libvcr_<tag>.so = the product library with ONE constexpr int of knn.hip changed (timing ablations / sweeps):
#   profiles/experiments/knn_variant_build.sh TAG NAME=VALUE [NAME=VALUE ...]
# Only knn.hip is recompiled; the other objects come from vcr-net_amd/build (python vcr-net_amd/build.py first).
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
tag=$1; shift
work=$ROOT/scratch/knnvar_$tag/a/b
rm -rf "$ROOT/scratch/knnvar_$tag"; mkdir -p "$work/csrc" "$work/../include"
cp "$ROOT"/vcr-net_amd/csrc/*.h "$work/csrc/"; cp "$ROOT/include/vcr_hip.h" "$work/../include/"
cp "$ROOT/vcr-net_amd/csrc/knn.hip" "$work/csrc/knn.hip"
for kv in "$@"; do
  name=${kv%%=*}; val=${kv#*=}
  grep -q "constexpr int $name = " "$work/csrc/knn.hip" || { echo "no constexpr int $name"; exit 1; }
  sed -i -E "s/(constexpr int $name = )[^;]+;/\1$val;/" "$work/csrc/knn.hip"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -c "$work/csrc/knn.hip" -o "$work/knn.o"
objs=$(ls "$ROOT"/vcr-net_amd/build/*.o | grep -v '/knn.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/scratch/libvcr_$tag.so" $objs "$work/knn.o"
echo "built scratch/libvcr_$tag.so ($*)"
