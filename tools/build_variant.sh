#!/bin/bash
# Builds a second copy of the library with extra flags on ONE source (A/B measurements on one box):
#   tools/build_variant.sh <tag> <source.hip> <flags...>   ->  video_rep_learning_amd/csrc/libmvf_hip_<tag>.so
# Select it at run time with MVF_HIP_LIB=<path>.
set -e
cd "$(dirname "$0")/../video_rep_learning_amd/csrc"
tag=$1; src=$2; shift 2
python3 -m video_rep_learning_amd.csrc.build >/dev/null 2>&1 || (cd ../.. && python3 -m video_rep_learning_amd.csrc.build)
mkdir -p build/$tag
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -mllvm -amdgpu-atomic-optimizer-strategy=None \
  -I. -I../../include "$@" -c $src -o build/$tag/${src%.hip}.o
objs=""
for o in build/*.o; do
  b=$(basename $o)
  if [ "$b" == "${src%.hip}.o" ]; then objs="$objs build/$tag/$b"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libmvf_hip_$tag.so $objs
echo built libmvf_hip_$tag.so
