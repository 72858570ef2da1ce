#!/bin/bash
# A/B builds of the library (select one at run time with MVF_HIP_LIB=<path>):
#   tools/build_variant.sh <name> "<extra hipcc flags>" file1.hip [file2.hip ..]  ->  tools/probes/libmvf_<name>.so
#       the named sources compiled with the extra flags, every other source's product object linked as is
#   tools/build_variant.sh <tag> <source.hip> <flags...>                          ->  video_rep_learning_amd/csrc/libmvf_hip_<tag>.so
#       (the round-3 form, kept for tools/corun_probe.sh: ONE source, flags as separate words)
# Both outputs are git-ignored and travel to the GPU box with the tree.
set -e
R=$(cd $(dirname $0)/.. && pwd)
C=$R/video_rep_learning_amd/csrc
python3 -c "import sys; sys.path.insert(0, '$R'); from video_rep_learning_amd.csrc import build; build.build()"
HIPCC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -mllvm -amdgpu-atomic-optimizer-strategy=None -I$C -I$R/include"
name=$1
if [[ "$2" == *.hip ]]; then
  srcs="$2"; shift 2; flags="$*"; out=$C/libmvf_hip_$name.so
else
  flags=$2; shift 2; srcs="$*"; out=$R/tools/probes/libmvf_$name.so
fi
tmp=/tmp/mvf_variant_$name
rm -rf $tmp; mkdir -p $tmp $R/tools/probes
objs=""
for o in $C/build/*.o; do
  b=$(basename $o .o); skip=0
  for f in $srcs; do [ "$b.hip" = "$f" ] && skip=1; done
  [ $skip = 0 ] && objs="$objs $o"
done
for f in $srcs; do
  $HIPCC $flags -c $C/$f -o $tmp/${f%.hip}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out $objs $tmp/*.o
ls -la $out
