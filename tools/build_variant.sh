# A/B builds: tools/build_variant.sh <name> "<extra hipcc flags>" file1.hip [file2.hip ..]
# compiles the named sources with the extra flags, links them with the product objects of every other source into
# tools/probes/libmvf_<name>.so (git-ignored; travels to the GPU box).  Use with MVF_HIP_LIB=tools/probes/libmvf_<name>.so
set -e
R=$(cd $(dirname $0)/.. && pwd)
C=$R/video_rep_learning_amd/csrc
name=$1; flags=$2; shift 2
python3 -c "import sys; sys.path.insert(0, '$R'); from video_rep_learning_amd.csrc import build; build.build()"
mkdir -p /tmp/mvf_variant_$name $R/tools/probes
objs=""
for o in $C/build/*.o; do
  b=$(basename $o .o); skip=0
  for f in "$@"; do [ "$b.hip" = "$f" ] && skip=1; done
  [ $skip = 0 ] && objs="$objs $o"
done
for f in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -mllvm -amdgpu-atomic-optimizer-strategy=None -I$C -I$R/include $flags -c $C/$f -o /tmp/mvf_variant_$name/${f%.hip}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/probes/libmvf_$name.so $objs /tmp/mvf_variant_$name/*.o
ls -la $R/tools/probes/libmvf_$name.so
