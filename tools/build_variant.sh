#!/bin/bash
# Build a kernel-variant copy of the library for A/B runs on the GPU box (select it with SG_LIB_PATH=<path>):
#   tools/build_variant.sh NAME [extra hipcc flags, e.g. -DSG_PIPE_DEFAULT=2]
# Objects go to /tmp/sgvar_NAME, the library to scene_generation_amd/csrc/variants/libsg2im_hip_NAME.so (git-ignored).
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/scene_generation_amd/csrc
obj=/tmp/sgvar_$name
mkdir -p $obj $src/variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -I$src $*"
for f in runtime igemm igemm_kn0 igemm_kn1 igemm_nk smallm norm graph gconv layout loss; do
  ( cd $src && /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o $obj/$f.o 2> $obj/$f.log || echo "FAILED $f" ) &
done
wait
if grep -l "error:" $obj/*.log; then exit 1; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $obj/*.o -o $src/variants/libsg2im_hip_$name.so
echo "built $src/variants/libsg2im_hip_$name.so"
