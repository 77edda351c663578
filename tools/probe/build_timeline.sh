#!/bin/bash
# Debugging variant of the library with -DSG_TIMELINE (igemm_core.h): per-workgroup s_memtime stamps.  Written next to the product
# library as libsg2im_hip_tl.so (objects under csrc/tl/); load it with SG_LIB_PATH=.../libsg2im_hip_tl.so.
set -e
cd "$(dirname "$0")/../../scene_generation_amd/csrc"
mkdir -p tl
UNITS="runtime igemm igemm_kn0 igemm_kn1 igemm_nk skinny smallm norm graph layout loss"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DSG_TIMELINE"
for f in $UNITS; do ( /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o tl/$f.o ) & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(for f in $UNITS; do echo tl/$f.o; done) -o libsg2im_hip_tl.so
echo "built $(pwd)/libsg2im_hip_tl.so"
