#!/bin/bash
# Build libsg2im_hip.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")"
OBJS=()
for f in runtime igemm smallm norm graph layout loss; do
  if [ ! -f $f.o ] || [ $f.hip -nt $f.o ] || [ common.h -nt $f.o ] || [ ../../include/sg2im_hip.h -nt $f.o ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -c $f.hip -o $f.o "$@" &
  fi
  OBJS+=($f.o)
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "${OBJS[@]}" -o libsg2im_hip.so
echo "built $(pwd)/libsg2im_hip.so"
