#!/bin/bash
# Build libsg2im_hip.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [--clean] [extra hipcc flags]
#   --clean : drop every object and the library first (what __graft_entry__.build() runs: a from-scratch build)
# Without --clean the build is incremental, keyed on CONTENT: an object is rebuilt when the sha256 of (its source, the headers
# it includes, the compiler flags) differs from the stamp written next to it -- not on mtimes, which do not survive a copy.
set -e
cd "$(dirname "$0")"
UNITS="runtime igemm igemm_kn0 igemm_kn1 igemm_nk skinny smallm norm graph layout loss"
if [ "$1" == "--clean" ]; then
  shift
  rm -f *.o *.stamp libsg2im_hip.so
fi
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $*"
HDRS="common.h igemm_core.h ../../include/sg2im_hip.h"
OBJS=()
built=0
for f in $UNITS; do
  want=$( (echo "$FLAGS"; cat $f.hip $HDRS) | sha256sum | cut -d' ' -f1)
  if [ ! -f $f.o ] || [ ! -f $f.stamp ] || [ "$(cat $f.stamp)" != "$want" ]; then
    rm -f $f.stamp                                  # a failed compile must not leave a stale object looking current
    ( /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o $f.o && echo "$want" > $f.stamp ) &
    built=$((built+1))
  fi
  OBJS+=($f.o)
done
wait
for f in $UNITS; do [ -f $f.o ] && [ -f $f.stamp ] || { echo "build of $f failed" >&2; exit 1; }; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "${OBJS[@]}" -o libsg2im_hip.so
echo "built $(pwd)/libsg2im_hip.so ($built of $(echo $UNITS | wc -w) units compiled)"
