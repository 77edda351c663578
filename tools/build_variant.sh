#!/bin/bash
# Build a VARIANT of libsg2im_hip.so with extra compiler flags into scene_generation_amd/csrc/variants/<name>.so (git-ignored;
# travels to the GPU box with gpurun).  Use with SG_LIB_PATH=<that file> python bench.py ...   Usage: build_variant.sh <name> <flags...>
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME="$1"; shift
W=/tmp/sg_variant_$NAME
rm -rf "$W"; mkdir -p "$W/scene_generation_amd" "$W/include"
cp -r "$ROOT/scene_generation_amd/csrc" "$W/scene_generation_amd/csrc"
cp "$ROOT/include/sg2im_hip.h" "$W/include/"
rm -rf "$W/scene_generation_amd/csrc/variants"
bash "$W/scene_generation_amd/csrc/build.sh" --clean "$@" | tail -1
mkdir -p "$ROOT/scene_generation_amd/csrc/variants"
cp "$W/scene_generation_amd/csrc/libsg2im_hip.so" "$ROOT/scene_generation_amd/csrc/variants/$NAME.so"
echo "variant: scene_generation_amd/csrc/variants/$NAME.so"
