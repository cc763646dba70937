#!/bin/bash
# Builds a VARIANT of libvdx.so beside the product library, for A/B timing of one kernel change on the GPU box:
#   bash tools/mkvariant.sh <name> <file.hip> "<extra compiler flags, e.g. -DVDX_H8_SB=4>"
# -> video_diffusion_nnx_amd/variants/libvdx_<name>.so (git-ignored; travels with gpurun).  Only <file.hip> is recompiled: the other
# objects are copied from the product build.  Run a tool against it with VDX_LIB=<path> (video_diffusion_nnx_amd/_lib.py).
set -e
name=$1; src=$2; flags=$3
root=$(cd "$(dirname "$0")/.." && pwd)
csrc=$root/video_diffusion_nnx_amd/csrc
make -C "$csrc" -j8 > /dev/null
rm -rf "$csrc/build_$name"; cp -rp "$csrc/build" "$csrc/build_$name"
rm -f "$csrc/build_$name/${src%.hip}.o"
mkdir -p "$root/video_diffusion_nnx_amd/variants"
make -C "$csrc" BUILD="build_$name" TARGET="../variants/libvdx_$name.so" EXTRA="$flags" > /dev/null
rm -rf "$csrc/build_$name"
ls -la "$root/video_diffusion_nnx_amd/variants/libvdx_$name.so"
