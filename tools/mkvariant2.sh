#!/bin/bash
# As tools/mkvariant.sh, for a variant that touches several sources:
#   bash tools/mkvariant2.sh <name> "<extra compiler flags>" <file.hip> [<file.hip> ...]
set -e
name=$1; flags=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
csrc=$root/video_diffusion_nnx_amd/csrc
make -C "$csrc" -j8 > /dev/null
rm -rf "$csrc/build_$name"; cp -rp "$csrc/build" "$csrc/build_$name"
for src in "$@"; do rm -f "$csrc/build_$name/${src%.hip}.o"; done
mkdir -p "$root/video_diffusion_nnx_amd/variants"
make -C "$csrc" -j4 BUILD="build_$name" TARGET="../variants/libvdx_$name.so" EXTRA="$flags" > /dev/null
rm -rf "$csrc/build_$name"
ls -la "$root/video_diffusion_nnx_amd/variants/libvdx_$name.so"
