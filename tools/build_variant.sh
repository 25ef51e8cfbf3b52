#!/bin/bash
# Build a variant of the engine library for A/B timing:
#   tools/build_variant.sh NAME [extra hipcc flags / -D defines ...]
# -> libh264_amd/lib/variants/libengine_NAME.so (git-ignored, travels with gpurun)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p "$ROOT/libh264_amd/lib/variants"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -mllvm -unroll-threshold=300 -std=c++17 -fPIC -shared \
	-Wno-unused-value -Wno-inline-asm -I"$ROOT/include" -I"$ROOT/libh264_amd/csrc" "$@" \
	"$ROOT/libh264_amd/csrc/engine.hip" \
	-o "$ROOT/libh264_amd/lib/variants/libengine_$name.so"
echo "built libengine_$name.so"
