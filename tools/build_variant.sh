#!/usr/bin/env bash
# Builds a variant of libikgpu.so BESIDE the production library (never over it) and prints its path:
#   tools/build_variant.sh <name> <extra hipcc flags for the kernel translation units>
# Select it at run time with IKGPU_LIB=<path> (read by ik_amd/capi.py).
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
name="$1"; shift
make -s -C "$ROOT/ik_amd/csrc" ARCH=gfx950 OUT="$ROOT/ik_amd/libikgpu_$name.so" OBJ="$ROOT/ik_amd/csrc/_obj_$name" KERNEL_EXTRA="$*" 2>&1 | grep -v "warning\|^\s*[0-9]* |\|\^\|~" || true
echo "$ROOT/ik_amd/libikgpu_$name.so"
