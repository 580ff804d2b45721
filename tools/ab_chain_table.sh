#!/usr/bin/env bash
# A/B of the general chain builds' table placement (scalar loads vs LDS); the variant is built BESIDE the production library.
cd "$GRAFT_REPO_ROOT"
lds=$(tools/build_variant.sh tablelds -DIKGPU_CHAIN_TABLE_IN_LDS | tail -1)
for v in "$PWD/ik_amd/libikgpu.so" "$lds" "$PWD/ik_amd/libikgpu.so" "$lds"; do echo "== $v"; IKGPU_LIB="$v" IKGPU_CHAIN_HOT=0 python tools/chain_variants_timing.py 2>/dev/null; done
