#!/usr/bin/env bash
cd "$GRAFT_REPO_ROOT"
cp ik_amd/libikgpu.so /tmp/lib_scalar.so
touch ik_amd/csrc/kernels.hip; make -s -C ik_amd/csrc KERNEL_EXTRA="-DIKGPU_CHAIN_TABLE_IN_LDS" >/dev/null 2>&1; cp ik_amd/libikgpu.so /tmp/lib_lds.so
for v in scalar lds scalar lds; do cp /tmp/lib_$v.so ik_amd/libikgpu.so; echo "== table $v"; python tools/chain_variants_timing.py 2>/dev/null; done
cp /tmp/lib_scalar.so ik_amd/libikgpu.so
