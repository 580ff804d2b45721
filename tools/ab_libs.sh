#!/usr/bin/env bash
# A/B of two builds of the library on ONE box: tools/ab_libs.sh <variant> <workload> [<workload> ...]
# (the variant is ik_amd/libikgpu_<variant>.so, built with tools/build_variant.sh or from another checkout)
cd $GRAFT_REPO_ROOT
v="$1"; shift
for w in "$@"; do
  for rep in 1 2; do
    for lib in "" "$v"; do
      if [ -n "$lib" ]; then export IKGPU_LIB=$GRAFT_REPO_ROOT/ik_amd/libikgpu_$lib.so; else unset IKGPU_LIB; fi
      timeout -k 10 300 python bench.py --workload $w --no-cpu --timed-only 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-28s %-10s %-70s %.4f ms' % ('$w', '${lib:-current}', d['config'].get('kernel'), d['ms_per_step']))" || exit 1
    done
  done
done
