#!/usr/bin/env bash
# A/B of full-body tree kernel variants on one box: tools/fullbody_ab.sh  (variants built with tools/build_variant.sh)
cd $GRAFT_REPO_ROOT
run() { label="$1"; shift; for i in 1 2; do env "$@" timeout -k 10 200 python bench.py --workload cassie_full_body --no-cpu --timed-only 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['config'].get('kernel'), '%.4f' % d['ms_per_step'])" || exit 1; done; }
run "never-stop build            " A=1
run "stop-capable build (r02 hot)" IKGPU_TREE_NEVER_OFF=1
for v in "$@"; do run "variant $v" IKGPU_LIB=$GRAFT_REPO_ROOT/ik_amd/libikgpu_$v.so; done
