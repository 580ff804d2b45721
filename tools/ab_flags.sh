#!/usr/bin/env bash
# A/B of compile options for the kernels on ONE box (kernel time, B = 65536, 50 iterations); usage: tools/ab_flags.sh
# Every variant is built BESIDE the production library (tools/build_variant.sh) and selected with IKGPU_LIB.
cd "$GRAFT_REPO_ROOT"
declare -a FL=("" "-mllvm -amdgpu-sched-strategy=max-ilp" "-mllvm -amdgpu-schedule-metric-bias=100" "-DIKD_NEAR_PI_BRANCHLESS" "-ffp-contract=fast" "-mllvm -amdgpu-use-amdgpu-trackers" "-O2" "-mllvm -amdgpu-sched-strategy=max-memory-clause")
declare -a LIBS=()
i=0
for f in "${FL[@]}"; do
  if [ -z "$f" ]; then LIBS+=("$PWD/ik_amd/libikgpu.so"); else LIBS+=("$(tools/build_variant.sh ab$i $f | tail -1)"); fi
  i=$((i+1))
done
for rep in 1 2; do
  for j in $(seq 0 $((i-1))); do
    [ -f "${LIBS[$j]}" ] || { echo "variant $j [${FL[$j]}] failed to build"; continue; }
    a=$(IKGPU_LIB="${LIBS[$j]}" python bench.py --timed-only --steps 40 2>/dev/null | grep -o '"kernel_ms": [0-9.]*' | cut -d' ' -f2)
    b=$(IKGPU_LIB="${LIBS[$j]}" python bench.py --timed-only --steps 40 --workload ur5 2>/dev/null | grep -o '"kernel_ms": [0-9.]*' | cut -d' ' -f2)
    c=$(IKGPU_LIB="${LIBS[$j]}" python bench.py --timed-only --steps 20 --workload cassie_full_body 2>/dev/null | grep -o '"kernel_ms": [0-9.]*' | cut -d' ' -f2)
    echo "rep $rep variant $j [${FL[$j]}] leg $a ur5 $b full $c"
  done
done
