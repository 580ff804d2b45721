#!/usr/bin/env bash
# A/B of compile options for the kernels on ONE box (kernel time, B = 65536, 50 iterations); usage: tools/ab_flags.sh
cd "$GRAFT_REPO_ROOT"
cp ik_amd/libikgpu.so /tmp/lib_0.so
i=0
declare -a FL=("" "-mllvm -amdgpu-sched-strategy=max-ilp" "-mllvm -amdgpu-schedule-metric-bias=100" "-DIKD_NEAR_PI_BRANCHLESS" "-ffp-contract=fast" "-mllvm -amdgpu-use-amdgpu-trackers" "-O2" "-mllvm -amdgpu-sched-strategy=max-memory-clause")
for f in "${FL[@]}"; do
  touch ik_amd/csrc/kernels.hip
  if make -s -C ik_amd/csrc KERNEL_EXTRA="$f" >/dev/null 2>&1; then cp ik_amd/libikgpu.so /tmp/lib_$i.so; else echo "variant $i [$f] failed to build"; rm -f /tmp/lib_$i.so; fi
  i=$((i+1))
done
for rep in 1 2; do
  for j in $(seq 0 $((i-1))); do
    [ -f /tmp/lib_$j.so ] || continue
    cp /tmp/lib_$j.so ik_amd/libikgpu.so
    a=$(python bench.py --timed-only --steps 40 2>/dev/null | grep -o '"kernel_ms": [0-9.]*' | cut -d' ' -f2)
    b=$(python bench.py --timed-only --steps 40 --workload ur5 2>/dev/null | grep -o '"kernel_ms": [0-9.]*' | cut -d' ' -f2)
    c=$(python bench.py --timed-only --steps 20 --workload cassie_full_body 2>/dev/null | grep -o '"kernel_ms": [0-9.]*' | cut -d' ' -f2)
    echo "rep $rep variant $j [${FL[$j]}] leg $a ur5 $b full $c"
  done
done
cp /tmp/lib_0.so ik_amd/libikgpu.so
