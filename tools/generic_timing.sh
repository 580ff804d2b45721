#!/usr/bin/env bash
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
  a=$(IKGPU_DLS_KERNEL=generic python bench.py --workload cassie_demo --timed-only --steps 5 2>/dev/null | grep -o '"kernel_ms": [0-9.]*' | cut -d' ' -f2)
  b=$(python bench.py --workload cassie_demo_pik --timed-only --steps 5 2>/dev/null | grep -o '"kernel_ms": [0-9.]*' | cut -d' ' -f2)
  echo "rep $rep generic-dls $a pik $b"
done
