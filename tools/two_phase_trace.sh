#!/usr/bin/env bash
# Per-kernel durations of the two launches of a two-phase solve (tools/two_phase_probe.py trace) for a few switch points:
#   tools/two_phase_trace.sh [tree]      -> gpurun_out/tp_trace_<K>_<N>/…kernel_stats.csv, summary on stdout
set -u
cd /tmp && export TMPDIR=/tmp
for kn in "4 16" "8 16" "8 8" "4 48"; do
  set -- $kn ${TREE:-}
  export IKGPU_TWO_PHASE_ITERS=$1 IKGPU_TWO_PHASE_ACTIVE=$2 IKGPU_REFILL=2
  OUT="$GRAFT_REPO_ROOT/gpurun_out/tp_trace_${1}_${2}${TREE:+_tree}"; rm -rf "$OUT"; mkdir -p "$OUT"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$GRAFT_REPO_ROOT/tools/two_phase_probe.py" trace uniform 262144 ${TREE:+tree} > "$OUT/run.log" 2>&1 || echo "rocprofv3 pass failed"
  echo "== K=$1 N=$2 $(grep -v amdgpu.ids "$OUT/run.log" | tail -1)"
  find "$OUT" -name "*kernel_stats.csv" | xargs -n1 head -4 | cut -c1-200
done
