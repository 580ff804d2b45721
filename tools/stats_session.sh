#!/usr/bin/env bash
# rocprofv3 --kernel-trace --stats of the bench's timed launches for one workload, plus the plain bench line:
#   tools/stats_session.sh <workload> [bench args for the plain run]
# writes gpurun_out/stats_<workload>/ (kernel_stats.csv) and gpurun_out/bench_<workload>.json
set -u
W="$1"; shift
OUT="$GRAFT_REPO_ROOT/gpurun_out/stats_$W"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$GRAFT_REPO_ROOT/bench.py" --timed-only --workload "$W" > "$OUT/run.log" 2>&1 || echo "rocprofv3 pass failed"
cd "$GRAFT_REPO_ROOT" && python3 bench.py --workload "$W" "$@" 2>/dev/null | grep '^{' > "gpurun_out/bench_$W.json"
find "$OUT" -name "*kernel_stats.csv" | head -2
cut -c1-300 "gpurun_out/bench_$W.json"
