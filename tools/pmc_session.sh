#!/usr/bin/env bash
# PMC passes for ONE bench workload on ONE build: separate rocprofv3 runs (--pmc only with --kernel-trace), each into a FRESH
# directory -- the session directory is removed first, so a pass directory never holds counter files of an earlier build
# (round 2's tools/pmc_to_stats.py averaged whatever had accumulated there).
#   tools/pmc_session.sh <workload> <outdir-under-gpurun_out> [extra bench args]
set -u
W="$1"; OUT="$GRAFT_REPO_ROOT/gpurun_out/$2"; shift 2
rm -rf "$OUT"; mkdir -p "$OUT"
python3 "$GRAFT_REPO_ROOT/tools/source_stamp.py" > "$OUT/source_stamp.json"
cd /tmp && export TMPDIR=/tmp
run() { name="$1"; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --timed-only --workload "$W" $EXTRA > "$OUT/$name.log" 2>&1 || echo "pass $name failed"; }
EXTRA="$*"
run fetch FETCH_SIZE
run write WRITE_SIZE
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES
run grbm GRBM_GUI_ACTIVE
run flops SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM
# the plain bench line of the same build next to the counters (its config.kernel names the ABI-level kernel of this workload)
cd "$GRAFT_REPO_ROOT" && python3 bench.py --workload "$W" --no-cpu $EXTRA 2>/dev/null | grep '^{' > "$OUT/bench.json"
find "$OUT" -name "*counter_collection.csv" | wc -l
