#!/usr/bin/env bash
# PMC passes over tools/generic_forms.py for one case (separate rocprofv3 runs, --pmc only with --kernel-trace):
#   tools/pmc_generic.sh <case> <outdir-under-gpurun_out>
set -u
C="$1"; OUT="$GRAFT_REPO_ROOT/gpurun_out/$2"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { name="$1"; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$GRAFT_REPO_ROOT/tools/generic_forms.py" "$C" > "$OUT/$name.log" 2>&1 || echo "pass $name failed"; }
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES
run mix SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU
run mem FETCH_SIZE WRITE_SIZE
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "generic" not in k and "coop" not in k: continue
        acc[(k[:60], r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "")), r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k)
    for c, v in sorted(d.items()):
        print("   %-24s mean %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
