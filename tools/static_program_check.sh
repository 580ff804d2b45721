#!/usr/bin/env bash
# Offline check of the run-time specialised generic programs (no GPU): for a set of problems, generate the source rtc.cpp would hand
# to hipRTC (IKGPU_RTC_DUMP), compile it with hipcc and the same flags, and report registers, scratch and the divergent regions left
# inside the iteration loop (tools/spill_exec_check.py: there must be none besides the store guard after the loop).
#   tools/static_program_check.sh [outdir]
set -e
cd "$(dirname "$0")/.."
OUT="${1:-/tmp/static_check}"; rm -rf "$OUT"; mkdir -p "$OUT/src" "$OUT/cache"
IKGPU_TREE_STATIC_ROWS=12 IKGPU_RTC_DUMP="$OUT/src" IKGPU_CACHE_DIR="$OUT/cache" python3 - <<'PY'
import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "tests"), os.path.join(os.getcwd(), "oracle")]
import ik_amd
from test_gpu_generic import CASES
from test_gpu_static import ROUTED
def make(name, ff, specs, edit=None, cons=None):
    xml = open("fixtures/models/%s.kin.urdf" % name, "rb").read()
    if edit: xml = edit(xml)
    model = ik_amd.Model.from_urdf_xml(xml, free_flyer=ff)
    problem = ik_amd.InverseKinematicsProblem(model, max(s[4] for s in specs))
    for i, (kind, f, r, t, p, w) in enumerate(specs):
        if kind == "com":
            task = problem.add_centre_of_mass_task(ik_amd.CentreOfMassTask.create(model, r), p)
            if w is not None: task.weighting()[:] = w
            continue
        if kind == "posture":
            task = problem.add_posture_task("t%d" % i, ik_amd.PostureTask.create(model, f), p); task.weighting()[:], task.mask[:] = w; continue
        task = problem.add_align_axis_task("t%d" % i, ik_amd.AlignAxisTask.create(model, f, ik_amd.AlignAxisType(t), r), p) if kind == "align" \
            else problem.add_frame_task("t%d" % i, ik_amd.FrameTask.create(model, f, ik_amd.KinematicType(t), r), p)
        if w is not None: task.weighting()[:] = w
    if cons: problem.add_frame_constraint("c", ik_amd.FrameConstraint.create(model, cons[0], ik_amd.KinematicType(cons[1])))
    return problem
for case, (name, ff, specs, edit) in sorted(CASES.items()):
    try: print(case, ik_amd.precompile(make(name, ff, specs, edit)))
    except Exception as e: print(case, "->", str(e)[:120])
for case, (name, ff, specs, cons) in sorted(ROUTED.items()):
    print(case, ik_amd.precompile(make(name, ff, specs, None, cons)))
PY
for f in "$OUT"/src/generic_static_*.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-fast-math -ffp-contract=on -fno-signed-zeros -fno-honor-nans -fno-honor-infinities \
    -mllvm -two-entry-phi-node-folding-threshold=100000 -mllvm -pragma-unroll-threshold=4000000 -Iik_amd/csrc/device -S --cuda-device-only "$f" -o "${f%.hip}.s" 2>/dev/null
  echo "$(basename "$f"): $(grep -E 'TotalNumVgprs' "${f%.hip}.s" | tr -d ';') $(grep -E 'ScratchSize' "${f%.hip}.s" | tr -d ';') saveexec in loop: $(awk '/Loop Header/ {l=1} /s_and_saveexec/ && l {n++} END {print n+0}' "${f%.hip}.s")"
done
python3 tools/spill_exec_check.py "$OUT"/src/*.s | grep -v "divergent regions:   [01]," || true
