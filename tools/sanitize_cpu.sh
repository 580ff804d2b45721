#!/usr/bin/env bash
# AddressSanitizer + UBSan pass over everything that can run on the CPU: the URDF loader and problem analysis
# (ik_amd/csrc/model.cpp, problem.cpp), the device lane programs (chain, tree, generic per-lane and cooperative, PIK) compiled for the host (tests/lane_emu) and
# the C oracle.  GPU sanitizers are not available on this pool; the device code shares these sources.
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"
g++ -O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-sanitize-recover=undefined -Iinclude -Iik_amd/csrc \
    -o /tmp/liblane_emu_asan.so tests/lane_emu/lane_emu.cpp ik_amd/csrc/model.cpp ik_amd/csrc/problem.cpp
make -s -C oracle asan
cat > /tmp/ik_asan_run.py <<PY
import os, sys, ctypes as C
sys.path[:0] = ["$ROOT/oracle", "$ROOT", "$ROOT/tests"]
import oracle as O
O._LIB = C.CDLL("$ROOT/oracle/libik_oracle_asan.so")
for f in ("iko_dls", "iko_dls_batch", "iko_task_rows", "iko_pik", "iko_pik_batch", "iko_dls_constrained", "iko_dls_batch_constrained"):
    getattr(O._LIB, f).restype = C.c_int
import test_lane_emulation as t
L = C.CDLL("/tmp/liblane_emu_asan.so"); L.lane_emu_last_error.restype = C.c_char_p
t.test_lane_program_stagewise(L, "cassie_fixed", "LeftFootFront")
t.test_lane_program_full_loop(L, "ur5", "tool0", 50, -1.0)
t.test_lane_program_types_and_weights(L, 1, [2.0, 1.0, 0.25])
t.test_tree_program_full_loop(L, 50, -1.0)
t.test_tree_program_types_weights_priorities(L)
t.test_tree_program_single_chain_plus_base_task(L)
for c in sorted(t.GENERIC_CASES): t.test_generic_program_matches_oracle(L, c)
for c in sorted(t.PIK_CASES): t.test_pik_program_matches_oracle(L, c)
for c in sorted(t.CONSTRAINT_CASES): t.test_constrained_dls_program_matches_oracle(L, c)
for c in t.COOP_CASES:
    if c != "posture_regulariser": t.test_cooperative_program_matches_oracle(L, c)
import test_lane_emulation_demo_tree as d
for c in sorted(d.CASES): d.test_tree_program_with_the_demo_extras_matches_oracle(L, c)
class MP:  # the tests' monkeypatch, for the LANE_EMU_TRIG switch to the device builds
    def setenv(self, k, v): os.environ[k] = v
    def delenv(self, k, raising=True): os.environ.pop(k, None)
for c in sorted(d.POSTURE_CASES): d.test_tree_program_with_posture_rows_matches_oracle(L, MP(), c)
for c in sorted(d.FIXED_BASE_CASES): d.test_tree_program_on_a_fixed_base_matches_oracle(L, MP(), c)
t.test_lane_program_device_general_build(L, MP(), "cassie_fixed", "LeftFootFront", 50, -1.0)
t.test_tree_program_device_general_build(L, MP(), 50, -1.0)
t.test_device_sincos_accuracy(L)
import test_oracle_pik as p, test_oracle_constraints as oc, test_oracle_com as com
p.test_damp_pseudoinverse_known_answers(None); p.test_rowspace_projector_properties_and_rank(None)
for k in range(len(p.LOOP_CASES)): p.test_oracle_pik_matches_the_twin(None, k)
for k in range(len(oc.CASES)): oc.test_constrained_dls_matches_the_twin_and_stays_in_the_null_space(None, k)
for c in com.CASES: com.test_dls_with_a_centre_of_mass_task_matches_the_twin_and_converges(None, *c)
print("asan/ubsan: clean")
PY
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libstdc++.so.6)" ASAN_OPTIONS=detect_leaks=0 python /tmp/ik_asan_run.py
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libstdc++.so.6)" ASAN_OPTIONS=detect_leaks=0 python tools/fuzz_urdf.py 30
