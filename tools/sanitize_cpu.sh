#!/usr/bin/env bash
# AddressSanitizer + UBSan pass over everything that can run on the CPU: the URDF loader and problem analysis
# (ik_amd/csrc/model.cpp, problem.cpp), the three device lane programs compiled for the host (tests/lane_emu) and
# the C oracle.  GPU sanitizers are not available on this pool; the device code shares these sources.
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"
g++ -O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-sanitize-recover=undefined -Iinclude -Iik_amd/csrc \
    -o /tmp/liblane_emu_asan.so tests/lane_emu/lane_emu.cpp ik_amd/csrc/model.cpp ik_amd/csrc/problem.cpp
make -s -C oracle asan
cat > /tmp/ik_asan_run.py <<PY
import sys, ctypes as C
sys.path[:0] = ["$ROOT/oracle", "$ROOT", "$ROOT/tests"]
import oracle as O
O._LIB = C.CDLL("$ROOT/oracle/libik_oracle_asan.so")
for f in ("iko_dls", "iko_dls_batch", "iko_task_rows"): getattr(O._LIB, f).restype = C.c_int
import test_lane_emulation as t
L = C.CDLL("/tmp/liblane_emu_asan.so"); L.lane_emu_last_error.restype = C.c_char_p
t.test_lane_program_stagewise(L, "cassie_fixed", "LeftFootFront")
t.test_lane_program_full_loop(L, "ur5", "tool0", 50, -1.0)
t.test_lane_program_types_and_weights(L, 1, [2.0, 1.0, 0.25])
t.test_tree_program_full_loop(L, 50, -1.0)
t.test_tree_program_types_weights_priorities(L)
t.test_tree_program_single_chain_plus_base_task(L)
for c in sorted(t.GENERIC_CASES): t.test_generic_program_matches_oracle(L, c)
print("asan/ubsan: clean")
PY
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libstdc++.so.6)" ASAN_OPTIONS=detect_leaks=0 python /tmp/ik_asan_run.py
