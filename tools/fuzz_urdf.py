"""Mutation fuzzer for the URDF reader and the problem analysis (ik_amd/csrc/model.cpp, problem.cpp) under AddressSanitizer +
UBSan: run by tools/sanitize_cpu.sh against the sanitizer build of the lane emulator (/tmp/liblane_emu_asan.so).
    python tools/fuzz_urdf.py [seconds]"""
import ctypes as C
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
from ik_amd import capi
L = C.CDLL("/tmp/liblane_emu_asan.so"); L.lane_emu_last_error.restype = C.c_char_p
base = [open(os.path.join(ROOT, "fixtures", "models", "%s.kin.urdf" % n), "rb").read() for n in ("ur5", "cassie_fixed", "cassie")]
rng = random.Random(1234)
task = (capi.Task * 1)(capi.Task(1, 0, 2, 0, (C.c_double * 6)(*[1.0] * 6)))
prm = capi.DlsParams(1, 1e-2, 1.0, -1.0)
t0 = time.time(); n = ok = 0
tokens = [b"<", b">", b"/", b'"', b"'", b"=", b" ", b"<joint", b"</joint>", b"<link", b"name=", b"xyz=", b"1e309", b"-", b"nan", b"\x00", b"&amp;", b"<!--", b"-->", b"<?xml", b"type=\"fixed\"", b"type=\"revolute\""]
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
while time.time() - t0 < budget:
    s = bytearray(rng.choice(base))
    for _ in range(rng.randint(1, 6)):
        op = rng.random(); i = rng.randrange(len(s))
        if op < 0.3: s[i] = rng.randrange(256)
        elif op < 0.55: del s[i:i + rng.randint(1, 40)]
        elif op < 0.85: s[i:i] = rng.choice(tokens)
        else: j = rng.randrange(len(s)); s[i:i] = s[j:j + rng.randint(1, 60)]
    buf = bytes(s)
    e = (C.c_double * 6)(); J = (C.c_double * 512)(); o = (C.c_double * 12)(); q = (C.c_double * 64)()
    rc = L.lane_emu_run(buf, C.c_size_t(len(buf)), rng.randint(0, 3), task, 1, 2, C.c_int64(0), q, q, C.byref(prm), q, None, None, e, J, o, 1)
    n += 1; ok += rc == 0
print("fuzzed %d inputs, %d parsed and analysed, no crash" % (n, ok))
