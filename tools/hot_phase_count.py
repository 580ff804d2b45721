#!/usr/bin/env python3
"""FP64 instructions of the headline loop by phase (the latency build's costing): tools/hot_phase_count.hip wraps
each function of device/chain_hot.hpp -- the seven sincos, hot_evaluate (which contains them), hot_gram, chol_solve<6>, hot_step -- for
the Cassie leg's structure code in a kernel of its own; this compiles it to assembly (no GPU) and counts.
    python tools/hot_phase_count.py"""
import collections
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-fast-math", "-ffp-contract=on", "-fno-signed-zeros", "-fno-honor-nans", "-fno-honor-infinities"]

with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "phase.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-I" + os.path.join(ROOT, "ik_amd", "csrc"), "-I" + os.path.join(ROOT, "include"), "-S",
                           "--cuda-device-only", os.path.join(ROOT, "tools", "hot_phase_count.hip"), "-o", out], stderr=subprocess.DEVNULL)
    text = open(out).read()
total = 0
for k in ("k_sincos", "k_evaluate", "k_gram", "k_chol", "k_step"):
    body = re.search(r"^%s:(.*?)s_endpgm" % k, text, re.S | re.M).group(1)
    ins = [l.split()[0] for l in body.splitlines() if l.startswith("\t") and l.strip() and not l.strip().startswith((".", ";"))]
    f64 = [i for i in ins if "_f64" in i]
    if k != "k_sincos":
        total += len(f64)
    print("%-11s %4d FP64 instructions  %s" % (k[2:], len(f64), dict(collections.Counter(f64).most_common(5))))
print("evaluate + gram + chol + step = %d (the loop executes 1 052 VALU instructions per wave-iteration: PMC, DESIGN.md section 3.1)" % total)
