#!/usr/bin/env python3
"""DESIGN.md section 3.1's two-phase table from profiles/r04_refill_timing_{chain,tree,static}.txt (tools/refill_timing.py): for every kernel,
batch above the resident lanes and target distribution (max_iterations = 100) the lock-step time, the better of the two refill grids,
the default policy, and the default against the better fixed mode.
    python tools/r04_two_phase_table.py"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
print("| kernel, B, targets | lock-step | refill from iteration 0 (better of 4 / 8 waves per CU) | **default (two phases)** | against the better fixed mode |")
print("|---|---|---|---|---|")
worst = {}
for name in ("chain", "tree", "static"):
    kernel = None
    for l in open(os.path.join(ROOT, "profiles", "r04_refill_timing_%s.txt" % name)):
        if l.startswith("== "):
            kernel = l[3:].strip()
            continue
        m = re.match(r"(\w+) B=\s*(\d+) max_it=100 .*lock-step ([\d.]+) ms \| refill 4 w/CU ([\d.]+) ms\S* \| refill 8 w/CU ([\d.]+) ms\S* \| default policy ([\d.]+) ms", l)
        if not m or int(m.group(2)) <= 65536:
            continue
        mode, B, lock, r4, r8, pol = m.group(1), int(m.group(2)), float(m.group(3)), float(m.group(4)), float(m.group(5)), float(m.group(6))
        best = min(lock, r4, r8)
        rel = 100.0 * (pol / best - 1.0)
        worst[kernel] = max(worst.get(kernel, -1e9), rel)
        print("| `%s`, %s, %s | %.3f ms | %.3f | **%.3f** | %+.0f %% |" % (kernel, "262 144" if B == 262144 else "2²⁰", mode, lock, min(r4, r8), pol, rel))
print()
for k, v in worst.items():
    print("worst cell of %s: %+.1f %%" % (k, v))
