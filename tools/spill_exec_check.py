#!/usr/bin/env python3
"""Static check of gfx950 assembly listings for a code-generation hazard met in round 3: a VGPR spill store (`scratch_store ... Folded
Spill`) that the register allocator placed inside a region whose EXEC mask is partial -- between an s_and_saveexec / s_or_saveexec /
s_xor exec and the s_or_b64 exec that restores it.  Lanes that are masked off there never write their slot, and the later reload
hands them a stale value (seen in a run-time specialised generic program: the spill sat in the flow block of a divergent branch the
compiler had made of a nested ?: in dacos, and frozen lanes got their INITIAL q back).  Reports, per kernel, the divergent regions inside
loops and every spill store inside one.

    hipcc -S --cuda-device-only ... -o k.s ; python tools/spill_exec_check.py k.s [more.s ...]
Exit status 1 when a spill store under a partial EXEC mask is found."""
import re
import sys


def check(path):
    bad = 0
    kernel, depth, in_loop = None, 0, False
    regions = spills = 0
    for n, line in enumerate(open(path), 1):
        t = line.strip()
        m = re.match(r"^(_Z\w+|ikgpu_\w+):", t)
        if m and "@" in line:
            kernel, depth, regions, spills = m.group(1), 0, 0, 0
            continue
        if kernel is None:
            continue
        if t.startswith(".Lfunc_end"):
            print("%-110s divergent regions: %3d, spill stores under a partial EXEC: %d" % (kernel[:110], regions, spills))
            kernel = None
            continue
        if re.search(r"\bs_(and|or)_saveexec_b64\b", t):
            depth += 1 if "s_and_saveexec" in t else 0
            regions += 1 if "s_and_saveexec" in t else 0
        elif re.match(r"s_or_b64\s+exec,\s*exec,", t) and depth > 0:
            depth -= 1
        elif "scratch_store" in t and "Spill" in t and depth > 0:
            spills += 1
            bad += 1
            print("  %s:%d  %s" % (path, n, t))
    return bad


if __name__ == "__main__":
    total = sum(check(p) for p in sys.argv[1:])
    sys.exit(1 if total else 0)
