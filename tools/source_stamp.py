#!/usr/bin/env python3
"""Identity of the device code of this checkout: sha256 over the kernel sources and the flags they are built with (the files the
Makefile's kernel objects depend on).  PMC sessions record it (tools/pmc_session.sh), tools/pmc_to_stats.py stamps it into
ik_amd/kernel_stats.json, and bench.py replays a counter figure only when the stamp matches the tree it runs from."""
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def device_source_sha16():
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "ik_amd", "csrc", "device", "*.hpp")) +
                   glob.glob(os.path.join(ROOT, "ik_amd", "csrc", "*.hip")) +
                   [os.path.join(ROOT, "ik_amd", "csrc", f) for f in ("kernels.hpp", "generic_tables.hpp", "Makefile", "rtc.cpp")])   # rtc.cpp: the run-time compiled kernels' source generator and flags
    for f in files:
        h.update(os.path.relpath(f, ROOT).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    json.dump({"device_source_sha16": device_source_sha16()}, sys.stdout)
