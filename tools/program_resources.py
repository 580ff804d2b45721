#!/usr/bin/env python3
"""Registers, scratch and instruction counts of the run-time generated lane programs, offline (no GPU): for each named problem the source
rtc.cpp hands to the compile worker is dumped (IKGPU_RTC_DUMP), compiled with hipcc and the worker's flags to assembly, and the
kernel's resource lines are printed.
    python tools/program_resources.py pik:feet_then_pelvis generic:three_feet_frames constrained:demo_everything_on ..."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
os.environ.setdefault("IKGPU_TREE_STATIC_ROWS", "12")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-fast-math", "-ffp-contract=on", "-fno-signed-zeros", "-fno-honor-nans",
         "-fno-honor-infinities", "-mllvm", "-two-entry-phi-node-folding-threshold=100000", "-mllvm", "-pragma-unroll-threshold=4000000"]


def main():
    import ik_amd
    from test_gpu_generic import CASES as GENERIC, build
    from test_gpu_constraints import CASES as CONSTRAINED
    from test_gpu_pik import PIK_CASES
    for spec in sys.argv[1:]:
        kind, case = spec.split(":")
        cspecs = []
        if kind == "pik":
            name, ff, specs, edit, _ = PIK_CASES[case]
        elif kind == "constrained":
            name, ff, specs, cspecs = CONSTRAINED[case]
            edit = None
        else:
            name, ff, specs, edit = GENERIC[case]
        with tempfile.TemporaryDirectory() as td:
            os.environ.update(IKGPU_RTC_DUMP=td, IKGPU_CACHE_DIR=os.path.join(td, "cache"))
            if kind != "pik":
                os.environ["IKGPU_DLS_KERNEL"] = "generic"
            os.makedirs(os.path.join(td, "cache"), mode=0o700)
            ik, _, model, problem, _, om, ot, q0, tg = build(name, ff, specs, 2, xml_edit=edit, device=False)
            for i, (f, t, r) in enumerate(cspecs):
                problem.add_frame_constraint("c%d" % i, ik_amd.FrameConstraint.create(model, f, ik_amd.KinematicType(t), r))
            try:
                print("%s: %s" % (spec, ik_amd.precompile(problem)))
            except Exception as e:
                print("%s: %s" % (spec, str(e)[:200]))
            os.environ.pop("IKGPU_DLS_KERNEL", None)
            want = "pik_static_" if kind == "pik" else "generic_static_"
            for src in sorted(glob.glob(os.path.join(td, want + "*.hip"))):
                if kind != "pik" and "refill" in os.path.basename(src):
                    continue
                asm = src[:-4] + ".s"
                subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, "-I" + os.path.join(ROOT, "ik_amd", "csrc", "device"), "-S", "--cuda-device-only", src, "-o", asm],
                                      stderr=subprocess.DEVNULL)
                text = open(asm).read()
                get = lambda k: re.search(r"\.%s:\s+(\d+)" % k, text).group(1)
                body = text[text.index("; %bb.0"):text.index(".Lfunc_end0")] if "; %bb.0" in text else text
                ins = [l.split()[0] for l in body.splitlines() if l.startswith("\t") and not l.strip().startswith((";", "."))]
                n64 = sum(1 for i in ins if i.endswith("_f64") or "_f64_" in i)
                print("    %-40s vgpr %s agpr %s sgpr %s scratch %s B | %d instructions, %d f64, %d v_accvgpr, %d scratch_" %
                      (os.path.basename(src), get("vgpr_count"), get("agpr_count"), get("sgpr_count"), get("private_segment_fixed_size"),
                       len(ins), n64, sum(1 for i in ins if i.startswith("v_accvgpr")), sum(1 for i in ins if i.startswith("scratch_"))))


if __name__ == "__main__":
    main()
