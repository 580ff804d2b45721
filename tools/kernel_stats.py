#!/usr/bin/env python3
"""Static statistics of the DLS kernels' iteration loop, from the gfx950 disassembly.

Counting rule for the FP64-VALU roofline (bench.py `valu_roofline`): every FP64 VALU instruction in
the loop body counts its arithmetic -- FMA/FMAC = 2 flop, MUL/ADD/MIN/MAX = 1, RCP/RSQ/RNDNE/LDEXP/
conversions = 1 -- per lane; the count is of *executed* instructions (what the hardware must issue),
not of an algorithmic minimum.  Both sides of wave-uniform branches are counted, so these static_* numbers are upper
bounds; the flop count bench.py uses is the measured one (tools/pmc_to_stats.py).  Merges into ik_amd/kernel_stats.json.

    python tools/kernel_stats.py
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = {  # display name -> (source file, extra flags, mangled-name fragment, trip count of loops nested in the iteration loop)
    # the structure-specialised chain builds (kernels_hot.hip; the never-stop instantiation sorts last: "Lb1E")
    "dls_chain<NJ=7,full>": ("kernels_hot.hip", ["-fno-signed-zeros", "-fno-honor-nans", "-fno-honor-infinities"], "dls_chain_hot_kernelILi7E", 1),
    "dls_chain<NJ=6,full>": ("kernels_hot.hip", ["-fno-signed-zeros", "-fno-honor-nans", "-fno-honor-infinities"], "dls_chain_hot_kernelILi6E", 1),
    "dls_tree<NJ=7,chains=2,base_task>": ("kernels.hip", [], "dls_tree_kernelILi7ELi2E", 2),
}
FLOPS = {"v_fma_f64": 2, "v_fmac_f64": 2, "v_mul_f64": 1, "v_add_f64": 1, "v_min_f64": 1, "v_max_f64": 1,
         "v_rcp_f64": 1, "v_rsq_f64": 1, "v_rndne_f64": 1, "v_ldexp_f64": 1, "v_cvt_i32_f64": 1, "v_cvt_f64_i32": 1,
         "v_sqrt_f64": 1, "v_fract_f64": 1, "v_trig_preop_f64": 1, "v_div_scale_f64": 1, "v_div_fmas_f64": 2,
         "v_div_fixup_f64": 1, "v_frexp_mant_f64": 1}


def main():
    texts = {}
    with tempfile.TemporaryDirectory() as td:
        for src, flags in {(v[0], tuple(v[1])) for v in KERNELS.values()}:
            asm = os.path.join(td, src + ".s")
            cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                   "-I" + os.path.join(ROOT, "ik_amd", "csrc"), "--offload-arch=gfx950", "-fno-fast-math", "-ffp-contract=" + os.environ.get("IKGPU_FP_CONTRACT", "on"),
                   *flags, "-S", "--cuda-device-only", os.path.join(ROOT, "ik_amd", "csrc", src), "-o", asm]
            subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
            texts[src] = open(asm).read()
    out = {}
    for disp, (src, _flags, frag, trips) in KERNELS.items():
        text = texts[src]
        # several instantiations share the fragment (third template argument: 0 = the general build, otherwise the mask of
        # the hot build the launcher picks for these workloads): take the hot one
        cands = list(re.finditer(r"^(_Z\w*%sLi(\d+)E\w*):" % re.escape(frag), text, re.M))
        if cands:
            m = max(cands, key=lambda mm: int(mm.group(2)))
        else:   # the structure-specialised kernels: (NJ, three structure words, never-stop flag) -- take the never-stop build
            cands = [mm for mm in re.finditer(r"^(_Z\w*%s\w*):" % re.escape(frag), text, re.M) if "Lb1E" in mm.group(1)]
            if not cands:
                continue
            m = cands[0]
        name = m.group(1)
        body = text[m.end():text.index(".Lfunc_end", m.end())].split("\n")
        # Blocks carry "; =>This Inner Loop Header" / "; =>This Loop Header" / ";   in Loop: Header=BBn_m Depth=d"
        # comments.  The DLS iteration loop is the depth-1 loop with the most instructions (the LDS staging loop is
        # the other one); blocks of loops nested inside it (the rolled 2-trip chain loops of the tree kernel) are
        # weighted by their trip count.
        per_header = collections.defaultdict(collections.Counter)   # header label -> instruction counts of its own blocks
        depth_of, parent_of = {}, {}
        cur = None
        for l in body:
            t = l.strip()
            mlab = re.match(r"^\.?L?(BB\d+_\d+):", t)
            if mlab or t.startswith("; %bb."):
                mh = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", t)
                if mlab and ("Loop Header: Depth=" in t):
                    cur = mlab.group(1)
                    depth_of[cur] = int(re.search(r"Loop Header: Depth=(\d+)", t).group(1))
                elif mh:
                    cur = mh.group(1)
                    depth_of.setdefault(cur, int(mh.group(2)))
                else:
                    cur = None
                continue
            mp = re.search(r"Parent Loop (BB\d+_\d+) Depth=(\d+)", t)
            if mp and cur is not None and t.startswith(";"):
                parent_of[cur] = mp.group(1)
                continue
            if cur is None or not t or t.startswith((".", ";", "//")):
                continue
            per_header[cur][re.sub(r"_e(32|64)$", "", t.split()[0])] += 1
        if not per_header:
            continue
        # children: loops whose header block says "Parent Loop X"; blocks "in Loop: Header=H" belong to H
        for h in list(per_header):
            if h not in parent_of and depth_of.get(h, 1) > 1:
                # a nested loop whose header line lacked the Parent comment: attach to the largest depth-1 loop
                parent_of[h] = None
        tops = [h for h in per_header if depth_of.get(h, 1) == 1]
        def total(h):
            c = collections.Counter(per_header[h])
            for k, par in parent_of.items():
                if par == h or (par is None and h == main):
                    sub = total(k)
                    for kk, vv in sub.items():
                        c[kk] += vv * trips
            return c
        main = max(tops, key=lambda h: sum(per_header[h].values()) + sum(sum(per_header[k].values()) for k in per_header if depth_of.get(k, 1) > 1))
        c = total(main)
        meta = text[text.index(name, text.index(".amdhsa_kernel")):]
        def grab(key):
            mm = re.search(r"%s\s+(\d+)" % key, meta)
            return int(mm.group(1)) if mm else None
        flops = sum(FLOPS.get(k, 0) * v for k, v in c.items())
        out[disp] = {
            "loop_instructions": sum(c.values()),
            "fp64_valu_instructions": sum(v for k, v in c.items() if k.endswith("_f64")),
            "fp64_fma_instructions": c["v_fma_f64"] + c["v_fmac_f64"],
            "flop_per_iteration": flops,
            "lds_reads": sum(v for k, v in c.items() if k.startswith("ds_read")),
            "next_free_vgpr": grab(r"\.amdhsa_next_free_vgpr"),
            "accum_offset": grab(r"\.amdhsa_accum_offset"),
            "top_instructions": dict(c.most_common(12)),
        }
    path = os.path.join(ROOT, "ik_amd", "kernel_stats.json")
    old = {}
    if os.path.exists(path):
        old = json.load(open(path))
    for k, v in out.items():  # keep what tools/pmc_to_stats.py measured (traffic, flop counts, raw counters)
        merged = dict(old.get(k, {}))
        merged.update({"static_" + kk if kk in ("flop_per_iteration", "loop_instructions", "fp64_valu_instructions",
                                                  "fp64_fma_instructions", "lds_reads", "top_instructions") else kk: vv
                       for kk, vv in v.items()})
        out[k] = merged
    for k, v in old.items():
        out.setdefault(k, v)
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    sys.exit(main())
