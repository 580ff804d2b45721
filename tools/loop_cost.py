#!/usr/bin/env python3
"""Static cost of a kernel's iteration loop for ONE wave per SIMD on gfx950, from the compiler's assembly listing.

Model (measured with tools/issue_probe.hip, profiles/r02_issue_probe.csv): a lone wave issues one instruction of ANY kind
(VALU, SALU, s_nop, s_waitcnt ...) per 4 cycles; FP64 transcendentals (v_rcp/rsq/sqrt_f64) hold the issue for 16 cycles and
v_mov_b64 for 8; the result of an FP64 VALU instruction can be consumed 8 cycles after its issue (20 for a transcendental);
scalar loads return after --sload cycles.  The loop is walked in program order; forward branches inside the loop are taken
only when their target is listed with --taken.

    hipcc -S ... -o kernels.s ; python tools/loop_cost.py kernels.s dls_chain_kernelILi7ELi2E [--taken LBB29_77] [--top 25]
"""
import argparse
import collections
import re
import sys

TRANS = ("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")
REG = re.compile(r"\b([vsa])(\d+)\b|\b([vsa])\[(\d+):(\d+)\]|\b(vcc|exec)(?:_lo|_hi)?\b|\b(scc)\b")


def regs_of(tok):
    out = []
    for m in REG.finditer(tok):
        if m.group(1):
            out.append((m.group(1), int(m.group(2))))
        elif m.group(3):
            out += [(m.group(3), k) for k in range(int(m.group(4)), int(m.group(5)) + 1)]
        elif m.group(6):
            out.append((m.group(6), 0))
        elif m.group(7):
            out.append(("scc", 0))
    return out


def parse_kernel(text, frag, pick="max"):
    cands = list(re.finditer(r"^(_Z\w*%s\w*):" % re.escape(frag), text, re.M))
    if not cands:
        sys.exit("no kernel matching " + frag)
    def tmpl3(m):
        mm = re.search(re.escape(frag) + r"Li(\d+)E", m.group(1))
        return int(mm.group(1)) if mm else 0
    m = max(cands, key=tmpl3) if pick == "max" else cands[0]
    body = text[m.end():text.index(".Lfunc_end", m.end())].split("\n")
    return m.group(1), body


def find_header(body):
    """Index of the depth-1 loop header whose loop has the most lines tagged with it (blocks may sit anywhere in the listing)."""
    headers = [i for i, l in enumerate(body) if "Loop Header: Depth=1" in l]
    if not headers:
        sys.exit("no loop found")
    def size(h):
        label = body[h].split(":")[0].strip().lstrip(".").lstrip("L")
        return sum(1 for l in body if ("Header=%s " % label) in l)
    return max(headers, key=size)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("kernel")
    ap.add_argument("--taken", action="append", default=[], help="label of a forward branch target that IS taken")
    ap.add_argument("--sload", type=int, default=200, help="scalar-load latency in cycles")
    ap.add_argument("--lds", type=int, default=64, help="ds_read latency in cycles")
    ap.add_argument("--f64lat", type=int, default=4, help="cycles from the issue of an FP64 VALU instruction to the issue of a dependent one")
    ap.add_argument("--inner-trips", type=int, default=1, help="trip count of loops nested in the iteration loop (their back edges are taken trips - 1 times)")
    ap.add_argument("--exit", action="append", default=[], help="LABEL:N -- the conditional branch to LABEL (the exit of a nested loop) is taken on every N-th visit")
    ap.add_argument("--top", type=int, default=0)
    ap.add_argument("--first", action="store_true", help="take the first matching instantiation instead of the one with the largest third template argument")
    ap.add_argument("--dump", action="store_true")
    args = ap.parse_args()
    name, body = parse_kernel(open(args.asm).read(), args.kernel, "first" if args.first else "max")
    lines = body
    # walk of the control-flow path from the loop header back to it
    label_at = {}
    for i, l in enumerate(lines):
        mm = re.match(r"^\.?(L?BB\d+_\d+):", l.strip())
        if mm:
            label_at[mm.group(1).lstrip("L")] = i
    def walk_loop(h):
        header = body[h].split(":")[0].strip().lstrip(".").lstrip("L")
        taken = [x.lstrip(".").lstrip("L") for x in args.taken]
        exits = {x.split(":")[0].lstrip(".").lstrip("L"): int(x.split(":")[1]) for x in args.exit}
        t = 0               # cycle at which the next instruction may issue
        ready = collections.defaultdict(int)    # register -> cycle its value is available
        pending_sload = []  # completion times of outstanding scalar loads (in order)
        pending_sregs = []
        pending_lds = []
        counts = collections.Counter()
        ophist = collections.Counter()
        stall_dep = stall_wait = 0
        sites = []
        n = 0
        i = h + 1
        steps = 0
        inner_taken = {}
        while i < len(lines):
            steps += 1
            if steps > 200000:
                return None
            raw = lines[i]
            if i == h:      # fell through into the header: one iteration walked
                break
            i += 1
            l = raw.split(";")[0].strip()
            if not l or l.startswith(".") or l.endswith(":"):
                continue
            op = l.split()[0]
            rest = l[len(op):]
            ophist[op] += 1
            if op.startswith("s_cbranch") or op == "s_branch":
                tgt = rest.strip().lstrip(".").lstrip("L")
                counts["branch"] += 1
                n += 1
                t += 4
                if tgt == header and (op == "s_branch" or tgt not in taken):
                    t += 16      # the back edge: a taken branch (instruction fetch restart; not measured precisely)
                    break
                if tgt in exits and op != "s_branch":
                    inner_taken[tgt] = inner_taken.get(tgt, 0) + 1
                    if inner_taken[tgt] % exits[tgt] == 0:
                        i = label_at[tgt]
                        t += 16
                    continue
                if tgt in label_at and (tgt in taken or op == "s_branch"):
                    i = label_at[tgt]
                    t += 16
                elif tgt in label_at and label_at[tgt] < i and tgt != header and args.inner_trips > 1:
                    # back edge of a nested loop: taken trips - 1 times
                    inner_taken[i] = inner_taken.get(i, 0) + 1
                    if inner_taken[i] % args.inner_trips != 0:
                        i = label_at[tgt]
                        t += 16
                continue
            ops = [o.strip() for o in rest.split(",")] if rest.strip() else []
            # destination: first operand (VOPC writes vcc / an SGPR pair given first as well; stores have none)
            is_store = op.startswith(("global_store", "ds_write", "buffer_store", "flat_store"))
            dst = [] if is_store or op in ("s_waitcnt", "s_nop", "s_cmp_lt_i32", "s_cmp_eq_u32", "s_cmp_lg_u32") or op.startswith("s_cmp") else (regs_of(ops[0]) if ops else [])
            srcs = []
            for o in (ops if is_store or not dst else ops[1:]):
                srcs += regs_of(o)
            if op.startswith(("v_fmac", "v_mac")) or op in ("v_writelane_b32",):
                srcs += dst          # accumulates into its destination
            if "vcc" in l and op.startswith("v_cndmask") and not any(r[0] == "vcc" for r in srcs):
                srcs.append(("vcc", 0))
            if op.startswith("v_cmp") and "_e32" in op:
                dst = [("vcc", 0)]
                srcs = [r for o in ops for r in regs_of(o)]
            if op.startswith("s_cmp") or op in ("s_cmp_lt_i32",):
                dst = [("scc", 0)]
                srcs = [r for o in ops for r in regs_of(o)]
            cost = 4
            lat = 4
            f64 = "_f64" in op
            if op.startswith(TRANS):
                cost, lat = 16, 20
                counts["trans_f64"] += 1
            elif op.startswith("v_mov_b64"):
                cost, lat = 8, 8
                counts["v_mov_b64"] += 1
            elif f64:
                lat = args.f64lat
                counts["fp64_fma" if op.startswith(("v_fma_f64", "v_fmac_f64")) else "fp64_other"] += 1
            elif op.startswith("v_"):
                counts["valu_other"] += 1
            elif op.startswith("s_load"):
                counts["s_load"] += 1
            elif op == "s_waitcnt":
                counts["s_waitcnt"] += 1
            elif op.startswith("s_"):
                counts["salu"] += 1
            elif op.startswith("ds_"):
                counts["lds"] += 1
            else:
                counts["other"] += 1
            if op == "s_nop":
                cost = 4 * (1 + int(ops[0], 0)) if ops else 4
            n += 1
            start = t
            dep = max([ready[r] for r in srcs] + [0])
            if dep > start:
                stall_dep += dep - start
                sites.append((dep - start, "dep", l))
                start = dep
            if op == "s_waitcnt":
                need = 0
                if "lgkmcnt(0)" in l or "lgkmcnt" in l:
                    need = max(pending_sload + pending_lds + [0])
                    pending_sload, pending_lds = [], []
                if need > start:
                    stall_wait += need - start
                    sites.append((need - start, "wait", l))
                    start = need
            if op.startswith("s_load"):
                done = max(start + args.sload, (pending_sload[-1] if pending_sload else 0))
                pending_sload.append(done)
                for r in dst:
                    ready[r] = 0     # guarded by s_waitcnt, not by the scoreboard
            elif op.startswith("ds_read"):
                pending_lds.append(start + args.lds)
                for r in dst:
                    ready[r] = 0
            else:
                for r in dst:
                    ready[r] = start + lat
            t = start + cost
        return dict(t=t, n=n, counts=counts, ophist=ophist, stall_dep=stall_dep, stall_wait=stall_wait, sites=sites)

    best = None
    for h in [i for i, l in enumerate(body) if "Loop Header: Depth=1" in l]:
        res = walk_loop(h)
        if res and (best is None or res["n"] > best["n"]):
            best = res
    if best is None:
        sys.exit("no loop whose walk returns to its header (give --taken labels)")
    t, n, counts, ophist, stall_dep, stall_wait, sites = (best[k] for k in ("t", "n", "counts", "ophist", "stall_dep", "stall_wait", "sites"))
    total = t
    print("kernel %s" % name)
    print("loop: %d instructions on the walked path, %d cycles = %d quads per iteration" % (n, total, total // 4))
    print("  issue: %d cycles; dependency stalls %d; s_waitcnt stalls %d (scalar load latency %d)" %
          (total - stall_dep - stall_wait, stall_dep, stall_wait, args.sload))
    print("  " + ", ".join("%s %d" % kv for kv in sorted(counts.items(), key=lambda kv: -kv[1])))
    if args.dump:
        print("  opcodes: " + ", ".join("%s %d" % kv for kv in ophist.most_common(60)))
    if args.top:
        agg = collections.Counter()
        for c, kind, l in sites:
            agg[(kind, l)] += c
        for (kind, l), c in agg.most_common(args.top):
            print("  %5d cycles %-4s %s" % (c, kind, l))


if __name__ == "__main__":
    main()
