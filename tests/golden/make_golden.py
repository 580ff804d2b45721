#!/usr/bin/env python3
"""Generates tests/golden/*.json with the independent numpy twin (oracle/twin.py).

The reference itself cannot run here (Pinocchio / Eigen absent: SURVEY.md 8c) and its tests hold no
numbers, so these vectors are produced by the *second* restatement -- written independently of the C
oracle (4x4 matrices, numpy.linalg.solve, xml.etree loader) -- and pin the C oracle, the product's
URDF loader and, on the GPU, the kernels.  Deterministic: re-running must reproduce the files.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import twin as T  # noqa: E402

MODELS = os.path.join(ROOT, "fixtures", "models")
NOMINAL = json.load(open(os.path.join(MODELS, "cassie.nominal.json")))["joints"]


def m12(M):
    return np.concatenate([M[:3, :3].reshape(9), M[:3, 3]]).tolist()


def case(name, urdf, free_flyer, frames, n, seed):
    rng = np.random.default_rng(seed)
    m = T.load_urdf(os.path.join(MODELS, urdf), free_flyer)
    out = dict(name=name, urdf=urdf, free_flyer=free_flyer, task_frames=frames,
               model=dict(njoints=m.njoints, nq=m.nq, nv=m.nv, nframes=len(m.frames), joint_names=m.names,
                          frame_names=[f["name"] for f in m.frames], idx_q=m.idx_q, idx_v=m.idx_v, parent=m.parent,
                          lower=m.lower.tolist(), upper=m.upper.tolist(),
                          joint_placement=[m12(M) for M in m.placement],
                          frame_parent=[f["parent"] for f in m.frames],
                          frame_placement=[m12(f["placement"]) for f in m.frames]),
               problems=[])
    nj0 = 7 if free_flyer else 0
    names = m.names[2:] if free_flyer else m.names[1:]
    if "cassie" in urdf:
        nominal = np.array([NOMINAL[x] for x in names])
    else:
        nominal = np.array([0.0, -np.pi / 2, np.pi / 2, 0.0, np.pi / 2, 0.0])
    lo, hi = m.lower[nj0:], m.upper[nj0:]
    for k in range(n):
        q0 = T.neutral(m)
        q0[nj0:] = np.clip(nominal + rng.uniform(-0.1, 0.1, nominal.size), lo, hi)
        qs = q0.copy()
        near = (k % 2 == 0) or ("ur5" in urdf)
        if near:
            qs[nj0:] = np.clip(q0[nj0:] + rng.uniform(-0.15, 0.15, nominal.size), lo, hi)
        else:
            qs[nj0:] = rng.uniform(lo, hi)
        if free_flyer:
            q0[:3] = [0.0, 0.0, 1.0] + rng.uniform(-0.02, 0.02, 3)
            quat = np.array([0, 0, 0, 1.0]) + rng.uniform(-0.02, 0.02, 4)
            q0[3:7] = quat / np.linalg.norm(quat)
            qs[:7] = q0[:7]
            v = np.zeros(m.nv)
            v[:3] = rng.uniform(-0.1, 0.1, 3)
            v[3:6] = rng.uniform(-0.2, 0.2, 3)
            qs = T.integrate(m, qs, v)
        tasks = [T.FrameTask(m, f) for f in frames]
        oMf = T.fk(m, qs)[1]
        for t in tasks:
            t.target = oMf[t.frame]
        e, J = T.evaluate(m, tasks, q0)
        trace = []
        q50, _, _ = T.dls(m, tasks, q0, 50, 1e-2, 1.0, -1.0, trace=trace)
        qd, okd, itd = T.dls(m, tasks, q0)
        q1, _, _ = T.dls(m, tasks, q0, 1, 1e-2, 1.0, -1.0)
        q3, _, _ = T.dls(m, tasks, q0, 3, 1e-2, 1.0, -1.0)
        out["problems"].append(dict(
            q0=q0.tolist(), qstar=qs.tolist(), targets=[m12(t.target) for t in tasks],
            oMf_q0=[m12(T.fk(m, q0)[1][t.frame]) for t in tasks],
            e=e.tolist(), J=J.tolist(), JJ=trace[0]["JJ"].tolist(), dq=trace[0]["dq"].tolist(),
            q_after_1=q1.tolist(), q_after_3=q3.tolist(), q_after_50=q50.tolist(),
            final_error_norm=float(np.linalg.norm(T.evaluate(m, tasks, q50)[0])),
            default_stop=dict(q=qd.tolist(), success=bool(okd), iterations=int(itd))))
    return out


def lie_vectors(seed):
    rng = np.random.default_rng(seed)
    rows = []
    for scale in (1e-9, 1e-5, 1e-3, 0.3, 1.0, 2.5, 3.1):
        for _ in range(3):
            w = rng.normal(size=3)
            w = w / np.linalg.norm(w) * scale
            nu = np.concatenate([rng.normal(size=3), w])
            M = T.exp6(nu)
            rows.append(dict(nu=nu.tolist(), M=m12(M), log6=T.log6(M).tolist(), Jlog6=T.Jlog6(M).tolist()))
    return rows


def pik_cases(seed):
    """ik::pik (reference ik/ik/pik.cpp:31-103) vectors from the twin: task levels, lambda per level, q after 1 / 4 / 30
    iterations and with the default stop rule.  Only cases whose result does not hinge on a noise-level rank decision."""
    rng = np.random.default_rng(seed)
    out = []
    specs = [
        ("ur5_pos_then_ori", "ur5.kin.urdf", False, [[("tool0", "universe", T.POSITION)], [("tool0", "universe", T.ORIENTATION)]], [0.1, 0.1]),
        ("cassie_fixed_two_feet", "cassie_fixed.kin.urdf", False,
         [[("LeftFootFront", "universe", T.FULL)], [("RightFootFront", "universe", T.POSITION)]], [0.05, 0.2]),
        ("cassie_demo_levels", "cassie.kin.urdf", True,
         [[("LeftFootFront", "pelvis", T.POSITION), ("pelvis", "universe", T.FULL)], [("LeftFootFront", "universe", T.ALIGN_X + 1)]], [0.1, 0.1]),
    ]
    for name, urdf, ff, levels, lam in specs:
        m = T.load_urdf(os.path.join(MODELS, urdf), ff)
        nj0 = 7 if ff else 0
        names = m.names[2:] if ff else m.names[1:]
        nominal = np.array([NOMINAL[x] for x in names]) if "cassie" in urdf else np.array([0.0, -np.pi / 2, np.pi / 2, 0.0, np.pi / 2, 0.0])
        problems = []
        for _ in range(4):
            q0 = T.neutral(m)
            q0[nj0:] = np.clip(nominal + rng.uniform(-0.1, 0.1, nominal.size), m.lower[nj0:], m.upper[nj0:])
            qs = q0.copy()
            qs[nj0:] = np.clip(q0[nj0:] + rng.uniform(-0.15, 0.15, nominal.size), m.lower[nj0:], m.upper[nj0:])
            if ff:
                q0[:3] = [0.0, 0.0, 1.0]
                qs = T.integrate(m, qs, np.concatenate([rng.uniform(-0.05, 0.05, 3), rng.uniform(-0.1, 0.1, 3), np.zeros(m.nv - 6)]))
            oMf = T.fk(m, qs)[1]
            tl = []
            for lv in levels:
                row = []
                for f, r, typ in lv:
                    t = T.FrameTask(m, f, typ, r)
                    if typ >= T.ALIGN_X:
                        t.target = np.eye(4)
                        t.target[:3, 3] = rng.normal(size=3)
                    else:
                        t.target = T.se3_inv(oMf[t.reference]) @ oMf[t.frame]
                    row.append(t)
                tl.append(row)
            q1 = T.pik(m, tl, q0, 1, 1.0, -1.0, lam)[0]
            q4 = T.pik(m, tl, q0, 4, 1.0, -1.0, lam)[0]
            q30 = T.pik(m, tl, q0, 30, 0.5, -1.0, lam)[0]
            qd, okd, itd = T.pik(m, tl, q0, 100, 1.0, 1e-4, lam)
            problems.append(dict(q0=q0.tolist(), targets=[m12(t.target) for lv in tl for t in lv],
                                 q_after_1=q1.tolist(), q_after_4=q4.tolist(), q_after_30_half_step=q30.tolist(),
                                 default_stop=dict(q=qd.tolist(), success=bool(okd), iterations=int(itd))))
        out.append(dict(name=name, urdf=urdf, free_flyer=ff, lam=lam,
                        levels=[[dict(frame=f, reference=r, type=int(typ)) for f, r, typ in lv] for lv in levels], problems=problems))
    return out


if __name__ == "__main__":
    cases = [
        case("S_cassie_leg", "cassie_fixed.kin.urdf", False, ["LeftFootFront"], 8, 11),
        case("U_ur5", "ur5.kin.urdf", False, ["tool0"], 8, 12),
        case("F_cassie_full", "cassie.kin.urdf", True, ["LeftFootFront", "RightFootFront", "pelvis"], 6, 13),
    ]
    for c in cases:
        with open(os.path.join(HERE, c["name"] + ".json"), "w") as fh:
            json.dump(c, fh)
        print(c["name"], len(c["problems"]), "problems; final errors", ["%.1e" % p["final_error_norm"] for p in c["problems"]])
    with open(os.path.join(HERE, "lie_maps.json"), "w") as fh:
        json.dump(lie_vectors(14), fh)
    with open(os.path.join(HERE, "P_pik.json"), "w") as fh:
        json.dump(pik_cases(15), fh)
