"""Full-size (B = 65536) lane-by-lane parity of BASELINE.json's configs 2/4 (Cassie leg), 3 (Cassie full body) and 5 (UR5 and UR10,
stock limits and limits narrowed to +-2 rad so the joint-limit projection binds) and of a NON-fixture arm (arm7: general joint-origin
rotations, oblique axes) against the CPU oracle, through the C ABI, at the metric's 50 fixed iterations -- for EVERY build of the chain
kernel a problem can get: "hot" (structure-specialised, compiled into the library), "hot-rtc" (the same kernel compiled for the
chain's structure code at run time) and "general" (IKGPU_CHAIN_HOT=0: what any chain ran on before round 3 and what runs when
hipRTC is absent).  Flags and iteration counts must be equal on every problem.

Parity rule (round 3; nothing of the system under test decides which lanes count):
  1. sensitivity probes of the ORACLE ALONE: a problem is *excluded from the 1e-6 rad bar* when the oracle's own answer moves by more
     than 1e-7 rad under a 1e-13 perturbation of its inputs (q0 +, target translations +, both -).  A lane stalled on a joint limit
     or far from a reachable pose amplifies rounding differences, so neither side has an answer good to 1e-6 there.
  2. every other problem, converged or not: |q_gpu - q_oracle| <= 1e-6 rad.
  3. every excluded problem (and any problem that passes the probes and still misses the bar) is ARBITRATED by the same oracle in
     _Float128 arithmetic (oracle/ik_oracle_ext.c, 113-bit significand): r = |q_gpu - q_ext| / max(|q_oracle - q_ext|, 1e-9).
     r <= 10 -- the device is no farther from the (near-)exact trajectory than ten times the double oracle -- is required where the
     excluded set is a handful (Cassie leg: 5 lanes), of all but one lane in five (r is a ratio of two rounding-error draws even
     there), and r <= 100 of every lane.  In the chaotic clamp workloads (thousands of excluded
     lanes whose 50-step map amplifies one rounding error by > 1e9) r is the ratio of two independent draws from the same heavy-tailed
     distribution, so the assertion is statistical: median r in [0.5, 2], at most 10 % of the arbitrated lanes with r > 10, and about
     as many with r < 0.1 (the double oracle ten times farther than the device) -- neither side is systematically nearer.
  The per-case counts (excluded by perturbation / arbitrated / failing the r <= 10 rule / mirror) are written to
  gpurun_out/parity_counts.json (committed as profiles/r03_parity_counts.json).

The proof that the chaotic lanes are chaos and not error is `test_step_synchronised_along_the_oracle_trajectory`: the device is fed
the ORACLE's iterate at each of the 50 steps and must return the next iterate to 1e-9 rad on ALL 65536 lanes, no exclusions; at
every tenth step the one-step errors of device and double oracle against the _Float128 oracle are compared as distributions (the
device's median and 99th percentile within 2x, its maximum within 5x of the double oracle's).  The oracle runs on all host cores."""
import json
import os
import re

import numpy as np
import pytest

from conftest import ROOT, urdf_path

pytestmark = pytest.mark.gpu

TOL = 1e-6           # rad, north_star
B = 65536
ITERS = 50
SEEDS = [0, 1, 2]    # seed 0 primary, 1 and 2 for repeats.  Every threshold below was set on seed 0 in round 3 and is
                     # FROZEN: seeds 1 and 2 run against the same constants (counts of all three in profiles/r04_parity_counts.json)


@pytest.fixture(scope="module")
def torch_cuda(native_built):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


COUNTS = {}


def _record(label, counts):
    COUNTS[label] = counts
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_counts.json"), "w") as fh:
        json.dump(COUNTS, fh, indent=1, sort_keys=True)


def oracle_sensitivity(O, solve, tg, q0, q_ref):
    """max |q - q_ref| of the oracle's own answers under three 1e-13 perturbations of the inputs."""
    sens = np.zeros(q0.shape[0])
    for dq, dt in ((1e-13, 0.0), (0.0, 1e-13), (-1e-13, -1e-13)):
        tgp = tg.copy()
        tgp[:, :, 9:] += dt
        qp, _, _ = solve(tgp, q0 + dq, None)
        sens = np.maximum(sens, np.abs(qp - q_ref).max(axis=1))
    return sens


def parity_counts(q_gpu, q_ref, sens, solve_ext, tg, q0):
    """The round-3 parity rule on one batch: returns the counts dict (see the module docstring)."""
    stable = sens <= 1e-7
    d = np.abs(q_gpu - q_ref).max(axis=1)
    beyond = stable & (d > TOL)
    arb = np.flatnonzero(~stable | beyond)
    c = {"problems": int(d.size), "stable": int(stable.sum()), "max_abs_dq_rad_stable_within_bar": float(d[stable & ~beyond].max()) if (stable & ~beyond).any() else 0.0,
         "excluded_by_perturbation": int((~stable).sum()), "stable_beyond_bar": int(beyond.sum()), "arbitrated": int(arb.size),
         "failing": 0, "mirror": 0, "median_ratio": None, "max_oracle_self_sensitivity_rad": float(sens.max())}
    if arb.size:
        q_ext, _, _ = solve_ext(tg[arb], q0[arb])
        eg, eo = np.abs(q_gpu[arb] - q_ext).max(axis=1), np.abs(q_ref[arb] - q_ext).max(axis=1)
        ratio, mirror = eg / np.maximum(eo, 1e-9), eo / np.maximum(eg, 1e-9)
        c.update(failing=int((ratio > 10).sum()), mirror=int((mirror > 10).sum()), median_ratio=float(np.median(ratio)),
                 max_ratio=float(ratio.max()), failing_lanes=[int(x) for x in arb[ratio > 10][:16]])
    return c


def assert_parity(c, label, max_excluded, statistical):
    """max_excluded: lanes (>= 1) or fraction (< 1) the perturbation probes may exclude; statistical: the chaotic-regime form of rule 3."""
    allowed = max_excluded if max_excluded >= 1 else max_excluded * c["problems"]
    assert c["excluded_by_perturbation"] <= allowed, (label, c)
    if not statistical:
        # a handful of excluded lanes: r is still the ratio of two rounding-error draws (P(r > 10) is a few per cent per lane for two
        # equally accurate implementations), so one lane in five may exceed 10 -- none may exceed 100
        assert c["stable_beyond_bar"] == 0, (label, c)
        assert c["failing"] <= max(1, 0.2 * c["arbitrated"]) and (c["arbitrated"] == 0 or c["max_ratio"] <= 100.0), (label, c)
        return
    assert c["stable_beyond_bar"] <= max(2, 2e-4 * c["problems"]), (label, c)   # (they are arbitrated with the excluded lanes)
    if c["arbitrated"] >= 100:
        assert 0.5 <= c["median_ratio"] <= 2.0, (label, c)
        assert c["failing"] <= 0.10 * c["arbitrated"], (label, c)
        assert c["failing"] <= 1.5 * c["mirror"] + 20, (label, c)
    else:
        assert c["failing"] == 0, (label, c)


def _compare(torch, model, problem, data, tasks, q0, targets_dev, max_excluded, label, statistical=False):
    import ik_amd
    import oracle as O
    om = O.OracleModel(model.flat())
    Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
    Q, ok, it = ik_amd.dls_batch(problem, Q0, targets_dev, data, ik_amd.never_stop_visitor(), ik_amd.dls_parameters(max_iterations=ITERS))
    q_gpu = Q.cpu().numpy().T
    tg = targets_dev.permute(2, 0, 1).contiguous().cpu().numpy()
    prm = O.params(ITERS, 1e-2, 1.0, -1.0)
    cores = os.cpu_count() or 1

    def solve(tg_, q0_, ext):
        return O.dls_batch(om, tasks, tg_, q0_, prm, cores, ext=ext)
    q_ref, ok_ref, it_ref = solve(tg, q0, None)
    assert np.array_equal(ok.cpu().numpy(), ok_ref) and np.array_equal(it.cpu().numpy(), it_ref), label
    sens = oracle_sensitivity(O, solve, tg, q0, q_ref)
    c = parity_counts(q_gpu, q_ref, sens, lambda t_, q_: solve(t_, q_, "q"), tg, q0)
    c["kernel"] = data.kernel
    print("%s: %s" % (label, json.dumps(c)))
    _record(label, c)
    assert_parity(c, label, max_excluded, statistical)
    return q_gpu


CHAIN_CASES = [
    # name, frame, narrowed limits, target distribution, lanes / fraction the probes may exclude, statistical form of rule 3
    ("cassie_fixed", "LeftFootFront", None, "uniform", 32, False),    # the bench workload of configs 2 / 4 (~6 % stall on a limit)
    ("cassie_fixed", "LeftFootFront", None, "near", 0, False),
    ("ur5", "tool0", None, "near", 0, False),                          # config 5, stock limits
    ("ur10", "tool0", None, "near", 0, False),
    ("ur5", "tool0", 2.0, "uniform", 0.12, True),                      # config 5 with the projection live: far targets are chaotic
    ("ur10", "tool0", 2.0, "uniform", 0.12, True),
    ("arm7", "tool", None, "near", 0, False),                          # a non-fixture chain: no pre-built structure-specialised kernel
    ("arm7", "tool", None, "uniform", 0.30, True),
]
BUILDS = ["default", "general"]   # default: hot for the fixture robots, hot-rtc for arm7 (general when hipRTC is absent)


def _chain_case(torch, name, frame, narrow, mode, build="default", seed=0):
    import ik_amd
    import oracle as O
    from ik_amd import workload
    xml = open(urdf_path(name)).read()
    if narrow:
        xml = re.sub(r'lower="[-0-9.e]+" upper="[-0-9.e]+"', 'lower="-%.1f" upper="%.1f"' % (narrow, narrow), xml)
    model = ik_amd.Model.from_urdf_xml(xml)
    problem = ik_amd.InverseKinematicsProblem(model)
    problem.add_frame_task("t", ik_amd.FrameTask.create(model, frame, ik_amd.KinematicType.Full))
    # the build is decided when the problem is created (ikgpu_problem_create) and is part of the kernel's name
    prev = os.environ.get("IKGPU_CHAIN_HOT")
    if build == "general":
        os.environ["IKGPU_CHAIN_HOT"] = "0"
    try:
        data = ik_amd.dls_data(problem, device=0)
    finally:
        if build == "general":
            if prev is None:
                del os.environ["IKGPU_CHAIN_HOT"]
            else:
                os.environ["IKGPU_CHAIN_HOT"] = prev
    want = ",general>" if build == "general" else (",hot-rtc>", ",general>") if name == "arm7" else ",hot>"
    assert data.kernel.endswith(want), (data.kernel, build)
    nominal = workload.UR5_NOMINAL if name.startswith("ur") else np.zeros(model.nq) if name == "arm7" else workload.cassie_nominal(model.names)
    q0, qs = workload.chain_workload(model.lowerPositionLimit, model.upperPositionLimit, nominal, np.arange(B), seed, mode)
    T = ik_amd.task_frames_fk_batch(problem, torch.from_numpy(np.ascontiguousarray(qs.T)).cuda(), data)
    tasks = O.make_tasks([(model.getFrameId(frame), 0, 2, 0, None)])
    return xml, model, problem, data, q0, T, tasks


def _step_synchronised(torch, model, problem, data, tasks, q0, T, steps, step_bar, label):
    """From the oracle's k-th iterate the device's next iterate, all lanes, no exclusions; every tenth step (first 8192 lanes) the
    one-step errors of device and double oracle against the _Float128 oracle, as distributions."""
    import ik_amd
    import oracle as O
    om = O.OracleModel(model.flat())
    tg = T.permute(2, 0, 1).contiguous().cpu().numpy()
    one = O.params(1, 1e-2, 1.0, -1.0)
    p1 = ik_amd.dls_parameters(max_iterations=1)
    cores = os.cpu_count() or 1
    q, worst = q0, 0.0
    out = None
    NX = 8192
    worst_ratio = 0.0
    for k in range(steps):
        q_next, _, _ = O.dls_batch(om, tasks, tg, q, one, cores)
        out = ik_amd.dls_batch(problem, torch.from_numpy(np.ascontiguousarray(q.T)).cuda(), T, data, ik_amd.never_stop_visitor(), p1, out=out)
        q_dev = out[0].cpu().numpy().T
        d = np.abs(q_dev - q_next).max()
        worst = max(worst, d)
        assert d <= step_bar, (label, k, d)
        if k % 10 == 0:
            q_x, _, _ = O.dls_batch(om, tasks, tg[:NX], q[:NX], one, cores, ext="q")
            e_dev, e_orc = np.abs(q_dev[:NX] - q_x).max(axis=1), np.abs(q_next[:NX] - q_x).max(axis=1)
            for pct, factor in ((50, 2.0), (99, 2.0), (100, 5.0)):     # (the maximum of 8192 draws is a noisy statistic)
                a, b = np.percentile(e_dev, pct), np.percentile(e_orc, pct)
                worst_ratio = max(worst_ratio, a / (b + 1e-15))
                assert a <= factor * b + 1e-15, (label, k, pct, a, b)
        q = q_next
    print("%s [%s]: worst one-step |dq| over %d steps x %d lanes: %.3e rad; one-step error vs _Float128, device / oracle, worst "
          "percentile ratio %.2f" % (label, data.kernel, steps, q0.shape[0], worst, worst_ratio))
    _record("step-synchronised " + label, {"kernel": data.kernel, "steps": steps, "lanes": int(q0.shape[0]), "worst_one_step_abs_dq_rad": float(worst),
                                           "bar_rad": step_bar, "worst_percentile_ratio_vs_float128": float(worst_ratio)})


@pytest.mark.parametrize("seed", SEEDS)
@pytest.mark.parametrize("build", BUILDS)
@pytest.mark.parametrize("name,frame,narrow,mode", [c[:4] for c in CHAIN_CASES])
def test_step_synchronised_along_the_oracle_trajectory(torch_cuda, name, frame, narrow, mode, build, seed):
    """All 65536 lanes, all 50 steps, no exclusions: from the oracle's k-th iterate the device's next iterate equals the
    oracle's to 1e-9 rad (one DLS step: evaluate, solve, integrate, project onto the limits -- reference ik/ik/dls.cpp:14-71).
    Every tenth step, on the first 8192 lanes: the device's and the double oracle's one-step errors against the _Float128 oracle."""
    torch = torch_cuda
    xml, model, problem, data, q0, T, tasks = _chain_case(torch, name, frame, narrow, mode, build, seed)
    # far targets drive the made-up arm through near-singular poses (damping 1e-2: condition ~1e4-1e5 on the rounding of J): one step
    # of either side is good to ~1e-9 there, so that case gets 1e-8; the distribution check against _Float128 is the sharp one
    step_bar = 1e-8 if (name, mode) == ("arm7", "uniform") else 1e-9
    _step_synchronised(torch, model, problem, data, tasks, q0, T, ITERS, step_bar, "%s %s narrow=%s %s [%s] seed %d" % (name, frame, narrow, mode, build, seed))


@pytest.mark.parametrize("seed", SEEDS)
@pytest.mark.parametrize("build", BUILDS)
@pytest.mark.parametrize("name,frame,narrow,mode,max_excluded,statistical", CHAIN_CASES)
def test_chain_configs_lane_by_lane(torch_cuda, name, frame, narrow, mode, max_excluded, statistical, build, seed):
    torch = torch_cuda
    xml, model, problem, data, q0, T, tasks = _chain_case(torch, name, frame, narrow, mode, build, seed)
    q = _compare(torch, model, problem, data, tasks, q0, T, max_excluded, "%s %s narrow=%s %s [%s] seed %d" % (name, frame, narrow, mode, build, seed), statistical)
    lo, hi = model.lowerPositionLimit, model.upperPositionLimit
    assert (q >= lo - 1e-15).all() and (q <= hi + 1e-15).all()
    if narrow:
        assert (np.abs(np.abs(q) - narrow) < 1e-15).any()          # the projection does bind


def _full_body(torch, mode, seed):
    import ik_amd
    import oracle as O
    from ik_amd import workload
    model = ik_amd.Model.from_urdf_file(urdf_path("cassie"), free_flyer=True)
    problem = ik_amd.InverseKinematicsProblem(model)
    frames = ["LeftFootFront", "RightFootFront", "pelvis"]
    for i, f in enumerate(frames):
        problem.add_frame_task("t%d" % i, ik_amd.FrameTask.create(model, f, ik_amd.KinematicType.Full))
    data = ik_amd.dls_data(problem, device=0)
    assert data.kernel == "dls_tree<NJ=7,chains=2,base_task>"
    q0, qs = workload.freeflyer_workload(model.lowerPositionLimit, model.upperPositionLimit, workload.cassie_nominal(model.names),
                                         np.arange(B), seed=seed, mode=mode)
    T = ik_amd.task_frames_fk_batch(problem, torch.from_numpy(np.ascontiguousarray(qs.T)).cuda(), data)
    tasks = O.make_tasks([(model.getFrameId(f), 0, 2, 0, None) for f in frames])
    return model, problem, data, q0, T, tasks


# Config 3's two target distributions: "near" (q* = q0 + U(+-0.15): every lane converges) and "uniform" (joints anywhere
# in their limits, base moved by U(+-0.1) m / U(+-0.2) rad: the primary distribution; about 6 % of the lanes never converge, and those
# are where the 50-step map amplifies rounding differences -- the chaotic regime of rule 3, as for the narrowed UR arms).
FULL_BODY_CASES = [("near", 0, False), ("uniform", 0.12, True)]


@pytest.mark.parametrize("seed", SEEDS)
@pytest.mark.parametrize("mode,max_excluded,statistical", FULL_BODY_CASES)
def test_full_body_lane_by_lane(torch_cuda, mode, max_excluded, statistical, seed):
    """Config 3: Cassie full body (free-flyer, nq = 23), SE(3) tasks on both feet and the pelvis; integrate is the SE(3) update of
    the floating base (reference ik/ik/dls.cpp:67-68)."""
    torch = torch_cuda
    model, problem, data, q0, T, tasks = _full_body(torch, mode, seed)
    q = _compare(torch, model, problem, data, tasks, q0, T, max_excluded, "cassie full body %s seed %d" % (mode, seed), statistical)
    assert np.abs(np.linalg.norm(q[:, 3:7], axis=1) - 1.0).max() < 1e-9
    lo, hi = model.lowerPositionLimit, model.upperPositionLimit
    assert (q[:, 7:] >= lo[7:] - 1e-15).all() and (q[:, 7:] <= hi[7:] + 1e-15).all()


@pytest.mark.parametrize("seed", SEEDS)
@pytest.mark.parametrize("mode", ["near", "uniform"])
def test_full_body_step_synchronised_along_the_oracle_trajectory(torch_cuda, mode, seed):
    """All 65536 lanes, 20 steps along the ORACLE's trajectory, no exclusions: one full step of the tree kernel -- both chains, the base
    task, the arrow solve, exp6 of the base twist and the quaternion update, the clamp -- equals the oracle's dense 18 x 18 dual solve
    and pinocchio-style integrate to 1e-9 rad / m (quaternion entries: 1e-9); the distribution check against _Float128 as for the chains."""
    torch = torch_cuda
    model, problem, data, q0, T, tasks = _full_body(torch, mode, seed)
    _step_synchronised(torch, model, problem, data, tasks, q0, T, 20, 1e-9, "cassie full body %s seed %d" % (mode, seed))


def test_full_body_never_stop_build_equals_the_stop_capable_build(torch_cuda):
    """The never-stop visitor has its own instantiation of the hot tree kernel (no stop test, no `active` selects); the arithmetic
    of a step is the same code, so the stop-capable build run with the same visitor (IKGPU_TREE_NEVER_OFF) returns the same bits."""
    torch = torch_cuda
    import ik_amd
    from ik_amd import workload
    model = ik_amd.Model.from_urdf_file(urdf_path("cassie"), free_flyer=True)
    problem = ik_amd.InverseKinematicsProblem(model)
    for i, f in enumerate(["LeftFootFront", "RightFootFront", "pelvis"]):
        problem.add_frame_task("t%d" % i, ik_amd.FrameTask.create(model, f, ik_amd.KinematicType.Full))
    data = ik_amd.dls_data(problem, device=0)
    n = 20000 + 37
    q0, qs = workload.freeflyer_workload(model.lowerPositionLimit, model.upperPositionLimit, workload.cassie_nominal(model.names),
                                         np.arange(n), seed=3, mode="uniform")
    Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
    T = ik_amd.task_frames_fk_batch(problem, torch.from_numpy(np.ascontiguousarray(qs.T)).cuda(), data)
    prm = ik_amd.dls_parameters(max_iterations=25)
    a = [t.clone() for t in ik_amd.dls_batch(problem, Q0, T, data, ik_amd.never_stop_visitor(), prm)]
    os.environ["IKGPU_TREE_NEVER_OFF"] = "1"
    try:
        b = ik_amd.dls_batch(problem, Q0, T, data, ik_amd.never_stop_visitor(), prm)
        torch.cuda.synchronize()
    finally:
        os.environ.pop("IKGPU_TREE_NEVER_OFF", None)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert int(a[2].min()) == 25 and int(a[1].sum()) == 0
