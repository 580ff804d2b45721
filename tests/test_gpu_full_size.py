"""Full-size (B = 65536) lane-by-lane parity of BASELINE.json's configs 2/4 (Cassie leg), 3 (Cassie full body) and 5 (UR5 and UR10,
stock limits and limits narrowed to +-2 rad so the joint-limit projection binds) against the CPU oracle, through the C ABI, at the
metric's 50 fixed iterations.  Flags and iteration counts must be equal on every problem; |q_gpu - q_oracle| <= 1e-6 rad
(BASELINE.json north_star) on every STABLE problem.

Stability rule (it replaces round 1's "converged on the CPU" split): a problem is unstable when the oracle's OWN answer moves by
more than 1e-7 rad under a 1e-13 perturbation of its inputs -- q0 + 1e-13, target translations + 1e-13, and both with the
opposite sign (the target perturbation matters: a first step that lands every joint on a limit erases a perturbation of q0,
while the target enters every iteration).  A lane stalled on a joint limit or far from a reachable pose amplifies rounding
differences, so neither side has an answer good to 1e-6 there; such lanes are counted and bounded separately.  Every other
problem, converged or not, is held to the bar.  On chain problems a fourth probe joins the three: the optimised CPU variant
(oracle/fast_cpu.cpp -- the same algorithm with a different order of arithmetic) against the faithful port; two CPU
restatements that disagree with each other mark exactly the lanes where rounding decides.

A perturbation test cannot prove stability (with the joint-limit projection live and targets far away the iteration is chaotic: a
few lanes in 65536 pass every probe and still differ), so the END-TO-END assertion on the clamp workloads tolerates a 1e-4
fraction of escapes and prints their number -- and the proof that those are chaos, not error, is the second test:
`test_step_synchronised_along_the_oracle_trajectory` feeds the device the ORACLE's iterate at each of the 50 steps and
demands the next iterate to 1e-9 rad on ALL 65536 lanes, no exclusions: every step of every problem agrees, only the
composition of 50 steps amplifies.  The oracle runs on all host cores (a few seconds per case)."""
import os
import re

import numpy as np
import pytest

from conftest import urdf_path

pytestmark = pytest.mark.gpu

TOL = 1e-6           # rad, north_star
B = 65536
ITERS = 50


@pytest.fixture(scope="module")
def torch_cuda(native_built):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def oracle_sensitivity(O, om, tasks, tg, q0, prm, cores, q_ref):
    """max |q - q_ref| of the oracle's own answers under three 1e-13 perturbations of the inputs."""
    sens = np.zeros(q0.shape[0])
    for dq, dt in ((1e-13, 0.0), (0.0, 1e-13), (-1e-13, -1e-13)):
        tgp = tg.copy()
        tgp[:, :, 9:] += dt
        qp, _, _ = O.dls_batch(om, tasks, tgp, q0 + dq, prm, cores)
        sens = np.maximum(sens, np.abs(qp - q_ref).max(axis=1))
    return sens


def _compare(torch, model, problem, data, tasks, q0, targets_dev, max_unstable_frac, label, fast=None, escapes=0.0):
    import ik_amd
    import oracle as O
    om = O.OracleModel(model.flat())
    Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
    Q, ok, it = ik_amd.dls_batch(problem, Q0, targets_dev, data, ik_amd.never_stop_visitor(), ik_amd.dls_parameters(max_iterations=ITERS))
    q_gpu = Q.cpu().numpy().T
    tg = targets_dev.permute(2, 0, 1).contiguous().cpu().numpy()
    prm = O.params(ITERS, 1e-2, 1.0, -1.0)
    cores = os.cpu_count() or 1
    q_ref, ok_ref, it_ref = O.dls_batch(om, tasks, tg, q0, prm, cores)
    sens = oracle_sensitivity(O, om, tasks, tg, q0, prm, cores, q_ref)
    if fast is not None:    # (urdf xml, frame id): the optimised CPU variant as a fourth probe
        q_fast, _, _ = O.fast_dls_chain_batch(fast[0], fast[1], tg, q0, prm, cores)
        sens = np.maximum(sens, np.abs(q_fast - q_ref).max(axis=1))
    assert np.array_equal(ok.cpu().numpy(), ok_ref) and np.array_equal(it.cpu().numpy(), it_ref), label
    stable = sens <= 1e-7
    d = np.abs(q_gpu - q_ref).max(axis=1)
    print("%s: kernel %s, %d problems, %d stable (max |dq| %.3e), %d unstable (max |dq| %.3e, max self-sensitivity %.3e)"
          % (label, data.kernel, d.size, stable.sum(), d[stable].max(), (~stable).sum(), d[~stable].max() if (~stable).any() else 0.0, sens.max()))
    assert (~stable).mean() <= max_unstable_frac, (label, (~stable).mean())
    bad = stable & (d > TOL)
    print("%s: %d stable lanes beyond the bar (allowed: %d)" % (label, bad.sum(), int(escapes * d.size)))
    assert bad.sum() <= int(escapes * d.size), (label, np.flatnonzero(bad)[:8], d[bad][:8])
    return q_gpu


CHAIN_CASES = [
    # name, frame, narrowed limits, target distribution, allowed unstable fraction, allowed escapes among the stable lanes
    ("cassie_fixed", "LeftFootFront", None, "uniform", 0.01, 0.0),   # the bench workload of configs 2 / 4 (~6 % stall on a limit)
    ("cassie_fixed", "LeftFootFront", None, "near", 0.0, 0.0),
    ("ur5", "tool0", None, "near", 0.0, 0.0),                         # config 5, stock limits
    ("ur10", "tool0", None, "near", 0.0, 0.0),
    ("ur5", "tool0", 2.0, "uniform", 0.6, 1e-4),                      # config 5 with the projection live: far targets are chaotic
    ("ur10", "tool0", 2.0, "uniform", 0.6, 1e-4),
]


def _chain_case(torch, name, frame, narrow, mode):
    import ik_amd
    import oracle as O
    from ik_amd import workload
    xml = open(urdf_path(name)).read()
    if narrow:
        xml = re.sub(r'lower="[-0-9.e]+" upper="[-0-9.e]+"', 'lower="-%.1f" upper="%.1f"' % (narrow, narrow), xml)
    model = ik_amd.Model.from_urdf_xml(xml)
    problem = ik_amd.InverseKinematicsProblem(model)
    problem.add_frame_task("t", ik_amd.FrameTask.create(model, frame, ik_amd.KinematicType.Full))
    data = ik_amd.dls_data(problem, device=0)
    nominal = workload.UR5_NOMINAL if name.startswith("ur") else workload.cassie_nominal(model.names)
    q0, qs = workload.chain_workload(model.lowerPositionLimit, model.upperPositionLimit, nominal, np.arange(B), 0, mode)
    T = ik_amd.task_frames_fk_batch(problem, torch.from_numpy(np.ascontiguousarray(qs.T)).cuda(), data)
    tasks = O.make_tasks([(model.getFrameId(frame), 0, 2, 0, None)])
    return xml, model, problem, data, q0, T, tasks


@pytest.mark.parametrize("name,frame,narrow,mode", [c[:4] for c in CHAIN_CASES])
def test_step_synchronised_along_the_oracle_trajectory(torch_cuda, name, frame, narrow, mode):
    """All 65536 lanes, all 50 steps, no exclusions: from the oracle's k-th iterate the device's next iterate equals the
    oracle's to 1e-9 rad (one DLS step: evaluate, solve, integrate, project onto the limits -- reference ik/ik/dls.cpp:14-71)."""
    torch = torch_cuda
    import ik_amd
    import oracle as O
    xml, model, problem, data, q0, T, tasks = _chain_case(torch, name, frame, narrow, mode)
    om = O.OracleModel(model.flat())
    tg = T.permute(2, 0, 1).contiguous().cpu().numpy()
    one = O.params(1, 1e-2, 1.0, -1.0)
    p1 = ik_amd.dls_parameters(max_iterations=1)
    cores = os.cpu_count() or 1
    q, worst = q0, 0.0
    out = None
    for k in range(ITERS):
        q_next, _, _ = O.dls_batch(om, tasks, tg, q, one, cores)
        out = ik_amd.dls_batch(problem, torch.from_numpy(np.ascontiguousarray(q.T)).cuda(), T, data, ik_amd.never_stop_visitor(), p1, out=out)
        d = np.abs(out[0].cpu().numpy().T - q_next).max()
        worst = max(worst, d)
        assert d <= 1e-9, (name, mode, narrow, k, d)
        q = q_next
    print("%s %s narrow=%s: worst one-step |dq| over %d steps x %d lanes: %.3e rad" % (name, mode, narrow, ITERS, B, worst))


@pytest.mark.parametrize("name,frame,narrow,mode,max_unstable,escapes", CHAIN_CASES)
def test_chain_configs_lane_by_lane(torch_cuda, name, frame, narrow, mode, max_unstable, escapes):
    torch = torch_cuda
    xml, model, problem, data, q0, T, tasks = _chain_case(torch, name, frame, narrow, mode)
    q = _compare(torch, model, problem, data, tasks, q0, T, max_unstable, "%s %s narrow=%s %s" % (name, frame, narrow, mode),
                 fast=(xml, model.getFrameId(frame)), escapes=escapes)
    lo, hi = model.lowerPositionLimit, model.upperPositionLimit
    assert (q >= lo - 1e-15).all() and (q <= hi + 1e-15).all()
    if narrow:
        assert (np.abs(np.abs(q) - narrow) < 1e-15).any()          # the projection does bind


def test_full_body_lane_by_lane(torch_cuda):
    """Config 3: Cassie full body (free-flyer, nq = 23), SE(3) tasks on both feet and the pelvis."""
    torch = torch_cuda
    import ik_amd
    import oracle as O
    from ik_amd import workload
    model = ik_amd.Model.from_urdf_file(urdf_path("cassie"), free_flyer=True)
    problem = ik_amd.InverseKinematicsProblem(model)
    frames = ["LeftFootFront", "RightFootFront", "pelvis"]
    for i, f in enumerate(frames):
        problem.add_frame_task("t%d" % i, ik_amd.FrameTask.create(model, f, ik_amd.KinematicType.Full))
    data = ik_amd.dls_data(problem, device=0)
    q0, qs = workload.freeflyer_workload(model.lowerPositionLimit, model.upperPositionLimit, workload.cassie_nominal(model.names),
                                         np.arange(B), seed=0, mode="near")
    T = ik_amd.task_frames_fk_batch(problem, torch.from_numpy(np.ascontiguousarray(qs.T)).cuda(), data)
    tasks = O.make_tasks([(model.getFrameId(f), 0, 2, 0, None) for f in frames])
    q = _compare(torch, model, problem, data, tasks, q0, T, 0.0, "cassie full body near")
    assert np.abs(np.linalg.norm(q[:, 3:7], axis=1) - 1.0).max() < 1e-9
