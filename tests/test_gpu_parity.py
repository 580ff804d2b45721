"""GPU parity tests: the gfx950 kernels, called through the C ABI (ik_amd.api -> libikgpu.so), against
the CPU oracle on identical seeded inputs.  Bars (BASELINE.json north_star): joint solutions within
1e-6 rad of the CPU solver on identical targets; success / iteration flags equal.

Far-from-solution UR5 problems make the DLS iteration chaotic (a 1e-13 perturbation grows ~10x every
3 iterations -- the oracle and the independent numpy twin diverge from each other in exactly the same
way), so for that distribution parity is asserted step-wise (1 and 3 iterations from identical q) and
statistically, and trajectory-wise only on the near-target distribution.
"""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import urdf_path

pytestmark = pytest.mark.gpu

TOL = 1e-6  # rad, north_star


@pytest.fixture(scope="module")
def torch_cuda(native_built):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def _setup(name, frame, ktype=None, weights=None):
    import ik_amd
    import oracle as O
    model = ik_amd.Model.from_urdf_file(urdf_path(name))
    problem = ik_amd.InverseKinematicsProblem(model)
    kt = ik_amd.KinematicType.Full if ktype is None else ktype
    task = problem.add_frame_task("t", ik_amd.FrameTask.create(model, frame, kt))
    if weights is not None:
        task.weighting()[:] = weights
    data = ik_amd.dls_data(problem, device=0)
    om = O.OracleModel(model.flat())
    fid = model.getFrameId(frame)
    ot = O.make_tasks([(fid, 0, int(kt), 0, weights)])
    return ik_amd, O, model, problem, data, om, ot, fid


def _inputs(model, name, B, mode, seed=0, narrow=None):
    from ik_amd import workload
    nominal = workload.UR5_NOMINAL if name == "ur5" else workload.cassie_nominal(model.names)
    return workload.chain_workload(model.lowerPositionLimit, model.upperPositionLimit, nominal, np.arange(B), seed, mode,
                                   narrow=narrow)


def _to_dev(torch, a_aos):  # [B, n] host -> [n, B] device (SoA)
    return torch.from_numpy(np.ascontiguousarray(a_aos.T)).cuda()


CASES = [("cassie_fixed", "LeftFootFront"), ("ur5", "tool0")]


@pytest.mark.parametrize("name,frame", CASES)
def test_task_frame_fk_matches_oracle(torch_cuda, name, frame):
    torch = torch_cuda
    ik_amd, O, model, problem, data, om, ot, fid = _setup(name, frame)
    _, qs = _inputs(model, name, 1000, "uniform")
    got = ik_amd.task_frames_fk_batch(problem, _to_dev(torch, qs), data).permute(2, 0, 1).cpu().numpy()
    want = O.fk_batch(om, qs, [fid])
    assert np.abs(got - want).max() < 1e-13


@pytest.mark.parametrize("name,frame", CASES)
def test_evaluate_matches_oracle(torch_cuda, name, frame):
    """Stage parity: e and the dense task Jacobian (reference ik/ik/data.cpp:25-58)."""
    torch = torch_cuda
    ik_amd, O, model, problem, data, om, ot, fid = _setup(name, frame)
    B = 300
    q0, qs = _inputs(model, name, B, "uniform", narrow=2.0 if name == "ur5" else None)
    tg = O.fk_batch(om, qs, [fid])
    T = torch.from_numpy(np.ascontiguousarray(tg.transpose(1, 2, 0))).cuda()
    e, J = ik_amd.evaluate_batch(problem, _to_dev(torch, q0), T, data)
    e, J = e.cpu().numpy().T, J.permute(2, 0, 1).cpu().numpy()
    for b in range(B):
        eo, Jo = O.evaluate(om, ot, tg[b], q0[b])
        assert np.abs(e[b] - eo).max() < 1e-10 and np.abs(J[b] - Jo).max() < 1e-10


@pytest.mark.parametrize("name,frame", CASES)
@pytest.mark.parametrize("mode,stop_tol,iters", [("near", -1.0, 50), ("near", 1e-4, 100), ("near", -1.0, 1)])
def test_dls_matches_oracle_near_targets(torch_cuda, name, frame, mode, stop_tol, iters):
    torch = torch_cuda
    ik_amd, O, model, problem, data, om, ot, fid = _setup(name, frame)
    B = 4096  # BASELINE.json configs[1]
    q0, qs = _inputs(model, name, B, mode)
    tg = O.fk_batch(om, qs, [fid])
    T = torch.from_numpy(np.ascontiguousarray(tg.transpose(1, 2, 0))).cuda()
    p = ik_amd.dls_parameters(max_iterations=iters)
    Q, ok, it = ik_amd.dls_batch(problem, _to_dev(torch, q0), T, data, ik_amd.inverse_kinematics_visitor(stop_tol), p)
    q_ref, ok_ref, it_ref = O.dls_batch(om, ot, tg, q0, O.params(iters, 1e-2, 1.0, stop_tol), os.cpu_count() or 1)
    assert np.array_equal(ok.cpu().numpy(), ok_ref)
    assert np.array_equal(it.cpu().numpy(), it_ref)
    assert np.abs(Q.cpu().numpy().T - q_ref).max() <= TOL


def test_dls_cassie_uniform_targets_fixed_50(torch_cuda):
    """The benchmark distribution (q* uniform in the joint limits); ~5 % of lanes stall on the clamp,
    they must still agree with the oracle."""
    torch = torch_cuda
    ik_amd, O, model, problem, data, om, ot, fid = _setup("cassie_fixed", "LeftFootFront")
    B = 4096
    q0, qs = _inputs(model, "cassie_fixed", B, "uniform")
    tg = O.fk_batch(om, qs, [fid])
    T = torch.from_numpy(np.ascontiguousarray(tg.transpose(1, 2, 0))).cuda()
    Q, ok, it = ik_amd.dls_batch(problem, _to_dev(torch, q0), T, data, ik_amd.never_stop_visitor(),
                                 ik_amd.dls_parameters(max_iterations=50))
    q_ref, ok_ref, it_ref = O.dls_batch(om, ot, tg, q0, O.params(50, 1e-2, 1.0, -1.0), os.cpu_count() or 1)
    assert np.abs(Q.cpu().numpy().T - q_ref).max() <= TOL
    assert not ok.any() and (it.cpu().numpy() == 50).all()


def test_dls_ur5_far_targets_stepwise_and_statistics(torch_cuda):
    torch = torch_cuda
    ik_amd, O, model, problem, data, om, ot, fid = _setup("ur5", "tool0")
    B = 2048
    q0, qs = _inputs(model, "ur5", B, "uniform", narrow=2.0)
    tg = O.fk_batch(om, qs, [fid])
    T = torch.from_numpy(np.ascontiguousarray(tg.transpose(1, 2, 0))).cuda()
    for iters, bar in ((1, 1e-9), (3, 1e-6)):
        Q, _, _ = ik_amd.dls_batch(problem, _to_dev(torch, q0), T, data, ik_amd.never_stop_visitor(),
                                   ik_amd.dls_parameters(max_iterations=iters))
        q_ref, _, _ = O.dls_batch(om, ot, tg, q0, O.params(iters, 1e-2, 1.0, -1.0))
        assert np.abs(Q.cpu().numpy().T - q_ref).max() <= bar
    Q, _, _ = ik_amd.dls_batch(problem, _to_dev(torch, q0), T, data, ik_amd.never_stop_visitor(),
                               ik_amd.dls_parameters(max_iterations=50))
    q_ref, _, _ = O.dls_batch(om, ot, tg, q0, O.params(50, 1e-2, 1.0, -1.0), os.cpu_count() or 1)
    d = np.abs(Q.cpu().numpy().T - q_ref).max(axis=1)
    # 50 iterations, lane by lane, by the rule tests/test_gpu_full_size.py states (and runs at B = 65536): probes of the ORACLE ALONE
    # decide which problems are held to the bar; the excluded ones are arbitrated by the same oracle in _Float128 arithmetic.
    from test_gpu_full_size import assert_parity, oracle_sensitivity, parity_counts
    prm = O.params(50, 1e-2, 1.0, -1.0)
    cores = os.cpu_count() or 1

    def solve(tg_, q0_, ext):
        return O.dls_batch(om, ot, tg_, q0_, prm, cores, ext=ext)
    sens = oracle_sensitivity(O, solve, tg, q0, q_ref)
    c = parity_counts(Q.cpu().numpy().T, q_ref, sens, lambda t_, q_: solve(t_, q_, "q"), tg, q0)
    print("ur5 far targets, B = %d: %s" % (B, c))
    assert c["stable"] > 0.5 * B
    assert_parity(c, "ur5 far targets", 0.5, statistical=True)
    # and the converged fraction is the same on both sides
    def converged(qm):
        res = O.fk_batch(om, qm, [fid])[:, 0, 9:] - tg[:, 0, 9:]
        return (np.linalg.norm(res, axis=1) < 1e-8).mean()
    assert abs(converged(Q.cpu().numpy().T) - converged(q_ref)) < 0.05


def test_layouts_tail_and_passthrough(torch_cuda):
    """AoS == SoA bit for bit; a batch that is not a multiple of 64; joints outside the support are only
    clamped (reference ik/ik/dls.cpp:71) -- and left untouched when the solve stops at iteration 0."""
    torch = torch_cuda
    ik_amd, O, model, problem, data, om, ot, fid = _setup("cassie_fixed", "LeftFootFront")
    B = 1000
    q0, qs = _inputs(model, "cassie_fixed", B, "near")
    q0[:, 8:] += 5.0 * (np.arange(B)[:, None] % 3 - 1)      # push right-leg entries outside their limits
    q0[:10, :8] = qs[:10, :8]                                # first ten problems start at the solution
    tg = O.fk_batch(om, qs, [fid])
    T = torch.from_numpy(np.ascontiguousarray(tg.transpose(1, 2, 0))).cuda()
    v, p = ik_amd.inverse_kinematics_visitor(), ik_amd.dls_parameters()
    Qs, oks, its = ik_amd.dls_batch(problem, _to_dev(torch, q0), T, data, v, p)
    Qa, oka, ita = ik_amd.dls_batch(problem, torch.from_numpy(q0).cuda(), torch.from_numpy(tg).cuda(), data, v, p, layout="aos")
    assert torch.equal(Qs.T.contiguous(), Qa) and torch.equal(oks, oka) and torch.equal(its, ita)
    q_ref, ok_ref, it_ref = O.dls_batch(om, ot, tg, q0, O.params())
    assert np.array_equal(its.cpu().numpy(), it_ref) and np.array_equal(oks.cpu().numpy(), ok_ref)
    assert np.abs(Qa.cpu().numpy() - q_ref).max() <= TOL
    assert (it_ref[:10] == 0).all() and np.array_equal(Qa.cpu().numpy()[:10, 8:], q0[:10, 8:])


@pytest.mark.parametrize("ktype,weights", [(0, None), (1, None), (2, [1.0, 2.0, 0.5, 1.5, 1.0, 3.0]), (0, [2.0, 1.0, 0.25])])
def test_kinematic_types_and_weights(torch_cuda, ktype, weights):
    torch = torch_cuda
    import ik_amd as ia
    ik_amd, O, model, problem, data, om, ot, fid = _setup("cassie_fixed", "LeftFootFront", ia.KinematicType(ktype), weights)
    B = 512
    q0, qs = _inputs(model, "cassie_fixed", B, "near")
    tg = O.fk_batch(om, qs, [fid])
    T = torch.from_numpy(np.ascontiguousarray(tg.transpose(1, 2, 0))).cuda()
    for iters in (1, 20):
        Q, ok, it = ik_amd.dls_batch(problem, _to_dev(torch, q0), T, data, ik_amd.never_stop_visitor(),
                                     ik_amd.dls_parameters(max_iterations=iters))
        q_ref, _, _ = O.dls_batch(om, ot, tg, q0, O.params(iters, 1e-2, 1.0, -1.0))
        assert np.abs(Q.cpu().numpy().T - q_ref).max() <= TOL


def test_single_problem_api_config1(torch_cuda):
    """BASELINE.json configs[0]: one Cassie left-leg problem through the reference-shaped API
    (ik::dls(problem, q0, data, visitor, params), reference ik/ik/dls.hpp:111-114), next to the CPU path."""
    ik_amd, O, model, problem, data, om, ot, fid = _setup("cassie_fixed", "LeftFootFront")
    q0, qs = _inputs(model, "cassie_fixed", 1, "uniform", seed=3)
    task = problem.get_frame_task("t")
    task.target = ik_amd.SE3.from12(O.fk_batch(om, qs, [fid])[0, 0])
    q = ik_amd.dls(problem, q0[0], data)
    q_ref, ok_ref, it_ref = O.dls(om, ot, task.target.to12()[None], q0[0], O.params())
    assert data.success == ok_ref and data.iterations == it_ref
    assert np.abs(q - q_ref).max() <= TOL


def test_full_size_properties(torch_cuda):
    """BASELINE.json's full batch (65536): size-independent properties instead of an oracle run.
    (1) round trip: FK(q_out) reaches the target on converged lanes; (2) idempotence: re-solving from
    q_out leaves q unchanged to 1e-8; (3) determinism: two runs are bit-identical;
    (4) every returned configuration is inside the joint limits."""
    torch = torch_cuda
    ik_amd, O, model, problem, data, om, ot, fid = _setup("cassie_fixed", "LeftFootFront")
    B = 65536
    q0, qs = _inputs(model, "cassie_fixed", B, "uniform")
    Q0, QS = _to_dev(torch, q0), _to_dev(torch, qs)
    T = ik_amd.task_frames_fk_batch(problem, QS, data)
    v, p = ik_amd.never_stop_visitor(), ik_amd.dls_parameters(max_iterations=50)
    Q1, _, _ = ik_amd.dls_batch(problem, Q0, T, data, v, p)
    Q2, _, _ = ik_amd.dls_batch(problem, Q0, T, data, v, p)
    assert torch.equal(Q1, Q2)
    reached = ik_amd.task_frames_fk_batch(problem, Q1, data)
    res = (reached - T).abs().amax(dim=(0, 1))
    conv = res < 1e-9
    assert conv.double().mean().item() > 0.90
    Q3, _, _ = ik_amd.dls_batch(problem, Q1, T, data, v, ik_amd.dls_parameters(max_iterations=5))
    assert (Q3 - Q1)[:, conv].abs().max().item() < 1e-8
    lo = torch.from_numpy(model.lowerPositionLimit).cuda()[:, None]
    hi = torch.from_numpy(model.upperPositionLimit).cuda()[:, None]
    assert bool(((Q1 >= lo) & (Q1 <= hi)).all())


def test_error_behaviour(torch_cuda):
    torch = torch_cuda
    import ik_amd
    from ik_amd import capi
    ia, O, model, problem, data, om, ot, fid = _setup("cassie_fixed", "LeftFootFront")
    Q0 = torch.zeros((16, 64), dtype=torch.float64, device="cuda")
    T = torch.zeros((1, 12, 64), dtype=torch.float64, device="cuda")
    with pytest.raises(capi.IkgpuError) as ei:
        ik_amd.dls_batch(problem, Q0, T, data, p=ik_amd.dls_parameters(damping=0.0))
    assert ei.value.code == capi.ERR_INVALID and "damping" in ei.value.message
    with pytest.raises(ValueError):
        ik_amd.dls_batch(problem, Q0[:15].contiguous(), T, data)
    with pytest.raises(capi.IkgpuError):
        ik_amd.dls_data(problem, device=99)
    # B = 0 is a no-op
    Q, ok, it = ik_amd.dls_batch(problem, Q0[:, :0].contiguous(), T[:, :, :0].contiguous(), data)
    assert Q.shape == (16, 0)


# ---------------------------------------------------------------------------------------------------
# Shape F (BASELINE.json configs[2]): Cassie full body, free-flyer base, nq = 23 / nv = 22,
# LeftFootFront + RightFootFront + pelvis SE(3) tasks (M = 18)
# ---------------------------------------------------------------------------------------------------
def _setup_full_body(types=(2, 2, 2), weights=(None, None, None), prios=(0, 0, 0),
                     frames=("LeftFootFront", "RightFootFront", "pelvis")):
    import ik_amd
    import oracle as O
    model = ik_amd.Model.from_urdf_file(urdf_path("cassie"), free_flyer=True)
    problem = ik_amd.InverseKinematicsProblem(model, max(prios))
    spec = []
    for i, f in enumerate(frames):
        t = problem.add_frame_task("t%d" % i, ik_amd.FrameTask.create(model, f, ik_amd.KinematicType(types[i])), prios[i])
        if weights[i] is not None:
            t.weighting()[:] = weights[i]
    for t, prio in problem.ordered_tasks():
        w = None if np.all(t.weighting() == 1) else list(t.weighting())
        spec.append((t._frame_id, 0, int(t.type), prio, w))
    data = ik_amd.dls_data(problem, device=0)
    om = O.OracleModel(model.flat())
    return ik_amd, O, model, problem, data, om, O.make_tasks(spec), [s[0] for s in spec]


def _full_body_inputs(model, B, mode="near", seed=0):
    from ik_amd import workload
    return workload.freeflyer_workload(model.lowerPositionLimit, model.upperPositionLimit,
                                       workload.cassie_nominal(model.names), np.arange(B), seed=seed, mode=mode)


def test_full_body_stages_match_oracle(torch_cuda):
    torch = torch_cuda
    ik_amd, O, model, problem, data, om, ot, fids = _setup_full_body()
    assert data.kernel == "dls_tree<NJ=7,chains=2,base_task>" and data.rows == 18
    B = 200
    q0, qs = _full_body_inputs(model, B)
    got = ik_amd.task_frames_fk_batch(problem, _to_dev(torch, qs), data).permute(2, 0, 1).cpu().numpy()
    tg = O.fk_batch(om, qs, fids)
    assert np.abs(got - tg).max() < 1e-13
    T = torch.from_numpy(np.ascontiguousarray(tg.transpose(1, 2, 0))).cuda()
    e, J = ik_amd.evaluate_batch(problem, _to_dev(torch, q0), T, data)
    e, J = e.cpu().numpy().T, J.permute(2, 0, 1).cpu().numpy()
    for b in range(B):
        eo, Jo = O.evaluate(om, ot, tg[b], q0[b])
        assert np.abs(e[b] - eo).max() < 1e-10 and np.abs(J[b] - Jo).max() < 1e-10


@pytest.mark.parametrize("mode,stop_tol,iters", [("near", -1.0, 50), ("near", 1e-4, 100), ("near", -1.0, 1), ("uniform", -1.0, 3)])
def test_full_body_dls_matches_oracle(torch_cuda, mode, stop_tol, iters):
    torch = torch_cuda
    ik_amd, O, model, problem, data, om, ot, fids = _setup_full_body()
    B = 2048
    q0, qs = _full_body_inputs(model, B, mode)
    tg = O.fk_batch(om, qs, fids)
    T = torch.from_numpy(np.ascontiguousarray(tg.transpose(1, 2, 0))).cuda()
    Q, ok, it = ik_amd.dls_batch(problem, _to_dev(torch, q0), T, data, ik_amd.inverse_kinematics_visitor(stop_tol),
                                 ik_amd.dls_parameters(max_iterations=iters))
    q_ref, ok_ref, it_ref = O.dls_batch(om, ot, tg, q0, O.params(iters, 1e-2, 1.0, stop_tol), os.cpu_count() or 1)
    assert np.array_equal(ok.cpu().numpy(), ok_ref) and np.array_equal(it.cpu().numpy(), it_ref)
    assert np.abs(Q.cpu().numpy().T - q_ref).max() <= TOL
    # AoS gives the same bits
    Qa, oka, ita = ik_amd.dls_batch(problem, torch.from_numpy(q0).cuda(), torch.from_numpy(tg).cuda(), data,
                                    ik_amd.inverse_kinematics_visitor(stop_tol), ik_amd.dls_parameters(max_iterations=iters), layout="aos")
    assert torch.equal(Q.T.contiguous(), Qa) and torch.equal(ok, oka) and torch.equal(it, ita)


def test_full_body_types_weights_priorities(torch_cuda):
    torch = torch_cuda
    ik_amd, O, model, problem, data, om, ot, fids = _setup_full_body(
        types=(0, 2, 1), weights=([2.0, 1.0, 0.5], [1, 1, 1, 0.3, 0.3, 0.3], None), prios=(0, 0, 1))
    B = 512
    q0, qs = _full_body_inputs(model, B)
    tg = O.fk_batch(om, qs, fids)
    T = torch.from_numpy(np.ascontiguousarray(tg.transpose(1, 2, 0))).cuda()
    for iters, tol in ((1, -1.0), (30, 1e-4)):
        Q, ok, it = ik_amd.dls_batch(problem, _to_dev(torch, q0), T, data, ik_amd.inverse_kinematics_visitor(tol),
                                     ik_amd.dls_parameters(max_iterations=iters))
        q_ref, ok_ref, it_ref = O.dls_batch(om, ot, tg, q0, O.params(iters, 1e-2, 1.0, tol))
        assert np.array_equal(ok.cpu().numpy(), ok_ref) and np.array_equal(it.cpu().numpy(), it_ref)
        assert np.abs(Q.cpu().numpy().T - q_ref).max() <= TOL


def test_full_body_full_size_properties(torch_cuda):
    """BASELINE.json configs[2] at its full batch (65536): round trip, idempotence, determinism, unit
    quaternions, joint limits."""
    torch = torch_cuda
    ik_amd, O, model, problem, data, om, ot, fids = _setup_full_body()
    B = 65536
    q0, qs = _full_body_inputs(model, B)
    Q0, QS = _to_dev(torch, q0), _to_dev(torch, qs)
    T = ik_amd.task_frames_fk_batch(problem, QS, data)
    v, p = ik_amd.never_stop_visitor(), ik_amd.dls_parameters(max_iterations=50)
    Q1, _, _ = ik_amd.dls_batch(problem, Q0, T, data, v, p)
    Q2, _, _ = ik_amd.dls_batch(problem, Q0, T, data, v, p)
    assert torch.equal(Q1, Q2)
    res = (ik_amd.task_frames_fk_batch(problem, Q1, data) - T).abs().amax(dim=(0, 1))
    conv = res < 1e-9
    assert conv.double().mean().item() > 0.95
    Q3, _, _ = ik_amd.dls_batch(problem, Q1, T, data, v, ik_amd.dls_parameters(max_iterations=5))
    assert (Q3 - Q1)[:, conv].abs().max().item() < 1e-8
    assert ((Q1[3:7] ** 2).sum(0).sqrt() - 1).abs().max().item() < 1e-12
    lo = torch.from_numpy(model.lowerPositionLimit).cuda()[7:, None]
    hi = torch.from_numpy(model.upperPositionLimit).cuda()[7:, None]
    assert bool(((Q1[7:] >= lo) & (Q1[7:] <= hi)).all())


# ---------------------------------------------------------------------------------------------------
# The C++ mirror of the reference API (ik_amd/csrc/host/ik/*.hpp) through a program written like the
# reference's own tests (tests/cpp/test_dls_api.cpp ~ reference ik/test/dls.cpp:10-76)
# ---------------------------------------------------------------------------------------------------
def _cpp_binary():
    import subprocess
    from conftest import ROOT
    src = os.path.join(ROOT, "tests", "cpp", "test_dls_api.cpp")
    exe = os.path.join(ROOT, "tests", "cpp", "test_dls_api")
    hdr = os.path.join(ROOT, "ik_amd", "csrc", "host", "ik", "ik_gpu.hpp")
    if not os.path.exists(exe) or max(os.path.getmtime(src), os.path.getmtime(hdr)) > os.path.getmtime(exe):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                               "-I" + os.path.join(ROOT, "ik_amd", "csrc", "host"), "-o", exe, src,
                               "-L" + os.path.join(ROOT, "ik_amd"), "-likgpu", "-Wl,-rpath," + os.path.join(ROOT, "ik_amd")])
    return exe


@pytest.mark.parametrize("name,ff,frames,types,prios", [
    ("ur5", False, ["ee_fixed_joint"], [2], [0]),
    ("cassie_fixed", False, ["LeftFootFront"], [0], [0]),
    ("cassie", True, ["LeftFootFront", "RightFootFront", "pelvis"], [2, 2, 2], [0, 0, 0]),
])
def test_cpp_api_program_matches_oracle(torch_cuda, name, ff, frames, types, prios):
    import json
    import subprocess
    import ik_amd
    import oracle as O
    from ik_amd import workload
    model = ik_amd.Model.from_urdf_file(urdf_path(name), free_flyer=ff)
    om = O.OracleModel(model.flat())
    fids = [model.getFrameId(f) for f in frames]
    if ff:
        q0, qs = _full_body_inputs(model, 4, seed=5)
    else:
        q0, qs = _inputs(model, name, 4, "near", seed=5)
    q0, tg = q0[3], O.fk_batch(om, qs, fids)[3]
    ot = O.make_tasks([(f, 0, t, p, None) for f, t, p in zip(fids, types, prios)])
    args = [_cpp_binary(), urdf_path(name), "1" if ff else "0", "30", "0.01", "1.0", "1e-4", str(len(frames))]
    for f, t, p, target in zip(frames, types, prios, tg):
        args += [f, str(t), str(p)] + ["%.17g" % x for x in target]
    args += ["%.17g" % x for x in q0]
    # a standalone C++ process uses the system ROCm runtime (no torch in that process)
    out = json.loads(subprocess.check_output(args, text=True))
    q1, ok1, it1 = O.dls(om, ot, tg, q0, O.params(30, 0.01, 1.0, 1e-4))
    q2, ok2, it2 = O.dls(om, ot, tg, q1, O.params(30, 0.01, 1.0, 1e-4))
    assert np.abs(np.array(out["q_first"]) - q1).max() <= TOL
    assert np.abs(np.array(out["q"]) - q2).max() <= TOL
    assert out["success"] == int(ok2) and out["iterations"] == it2
