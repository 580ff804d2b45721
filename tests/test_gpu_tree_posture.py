"""PostureTask rows (reference ik/ik/posture.hpp:17-85) in the register-resident tree kernel (device/tree_solver.hpp, posture build):
rows on chain joints join the chain's normal equations, rows on the other joints are 1x1 systems stepped in the q_out column.
Through the C ABI against the CPU oracle (dense dual solve with every row stacked), 1e-6 rad; and against the generic kernel."""
import os

import numpy as np
import pytest

from test_gpu_generic import build

pytestmark = pytest.mark.gpu
TOL = 1e-6

CASES = {
    # the demo with the regulariser its source declares and leaves commented out (ik_ros/src/cassie.cpp:63-64,76): priority 1
    "demo_with_posture": ([("frame", "LeftFootFront", "pelvis", 0, 0, None), ("frame", "pelvis", "universe", 2, 0, None),
                           ("align", "LeftFootFront", "universe", 1, 0, None), ("posture", 16, None, None, 1, ([1.0] * 16, [1.0] * 16))],
                          "dls_tree<NJ=7,chains=1,base_task,base_reference,align_axis,posture>"),
    # both legs as chains, the rows at priority 0 (they enter the stop test), uneven weights, a mask with holes
    "two_chains_posture_in_the_stop_test": ([("frame", "LeftFootFront", "universe", 2, 0, None), ("frame", "RightFootFront", "universe", 2, 0, None),
                                             ("posture", 16, None, None, 0, ([0.05 + 0.02 * k for k in range(16)],
                                                                             [0.0 if k in (2, 7, 12) else 1.0 for k in range(16)]))],
                                            "dls_tree<NJ=7,chains=2,posture>"),
    "full_body_with_posture": ([("frame", "LeftFootFront", "universe", 2, 0, None), ("frame", "RightFootFront", "universe", 2, 0, None),
                                ("frame", "pelvis", "universe", 2, 0, None), ("posture", 16, None, None, 1, ([0.2] * 16, [1.0] * 16))],
                               "dls_tree<NJ=7,chains=2,base_task,posture>"),
    # a PostureTask over the last nine joints only
    "posture_on_a_suffix_of_the_joints": ([("frame", "LeftFootFront", "universe", 2, 0, None), ("posture", 9, None, None, 0, ([0.3] * 9, [1.0] * 9))],
                                          "dls_tree<NJ=7,chains=1,posture>"),
}


@pytest.fixture(scope="module")
def torch_cuda(native_built):
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


@pytest.mark.parametrize("case", sorted(CASES))
def test_tree_kernel_with_posture_rows_matches_oracle(torch_cuda, case, monkeypatch):
    torch = torch_cuda
    monkeypatch.delenv("IKGPU_DLS_KERNEL", raising=False)
    specs, kernel = CASES[case]
    B = 700  # not a multiple of the 128-lane workgroup: tail lanes shadow the last problem and must not touch its q_out column
    ik_amd, O, model, problem, data, om, ot, q0, tg = build("cassie", True, specs, B, seed=13)
    assert data.kernel == kernel
    q0[::5, 15] += 3.0      # a joint outside every chain, beyond its limit: clamped by the first step that is taken
    Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
    T = torch.from_numpy(np.ascontiguousarray(tg.transpose(1, 2, 0))).cuda()
    chaotic = any(s[0] == "align" for s in specs)   # a direction the foot cannot reach, at full step (test_gpu_generic.py)
    for iters, damping, step, tol in ((0, 1e-2, 1.0, 1e-4), (1, 1e-2, 1.0, -1.0), (3, 1e-2, 1.0, -1.0), (200, 1e-1, 1e-1, 1e-4), (40, 1e-2, 1.0, 1e-6)):
        p = ik_amd.dls_parameters(max_iterations=iters, damping=damping, step_length=step)
        Q, ok, it = ik_amd.dls_batch(problem, Q0, T, data, ik_amd.inverse_kinematics_visitor(tol), p)
        q_ref, ok_ref, it_ref = O.dls_batch(om, ot, tg, q0, O.params(iters, damping, step, tol), os.cpu_count() or 1)
        d = np.abs(Q.cpu().numpy().T - q_ref).max(axis=1)
        if chaotic and step == 1.0 and iters > 3:
            assert (d <= TOL).mean() > 0.7, (case, iters, (d <= TOL).mean())
            continue
        assert np.array_equal(ok.cpu().numpy(), ok_ref) and np.array_equal(it.cpu().numpy(), it_ref), (case, iters)
        assert d.max() <= TOL, (case, iters, d.max())
    # run-to-run bit-identical (the outside joints are read-modify-written in HBM by their own lane only), AoS the same bits
    Q2, _, _ = ik_amd.dls_batch(problem, Q0, T, data, ik_amd.inverse_kinematics_visitor(tol), p)
    assert torch.equal(Q, Q2)
    Qa, oka, ita = ik_amd.dls_batch(problem, torch.from_numpy(q0).cuda(), torch.from_numpy(tg).cuda(), data,
                                    ik_amd.inverse_kinematics_visitor(tol), p, layout="aos")
    assert torch.equal(Q.T.contiguous(), Qa) and torch.equal(ok, oka) and torch.equal(it, ita)
    # in place (q_out aliasing q0) gives the same result
    Qi = Q0.clone()
    ik_amd.dls_batch(problem, Qi, T, data, ik_amd.inverse_kinematics_visitor(tol), p,
                     out=(Qi, torch.empty(B, dtype=torch.uint8, device="cuda"), torch.empty(B, dtype=torch.int32, device="cuda")))
    assert torch.equal(Qi, Q)
    # the generic kernel on the same problem (IKGPU_DLS_KERNEL=generic) agrees on the small-step configuration
    p = ik_amd.dls_parameters(max_iterations=200, damping=1e-1, step_length=1e-1)
    Qt, okt, itt = ik_amd.dls_batch(problem, Q0, T, data, ik_amd.inverse_kinematics_visitor(1e-4), p)
    monkeypatch.setenv("IKGPU_DLS_KERNEL", "generic")
    gen = ik_amd.dls_data(problem, device=0)
    assert gen.kernel.startswith("dls_generic<")
    Qg, okg, itg = ik_amd.dls_batch(problem, Q0, T, gen, ik_amd.inverse_kinematics_visitor(1e-4), p)
    assert torch.equal(okt, okg) and torch.equal(itt, itg) and (Qt - Qg).abs().max().item() < 1e-8


def test_demo_with_posture_at_full_size(torch_cuda, monkeypatch):
    """B = 65536: with reachable pose targets and a light regulariser at priority 1 the pose errors shrink as without it, and
    the joints outside the leg move towards the posture target instead of staying put."""
    torch = torch_cuda
    monkeypatch.delenv("IKGPU_DLS_KERNEL", raising=False)
    import ik_amd
    from conftest import urdf_path
    from ik_amd import workload
    model = ik_amd.Model.from_urdf_file(urdf_path("cassie"), free_flyer=True)
    problem = ik_amd.InverseKinematicsProblem(model, 1)
    problem.add_frame_task("fl", ik_amd.FrameTask.create(model, "LeftFootFront", ik_amd.KinematicType.Full))
    problem.add_frame_task("pelvis", ik_amd.FrameTask.create(model, "pelvis", ik_amd.KinematicType.Full))
    posture = problem.add_posture_task("posture", ik_amd.PostureTask.create(model, 16), 1)
    posture.weighting()[:] = 0.05
    data = ik_amd.dls_data(problem, device=0)
    assert data.kernel == "dls_tree<NJ=7,chains=1,base_task,posture>"
    B = 65536
    nom = workload.cassie_nominal(model.names)
    q0, qs = workload.freeflyer_workload(model.lowerPositionLimit, model.upperPositionLimit, nom, np.arange(B), seed=0)
    Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
    QS = torch.from_numpy(np.ascontiguousarray(qs.T)).cuda()
    T = torch.zeros((2 + 16, 12, B), dtype=torch.float64, device="cuda")
    T[:2] = ik_amd.task_frames_fk_batch(problem, QS, data)[:2]
    T[2:, 9, :] = QS[7:]                                   # posture target = the generating configuration: everything is reachable
    Q, ok, it = ik_amd.dls_batch(problem, Q0, T, data, ik_amd.never_stop_visitor(), ik_amd.dls_parameters(max_iterations=50))
    reached = ik_amd.task_frames_fk_batch(problem, Q, data)[:2]
    assert (reached - T[:2]).abs().max().item() < 1e-6
    outside = [i for i in range(7, 23) if i not in (7, 8, 9, 10, 11, 12, 14)]      # q entries off the left-leg chain
    before = (Q0[outside] - QS[outside]).abs().max().item()
    after = (Q[outside] - QS[outside]).abs().max().item()
    assert after < 0.5 * before and torch.isfinite(Q).all()
