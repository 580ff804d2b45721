"""The derived-visitor family of ikgpu_dls_params: inverse_kinematics_visitor::should_stop(ik, e, dq) is virtual and is handed every
level's error and the step (reference ik/ik/visitor.hpp:15-21, called at ik/ik/dls.cpp:61-64); the ABI carries the closed family
"||e[l]||^2 < tol_l on every listed level, OR ||dq||^2 < step tolerance".  Device against the oracle's restatement of the same
family (oracle/ik_oracle.c visitor_should_stop): iteration counts, flags and q."""
import os

import numpy as np
import pytest

from conftest import urdf_path

pytestmark = pytest.mark.gpu
TOL = 1e-6


@pytest.fixture(scope="module")
def torch_cuda(native_built):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def _leg(torch, B):
    import ik_amd
    import oracle as O
    from ik_amd import workload
    model = ik_amd.Model.from_urdf_file(urdf_path("cassie_fixed"))
    problem = ik_amd.InverseKinematicsProblem(model)
    problem.add_frame_task("t", ik_amd.FrameTask.create(model, "LeftFootFront", ik_amd.KinematicType.Full))
    data = ik_amd.dls_data(problem, device=0)
    q0, qs = workload.chain_workload(model.lowerPositionLimit, model.upperPositionLimit, workload.cassie_nominal(model.names), np.arange(B), 0, "near")
    om = O.OracleModel(model.flat())
    fid = model.getFrameId("LeftFootFront")
    tg = O.fk_batch(om, qs, [fid])
    return ik_amd, O, model, problem, data, om, O.make_tasks([(fid, 0, 2, 0, None)]), q0, tg


def _two_levels(torch, B):
    """The demo's task set over two priority levels (pose tasks, then the alignment row): ik::dls stacks the levels, the visitor
    sees them separately."""
    from test_gpu_generic import build
    specs = [("frame", "LeftFootFront", "pelvis", 0, 0, None), ("frame", "pelvis", "universe", 2, 0, None), ("align", "LeftFootFront", "universe", 1, 1, None)]
    os.environ["IKGPU_TREE_STATIC_ROWS"] = "12"
    try:
        return build("cassie", True, specs, B, seed=2)
    finally:
        os.environ["IKGPU_TREE_STATIC_ROWS"] = "0"


@pytest.mark.parametrize("which", ["leg_chain", "demo_two_levels"])
@pytest.mark.parametrize("visitor", [dict(step_tolerance=1e-3), dict(level_tolerances=(1e-5, 1e-3)), dict(tolerance=-1.0, step_tolerance=1e-5),
                                     dict(level_tolerances=(1e-6,), step_tolerance=1e-7)])
def test_derived_visitors_match_the_oracle(torch_cuda, which, visitor):
    torch = torch_cuda
    B = 1500
    ik, O, model, problem, data, om, ot, q0, tg = (_leg if which == "leg_chain" else _two_levels)(torch, B)
    Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
    T = torch.from_numpy(np.ascontiguousarray(tg.transpose(1, 2, 0))).cuda()
    v = ik.inverse_kinematics_visitor(**visitor)
    p = ik.dls_parameters(max_iterations=60, damping=1e-1, step_length=0.5)
    Q, ok, it = ik.dls_batch(problem, Q0, T, data, v, p)
    plain = ik.dls_batch(problem, Q0, T, data, ik.inverse_kinematics_visitor(v.tolerance), p)
    O.set_visitor(dq_sq_tol=v.step_tolerance if v.step_tolerance > 0 else -1.0, level_sq_tol=v.level_tolerances)
    try:
        q_ref, ok_ref, it_ref = O.dls_batch(om, ot, tg, q0, O.params(60, 1e-1, 0.5, v.tolerance), os.cpu_count() or 1)
    finally:
        O.set_visitor()
    it_gpu = it.cpu().numpy()
    same = it_gpu == it_ref
    assert same.mean() > 0.995, (which, visitor, same.mean())                       # (a test within rounding of its tolerance may flip)
    assert np.array_equal(ok.cpu().numpy()[same], ok_ref[same])
    d = np.abs(Q.cpu().numpy().T - q_ref).max(axis=1)
    assert (d[same] <= TOL).mean() >= 0.998 and np.median(d) < 1e-12, (which, visitor, d.max())   # (all but the odd ill-conditioned problem)
    _, _, it_plain_ref = O.dls_batch(om, ot, tg, q0, O.params(60, 1e-1, 0.5, v.tolerance), os.cpu_count() or 1)
    assert (it_ref != it_plain_ref).any(), "this visitor should change when problems stop"
    assert (it_gpu != plain[2].cpu().numpy()).any()
    assert 0 < ok_ref.mean() <= 1.0
    print("%s %s: kernel %s, success %.3f, mean iterations %.1f (reference's visitor: %.1f)" % (which, visitor, data.kernel, ok_ref.mean(), it_ref.mean(),
                                                                                        plain[2].double().mean().item()))
