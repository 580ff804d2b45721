"""Free-flyer problems with the floating base in ANY orientation.  Every other free-flyer workload of the suite starts the base within
0.02 of the identity quaternion (ik_amd/workload.py freeflyer_workload), so the rotation -> quaternion conversion inside the SE(3)
update (pinocchio::integrate for JointModelFreeFlyer: exp6, then Eigen's four-way branch on the trace and the largest diagonal entry,
SURVEY.md A.5; reference ik/ik/dls.cpp:67-69) only ever took its `trace > 0` branch on the device.  Here the base orientation is uniform
on SO(3) -- all four branches, both hemispheres -- on the tree kernel (hot and general builds), the static lane program and the per-lane
interpreter, one step and twenty, against the oracle."""
import os

import numpy as np
import pytest

from conftest import urdf_path

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda(native_built):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def _inputs(model, B, seed, turn):
    from ik_amd import workload
    rng = np.random.default_rng(seed)
    q0, _ = workload.freeflyer_workload(model.lowerPositionLimit, model.upperPositionLimit, workload.cassie_nominal(model.names), np.arange(B), seed=seed, mode="near")
    quat = rng.normal(size=(B, 4))
    quat /= np.linalg.norm(quat, axis=1)[:, None]
    # a share of the lanes ON the branch boundaries of the conversion: rotations by pi about a coordinate axis and about a diagonal
    # (w = 0: trace = -1; two equal diagonal entries), and the identity with either sign
    special = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1], [0, 0, 0, -1], [np.sqrt(.5), np.sqrt(.5), 0, 0],
                        [np.sqrt(.5), 0, np.sqrt(.5), 0], [0, np.sqrt(.5), np.sqrt(.5), 0], [.5, .5, .5, .5], [.5, .5, .5, -.5]], dtype=float)
    quat[:special.shape[0] * 8] = np.repeat(special, 8, axis=0)
    q0[:, 3:7] = quat
    q0[:, :3] += rng.uniform(-0.5, 0.5, (B, 3))
    v = np.zeros((B, model.nv))
    v[:, :3] = rng.uniform(-0.1, 0.1, (B, 3))
    v[:, 3:6] = rng.uniform(-turn, turn, (B, 3))
    v[:, 6:] = rng.uniform(-0.15, 0.15, (B, model.nv - 6))
    qs = workload.freeflyer_integrate_batch(q0, v)
    qs[:, 7:] = np.clip(qs[:, 7:], model.lowerPositionLimit[7:], model.upperPositionLimit[7:])
    return q0, qs


CASES = {
    "full_body_tree_hot": (["LeftFootFront", "RightFootFront", "pelvis"], [2, 2, 2], {}),
    "foot_and_pelvis_tree_general": (["LeftFootFront", "pelvis"], [0, 2], {}),
    "full_body_static_program": (["LeftFootFront", "RightFootFront", "pelvis"], [2, 2, 2], {"IKGPU_DLS_KERNEL": "generic"}),
    "full_body_interpreter": (["LeftFootFront", "RightFootFront", "pelvis"], [2, 2, 2], {"IKGPU_DLS_KERNEL": "generic", "IKGPU_GENERIC_STATIC": "0"}),
}


@pytest.mark.parametrize("turn", [0.3, 3.0], ids=["near", "far"])   # rad: how far the targets are turned from the start pose (far: the first
@pytest.mark.parametrize("case", list(CASES))                        # steps turn the base by radians -- exp6 and the conversion at large angles)
def test_any_base_orientation_against_the_oracle(torch_cuda, case, turn):
    torch = torch_cuda
    import ik_amd
    import oracle as O
    frames, types, env = CASES[case]
    B = 4096 if "interpreter" not in case else 1024
    model = ik_amd.Model.from_urdf_file(urdf_path("cassie"), free_flyer=True)
    problem = ik_amd.InverseKinematicsProblem(model)
    for i, (f, t) in enumerate(zip(frames, types)):
        problem.add_frame_task("t%d" % i, ik_amd.FrameTask.create(model, f, ik_amd.KinematicType(t)))
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        data = ik_amd.dls_data(problem, device=0)
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    q0, qs = _inputs(model, B, seed=21, turn=turn)
    om = O.OracleModel(model.flat())
    fids = [model.getFrameId(f) for f in frames]
    tasks = O.make_tasks([(fid, 0, t, 0, None) for fid, t in zip(fids, types)])
    tg = O.fk_batch(om, qs, fids)
    T = torch.from_numpy(np.ascontiguousarray(tg.transpose(1, 2, 0))).cuda()
    Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
    cores = os.cpu_count() or 1
    worst = {}
    for iters, bar in ((1, 1e-9), (20, 1e-6)) if turn < 1.0 else ((1, 1e-9), (3, 1e-8)):
        Q, _, _ = ik_amd.dls_batch(problem, Q0, T, data, ik_amd.never_stop_visitor(), ik_amd.dls_parameters(max_iterations=iters))
        q_dev = Q.cpu().numpy().T
        q_ref, _, _ = O.dls_batch(om, tasks, tg, q0, O.params(iters, 1e-2, 1.0, -1.0), cores)
        assert np.isfinite(q_dev).all()
        d = np.abs(q_dev - q_ref).max(axis=1)
        worst[iters] = d.max()
        assert np.abs(np.linalg.norm(q_dev[:, 3:7], axis=1) - 1.0).max() < 1e-12
        assert d.max() < bar, (data.kernel, iters, d.max(), int(np.argmax(d)), q0[np.argmax(d), 3:7], q_dev[np.argmax(d), 3:7], q_ref[np.argmax(d), 3:7])
    # every branch of the conversion was taken by the oracle's results: w dominant / x / y / z dominant (Eigen: trace > 0, else the largest diagonal)
    qf = q_ref[:, 3:7]
    dom = np.argmax(np.abs(qf), axis=1)
    assert all((dom == k).sum() > B // 40 for k in range(4)), np.bincount(dom, minlength=4)
    print("%s %s [%s]: max |dq| vs oracle %s; dominant quaternion component counts %s" % (case, turn, data.kernel, {k: "%.1e" % v for k, v in worst.items()}, np.bincount(dom, minlength=4)))
