"""BASELINE.json's config 5 names a UR10 arm; the reference ships a UR5 only (ik/test/ur5.urdf).  fixtures/models/ur10.kin.urdf
is authored from the public ur_description constants (fixtures/make_ur10_urdf.py, NOT a reference file).  Here: the
known-answer zero pose, oracle <-> independent twin agreement on the new model (CPU), and GPU <-> oracle parity with the joint
clamp live (targets within +-2 rad, limits narrowed to +-2 rad in a second model so the projection binds)."""
import os

import numpy as np
import pytest

from conftest import urdf_path

TOL = 1e-6  # rad, BASELINE.json north_star
NOMINAL = np.array([0.0, -np.pi / 2, np.pi / 2, 0.0, np.pi / 2, 0.0])


def _models():
    import oracle as O
    import twin as T
    import ik_amd
    model = ik_amd.Model.from_urdf_file(urdf_path("ur10"))
    return ik_amd, O, T, model, O.OracleModel(model.flat()), T.load_urdf(urdf_path("ur10"))


def test_ur10_fixture_is_reproducible(tmp_path):
    """The committed URDF is exactly what the generator writes."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    committed = open(urdf_path("ur10")).read()
    src = open(os.path.join(root, "fixtures", "make_ur10_urdf.py")).read().replace(
        'os.path.join(os.path.dirname(os.path.abspath(__file__)), "models", "ur10.kin.urdf")', repr(str(tmp_path / "ur10.kin.urdf")))
    subprocess.check_call([sys.executable, "-c", src])
    assert open(tmp_path / "ur10.kin.urdf").read() == committed


def test_ur10_zero_pose_known_answer(native_built):
    """tool0 at q = 0 from the DH constants: x = -a2 - a3, y = d4 + d6, z = d1 - d5 (host parser, oracle and twin)."""
    ik_amd, O, T, model, om, tm = _models()
    assert (model.nq, model.nv, model.njoints) == (6, 6, 7)
    assert model.names[1:] == ["shoulder_pan_joint", "shoulder_lift_joint", "elbow_joint", "wrist_1_joint", "wrist_2_joint", "wrist_3_joint"]
    want = np.array([0.612 + 0.5723, 0.163941 + 0.0922, 0.1273 - 0.1157])
    fid = model.getFrameId("tool0")
    got = O.fk(om, np.zeros(6))[1][fid]
    assert np.abs(got[9:] - want).max() < 1e-12
    M = T.fk(tm, np.zeros(6))[1][T.frame_id(tm, "tool0")]
    assert np.abs(M[:3, 3] - want).max() < 1e-12 and np.abs(M[:3, :3].reshape(-1) - got[:9]).max() < 1e-12
    # elbow limited to +-pi, the rest to +-2 pi
    assert np.allclose(model.upperPositionLimit, [2 * np.pi, 2 * np.pi, np.pi, 2 * np.pi, 2 * np.pi, 2 * np.pi], atol=1e-9)


def test_ur10_oracle_matches_twin(native_built):
    ik_amd, O, T, model, om, tm = _models()
    fid = model.getFrameId("tool0")
    ot = O.make_tasks([(fid, 0, 2, 0, None)])
    task = T.FrameTask(tm, "tool0", T.FULL, "universe")
    rng = np.random.default_rng(10)
    for k in range(12):
        q0 = NOMINAL + rng.uniform(-0.1, 0.1, 6)
        qs = q0 + rng.uniform(-0.15, 0.15, 6)
        tg = O.fk(om, qs)[1][[fid]]
        task.target = np.eye(4)
        task.target[:3, :3], task.target[:3, 3] = tg[0][:9].reshape(3, 3), tg[0][9:]
        q_o, ok_o, it_o = O.dls(om, ot, tg, q0, O.params(50, 1e-2, 1.0, 1e-12))
        q_t, ok_t, it_t = T.dls(tm, [task], q0, max_iterations=50, damping=1e-2, step_length=1.0, stop_sq_tol=1e-12)
        assert ok_o == ok_t and it_o == it_t and np.abs(q_o - q_t).max() < 1e-9
        assert np.abs(q_o - qs).max() < 1e-5          # near targets: converges to the generating configuration


def test_ur10_plans_onto_the_six_joint_chain_kernel(native_built):
    ik_amd, O, T, model, om, tm = _models()
    problem = ik_amd.InverseKinematicsProblem(model)
    problem.add_frame_task("t", ik_amd.FrameTask.create(model, "tool0", ik_amd.KinematicType.Full))
    assert ik_amd.plan(problem) == "dls_chain<NJ=6,full,hot>"


@pytest.mark.gpu
@pytest.mark.parametrize("narrow_limits", [False, True])
def test_ur10_dls_matches_oracle_with_the_clamp_live(native_built, narrow_limits):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import ik_amd
    import oracle as O
    from ik_amd import workload
    xml = open(urdf_path("ur10")).read()
    if narrow_limits:   # +-2 rad limits so that the projection onto the limits actually binds
        import re
        xml = re.sub(r'lower="[-0-9.e]+" upper="[-0-9.e]+"', 'lower="-2.0" upper="2.0"', xml)
    model = ik_amd.Model.from_urdf_xml(xml)
    problem = ik_amd.InverseKinematicsProblem(model)
    problem.add_frame_task("t", ik_amd.FrameTask.create(model, "tool0", ik_amd.KinematicType.Full))
    data = ik_amd.dls_data(problem, device=0)
    assert data.kernel == "dls_chain<NJ=6,full,hot>"
    om = O.OracleModel(model.flat())
    fid = model.getFrameId("tool0")
    ot = O.make_tasks([(fid, 0, 2, 0, None)])
    B = 4096
    for mode, iters, tol in (("near", 50, -1.0), ("near", 100, 1e-4), ("uniform", 1, -1.0), ("uniform", 3, -1.0)):
        q0, qs = workload.chain_workload(model.lowerPositionLimit, model.upperPositionLimit, NOMINAL, np.arange(B), 5, mode, narrow=2.0)
        tg = O.fk_batch(om, qs, [fid])
        Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
        T = torch.from_numpy(np.ascontiguousarray(tg.transpose(1, 2, 0))).cuda()
        Q, ok, it = ik_amd.dls_batch(problem, Q0, T, data, ik_amd.inverse_kinematics_visitor(tol), ik_amd.dls_parameters(max_iterations=iters))
        q_ref, ok_ref, it_ref = O.dls_batch(om, ot, tg, q0, O.params(iters, 1e-2, 1.0, tol), os.cpu_count() or 1)
        assert np.array_equal(ok.cpu().numpy(), ok_ref) and np.array_equal(it.cpu().numpy(), it_ref), (mode, iters)
        assert np.abs(Q.cpu().numpy().T - q_ref).max() <= TOL, (mode, iters)
        q = Q.cpu().numpy().T
        assert (q >= model.lowerPositionLimit - 1e-15).all() and (q <= model.upperPositionLimit + 1e-15).all()
        if narrow_limits and mode == "uniform":
            assert (np.abs(np.abs(q) - 2.0) < 1e-15).any()        # some joints do sit on a limit after the step


@pytest.mark.gpu
def test_ur10_full_size_round_trip(native_built):
    """BASELINE.json config 5 at its full size (B = 65536): targets FK(q*) near the nominal pose are reached, i.e. FK(q) = target."""
    import torch
    import ik_amd
    from ik_amd import workload
    model = ik_amd.Model.from_urdf_file(urdf_path("ur10"))
    problem = ik_amd.InverseKinematicsProblem(model)
    problem.add_frame_task("t", ik_amd.FrameTask.create(model, "tool0", ik_amd.KinematicType.Full))
    data = ik_amd.dls_data(problem, device=0)
    B = 65536
    q0, qs = workload.chain_workload(model.lowerPositionLimit, model.upperPositionLimit, NOMINAL, np.arange(B), 0, "near")
    Q0 = torch.from_numpy(np.ascontiguousarray(q0.T)).cuda()
    QS = torch.from_numpy(np.ascontiguousarray(qs.T)).cuda()
    T = ik_amd.task_frames_fk_batch(problem, QS, data)
    Q, ok, it = ik_amd.dls_batch(problem, Q0, T, data, ik_amd.never_stop_visitor(), ik_amd.dls_parameters(max_iterations=50))
    reached = ik_amd.task_frames_fk_batch(problem, Q, data)
    assert (reached - T).abs().max().item() < 1e-9
    assert (Q - QS).abs().max().item() < 1e-6
