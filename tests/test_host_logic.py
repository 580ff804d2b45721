"""CPU-side checks of the product's host logic: the C-ABI library loads and exports every symbol
include/ikgpu.h declares, the URDF loader reproduces the golden model tables (Pinocchio's
conventions), problem analysis picks / rejects kernels, errors are reported as the
header promises.  No compute call is made (there is no GPU here and the product has no CPU path)."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

from conftest import MODELS, ROOT, urdf_path

import twin as T


def _g(name):
    """A generic problem's kernel name without the build suffix (",static>": the lane program specialised at run time, when hipRTC is there)."""
    return name.replace(",static>", ">")


@pytest.fixture(scope="module")
def ik(native_built):
    import ik_amd
    return ik_amd


def test_library_exports_every_declared_symbol(native_built):
    from ik_amd import capi
    header = open(os.path.join(ROOT, "include", "ikgpu.h")).read()
    declared = sorted(set(re.findall(r"\b(ikgpu_[a-z_0-9]+)\s*\(", header)))
    assert declared, "no prototypes found"
    lib = C.CDLL(native_built)
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(capi.SYMBOLS) == declared
    lib.ikgpu_abi_version.restype = C.c_int
    assert lib.ikgpu_abi_version() == 2   # 2: ikgpu_dls_params carries the derived-visitor members


def test_default_parameters_are_the_reference_defaults(ik):
    from ik_amd import capi
    p = capi.DlsParams()
    capi.lib().ikgpu_dls_params_default(C.byref(p))
    # reference ik/ik/common.hpp:61,65; ik/ik/dls.hpp:25; ik/ik/visitor.hpp:19
    assert (p.max_iterations, p.damping, p.step_length, p.stop_sq_tol) == (100, 1e-2, 1.0, 1e-4)
    d = ik.dls_parameters()
    assert (d.max_iterations, d.damping, d.step_length) == (100, 1e-2, 1.0)
    assert ik.inverse_kinematics_visitor().tolerance == 1e-4


@pytest.mark.parametrize("case", ["S_cassie_leg", "U_ur5", "F_cassie_full"])
def test_urdf_loader_matches_golden_model_tables(ik, case):
    g = json.load(open(os.path.join(ROOT, "tests", "golden", case + ".json")))
    m = ik.Model.from_urdf_file(os.path.join(MODELS, g["urdf"]), free_flyer=g["free_flyer"])
    gm = g["model"]
    assert (m.njoints, m.nq, m.nv, m.nframes) == (gm["njoints"], gm["nq"], gm["nv"], gm["nframes"])
    assert m.names == gm["joint_names"] and m.frame_names == gm["frame_names"]
    f = m.flat()
    assert f["idx_q"].tolist() == gm["idx_q"] and f["idx_v"].tolist() == gm["idx_v"] and f["parent"].tolist() == gm["parent"]
    assert np.array_equal(f["lower"], np.array(gm["lower"])) and np.array_equal(f["upper"], np.array(gm["upper"]))
    assert np.array_equal(f["placement"], np.array(gm["joint_placement"]))          # bit-exact
    assert np.array_equal(f["frame_placement"], np.array(gm["frame_placement"]))
    assert f["frame_parent"].tolist() == gm["frame_parent"]
    assert m.getFrameId("universe") == 0 and m.getFrameId("no_such_frame") == m.nframes  # as Model::getFrameId


@pytest.mark.parametrize("name,ff", [("cassie", True), ("cassie_fixed", False), ("ur5", False)])
def test_loader_masses_match_the_twin(ik, name, ff):
    """Mass and lever of the bodies on each joint (what pinocchio::centerOfMass reads of model.inertias): the product's
    loader (appendBodyToJoint-style accumulation) against the twin's (sums of first moments)."""
    f = ik.Model.from_urdf_file(urdf_path(name), free_flyer=ff).flat()
    t = T.load_urdf(urdf_path(name), free_flyer=ff)
    assert np.abs(f["mass"] - t.mass).max() < 1e-13
    assert np.abs(f["lever"] - t.lever).max() < 1e-15
    total = {"cassie": 34.676752, "cassie_fixed": 24.346752, "ur5": 16.9939}[name]   # links below the first moving joint
    assert abs(f["mass"][1:].sum() - total) < 1e-9
    if name == "cassie_fixed":
        assert abs(f["mass"][0] - 10.33) < 1e-12   # the pelvis is welded to the universe: Pinocchio leaves it out of the CoM


def test_centre_of_mass_task_plan_and_errors(ik):
    from ik_amd import api, capi
    m = ik.Model.from_urdf_file(urdf_path("cassie"), free_flyer=True)
    p = ik.InverseKinematicsProblem(m, 1)
    p.add_frame_task("fl", ik.FrameTask.create(m, "LeftFootFront"))
    com = p.add_centre_of_mass_task(ik.CentreOfMassTask.create(m), 1)
    assert p.get_centre_of_mass_task() is com and com.dimension() == 3 and p.e_size(1) == 3
    com.target[:] = [0.0, 0.0, 1.0]                                              # reference ik_ros/src/cassie.cpp:101
    assert api._abi_rows(com, 1) == [(0, 0, capi.CENTRE_OF_MASS, 1, [1.0] * 6)]
    assert api._target_slots(com)[0, 9:].tolist() == [0.0, 0.0, 1.0]
    assert _g(ik.plan(p)) == "dls_generic<M=9,nv=22,joints=17>"
    with pytest.raises(ValueError):
        p.add_centre_of_mass_task(ik.CentreOfMassTask.create(m), 1)
    # a model without masses cannot carry the task
    bare = ik.Model.from_urdf_xml(b"<robot><link name='a'/><link name='b'/><joint name='j' type='revolute'><parent link='a'/>"
                                  b"<child link='b'/><axis xyz='0 0 1'/><limit lower='-1' upper='1'/></joint></robot>")
    q = ik.InverseKinematicsProblem(bare)
    q.add_centre_of_mass_task(ik.CentreOfMassTask.create(bare))
    with pytest.raises(capi.IkgpuError) as ei:
        ik.plan(q)
    assert "needs joint masses" in ei.value.message


def test_loader_on_the_reference_urdfs_when_present(ik):
    """The stripped fixtures and the reference's full URDFs (visual / inertial / transmission elements,
    comments, nested <joint> inside <transmission>) give the identical model."""
    ref = "/root/reference"
    pairs = [("cassie-description/urdf/cassie.urdf", "cassie", True), ("cassie-description/urdf/cassie_fixed.urdf", "cassie_fixed", False),
             ("ik/test/ur5.urdf", "ur5", False)]
    if not os.path.isdir(ref):
        pytest.skip("reference tree not present (GPU box)")
    for rel, name, ff in pairs:
        a = ik.Model.from_urdf_file(os.path.join(ref, rel), ff).flat()
        b = ik.Model.from_urdf_file(urdf_path(name), ff).flat()
        for k in a:
            assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), (name, k)


def test_model_roundtrip_through_flat_arrays(ik):
    from ik_amd import capi
    m = ik.Model.from_urdf_file(urdf_path("cassie"), free_flyer=True)
    h = C.c_void_p()
    capi.check(capi.lib().ikgpu_model_create(C.byref(m._flat), C.byref(h)))
    m2 = ik.Model(h)
    a, b = m.flat(), m2.flat()
    for k in a:
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k


BAD = {
    "not xml": (b"this is not xml", "XML parse error"),
    "wrong root": (b"<model/>", "must be <robot>"),
    "mismatched tag": (b"<robot><link name='a'></robot>", "mismatched end tag"),
    "no links": (b"<robot name='r'/>", "no links"),
    "planar joint": (b"<robot><link name='a'/><link name='b'/><joint name='j' type='planar'><parent link='a'/>"
                     b"<child link='b'/><axis xyz='0 0 1'/></joint></robot>", "unsupported URDF joint type"),
    "unknown link": (b"<robot><link name='a'/><joint name='j' type='fixed'><parent link='a'/><child link='zz'/></joint></robot>",
                     "unknown link"),
    "two roots": (b"<robot><link name='a'/><link name='b'/></robot>", "more than one root"),
    "bad number": (b"<robot><link name='a'/><link name='b'/><joint name='j' type='revolute'><origin xyz='0 x 0'/>"
                   b"<parent link='a'/><child link='b'/></joint></robot>", "cannot parse number"),
}


@pytest.mark.parametrize("label", sorted(BAD))
def test_malformed_urdf_is_an_error_not_a_crash(ik, label):
    from ik_amd import capi
    xml, needle = BAD[label]
    with pytest.raises(capi.IkgpuError) as ei:
        ik.Model.from_urdf_xml(xml)
    assert ei.value.code == capi.ERR_PARSE and needle in ei.value.message


def test_xml_features_attributes_comments_entities(ik):
    xml = b"""<?xml version="1.0"?><!DOCTYPE robot><!-- c --><robot name='r'>
      <link name="a"><visual><geometry><mesh filename="x &amp; y.stl"/></geometry></visual></link><link name='b'/>
      <!-- <joint name="ghost" type="revolute"/> -->
      <transmission name="t"><joint name="j"><hardwareInterface>P</hardwareInterface></joint></transmission>
      <joint name="j" type="revolute"><origin xyz="+1 2e-1 -.5" rpy="0 0 0"/><parent link="a"/><child link="b"/>
        <axis xyz="0 1 0"/><limit lower="-1" upper="2" effort="1"/></joint></robot>"""
    m = ik.Model.from_urdf_xml(xml)
    assert m.names == ["universe", "j"] and m.nq == 1
    f = m.flat()
    assert f["placement"][1, 9:].tolist() == [1.0, 0.2, -0.5] and f["axis"][1].tolist() == [0, 1, 0]
    assert (f["lower"][0], f["upper"][0]) == (-1.0, 2.0)


def _problem(ik, name, frames, ff=False, types=None, reference="universe", max_prio=0, prios=None):
    m = ik.Model.from_urdf_file(urdf_path(name), free_flyer=ff)
    p = ik.InverseKinematicsProblem(m, max_prio)
    for i, f in enumerate(frames):
        t = ik.FrameTask.create(m, f, types[i] if types else ik.KinematicType.Full, reference)
        p.add_frame_task("t%d" % i, t, prios[i] if prios else 0)
    return m, p


def test_plan_names_the_kernel_for_the_benchmark_shapes(ik):
    # the name says which BUILD runs: "hot" = structure-specialised and compiled into the library (the fixture shapes), "hot-rtc" =
    # the same kernel compiled for this chain's structure code at run time (hipRTC), "general" = device/chain_solver.hpp
    assert ik.plan(_problem(ik, "cassie_fixed", ["LeftFootFront"])[1]) == "dls_chain<NJ=7,full,hot>"
    assert ik.plan(_problem(ik, "ur5", ["tool0"])[1]) == "dls_chain<NJ=6,full,hot>"
    assert ik.plan(_problem(ik, "cassie_fixed", ["RightFootFront"], types=[ik.KinematicType.Position])[1]) == "dls_chain<NJ=7,position,general>"
    assert ik.plan(_problem(ik, "ur5", ["wrist_1_link"], types=[ik.KinematicType.Orientation])[1]) == "dls_chain<NJ=4,orientation,general>"
    assert ik.plan(_problem(ik, "arm7", ["tool"])[1]) in ("dls_chain<NJ=7,full,hot-rtc>", "dls_chain<NJ=7,full,general>")   # (general when libhiprtc is absent)


def test_problem_api_mirrors_the_reference_container(ik):
    m, p = _problem(ik, "cassie_fixed", ["LeftFootFront", "RightFootFront"], max_prio=1, prios=[1, 0],
                    types=[ik.KinematicType.Full, ik.KinematicType.Position])
    assert p.max_priority_level() == 1 and p.e_size(0) == 3 and p.e_size(1) == 6 and p.c_size() == 0
    assert [t.frame for t, _ in p.ordered_tasks()] == ["RightFootFront", "LeftFootFront"]  # priority order (dls.cpp:20-24)
    assert p.get_frame_task("t0").frame == "LeftFootFront" and p.model() is m
    t = p.get_frame_task("t1")
    assert t.dimension() == 3 and t.weighting().tolist() == [1, 1, 1] and np.array_equal(t.target.to12(), ik.SE3().to12())
    with pytest.raises(ValueError):
        ik.FrameTask.create(m, "nope")
    with pytest.raises(ValueError):
        p.add_frame_task("x", t, priority=2)


def test_every_other_shape_maps_to_the_generic_kernel(ik):
    """Shapes without a register-resident specialisation run on the memory-resident generic kernel -- on the
    device, never on a CPU path."""
    m, p = _problem(ik, "ur5", ["tool0"], reference="wrist_1_link")          # reference frame moves with q
    assert _g(ik.plan(p)) == "dls_generic<M=6,nv=6,joints=6>"
    m, p = _problem(ik, "cassie_fixed", ["LeftFootFront", "LeftFootBack"])   # fixed base, two tasks sharing their joints
    assert _g(ik.plan(p)) == "dls_generic<M=12,nv=16,joints=16>"
    m, p = _problem(ik, "cassie_fixed", ["LeftFootFront", "RightFootFront"])  # fixed base, two disjoint chains: the tree kernel
    assert ik.plan(p) == "dls_tree<NJ=7,chains=2,fixed_base>"                 # with its base block dropped
    m, p = _problem(ik, "cassie", ["LeftFootFront", "RightFootFront", "LeftFootBack"], ff=True)   # three chain tasks
    assert _g(ik.plan(p)) == "dls_generic<M=18,nv=22,joints=17>"
    m, p = _problem(ik, "cassie", ["LeftFootFront", "pelvis"], ff=True)
    p.add_align_axis_task("align", ik.AlignAxisTask.create(m, "RightFootFront", ik.AlignAxisType.AxisY))   # row on a frame with no task
    assert _g(ik.plan(p)) == "dls_generic<M=13,nv=22,joints=17>"


def test_the_demo_task_set_has_a_register_resident_kernel(ik):
    """The reference's demo (ik_ros/src/cassie.cpp:45-81): foot position w.r.t. the pelvis (a reference frame riding on
    the floating base), pelvis pose in the world, foot Y axis aligned with a direction -- the tree kernel's general
    build takes the base-relative reference and the alignment row on the chain task's own frame."""
    m = ik.Model.from_urdf_file(urdf_path("cassie"), free_flyer=True)
    p = ik.InverseKinematicsProblem(m, 1)
    p.add_frame_task("fl", ik.FrameTask.create(m, "LeftFootFront", ik.KinematicType.Position, "pelvis"))
    p.add_frame_task("pelvis", ik.FrameTask.create(m, "pelvis", ik.KinematicType.Full))
    p.add_align_axis_task("align", ik.AlignAxisTask.create(m, "LeftFootFront", ik.AlignAxisType.AxisY))
    assert ik.plan(p) == "dls_tree<NJ=7,chains=1,base_task,base_reference,align_axis>"
    assert p.get_align_axis_task("align").dimension() == 1 and p.e_size(0) == 10
    m2, p2 = _problem(ik, "cassie", ["LeftFootFront", "pelvis"], ff=True)
    p2.add_align_axis_task("align", ik.AlignAxisTask.create(m2, "LeftFootFront", ik.AlignAxisType.AxisY), 0)
    assert ik.plan(p2) == "dls_tree<NJ=7,chains=1,base_task,align_axis>"
    # an alignment direction given in a moving frame stays on the generic kernel
    p3 = ik.InverseKinematicsProblem(m)
    p3.add_frame_task("fl", ik.FrameTask.create(m, "LeftFootFront"))
    p3.add_align_axis_task("align", ik.AlignAxisTask.create(m, "LeftFootFront", ik.AlignAxisType.AxisX, "pelvis"))
    assert ik.plan(p3).startswith("dls_generic<")


def test_invalid_task_tables_are_errors(ik):
    from ik_amd import capi
    m = ik.Model.from_urdf_file(urdf_path("ur5"))
    bad = (capi.Task * 1)(capi.Task(m.nframes + 3, 0, 2, 0, (C.c_double * 6)(*[1.0] * 6)))
    buf = C.create_string_buffer(64)
    rc = capi.lib().ikgpu_problem_plan(m._h, bad, 1, buf, 64)
    assert rc == capi.ERR_INVALID and b"frame id out of range" in capi.lib().ikgpu_last_error()
    bad[0].frame, bad[0].type = 1, 9
    assert capi.lib().ikgpu_problem_plan(m._h, bad, 1, buf, 64) == capi.ERR_INVALID
    # posture rows index the tangent vector and q directly
    bad[0].frame, bad[0].reference, bad[0].type = m.nv, 0, capi.POSTURE_ROW
    assert capi.lib().ikgpu_problem_plan(m._h, bad, 1, buf, 64) == capi.ERR_INVALID
    bad[0].frame, bad[0].reference = 0, m.nq
    assert capi.lib().ikgpu_problem_plan(m._h, bad, 1, buf, 64) == capi.ERR_INVALID


def test_posture_task_expands_to_one_row_per_joint(ik):
    """ik::PostureTask (reference ik/ik/posture.hpp:17-85) crosses the ABI as nj IKGPU_POSTURE_ROW tasks."""
    from ik_amd import api, capi
    m = ik.Model.from_urdf_file(urdf_path("cassie"), free_flyer=True)
    p = ik.InverseKinematicsProblem(m, 1)
    p.add_frame_task("fl", ik.FrameTask.create(m, "LeftFootFront"))
    t = p.add_posture_task("posture", ik.PostureTask.create(m, 16), 1)
    assert t.dimension() == 16 and (t.target == 0).all() and (t.mask == 1).all() and (t.weighting() == 1).all()
    t.weighting()[:] = np.arange(16) + 1.0
    t.mask[5] = 0.0
    t.target[:] = np.linspace(-1, 1, 16)
    rows = api._abi_rows(t, 1)
    assert [r[0] for r in rows] == list(range(6, 22)) and [r[1] for r in rows] == list(range(7, 23))
    assert all(r[2] == capi.POSTURE_ROW and r[3] == 1 for r in rows)
    assert [r[4][0] for r in rows] == list(np.arange(16) + 1.0) and [r[4][1] for r in rows] == [0.0 if k == 5 else 1.0 for k in range(16)]
    slots = api._target_slots(t)
    assert slots.shape == (16, 12) and np.array_equal(slots[:, 9], t.target) and np.count_nonzero(slots) == np.count_nonzero(t.target)
    assert p.e_size(0) == 6 and p.e_size(1) == 16 and p.target_slots() == 17
    # one chain task on a free-flyer model + posture rows: the tree kernel takes the rows (7 on the chain's joints, 9 outside)
    assert ik.plan(p) == "dls_tree<NJ=7,chains=1,posture>"
    fixed = ik.Model.from_urdf_file(urdf_path("cassie_fixed"))
    pf = ik.InverseKinematicsProblem(fixed)
    pf.add_frame_task("fl", ik.FrameTask.create(fixed, "LeftFootFront"))
    pf.add_posture_task("posture", ik.PostureTask.create(fixed, 16))
    assert ik.plan(pf) == "dls_tree<NJ=7,chains=1,posture,fixed_base>"  # a fixed-base model: same kernel, base block dropped
    with pytest.raises(ValueError):
        ik.PostureTask(m, 40)
    with pytest.raises(ValueError):
        p.add_posture_task("again", t, 2)


def test_no_gpu_means_a_loud_error_not_a_fallback(ik):
    import torch
    from ik_amd import capi
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    m, p = _problem(ik, "cassie_fixed", ["LeftFootFront"])
    with pytest.raises(capi.IkgpuError) as ei:
        ik.dls_data(p)
    assert ei.value.code == capi.ERR_DEVICE and "no CPU path" in ei.value.message


def test_workload_is_counter_based_and_shardable():
    from ik_amd import workload
    lo, hi, nom = -np.ones(5), np.ones(5), np.zeros(5)
    q0, qs = workload.chain_workload(lo, hi, nom, np.arange(100), seed=7)
    q0b, qsb = workload.chain_workload(lo, hi, nom, np.arange(40, 60), seed=7)
    assert np.array_equal(q0[40:60], q0b) and np.array_equal(qs[40:60], qsb)      # any shard, same numbers
    q0c, _ = workload.chain_workload(lo, hi, nom, np.arange(100), seed=8)
    assert not np.array_equal(q0, q0c)
    assert (np.abs(q0) <= 0.1).all() and (qs >= lo).all() and (qs <= hi).all()
    u = workload.uniform01(0, np.arange(200000), np.arange(2))
    assert abs(u.mean() - 0.5) < 2e-3 and abs(u.var() - 1 / 12) < 2e-3 and u.min() >= 0 and u.max() < 1


def test_arm7_loader_matches_the_twin(ik):
    """The non-fixture arm (general joint-origin rotations, oblique axes; fixtures/make_arm7_urdf.py): the product's URDF loader
    against the independent twin's."""
    f = ik.Model.from_urdf_file(urdf_path("arm7")).flat()
    t = T.load_urdf(urdf_path("arm7"))
    tf = __import__("oracle").flat_from_twin(t)
    assert (f["nq"], f["nv"]) == (7, 7) and f["idx_q"].tolist() == tf["idx_q"].tolist() and f["parent"].tolist() == tf["parent"].tolist()
    assert np.abs(f["placement"] - tf["placement"]).max() < 1e-15 and np.abs(f["axis"] - tf["axis"]).max() < 1e-15
    assert np.abs(f["frame_placement"] - tf["frame_placement"]).max() < 1e-15
    assert np.array_equal(f["lower"], tf["lower"]) and np.array_equal(f["upper"], tf["upper"])


def test_chain_build_switches_and_precompile(ik, tmp_path, monkeypatch):
    """Which build of the chain kernel a problem gets is decided at creation and is part of the name; IKGPU_CHAIN_HOT=0 forces the
    general build, IKGPU_RTC=0 keeps run-time compilation off; ikgpu_problem_precompile fills the on-disk cache without a device."""
    from ik_amd import capi
    leg = _problem(ik, "cassie_fixed", ["LeftFootFront"])[1]
    arm = _problem(ik, "arm7", ["tool"])[1]
    monkeypatch.setenv("IKGPU_CHAIN_HOT", "0")
    assert ik.plan(leg) == "dls_chain<NJ=7,full,general>" and ik.plan(arm) == "dls_chain<NJ=7,full,general>"
    monkeypatch.delenv("IKGPU_CHAIN_HOT")
    monkeypatch.setenv("IKGPU_RTC", "0")
    assert ik.plan(leg) == "dls_chain<NJ=7,full,hot>" and ik.plan(arm) == "dls_chain<NJ=7,full,general>"
    monkeypatch.delenv("IKGPU_RTC")
    if ik.plan(arm) != "dls_chain<NJ=7,full,hot-rtc>":
        pytest.skip("libhiprtc not loadable here")
    monkeypatch.setenv("IKGPU_CACHE_DIR", str(tmp_path))
    assert ik.precompile(leg) == "dls_chain<NJ=7,full,hot>" and not os.listdir(tmp_path)        # pre-built: nothing to compile
    assert ik.precompile(arm) == "dls_chain<NJ=7,full,hot-rtc>"
    files = os.listdir(tmp_path)
    assert len(files) == 1 and files[0].startswith("chain_hot_") and files[0].endswith(".hsaco")
    raw = open(os.path.join(tmp_path, files[0]), "rb").read(36)
    assert raw[:8] == b"IKGPUCO2" and raw[32:36] == b"\x7fELF"      # the cache header (magic, key, length, checksum), then the code object
    weighted = _problem(ik, "arm7", ["tool"])[1]
    weighted.get_frame_task("t0").weighting()[0] = 2.0                                           # the hot program needs unit weights
    assert ik.precompile(weighted) == "dls_chain<NJ=7,full,general>"


def test_shard_rule_and_slot_layout_of_the_c_abi(ik):
    """ikgpu_shard_range / ikgpu_shard_slot_layout / ikgpu_shard_slot_bytes (what ikgpu_dls_solve_batch_sharded splits and packs by)
    against their definitions; ik_amd.distributed uses the same functions for the one-process-per-GPU path."""
    from ik_amd import capi, distributed as D
    L = capi.lib()
    for total in (1, 7, 10, 64, 65536, 262144, 262145, 1000003):
        for world in (1, 2, 3, 4, 8):
            if total < world:
                continue
            base, rem = divmod(total, world)
            prev_hi = 0
            for r in range(world):
                lo, hi = D.shard_range(total, r, world)
                assert lo == prev_hi == r * base + min(r, rem) and hi - lo == base + (1 if r < rem else 0)
                prev_hi = hi
            assert prev_hi == total
            for rows in (6, 7, 16, 23):
                b_max = base + (1 if rem else 0)
                (oq, oi, os_), used = D._layout(rows, b_max)
                assert (oq, oi, os_, used) == (0, rows * b_max * 8, rows * b_max * 8 + 4 * b_max, rows * b_max * 8 + 5 * b_max)
                assert L.ikgpu_shard_slot_bytes(rows, total, world) == (used + 15) // 16 * 16


def test_derived_visitor_defaults_and_oracle_family(ik):
    """ikgpu_dls_params_default switches the derived-visitor members off (= the reference's visitor, ik/ik/visitor.hpp:15-21); the
    oracle's restatement of the family stops earlier on a step tolerance and later on a stricter second-level tolerance."""
    from ik_amd import capi
    import oracle as O
    p = capi.DlsParams(7, 0.5, 0.25, 3.0, 9.0, 5)
    capi.lib().ikgpu_dls_params_default(C.byref(p))
    assert (p.dq_sq_tol, p.num_level_tols, list(p.level_sq_tol)) == (0.0, 0, [0.0] * 8)
    assert capi.lib().ikgpu_abi_version() == 2
    v = ik.inverse_kinematics_visitor()
    assert (v.tolerance, v.step_tolerance, v.level_tolerances) == (1e-4, 0.0, ())
    assert ik.Problem is ik.InverseKinematicsProblem
    model = ik.Model.from_urdf_file(urdf_path("ur5"))
    om = O.OracleModel(model.flat())
    fid = model.getFrameId("tool0")
    tasks = O.make_tasks([(fid, 0, 0, 0, None), (fid, 0, 1, 1, None)])       # position at level 0, orientation at level 1
    rng = np.random.default_rng(1)
    q0 = rng.uniform(-1, 1, (64, 6))
    tg = np.repeat(O.fk_batch(om, q0 + rng.uniform(-0.2, 0.2, (64, 6)), [fid]), 2, axis=1)
    prm = O.params(200, 1e-1, 0.5, 1e-4)
    _, ok0, it0 = O.dls_batch(om, tasks, tg, q0, prm)
    try:
        O.set_visitor(level_sq_tol=(1e-4, 1e-6))
        _, ok1, it1 = O.dls_batch(om, tasks, tg, q0, prm)
        O.set_visitor(dq_sq_tol=1e-2)
        _, ok2, it2 = O.dls_batch(om, tasks, tg, q0, prm)
    finally:
        O.set_visitor()
    assert (it1 >= it0).all() and (it1 > it0).any()          # the second level has to converge too
    assert (it2 <= it0).all() and (it2 < it0).any()          # a coarse step tolerance fires first
    _, ok3, it3 = O.dls_batch(om, tasks, tg, q0, prm)
    assert np.array_equal(it3, it0)                          # restored
