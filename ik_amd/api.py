"""Host-side mirror of the reference's API for the DLS path, over the C ABI (include/ikgpu.h).

Names, argument meaning and outcome reporting follow dazzmo/ik so that the parity tests read
like the reference's own (commented-out) tests (ik/test/dls.cpp:10-76):

    reference (C++)                                   here (Python over libikgpu.so)
    ------------------------------------------------  -----------------------------------------
    pinocchio::urdf::buildModelFromXML  cassie.cpp:34  Model.from_urdf_xml(xml, free_flyer)
    ik::InverseKinematicsProblem        problem.hpp:9  InverseKinematicsProblem(model, max_priority_level)
    ik::FrameTask::create               frame.hpp:123  FrameTask.create(model, frame, type, reference_frame)
    ik::KinematicType                   frame.hpp:20   KinematicType
    ik::dls_parameters                  dls.hpp:24     dls_parameters
    ik::AlignAxisTask::create           frame.hpp:245  AlignAxisTask.create(model, frame, axis, reference_frame)
    ik::inverse_kinematics_visitor      visitor.hpp:7  inverse_kinematics_visitor (one-parameter family)
    ik::dls_data                        dls.hpp:34     dls_data(problem)  (owns the device handle)
    ik::dls(problem, q0, data, v, p)    dls.hpp:111    dls(problem, q0, data, visitor, p)
    --                                                 dls_batch(problem, Q0, targets, data, visitor, p)

The C++ mirror of the same classes lives in ik_amd/csrc/host/ik/.  No arithmetic happens here.
"""
import ctypes as C
import enum

import numpy as np

from . import capi


class KinematicType(enum.IntEnum):  # reference ik/ik/frame.hpp:20
    Position = 0
    Orientation = 1
    Full = 2


class AlignAxisType(enum.IntEnum):  # reference ik/ik/frame.hpp:202
    AxisX = 0
    AxisY = 1
    AxisZ = 2


class SE3:
    """pinocchio::SE3 as the path uses it: a rotation matrix and a translation."""

    def __init__(self, rotation=None, translation=None):
        self.rotation = np.eye(3) if rotation is None else np.array(rotation, dtype=np.float64).reshape(3, 3)
        self.translation = np.zeros(3) if translation is None else np.array(translation, dtype=np.float64).reshape(3)

    @staticmethod
    def Identity():
        return SE3()

    @staticmethod
    def from12(v):
        v = np.asarray(v, dtype=np.float64).reshape(12)
        return SE3(v[:9].reshape(3, 3), v[9:])

    def to12(self):
        return np.concatenate([self.rotation.reshape(9), self.translation])


class Model:
    """What the path reads of pinocchio::Model (reference ik/ik/common.hpp:16)."""

    def __init__(self, handle):
        self._h = handle
        L = capi.lib()
        f = capi.FlatModel()
        capi.check(L.ikgpu_model_get_flat(self._h, C.byref(f)))
        self.njoints, self.nq, self.nv, self.nframes = f.njoints, f.nq, f.nv, f.nframes
        self.names = [f.joint_names[i].decode() for i in range(f.njoints)]
        self.frame_names = [f.frame_names[i].decode() for i in range(f.nframes)]
        self.lowerPositionLimit = np.array(f.lower[:f.nq], dtype=np.float64)
        self.upperPositionLimit = np.array(f.upper[:f.nq], dtype=np.float64)
        self._flat = f

    @staticmethod
    def from_urdf_xml(xml, free_flyer=False):
        """pinocchio::urdf::buildModelFromXML(xml, [JointModelFreeFlyer(),] model)."""
        if isinstance(xml, str):
            xml = xml.encode("utf-8")
        h = C.c_void_p()
        capi.check(capi.lib().ikgpu_model_from_urdf(xml, len(xml), capi.ROOT_FREEFLYER if free_flyer else capi.ROOT_FIXED,
                                                    C.byref(h)))
        return Model(h)

    @staticmethod
    def from_urdf_file(path, free_flyer=False):
        with open(path, "rb") as fh:
            return Model.from_urdf_xml(fh.read(), free_flyer)

    def getFrameId(self, name):
        return int(capi.lib().ikgpu_model_frame_id(self._h, name.encode()))

    def getJointId(self, name):
        return int(capi.lib().ikgpu_model_joint_id(self._h, name.encode()))

    def existFrame(self, name):
        return self.getFrameId(name) < self.nframes

    def flat(self):
        """numpy copies of the flat arrays (same keys as the oracle's model dict)."""
        f, nj, nf = self._flat, self.njoints, self.nframes
        return dict(
            nq=self.nq, nv=self.nv,
            jtype=np.array(f.joint_type[:nj], np.int32), parent=np.array(f.joint_parent[:nj], np.int32),
            idx_q=np.array(f.joint_idx_q[:nj], np.int32), idx_v=np.array(f.joint_idx_v[:nj], np.int32),
            placement=np.array(f.joint_placement[:12 * nj]).reshape(nj, 12),
            axis=np.array(f.joint_axis[:3 * nj]).reshape(nj, 3),
            mass=np.array(f.joint_mass[:nj]), lever=np.array(f.joint_com[:3 * nj]).reshape(nj, 3),
            lower=self.lowerPositionLimit.copy(), upper=self.upperPositionLimit.copy(),
            frame_parent=np.array(f.frame_parent[:nf], np.int32),
            frame_placement=np.array(f.frame_placement[:12 * nf]).reshape(nf, 12),
            frame_names=list(self.frame_names), joint_names=list(self.names))

    def __del__(self):
        try:
            if self._h:
                capi.lib().ikgpu_model_destroy(self._h)
                self._h = None
        except Exception:
            pass


class FrameTask:
    """ik::FrameTask (reference ik/ik/frame.hpp:78-200)."""

    def __init__(self, model, frame, type=KinematicType.Full, reference_frame="universe"):
        self.frame, self.reference_frame, self.type = frame, reference_frame, KinematicType(type)
        self._frame_id, self._ref_id = model.getFrameId(frame), model.getFrameId(reference_frame)
        if self._frame_id >= model.nframes:
            raise ValueError("frame %r not found in model" % frame)
        if self._ref_id >= model.nframes:
            raise ValueError("reference frame %r not found in model" % reference_frame)
        self._dimension = 6 if self.type == KinematicType.Full else 3  # frame.hpp:102-108
        self._weighting = np.ones(self._dimension)                      # task.hpp:48-51
        self.target = SE3.Identity()                                    # frame.hpp:189

    @staticmethod
    def create(model, frame, type=KinematicType.Full, reference_frame="universe"):
        return FrameTask(model, frame, type, reference_frame)

    def dimension(self):
        return self._dimension

    def weighting(self):
        return self._weighting


class AlignAxisTask:
    """ik::AlignAxisTask (reference ik/ik/frame.hpp:210-319): one row, e = 1 - axis . target/|target| with the
    frame's chosen axis expressed in the reference frame.  Runs on the generic kernel."""

    def __init__(self, model, frame, axis, reference_frame="universe"):
        self.frame, self.reference_frame, self.axis = frame, reference_frame, AlignAxisType(axis)
        self._frame_id, self._ref_id = model.getFrameId(frame), model.getFrameId(reference_frame)
        if self._frame_id >= model.nframes:
            raise ValueError("frame %r not found in model" % frame)
        if self._ref_id >= model.nframes:
            raise ValueError("reference frame %r not found in model" % reference_frame)
        self._weighting = np.ones(1)
        self.target = np.array([1.0, 0.0, 0.0])  # frame.hpp:307 (uninitialised in the reference)

    @staticmethod
    def create(model, frame, axis, reference_frame="universe"):
        return AlignAxisTask(model, frame, axis, reference_frame)

    def dimension(self):
        return 1

    def weighting(self):
        return self._weighting


class PostureTask:
    """ik::PostureTask (reference ik/ik/posture.hpp:17-85): e = (q.tail(nj) - target) * mask, J.rightCols(nj) = I
    (the reference does not apply the mask to J).  Crosses the ABI as nj one-row tasks (IKGPU_POSTURE_ROW) and
    runs on the generic kernel."""

    def __init__(self, model, nj):
        if not 0 < nj <= min(model.nq, model.nv):
            raise ValueError("PostureTask over %d joints on a model with nv = %d" % (nj, model.nv))
        self.nj = int(nj)
        self._q0, self._v0 = model.nq - self.nj, model.nv - self.nj
        self.target = np.zeros(self.nj)   # posture.hpp:75
        self.mask = np.ones(self.nj)      # posture.hpp:82
        self._weighting = np.ones(self.nj)

    @staticmethod
    def create(model, nj):
        return PostureTask(model, nj)

    def dimension(self):
        return self.nj

    def weighting(self):
        return self._weighting


class FrameConstraint:
    """ik::FrameConstraint (reference ik/ik/frame.hpp:325-449): the frame may not move relative to its reference frame in
    the selected coordinates.  ik::dls keeps its step in the null space of the constraint Jacobian (reference
    ik/ik/dls.cpp:26-34,43-53); `target` is carried for source compatibility -- the loop never reads it."""

    def __init__(self, model, frame, type=KinematicType.Full, reference_frame="universe"):
        self.frame, self.reference_frame, self.type = frame, reference_frame, KinematicType(type)
        self._frame_id, self._ref_id = model.getFrameId(frame), model.getFrameId(reference_frame)
        if self._frame_id >= model.nframes:
            raise ValueError("Frame not found in model: %s" % frame)
        if self._ref_id >= model.nframes:
            raise ValueError("Reference frame not found in model: %s" % reference_frame)
        self.target = SE3.Identity()

    @staticmethod
    def create(model, frame, type=KinematicType.Full, reference_frame="universe"):
        return FrameConstraint(model, frame, type, reference_frame)

    def dimension(self):
        return 6 if self.type == KinematicType.Full else 3


def _constraint_table(problem):
    cons = problem.get_all_constraints()
    arr = (capi.Task * max(1, len(cons)))()
    for i, c in enumerate(cons):
        arr[i].frame, arr[i].reference, arr[i].type, arr[i].priority = c._frame_id, c._ref_id, int(c.type), 0
        for k in range(6):
            arr[i].weight[k] = 1.0
    return arr, len(cons)


class CentreOfMassTask:
    """ik::CentreOfMassTask (reference ik/ik/centre_of_mass.hpp:14-62): e = oMr^-1 com(q) - target (three rows),
    J = R(oMr)^T Jcom.  Needs link masses in the model (URDF <inertial>).  Runs on the generic kernel."""

    def __init__(self, model, reference_frame="universe"):
        self.reference_frame = reference_frame
        self._ref_id = model.getFrameId(reference_frame)
        if self._ref_id >= model.nframes:
            raise ValueError("Reference frame not found in model: %s" % reference_frame)
        self.target = np.zeros(3)   # the reference leaves it uninitialised; a caller sets it (cassie.cpp:101)
        self._weighting = np.ones(3)

    @staticmethod
    def create(model, reference_frame="universe"):
        return CentreOfMassTask(model, reference_frame)

    def dimension(self):
        return 3

    def weighting(self):
        return self._weighting


def _abi_rows(task, prio):
    """The rows a task contributes to the ABI's task table: (frame, reference, type, priority, weight[6])."""
    if isinstance(task, CentreOfMassTask):
        return [(0, task._ref_id, capi.CENTRE_OF_MASS, prio, [float(x) for x in task._weighting] + [1.0] * 3)]
    if isinstance(task, PostureTask):
        return [(task._v0 + k, task._q0 + k, capi.POSTURE_ROW, prio, [float(task._weighting[k]), float(task.mask[k]), 1, 1, 1, 1])
                for k in range(task.nj)]
    w = list(np.asarray(task.weighting(), dtype=np.float64)) + [1.0] * 6
    typ = 3 + int(task.axis) if isinstance(task, AlignAxisTask) else int(task.type)
    return [(task._frame_id, task._ref_id, typ, prio, w[:6])]


def _target_slots(task):
    """The 12-double target slots of a task, one per ABI row."""
    if isinstance(task, PostureTask):
        out = np.zeros((task.nj, 12))
        out[:, 9] = np.asarray(task.target, dtype=np.float64)
        return out
    if isinstance(task, (AlignAxisTask, CentreOfMassTask)):
        return np.concatenate([np.eye(3).reshape(9), np.asarray(task.target, dtype=np.float64).reshape(3)])[None, :]
    return task.target.to12()[None, :]


def _task_table(problem):
    rows = [r for t, prio in problem.ordered_tasks() for r in _abi_rows(t, prio)]
    arr = (capi.Task * len(rows))()
    for i, (f, r, typ, prio, w) in enumerate(rows):
        arr[i].frame, arr[i].reference, arr[i].type, arr[i].priority = f, r, typ, prio
        for k in range(6):
            arr[i].weight[k] = w[k]
    return arr


class InverseKinematicsProblem:
    """ik::InverseKinematicsProblem (reference ik/ik/problem.hpp:9-206), frame tasks only --
    the other task kinds are outside the accelerated path."""

    def __init__(self, model, max_priority_level=0):
        self._model = model
        self._max_priority_level = int(max_priority_level)
        self._tasks = [[] for _ in range(self._max_priority_level + 1)]
        self._frame_tasks = []
        self._frame_tasks_map = {}
        self._axis_tasks = []
        self._axis_tasks_map = {}
        self._posture_tasks = []
        self._posture_tasks_map = {}
        self._frame_constraints = []
        self._frame_constraints_map = {}
        self._com_task = None
        self._generation = 0

    def max_priority_level(self):
        return self._max_priority_level

    def model(self):
        return self._model

    def add_frame_task(self, name, task, priority=0):
        if not 0 <= priority <= self._max_priority_level:
            raise ValueError("Maximum priority level exceeded!")
        self._frame_tasks_map.setdefault(name, len(self._frame_tasks))
        self._frame_tasks.append(task)
        self._tasks[priority].append(task)
        self._generation += 1
        return task

    def get_frame_task(self, name):
        return self._frame_tasks[self._frame_tasks_map[name]]

    def add_align_axis_task(self, name, task, priority=0):  # reference ik/ik/problem.hpp:94-105
        if not 0 <= priority <= self._max_priority_level:
            raise ValueError("Maximum priority level exceeded!")
        self._axis_tasks_map.setdefault(name, len(self._axis_tasks))
        self._axis_tasks.append(task)
        self._tasks[priority].append(task)
        self._generation += 1
        return task

    def get_align_axis_task(self, name):
        return self._axis_tasks[self._axis_tasks_map[name]]

    def get_all_tasks(self, priority):
        return self._tasks[priority]

    def add_frame_constraint(self, name, constraint):  # reference ik/ik/problem.hpp:68-77
        self._frame_constraints_map.setdefault(name, len(self._frame_constraints))
        self._frame_constraints.append(constraint)
        self._generation += 1
        return constraint

    def get_frame_constraint(self, name):
        return self._frame_constraints[self._frame_constraints_map[name]]

    def get_all_constraints(self):  # reference ik/ik/problem.hpp:167-173
        return list(self._frame_constraints)

    def e_size(self, priority):
        return sum(t.dimension() for t in self._tasks[priority])

    def c_size(self):  # reference ik/ik/problem.hpp:47-53
        return sum(c.dimension() for c in self._frame_constraints)

    def ordered_tasks(self):
        """(task, priority) in the row order of the stacked system (reference ik/ik/dls.cpp:20-24)."""
        return [(t, p) for p in range(self._max_priority_level + 1) for t in self._tasks[p]]

    def target_slots(self):
        """Number of 12-double target slots a batch call takes per problem: one per frame / axis task, one per
        joint of a posture task (value in double 9 of its slot)."""
        return sum(t.nj if isinstance(t, PostureTask) else 1 for t, _ in self.ordered_tasks())

    def add_posture_task(self, name, task, priority=0):  # reference ik/ik/problem.hpp:134-145
        if not 0 <= priority <= self._max_priority_level:
            raise ValueError("Maximum priority level exceeded!")
        self._posture_tasks_map.setdefault(name, len(self._posture_tasks))
        self._posture_tasks.append(task)
        self._tasks[priority].append(task)
        self._generation += 1
        return task

    def get_posture_task(self, name):
        return self._posture_tasks[self._posture_tasks_map[name]]

    def add_centre_of_mass_task(self, task, priority=0):  # reference ik/ik/problem.hpp:121-128
        if not 0 <= priority <= self._max_priority_level:
            raise ValueError("Maximum priority level exceeded!")
        if self._com_task is not None:
            raise ValueError("a problem holds one centre-of-mass task")
        self._com_task = task
        self._tasks[priority].append(task)
        self._generation += 1
        return task

    def get_centre_of_mass_task(self):  # reference ik/ik/problem.hpp:130-132
        return self._com_task


Problem = InverseKinematicsProblem   # the name BASELINE.json's north_star uses for reference ik/ik/problem.hpp:9's class


class dls_parameters:
    """ik::dls_parameters (reference ik/ik/dls.hpp:24-28, ik/ik/common.hpp:59-66)."""

    def __init__(self, max_iterations=100, step_length=1.0, damping=1e-2, max_time=1.0, random_restart=False):
        self.max_iterations = max_iterations
        self.max_time = max_time              # unused by the reference loop
        self.step_length = step_length
        self.damping = damping
        self.random_restart = random_restart  # unused by the reference loop


class inverse_kinematics_visitor:
    """ik::inverse_kinematics_visitor (reference ik/ik/visitor.hpp:7-22).  A C++ visitor cannot run
    inside the kernel; the one-parameter family `||e[0]||^2 < tolerance` is what crosses the ABI.
    tolerance < 0 is a visitor whose should_stop() always returns false."""

    def __init__(self, tolerance=1e-4, step_tolerance=0.0, level_tolerances=()):
        self.tolerance = tolerance
        # the rest of what should_stop(ik, e, dq) is handed (reference ik/ik/visitor.hpp:15-21): also stop when ||dq||^2 <
        # step_tolerance (<= 0: off); test ||e[l]||^2 < level_tolerances[l] on every level instead of level 0 alone (empty: off)
        self.step_tolerance = step_tolerance
        self.level_tolerances = tuple(level_tolerances)


class never_stop_visitor(inverse_kinematics_visitor):
    def __init__(self):
        super().__init__(-1.0)


class dls_data:
    """ik::dls_data (reference ik/ik/dls.hpp:34-65): the reusable workspace.  Here it owns the
    device-side problem handle (constant tables in HBM) and reports the outcome of the last call."""

    def __init__(self, problem, device=0):
        self.success = False
        self.iterations = 0
        self.q = None
        self._device = int(device)
        self._h = None
        self._generation = -1
        self._bind(problem)

    def _bind(self, problem):
        # everything ikgpu_problem_create_constrained reads: the model, every ABI row (frame, reference, type, priority,
        # weight[6]) and every constraint row -- the comparison the C++ mirror makes (ik_gpu.hpp dls_data::same)
        key = (id(problem.model()), int(problem.model()._h.value or 0),
               tuple((f, r, typ, prio, tuple(w)) for t, p in problem.ordered_tasks() for f, r, typ, prio, w in _abi_rows(t, p)),
               tuple((c._frame_id, c._ref_id, int(c.type)) for c in problem.get_all_constraints()))
        if self._h is not None and key == self._generation:
            return
        self._release()
        arr = _task_table(problem)
        cons, ncons = _constraint_table(problem)
        h = C.c_void_p()
        capi.check(capi.lib().ikgpu_problem_create_constrained(problem.model()._h, arr, len(arr), cons, ncons, self._device, C.byref(h)))
        self._h = h
        self._generation = key
        self.rows = int(capi.lib().ikgpu_problem_rows(h))
        self.kernel = capi.lib().ikgpu_problem_kernel(h).decode()
        sup = (C.c_uint8 * problem.model().nq)()
        capi.check(capi.lib().ikgpu_problem_support(h, sup))
        self.support = np.frombuffer(sup, dtype=np.uint8).astype(bool)   # [nq]: entries of q a solve can move (the rest is only clipped)

    def _release(self):
        if self._h is not None:
            capi.lib().ikgpu_problem_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass


def plan(problem):
    """Name of the kernel specialisation the problem maps to (host-only; raises IkgpuError when the
    shape has no device kernel)."""
    arr = _task_table(problem)
    cons, ncons = _constraint_table(problem)
    buf = C.create_string_buffer(160)
    capi.check(capi.lib().ikgpu_problem_plan_constrained(problem.model()._h, arr, len(arr), cons, ncons, buf, len(buf)))
    return buf.value.decode()


def precompile(problem):
    """Compile ahead of time (host-only, no device) what dls_data(problem) would compile at run time -- the structure-specialised
    chain kernel of a chain without a pre-built instantiation -- into the on-disk cache; returns the kernel name.  Raises
    IkgpuError (UNSUPPORTED) with the compiler's log when the compilation fails (the problem then runs on the general build)."""
    arr = _task_table(problem)
    cons, ncons = _constraint_table(problem)
    buf = C.create_string_buffer(160)
    capi.check(capi.lib().ikgpu_problem_precompile(problem.model()._h, arr, len(arr), cons, ncons, buf, len(buf)))
    return buf.value.decode()


def _params(visitor, p):
    lt = tuple(getattr(visitor, "level_tolerances", ()))
    if len(lt) > capi.MAX_VISITOR_LEVELS:
        raise ValueError("at most %d level tolerances" % capi.MAX_VISITOR_LEVELS)
    return capi.DlsParams(int(p.max_iterations), float(p.damping), float(p.step_length), float(visitor.tolerance),
                          float(getattr(visitor, "step_tolerance", 0.0)), len(lt),
                          (C.c_double * capi.MAX_VISITOR_LEVELS)(*(list(lt) + [0.0] * (capi.MAX_VISITOR_LEVELS - len(lt)))))


def dls(problem, q0, data, visitor=None, p=None):
    """ik::dls (reference ik/ik/dls.hpp:111-114): one problem, targets taken from each task's
    `target`; returns q and sets data.success exactly as reference ik/ik/dls.cpp:62-63,76-77.
    Runs as a batch of one on the device."""
    visitor = visitor or inverse_kinematics_visitor()
    p = p or dls_parameters()
    data._bind(problem)
    model = problem.model()
    q0 = np.ascontiguousarray(q0, dtype=np.float64).reshape(model.nq)
    tg = np.ascontiguousarray(np.concatenate([_target_slots(t) for t, _ in problem.ordered_tasks()]))
    q = np.empty(model.nq)
    ok = np.zeros(1, np.uint8)
    it = np.zeros(1, np.int32)
    prm = _params(visitor, p)
    capi.check(capi.lib().ikgpu_dls_solve_batch_host(
        data._h, 1, q0.ctypes.data, tg.ctypes.data, C.byref(prm), q.ctypes.data, ok.ctypes.data, it.ctypes.data, capi.AOS))
    data.success, data.iterations, data.q = bool(ok[0]), int(it[0]), q
    return q


def dls_batch(problem, Q0, targets, data, visitor=None, p=None, layout="soa", out=None, stream=None):
    """B independent ik::dls() calls in lockstep on the device.

    torch CUDA tensors (float64, contiguous) go straight through as device pointers on the current
    stream; numpy arrays take the host-pointer entry point (copy in / solve / copy out).
      layout "soa": Q0 [nq, B], targets [ntasks, 12, B]   |   "aos": Q0 [B, nq], targets [B, ntasks, 12]
    Returns (Q, success uint8 [B], iterations int32 [B]) in the same container type as Q0.
    """
    visitor = visitor or inverse_kinematics_visitor()
    p = p or dls_parameters()
    data._bind(problem)
    model = problem.model()
    ntasks = problem.target_slots()
    lay ={"soa": capi.SOA, "aos": capi.AOS}[layout]
    prm = _params(visitor, p)
    L = capi.lib()
    if isinstance(Q0, np.ndarray):
        Q0 = np.ascontiguousarray(Q0, dtype=np.float64)
        targets = np.ascontiguousarray(targets, dtype=np.float64)
        B = Q0.shape[1] if lay == capi.SOA else Q0.shape[0]
        _check_shapes(Q0.shape, targets.shape, model.nq, ntasks, B, lay)
        Q = np.empty_like(Q0)
        ok = np.zeros(B, np.uint8)
        it = np.zeros(B, np.int32)
        capi.check(L.ikgpu_dls_solve_batch_host(data._h, B, Q0.ctypes.data, targets.ctypes.data, C.byref(prm),
                                                Q.ctypes.data, ok.ctypes.data, it.ctypes.data, lay))
        return Q, ok, it
    import torch
    if not (Q0.is_cuda and targets.is_cuda and Q0.dtype == torch.float64 and targets.dtype == torch.float64):
        raise TypeError("dls_batch needs float64 CUDA tensors (or numpy arrays)")
    if not (Q0.is_contiguous() and targets.is_contiguous()):
        raise ValueError("dls_batch needs contiguous tensors")
    if Q0.device.index != data._device:
        raise ValueError("tensors live on cuda:%s but the problem was created on device %d" % (Q0.device.index, data._device))
    B = Q0.shape[1] if lay == capi.SOA else Q0.shape[0]
    _check_shapes(tuple(Q0.shape), tuple(targets.shape), model.nq, ntasks, B, lay)
    if out is None:
        Q = torch.empty_like(Q0)
        ok = torch.empty(B, dtype=torch.uint8, device=Q0.device)
        it = torch.empty(B, dtype=torch.int32, device=Q0.device)
    else:
        Q, ok, it = _check_out(out, model.nq, B, lay, data._device)
    s = torch.cuda.current_stream(Q0.device).cuda_stream if stream is None else stream
    capi.check(L.ikgpu_dls_solve_batch(data._h, B, Q0.data_ptr(), targets.data_ptr(), C.byref(prm), Q.data_ptr(),
                                       ok.data_ptr(), it.data_ptr(), lay, C.c_void_p(s)))
    return Q, ok, it


def _check_tensor(name, t, shape, dtype, device):
    """A device buffer handed to the C ABI as a raw pointer: wrong dtype / device / stride / size would make the kernel read or
    write out of bounds, so it is refused here."""
    import torch
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError("%s must be a CUDA tensor" % name)
    if t.dtype != dtype:
        raise TypeError("%s has dtype %s, expected %s" % (name, t.dtype, dtype))
    if t.device.index != device:
        raise ValueError("%s lives on cuda:%s but the problem was created on device %d" % (name, t.device.index, device))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    if tuple(t.shape) != tuple(shape):
        raise ValueError("%s has shape %s, expected %s" % (name, tuple(t.shape), tuple(shape)))


def _check_out(out, nq, B, lay, device):
    import torch
    Q, ok, it = out
    _check_tensor("out[0] (Q)", Q, (nq, B) if lay == capi.SOA else (B, nq), torch.float64, device)
    _check_tensor("out[1] (success)", ok, (B,), torch.uint8, device)
    _check_tensor("out[2] (iterations)", it, (B,), torch.int32, device)
    return Q, ok, it


def _check_shapes(qs, ts, nq, ntasks, B, lay):
    want_q = (nq, B) if lay == capi.SOA else (B, nq)
    want_t = (ntasks, 12, B) if lay == capi.SOA else (B, ntasks, 12)
    if tuple(qs) != want_q:
        raise ValueError("Q0 has shape %s, expected %s" % (tuple(qs), want_q))
    if tuple(ts) != want_t:
        raise ValueError("targets has shape %s, expected %s" % (tuple(ts), want_t))


class pik_parameters:
    """ik::pik_parameters (reference ik/ik/pik.hpp:11-16).  `damping` and `max_time` are carried for source
    compatibility: the reference loop reads neither (the damping factors live in pik_data.lambda_)."""

    def __init__(self, max_iterations=100, damping=1e-2, step_length=1.0, max_time=1.0):
        self.max_iterations = max_iterations
        self.damping = damping
        self.step_length = step_length
        self.max_time = max_time


class pik_data(dls_data):
    """ik::pik_data (reference ik/ik/pik.hpp:21-50): the workspace of ik::pik.  `lambda_` holds the damping factor of
    each priority level (`lambda` in the reference, 1.0 each) and `da` the secondary step projected into the null
    space of all levels (zero); both are read at every call."""

    def __init__(self, problem, device=0):
        super().__init__(problem, device)
        self.lambda_ = [1.0] * (problem.max_priority_level() + 1)
        self.da = np.zeros(problem.model().nv)

    @property
    def kernel(self):
        """Name of the kernel the next call runs with the current `lambda_` / `da`: the problem's DLS kernel when there is one
        priority level and no secondary step (ik::pik is then the DLS iteration, include/ikgpu.h), the PIK kernel otherwise."""
        if self._h is None or len(self.lambda_) > capi.MAX_PIK_LEVELS:
            return ""
        prm = self._params(inverse_kinematics_visitor(), pik_parameters())
        return capi.lib().ikgpu_pik_kernel(self._h, C.byref(prm)).decode()

    @kernel.setter
    def kernel(self, _):
        pass  # dls_data._bind records the DLS kernel's name; this class derives its name from the handle at every read

    def _params(self, visitor, p):
        prm = capi.PikParams()
        prm.max_iterations, prm.step_length, prm.stop_sq_tol = int(p.max_iterations), float(p.step_length), float(visitor.tolerance)
        prm.num_levels = len(self.lambda_)
        if prm.num_levels > capi.MAX_PIK_LEVELS:
            raise ValueError("ik::pik on the device takes at most %d priority levels" % capi.MAX_PIK_LEVELS)
        for i, l in enumerate(self.lambda_):
            prm.lam[i] = float(l)
        da = np.ascontiguousarray(self.da, dtype=np.float64)
        if da.any():
            prm._da_keepalive = da
            prm.da = da.ctypes.data_as(C.POINTER(C.c_double))
        return prm


def plan_generic(problem):
    """Name of the generic kernel instance of the problem (what ik::pik runs on)."""
    name = plan(problem)
    if name.startswith("dls_generic<"):
        return name
    rows = sum(t.dimension() for t, _ in problem.ordered_tasks())
    m = problem.model()
    return "dls_generic<M=%d,nv=%d,joints=%d>" % (rows, m.nv, m.njoints - 1)


def pik(problem, q0, data, visitor=None, p=None):
    """ik::pik (reference ik/ik/pik.hpp:56-59, ik/ik/pik.cpp:31-103): one problem, targets from each task's `target`;
    returns q and sets data.success as the reference does.  A batch of one on the device."""
    visitor = visitor or inverse_kinematics_visitor()
    p = p or pik_parameters()
    data._bind(problem)
    model = problem.model()
    q0 = np.ascontiguousarray(q0, dtype=np.float64).reshape(model.nq)
    tg = np.ascontiguousarray(np.concatenate([_target_slots(t) for t, _ in problem.ordered_tasks()]))
    q = np.empty(model.nq)
    ok = np.zeros(1, np.uint8)
    it = np.zeros(1, np.int32)
    prm = data._params(visitor, p)
    capi.check(capi.lib().ikgpu_pik_solve_batch_host(
        data._h, 1, q0.ctypes.data, tg.ctypes.data, C.byref(prm), q.ctypes.data, ok.ctypes.data, it.ctypes.data, capi.AOS))
    data.success, data.iterations, data.q = bool(ok[0]), int(it[0]), q
    return q


def pik_batch(problem, Q0, targets, data, visitor=None, p=None, layout="soa", out=None, stream=None):
    """B independent ik::pik() calls in lockstep on the device; arguments and return value as dls_batch."""
    visitor = visitor or inverse_kinematics_visitor()
    p = p or pik_parameters()
    data._bind(problem)
    model = problem.model()
    ntasks = problem.target_slots()
    lay = {"soa": capi.SOA, "aos": capi.AOS}[layout]
    prm = data._params(visitor, p)
    L = capi.lib()
    if isinstance(Q0, np.ndarray):
        Q0 = np.ascontiguousarray(Q0, dtype=np.float64)
        targets = np.ascontiguousarray(targets, dtype=np.float64)
        B = Q0.shape[1] if lay == capi.SOA else Q0.shape[0]
        _check_shapes(Q0.shape, targets.shape, model.nq, ntasks, B, lay)
        Q = np.empty_like(Q0)
        ok = np.zeros(B, np.uint8)
        it = np.zeros(B, np.int32)
        capi.check(L.ikgpu_pik_solve_batch_host(data._h, B, Q0.ctypes.data, targets.ctypes.data, C.byref(prm),
                                                Q.ctypes.data, ok.ctypes.data, it.ctypes.data, lay))
        return Q, ok, it
    import torch
    if not (Q0.is_cuda and targets.is_cuda and Q0.dtype == torch.float64 and targets.dtype == torch.float64):
        raise TypeError("pik_batch needs float64 CUDA tensors (or numpy arrays)")
    if not (Q0.is_contiguous() and targets.is_contiguous()):
        raise ValueError("pik_batch needs contiguous tensors")
    if Q0.device.index != data._device:
        raise ValueError("tensors live on cuda:%s but the problem was created on device %d" % (Q0.device.index, data._device))
    B = Q0.shape[1] if lay == capi.SOA else Q0.shape[0]
    _check_shapes(tuple(Q0.shape), tuple(targets.shape), model.nq, ntasks, B, lay)
    if out is None:
        Q = torch.empty_like(Q0)
        ok = torch.empty(B, dtype=torch.uint8, device=Q0.device)
        it = torch.empty(B, dtype=torch.int32, device=Q0.device)
    else:
        Q, ok, it = _check_out(out, model.nq, B, lay, data._device)
    s = torch.cuda.current_stream(Q0.device).cuda_stream if stream is None else stream
    capi.check(L.ikgpu_pik_solve_batch(data._h, B, Q0.data_ptr(), targets.data_ptr(), C.byref(prm), Q.data_ptr(),
                                       ok.data_ptr(), it.data_ptr(), lay, C.c_void_p(s)))
    return Q, ok, it


def evaluate_batch(problem, Q, targets, data, layout="soa", jacobian=True):
    """evaluate_problem_data + stacking (reference ik/ik/data.cpp:25-58, ik/ik/dls.cpp:18-24) for a
    batch, on the device: returns (e [M, B], J [M, nv, B]) for "soa" ([B, M], [B, M, nv] for "aos")."""
    import torch
    data._bind(problem)
    model = problem.model()
    lay = {"soa": capi.SOA, "aos": capi.AOS}[layout]
    B = Q.shape[1] if lay == capi.SOA else Q.shape[0]
    M = data.rows
    nt = problem.target_slots()
    _check_tensor("Q", Q, (model.nq, B) if lay == capi.SOA else (B, model.nq), torch.float64, data._device)
    _check_tensor("targets", targets, (nt, 12, B) if lay == capi.SOA else (B, nt, 12), torch.float64, data._device)
    e = torch.empty((M, B) if lay == capi.SOA else (B, M), dtype=torch.float64, device=Q.device)
    J = None
    if jacobian:
        J = torch.empty((M, model.nv, B) if lay == capi.SOA else (B, M, model.nv), dtype=torch.float64, device=Q.device)
    s = torch.cuda.current_stream(Q.device).cuda_stream
    capi.check(capi.lib().ikgpu_evaluate_batch(data._h, B, Q.data_ptr(), targets.data_ptr(), e.data_ptr(),
                                               J.data_ptr() if jacobian else None, lay, C.c_void_p(s)))
    return e, J


def task_frames_fk_batch(problem, Q, data, layout="soa"):
    """World placements of the task frames (one entry of data.oMf per task, reference
    ik/ik/data.cpp:28-29): [ntasks, 12, B] ("soa") or [B, ntasks, 12] ("aos")."""
    import torch
    data._bind(problem)
    lay = {"soa": capi.SOA, "aos": capi.AOS}[layout]
    B = Q.shape[1] if lay == capi.SOA else Q.shape[0]
    nt = problem.target_slots()
    nq = problem.model().nq
    _check_tensor("Q", Q, (nq, B) if lay == capi.SOA else (B, nq), torch.float64, data._device)
    out = torch.empty((nt, 12, B) if lay == capi.SOA else (B, nt, 12), dtype=torch.float64, device=Q.device)
    s = torch.cuda.current_stream(Q.device).cuda_stream
    capi.check(capi.lib().ikgpu_task_frames_fk_batch(data._h, B, Q.data_ptr(), out.data_ptr(), lay, C.c_void_p(s)))
    return out
