"""ctypes binding of the plain-C oracle (oracle/ik_oracle.c).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED -- see oracle/ik_oracle.h.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libik_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.iko_dls.restype = C.c_int
        _LIB.iko_dls_batch.restype = C.c_int
        _LIB.iko_task_rows.restype = C.c_int
    return _LIB


class _Model(C.Structure):
    _fields_ = [("njoints", C.c_int), ("nq", C.c_int), ("nv", C.c_int), ("nframes", C.c_int),
                ("jtype", C.c_void_p), ("parent", C.c_void_p), ("idx_q", C.c_void_p), ("idx_v", C.c_void_p),
                ("placement", C.c_void_p), ("axis", C.c_void_p), ("lower", C.c_void_p), ("upper", C.c_void_p),
                ("frame_parent", C.c_void_p), ("frame_placement", C.c_void_p), ("mass", C.c_void_p), ("lever", C.c_void_p)]


class _Task(C.Structure):
    _fields_ = [("frame", C.c_int), ("reference", C.c_int), ("type", C.c_int), ("priority", C.c_int),
                ("weight", C.c_double * 6)]


class _Params(C.Structure):
    _fields_ = [("max_iterations", C.c_int), ("damping", C.c_double), ("step_length", C.c_double),
                ("stop_sq_tol", C.c_double)]


class _PikParams(C.Structure):
    _fields_ = [("max_iterations", C.c_int), ("step_length", C.c_double), ("stop_sq_tol", C.c_double),
                ("nlevels", C.c_int), ("lam", C.POINTER(C.c_double)), ("da", C.POINTER(C.c_double))]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# ---- the same restatement in extended precision (oracle/ik_oracle_ext.c): "q" = _Float128 (113-bit significand), "ld" = x87 long
# double (64-bit).  Same ABI, same names; used by the parity tests to arbitrate lanes the perturbation probes exclude.
_EXT = {}


def ext_lib(kind="q"):
    if kind not in _EXT:
        path = os.path.join(_HERE, "libik_oracle_%s.so" % kind)
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        for f in (L.iko_dls_batch, L.iko_dls_batch_constrained, L.iko_pik_batch, L.iko_task_rows, L.iko_ext_bits):
            f.restype = C.c_int
        _EXT[kind] = L
    return _EXT[kind]


class OracleModel:
    """Wraps a flat model dict: jtype, parent, idx_q, idx_v (int32 [nj]); placement [nj,12];
    axis [nj,3]; lower, upper [nq]; frame_parent int32 [nf]; frame_placement [nf,12];
    frame_names (list of str); nq, nv."""

    def __init__(self, flat):
        self.flat = flat
        self.nq, self.nv = int(flat["nq"]), int(flat["nv"])
        self.njoints = len(flat["jtype"])
        self.nframes = len(flat["frame_parent"])
        self.frame_names = list(flat["frame_names"])
        k = {}
        for name in ("jtype", "parent", "idx_q", "idx_v", "frame_parent"):
            k[name] = np.ascontiguousarray(flat[name], dtype=np.int32)
        for name in ("placement", "axis", "lower", "upper", "frame_placement"):
            k[name] = np.ascontiguousarray(flat[name], dtype=np.float64)
        k["mass"] = np.ascontiguousarray(flat.get("mass", np.zeros(self.njoints)), dtype=np.float64)
        k["lever"] = np.ascontiguousarray(flat.get("lever", np.zeros((self.njoints, 3))), dtype=np.float64)
        self._keep = k
        self.c = _Model(self.njoints, self.nq, self.nv, self.nframes, _p(k["jtype"]), _p(k["parent"]),
                        _p(k["idx_q"]), _p(k["idx_v"]), _p(k["placement"]), _p(k["axis"]), _p(k["lower"]),
                        _p(k["upper"]), _p(k["frame_parent"]), _p(k["frame_placement"]), _p(k["mass"]), _p(k["lever"]))
        self.lower, self.upper = k["lower"], k["upper"]

    def frame_id(self, name):
        return self.frame_names.index(name)


def make_tasks(specs):
    """specs: list of (frame_id, reference_id, type, priority, weights or None)."""
    arr = (_Task * len(specs))()
    for i, (f, r, t, p, w) in enumerate(specs):
        arr[i].frame, arr[i].reference, arr[i].type, arr[i].priority = f, r, t, p
        ww = [1.0] * 6 if w is None else list(w) + [1.0] * (6 - len(w))
        for k in range(6):
            arr[i].weight[k] = ww[k]
    return arr


def params(max_iterations=100, damping=1e-2, step_length=1.0, stop_sq_tol=1e-4):
    return _Params(max_iterations, damping, step_length, stop_sq_tol)


class _Visitor(C.Structure):
    _fields_ = [("dq_sq_tol", C.c_double), ("nlevels", C.c_int), ("level_sq_tol", C.c_double * 8)]


def set_visitor(dq_sq_tol=-1.0, level_sq_tol=()):
    """The derived-visitor family of ik_oracle.h (process-wide; call set_visitor() with no arguments to restore the reference's)."""
    v = _Visitor(float(dq_sq_tol), len(level_sq_tol), (C.c_double * 8)(*list(level_sq_tol)))
    lib().iko_set_visitor(C.byref(v))


def task_rows(tasks):
    return lib().iko_task_rows(tasks, len(tasks))


def fk(model, q):
    q = np.ascontiguousarray(q, dtype=np.float64)
    oMi = np.empty((model.njoints, 12))
    oMf = np.empty((model.nframes, 12))
    lib().iko_fk(C.byref(model.c), _p(q), _p(oMi), _p(oMf))
    return oMi, oMf


def fk_batch(model, q, frames):
    q = np.ascontiguousarray(q, dtype=np.float64)
    fr = np.ascontiguousarray(frames, dtype=np.int32)
    out = np.empty((q.shape[0], len(fr), 12))
    lib().iko_fk_batch(C.byref(model.c), C.c_long(q.shape[0]), _p(q), _p(fr), C.c_int(len(fr)), _p(out))
    return out


def evaluate(model, tasks, targets, q):
    M = task_rows(tasks)
    targets = np.ascontiguousarray(targets, dtype=np.float64)
    q = np.ascontiguousarray(q, dtype=np.float64)
    et = np.empty(M)
    Jt = np.empty((M, model.nv))
    lib().iko_evaluate(C.byref(model.c), tasks, C.c_int(len(tasks)), _p(targets), _p(q), _p(et), _p(Jt))
    return et, Jt


def dls(model, tasks, targets, q0, prm, trace=False):
    targets = np.ascontiguousarray(targets, dtype=np.float64)
    q0 = np.ascontiguousarray(q0, dtype=np.float64)
    q = np.empty(model.nq)
    ok, it = C.c_int(0), C.c_int(0)
    M = task_rows(tasks)
    tr = np.full((prm.max_iterations, model.nq + M + model.nv), np.nan) if trace else None
    lib().iko_dls(C.byref(model.c), tasks, C.c_int(len(tasks)), _p(targets), _p(q0), C.byref(prm), _p(q),
                  C.byref(ok), C.byref(it), _p(tr) if trace else None)
    if trace:
        return q, bool(ok.value), it.value, tr
    return q, bool(ok.value), it.value


def dls_batch(model, tasks, targets, q0, prm, nthreads=1, ext=None):
    """targets [B, ntasks, 12], q0 [B, nq] (array-of-structures).  ext: None (the double oracle), "q" or "ld" (the same
    statements in _Float128 / long double arithmetic, results rounded to double once at the end)."""
    targets = np.ascontiguousarray(targets, dtype=np.float64)
    q0 = np.ascontiguousarray(q0, dtype=np.float64)
    B = q0.shape[0]
    assert targets.shape == (B, len(tasks), 12) and q0.shape == (B, model.nq)
    q = np.empty_like(q0)
    ok = np.zeros(B, dtype=np.uint8)
    it = np.zeros(B, dtype=np.int32)
    (ext_lib(ext) if ext else lib()).iko_dls_batch(C.byref(model.c), tasks, C.c_int(len(tasks)), C.c_long(B), _p(targets), _p(q0),
                        C.byref(prm), _p(q), _p(ok), _p(it), C.c_int(nthreads))
    return q, ok, it


def dls_constrained(model, tasks, constraints, targets, q0, prm):
    """ik::dls with ik::FrameConstraint entries (make_tasks records: frame, reference, type)."""
    targets = np.ascontiguousarray(targets, dtype=np.float64)
    q0 = np.ascontiguousarray(q0, dtype=np.float64)
    q = np.empty(model.nq)
    ok, it = C.c_int(0), C.c_int(0)
    lib().iko_dls_constrained(C.byref(model.c), tasks, C.c_int(len(tasks)), constraints, C.c_int(len(constraints)), _p(targets), _p(q0),
                              C.byref(prm), _p(q), C.byref(ok), C.byref(it), None)
    return q, bool(ok.value), it.value


def dls_batch_constrained(model, tasks, constraints, targets, q0, prm, nthreads=1, ext=None):
    targets = np.ascontiguousarray(targets, dtype=np.float64)
    q0 = np.ascontiguousarray(q0, dtype=np.float64)
    B = q0.shape[0]
    assert targets.shape == (B, len(tasks), 12) and q0.shape == (B, model.nq)
    q = np.empty_like(q0)
    ok = np.zeros(B, dtype=np.uint8)
    it = np.zeros(B, dtype=np.int32)
    (ext_lib(ext) if ext else lib()).iko_dls_batch_constrained(C.byref(model.c), tasks, C.c_int(len(tasks)), constraints, C.c_int(len(constraints)), C.c_long(B),
                                    _p(targets), _p(q0), C.byref(prm), _p(q), _p(ok), _p(it), C.c_int(nthreads))
    return q, ok, it


def constraint_jacobian(model, constraints, q):
    q = np.ascontiguousarray(q, dtype=np.float64)
    Mc = task_rows(constraints)
    Jc = np.empty((Mc, model.nv))
    lib().iko_constraint_jacobian(C.byref(model.c), constraints, C.c_int(len(constraints)), _p(q), _p(Jc))
    return Jc


def pik_params(max_iterations=100, step_length=1.0, stop_sq_tol=1e-4, lam=(1.0,), da=None):
    """ik::pik_parameters + pik_data::lambda / da (reference ik/ik/pik.hpp:11-16,41-44).  Keeps the arrays alive."""
    prm = _PikParams()
    prm.max_iterations, prm.step_length, prm.stop_sq_tol, prm.nlevels = max_iterations, step_length, stop_sq_tol, len(lam)
    prm._lam = (C.c_double * len(lam))(*lam)
    prm.lam = C.cast(prm._lam, C.POINTER(C.c_double))
    if da is not None:
        prm._da = (C.c_double * len(da))(*da)
        prm.da = C.cast(prm._da, C.POINTER(C.c_double))
    return prm


def pik(model, tasks, targets, q0, prm, trace=False):
    targets = np.ascontiguousarray(targets, dtype=np.float64)
    q0 = np.ascontiguousarray(q0, dtype=np.float64)
    q = np.empty(model.nq)
    ok, it = C.c_int(0), C.c_int(0)
    M = task_rows(tasks)
    tr = np.full((prm.max_iterations, model.nq + M + model.nv), np.nan) if trace else None
    rc = lib().iko_pik(C.byref(model.c), tasks, C.c_int(len(tasks)), _p(targets), _p(q0), C.byref(prm), _p(q),
                       C.byref(ok), C.byref(it), _p(tr) if trace else None)
    if rc != 0:
        raise ValueError("iko_pik: nlevels does not match the task table")
    if trace:
        return q, bool(ok.value), it.value, tr
    return q, bool(ok.value), it.value


def pik_batch(model, tasks, targets, q0, prm, nthreads=1, ext=None):
    """targets [B, ntasks, 12], q0 [B, nq] (array-of-structures)."""
    targets = np.ascontiguousarray(targets, dtype=np.float64)
    q0 = np.ascontiguousarray(q0, dtype=np.float64)
    B = q0.shape[0]
    assert targets.shape == (B, len(tasks), 12) and q0.shape == (B, model.nq)
    q = np.empty_like(q0)
    ok = np.zeros(B, dtype=np.uint8)
    it = np.zeros(B, dtype=np.int32)
    rc = (ext_lib(ext) if ext else lib()).iko_pik_batch(C.byref(model.c), tasks, C.c_int(len(tasks)), C.c_long(B), _p(targets), _p(q0),
                             C.byref(prm), _p(q), _p(ok), _p(it), C.c_int(nthreads))
    if rc != 0:
        raise ValueError("iko_pik_batch: nlevels does not match the task table")
    return q, ok, it


def damp_pseudoinverse(A, lam):
    A = np.ascontiguousarray(A, dtype=np.float64)
    res = np.empty((A.shape[1], A.shape[0]))
    lib().iko_damp_pseudoinverse(_p(A), C.c_int(A.shape[0]), C.c_int(A.shape[1]), C.c_double(lam), _p(res))
    return res


def rowspace_projector(A):
    A = np.ascontiguousarray(A, dtype=np.float64)
    res = np.empty((A.shape[1], A.shape[1]))
    lib().iko_rowspace_projector(_p(A), C.c_int(A.shape[0]), C.c_int(A.shape[1]), _p(res))
    return res


def log6(M12):
    M12 = np.ascontiguousarray(M12, dtype=np.float64)
    out = np.empty(6)
    lib().iko_log6(_p(M12), _p(out))
    return out


def Jlog6(M12):
    M12 = np.ascontiguousarray(M12, dtype=np.float64)
    out = np.empty((6, 6))
    lib().iko_Jlog6(_p(M12), _p(out))
    return out


def exp6(nu):
    nu = np.ascontiguousarray(nu, dtype=np.float64)
    out = np.empty(12)
    lib().iko_exp6(_p(nu), _p(out))
    return out


def integrate(model, q, v):
    q = np.ascontiguousarray(q, dtype=np.float64)
    v = np.ascontiguousarray(v, dtype=np.float64)
    out = np.array(q)
    lib().iko_integrate(C.byref(model.c), _p(q), _p(v), _p(out))
    return out


def flat_from_twin(m):
    """Flat model dict from the numpy twin's independently loaded Model (oracle/twin.py)."""
    def m12(M):
        return np.concatenate([M[:3, :3].reshape(9), M[:3, 3]])
    return dict(
        nq=m.nq, nv=m.nv,
        jtype=np.array(m.jtype, np.int32), parent=np.array(m.parent, np.int32),
        idx_q=np.array(m.idx_q, np.int32), idx_v=np.array(m.idx_v, np.int32),
        placement=np.array([m12(M) for M in m.placement]), axis=np.array(m.axis),
        lower=np.array(m.lower), upper=np.array(m.upper),
        frame_parent=np.array([f["parent"] for f in m.frames], np.int32),
        frame_placement=np.array([m12(f["placement"]) for f in m.frames]),
        frame_names=[f["name"] for f in m.frames], joint_names=list(m.names),
        mass=np.array(m.mass), lever=np.array(m.lever))


# ---- the optimised CPU variant (oracle/fast_cpu.cpp): bench baseline only --------------------------------------------------
_FAST = None


def fast_lib():
    global _FAST
    if _FAST is None:
        path = os.path.join(_HERE, "libik_fastcpu.so")
        if not os.path.exists(path):
            build()
        _FAST = C.CDLL(path)
        _FAST.fastcpu_last_error.restype = C.c_char_p
        _FAST.fastcpu_dls_chain.restype = C.c_int
    return _FAST


def fast_dls_chain_batch(urdf_xml, frame_id, targets, q0, prm, nthreads=1):
    """One Full FrameTask (w.r.t. the universe) on a fixed-base chain; targets [B, 1, 12], q0 [B, nq] (array-of-structures).
    Returns (q, success, iterations) like dls_batch."""
    if isinstance(urdf_xml, str):
        urdf_xml = urdf_xml.encode("utf-8")
    targets = np.ascontiguousarray(targets, dtype=np.float64)
    q0 = np.ascontiguousarray(q0, dtype=np.float64)
    B = q0.shape[0]
    task = make_tasks([(frame_id, 0, 2, 0, None)])
    q = np.empty_like(q0)
    ok = np.zeros(B, dtype=np.uint8)
    it = np.zeros(B, dtype=np.int32)
    rc = fast_lib().fastcpu_dls_chain(urdf_xml, C.c_size_t(len(urdf_xml)), task, C.c_int64(B), _p(q0), _p(targets), C.byref(prm),
                                      _p(q), _p(ok), _p(it), C.c_int(1), C.c_int(nthreads))
    if rc != 0:
        raise RuntimeError(fast_lib().fastcpu_last_error().decode())
    return q, ok, it
