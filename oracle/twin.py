"""Independent numpy twin of the reference IK hot path.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference (dazzmo/ik) cannot be compiled here (Pinocchio, Eigen, Boost,
glog are absent) and its own tests hold no numeric expectations (every TEST body in
ik/test/*.cpp is commented out).  This file is the *second*, independently written restatement
(4x4 homogeneous matrices, numpy.linalg.solve, xml.etree URDF reader); the first is the plain-C
oracle in oracle/ik_oracle.c.  The two are cross-checked against each other, against
scipy.linalg.expm/logm and finite differences (tests/test_oracle_*.py), and this twin emits the
committed golden vectors under tests/golden/ (tests/golden/make_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

What is restated (file:line relative to the reference root):
  * ik/ik/dls.cpp:5-78        -- dls(): loop order, lambda^2 damping, stop-before-step, clamp-after-step
  * ik/ik/data.cpp:25-58      -- evaluate_problem_data(): FK, joint Jacobians, per-task e / J, weighting
  * ik/ik/frame.hpp:37-62     -- compute_frame_error(): e = log6(oMf^-1 * oMr * target)
  * ik/ik/frame.hpp:152-182   -- FrameTask::compute_jacobian(): J = -Jlog6(tMf) * J_local
  * ik/ik/common.hpp:53-56    -- apply_joint_clipping()
  * ik/ik/visitor.hpp:15-21   -- should_stop(): ||e[0]||^2 < 1e-4
Pinocchio/Eigen semantics (third-party, un-vendored, version unpinned by ik/ik/CMakeLists.txt:1-3)
follow SURVEY.md Appendix A.
"""
import math
import xml.etree.ElementTree as ET

import numpy as np

TAYLOR_PREC3 = float(np.finfo(np.float64).eps) ** 0.25  # TaylorSeriesExpansion<double>::precision<3>()
DBL_MAX = float(np.finfo(np.float64).max)

POSITION, ORIENTATION, FULL = 0, 1, 2  # ik::KinematicType (ik/ik/frame.hpp:20)
ALIGN_X, ALIGN_Y, ALIGN_Z = 3, 4, 5     # ik::AlignAxisTask, AlignAxisType X / Y / Z (ik/ik/frame.hpp:202)


class PostureTask:
    """ik::PostureTask (ik/ik/posture.hpp:17-85): e = (q.tail(nj) - target) * mask, J.rightCols(nj) = I (mask not applied)."""

    def __init__(self, m, nj, target=None, mask=None, weights=None):
        self.nj = nj
        self.dim = nj
        self.target = np.zeros(nj) if target is None else np.array(target, float)
        self.mask = np.ones(nj) if mask is None else np.array(mask, float)
        self.w = np.ones(nj) if weights is None else np.array(weights, float)


# ----------------------------------------------------------------------------------------------
# SE(3) helpers on 4x4 homogeneous matrices
# ----------------------------------------------------------------------------------------------
def se3(R, p):
    M = np.eye(4)
    M[:3, :3] = R
    M[:3, 3] = p
    return M


def se3_inv(M):
    R = M[:3, :3]
    Mi = np.eye(4)
    Mi[:3, :3] = R.T
    Mi[:3, 3] = -R.T @ M[:3, 3]
    return Mi


def skew(v):
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


def quat_from_rpy(r, p, y):
    """urdfdom Rotation::setFromRPY (x, y, z, w), normalised."""
    phi, the, psi = r / 2.0, p / 2.0, y / 2.0
    x = math.sin(phi) * math.cos(the) * math.cos(psi) - math.cos(phi) * math.sin(the) * math.sin(psi)
    yy = math.cos(phi) * math.sin(the) * math.cos(psi) + math.sin(phi) * math.cos(the) * math.sin(psi)
    z = math.cos(phi) * math.cos(the) * math.sin(psi) - math.sin(phi) * math.sin(the) * math.cos(psi)
    w = math.cos(phi) * math.cos(the) * math.cos(psi) + math.sin(phi) * math.sin(the) * math.sin(psi)
    n = math.sqrt(x * x + yy * yy + z * z + w * w)
    return x / n, yy / n, z / n, w / n


def quat_to_matrix(x, y, z, w):
    """Eigen::Quaternion::toRotationMatrix (no normalisation)."""
    tx, ty, tz = 2.0 * x, 2.0 * y, 2.0 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array([
        [1.0 - (tyy + tzz), txy - twz, txz + twy],
        [txy + twz, 1.0 - (txx + tzz), tyz - twx],
        [txz - twy, tyz + twx, 1.0 - (txx + tyy)],
    ])


def matrix_to_quat(R):
    """Eigen rotation matrix -> quaternion (x, y, z, w)."""
    t = R[0, 0] + R[1, 1] + R[2, 2]
    q = [0.0, 0.0, 0.0, 0.0]  # x y z w
    if t > 0.0:
        t = math.sqrt(t + 1.0)
        q[3] = 0.5 * t
        t = 0.5 / t
        q[0] = (R[2, 1] - R[1, 2]) * t
        q[1] = (R[0, 2] - R[2, 0]) * t
        q[2] = (R[1, 0] - R[0, 1]) * t
    else:
        i = 0
        if R[1, 1] > R[0, 0]:
            i = 1
        if R[2, 2] > R[i, i]:
            i = 2
        j = (i + 1) % 3
        k = (j + 1) % 3
        t = math.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q[i] = 0.5 * t
        t = 0.5 / t
        q[3] = (R[k, j] - R[j, k]) * t
        q[j] = (R[j, i] + R[i, j]) * t
        q[k] = (R[k, i] + R[i, k]) * t
    return q


def axis_rotation(axis, c, s):
    """Rotation by angle (cos c, sin s) about unit axis (Rodrigues); exact for aligned axes."""
    a = np.asarray(axis, dtype=float)
    K = skew(a)
    return c * np.eye(3) + s * K + (1.0 - c) * np.outer(a, a)


# ----------------------------------------------------------------------------------------------
# Lie-group maps
# ----------------------------------------------------------------------------------------------
def log3(R):
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    if tr >= 3.0:
        theta = 0.0
    elif tr <= -1.0:
        theta = math.pi
    else:
        theta = math.acos((tr - 1.0) / 2.0)
    w = np.zeros(3)
    if theta >= math.pi - 1e-2:
        cphi = -(tr - 1.0) / 2.0
        beta = theta * theta / (1.0 + cphi)
        tmp = (np.diag(R) + cphi) * beta
        sg = (1.0 if R[2, 1] > R[1, 2] else -1.0,
              1.0 if R[0, 2] > R[2, 0] else -1.0,
              1.0 if R[1, 0] > R[0, 1] else -1.0)
        for k in range(3):
            w[k] = sg[k] * (math.sqrt(tmp[k]) if tmp[k] > 0.0 else 0.0)
    else:
        t = (theta / math.sin(theta) if theta > TAYLOR_PREC3 else 1.0) / 2.0
        w[0] = t * (R[2, 1] - R[1, 2])
        w[1] = t * (R[0, 2] - R[2, 0])
        w[2] = t * (R[1, 0] - R[0, 1])
    return w, theta


def log6(M):
    """Returns [v; w] (linear first), as Motion::toVector()."""
    R, p = M[:3, :3], M[:3, 3]
    w, t = log3(R)
    t2 = t * t
    if t < TAYLOR_PREC3:
        alpha = 1.0 - t2 / 12.0 - t2 * t2 / 720.0
        beta = 1.0 / 12.0 + t2 / 720.0
    else:
        st, ct = math.sin(t), math.cos(t)
        alpha = t * st / (2.0 * (1.0 - ct))
        beta = 1.0 / t2 - st / (2.0 * t * (1.0 - ct))
    v = alpha * p - 0.5 * np.cross(w, p) + (beta * np.dot(w, p)) * w
    return np.concatenate([v, w])


def Jlog3(theta, w):
    if theta < TAYLOR_PREC3:
        alpha = 1.0 / 12.0 + theta * theta / 720.0
        diag = 0.5 * (2.0 - theta * theta / 6.0)
    else:
        st, ct = math.sin(theta), math.cos(theta)
        st_1mct = st / (1.0 - ct)
        alpha = 1.0 / (theta * theta) - st_1mct / (2.0 * theta)
        diag = 0.5 * (theta * st_1mct)
    return alpha * np.outer(w, w) + diag * np.eye(3) + skew(0.5 * w)


def Jlog6(M):
    R, p = M[:3, :3], M[:3, 3]
    w, t = log3(R)
    A = Jlog3(t, w)
    t2 = t * t
    if t < TAYLOR_PREC3:
        beta = 1.0 / 12.0 + t2 / 720.0
        bdot = 1.0 / 360.0
    else:
        tinv = 1.0 / t
        t2inv = tinv * tinv
        st, ct = math.sin(t), math.cos(t)
        inv_2_2ct = 1.0 / (2.0 * (1.0 - ct))
        beta = t2inv - st * tinv * inv_2_2ct
        bdot = -2.0 * t2inv * t2inv + (1.0 + st * tinv) * t2inv * inv_2_2ct
    wTp = float(np.dot(w, p))
    v3 = (bdot * wTp) * w - (t2 * bdot + 2.0 * beta) * p
    C = np.outer(v3, w) + beta * np.outer(w, p) + (wTp * beta) * np.eye(3) + skew(0.5 * p)
    J = np.zeros((6, 6))
    J[:3, :3] = A
    J[3:, 3:] = A
    J[:3, 3:] = C @ A
    return J


def exp3(w):
    t2 = float(np.dot(w, w))
    t = math.sqrt(t2)
    if t < TAYLOR_PREC3:
        a_wxv, a_v, diag = 0.5 - t2 / 24.0, 1.0 - t2 / 6.0, 1.0 - t2 / 2.0
    else:
        a_wxv, a_v, diag = (1.0 - math.cos(t)) / t2, math.sin(t) / t, math.cos(t)
    return a_wxv * np.outer(w, w) + a_v * skew(w) + diag * np.eye(3)


def exp6(nu):
    v, w = np.asarray(nu[:3], float), np.asarray(nu[3:], float)
    t2 = float(np.dot(w, w))
    t = math.sqrt(t2)
    if t < TAYLOR_PREC3:
        a_wxv = 0.5 - t2 / 24.0
        a_v = 1.0 - t2 / 6.0
        a_w = 1.0 / 6.0 - t2 / 120.0
        diag = 1.0 - t2 / 2.0
    else:
        st, ct = math.sin(t), math.cos(t)
        a_wxv = (1.0 - ct) / t2
        a_v = st / t
        a_w = (1.0 - a_v) / t2
        diag = ct
    trans = a_v * v + (a_w * float(np.dot(w, v))) * w + a_wxv * np.cross(w, v)
    R = a_wxv * np.outer(w, w) + a_v * skew(w) + diag * np.eye(3)
    return se3(R, trans)


# ----------------------------------------------------------------------------------------------
# URDF -> model with Pinocchio's conventions
# ----------------------------------------------------------------------------------------------
J_UNIVERSE, J_REVOLUTE, J_PRISMATIC, J_FREEFLYER, J_REVOLUTE_UNBOUNDED = 0, 1, 2, 3, 4   # 4: a URDF "continuous" joint, q = (cos, sin)


class Model:
    pass


def _floats(s, n):
    v = [float(x) for x in s.split()]
    assert len(v) == n
    return v


def load_urdf(path_or_xml, free_flyer=False):
    if path_or_xml.lstrip().startswith("<"):
        root = ET.fromstring(path_or_xml)
    else:
        root = ET.parse(path_or_xml).getroot()
    links = [e.get("name") for e in root if e.tag == "link"]
    joints = {}
    for e in root:
        if e.tag != "joint":
            continue
        o = e.find("origin")
        xyz = _floats(o.get("xyz", "0 0 0"), 3) if o is not None else [0.0, 0.0, 0.0]
        rpy = _floats(o.get("rpy", "0 0 0"), 3) if o is not None else [0.0, 0.0, 0.0]
        ax = e.find("axis")
        axis = _floats(ax.get("xyz"), 3) if ax is not None else [1.0, 0.0, 0.0]
        lim = e.find("limit")
        lo = float(lim.get("lower", "0")) if lim is not None else 0.0
        hi = float(lim.get("upper", "0")) if lim is not None else 0.0
        joints[e.get("name")] = dict(
            name=e.get("name"), type=e.get("type"), parent=e.find("parent").get("link"),
            child=e.find("child").get("link"),
            M=se3(quat_to_matrix(*quat_from_rpy(*rpy)), xyz), axis=axis, lo=lo, hi=hi)
    children = {l: [] for l in links}
    has_parent = set()
    for name in sorted(joints):  # urdfdom keeps joints in a std::map: byte-lexicographic order
        j = joints[name]
        children[j["parent"]].append(j)
        has_parent.add(j["child"])
    roots = [l for l in links if l not in has_parent]
    assert len(roots) == 1, roots
    root_link = roots[0]

    m = Model()
    m.names = ["universe"]
    m.jtype = [J_UNIVERSE]
    m.parent = [0]
    m.placement = [np.eye(4)]
    m.axis = [np.zeros(3)]
    m.idx_q = [0]
    m.idx_v = [0]
    m.nq = 0
    m.nv = 0
    m.lower = []
    m.upper = []
    m.frames = [dict(name="universe", parent=0, placement=np.eye(4), type="fixed_joint")]

    def add_joint(parent, jtype, placement, name, axis, lo, hi):
        m.names.append(name)
        m.jtype.append(jtype)
        m.parent.append(parent)
        m.placement.append(placement)
        m.axis.append(np.asarray(axis, float))
        m.idx_q.append(m.nq)
        m.idx_v.append(m.nv)
        if jtype == J_FREEFLYER:
            m.nq += 7
            m.nv += 6
            m.lower += [-DBL_MAX] * 7
            m.upper += [DBL_MAX] * 7
        elif jtype == J_REVOLUTE_UNBOUNDED:   # Pinocchio's limits for a continuous joint
            m.nq += 2
            m.nv += 1
            m.lower += [-1.01] * 2
            m.upper += [1.01] * 2
        else:
            m.nq += 1
            m.nv += 1
            m.lower.append(lo)
            m.upper.append(hi)
        return len(m.names) - 1

    if free_flyer:
        jid = add_joint(0, J_FREEFLYER, np.eye(4), "root_joint", np.zeros(3), 0, 0)
        m.frames.append(dict(name="root_joint", parent=jid, placement=np.eye(4), type="joint"))
        m.frames.append(dict(name=root_link, parent=jid, placement=np.eye(4), type="body"))
    else:
        m.frames.append(dict(name="root_joint", parent=0, placement=np.eye(4), type="fixed_joint"))
        m.frames.append(dict(name=root_link, parent=0, placement=np.eye(4), type="body"))

    def frame_of_link(name):
        for i, f in enumerate(m.frames):
            if f["type"] == "body" and f["name"] == name:
                return i
        raise KeyError(name)

    def visit(link):
        pf = m.frames[frame_of_link(link)]
        for j in children[link]:
            if j["type"] == "fixed":
                pl = pf["placement"] @ j["M"]
                m.frames.append(dict(name=j["name"], parent=pf["parent"], placement=pl, type="fixed_joint"))
                m.frames.append(dict(name=j["child"], parent=pf["parent"], placement=pl, type="body"))
            elif j["type"] in ("revolute", "prismatic", "continuous"):
                a = np.asarray(j["axis"], float)
                a = a / np.linalg.norm(a) if not any(np.array_equal(a, e) for e in np.eye(3)) else a
                jt = {"revolute": J_REVOLUTE, "prismatic": J_PRISMATIC, "continuous": J_REVOLUTE_UNBOUNDED}[j["type"]]
                jid = add_joint(pf["parent"], jt, pf["placement"] @ j["M"], j["name"], a, j["lo"], j["hi"])
                m.frames.append(dict(name=j["name"], parent=jid, placement=np.eye(4), type="joint"))
                m.frames.append(dict(name=j["child"], parent=jid, placement=np.eye(4), type="body"))
            else:
                raise ValueError("unsupported joint type %r (%s)" % (j["type"], j["name"]))
            visit(j["child"])

    visit(root_link)
    m.njoints = len(m.names)
    m.lower = np.array(m.lower)
    m.upper = np.array(m.upper)
    # what pinocchio::centerOfMass reads of model.inertias[j]: total mass of the links welded to joint j and their common
    # centre of mass in the joint frame (appendBodyToJoint with each link's body placement)
    inertial = {}
    for e in root:
        if e.tag == "link" and e.find("inertial") is not None and e.find("inertial").find("mass") is not None:
            o = e.find("inertial").find("origin")
            inertial[e.get("name")] = (float(e.find("inertial").find("mass").get("value")),
                                       np.array(_floats(o.get("xyz", "0 0 0"), 3) if o is not None else [0.0, 0.0, 0.0]))
    m.mass = np.zeros(m.njoints)
    first_moment = np.zeros((m.njoints, 3))
    for f in m.frames:
        if f["type"] == "body" and f["name"] in inertial:
            mass, c = inertial[f["name"]]
            m.mass[f["parent"]] += mass
            first_moment[f["parent"]] += mass * (f["placement"][:3, :3] @ c + f["placement"][:3, 3])
    m.lever = np.where(m.mass[:, None] > 0, first_moment / np.where(m.mass > 0, m.mass, 1.0)[:, None], 0.0)
    return m


def centre_of_mass(m, q):
    """pinocchio::centerOfMass + jacobianCenterOfMass (ik/ik/data.cpp:31-34): data.com[0] and data.Jcom (3 x nv).
    Bodies welded to the universe (joint 0) do not count, as in Pinocchio."""
    oMi, _ = fk(m, q)
    Jw = joint_jacobians_world(m, oMi)
    sub_mass = m.mass.copy()
    sub_mass[0] = 0.0
    sub_first = np.array([m.mass[j] * (oMi[j][:3, :3] @ m.lever[j] + oMi[j][:3, 3]) for j in range(m.njoints)])
    sub_first[0] = 0.0
    for j in range(m.njoints - 1, 0, -1):
        sub_mass[m.parent[j]] += sub_mass[j]
        sub_first[m.parent[j]] += sub_first[j]
    M = sub_mass[0]
    Jcom = np.zeros((3, m.nv))
    for j in range(1, m.njoints):
        n = 6 if m.jtype[j] == J_FREEFLYER else 1
        for c in range(m.idx_v[j], m.idx_v[j] + n):
            Jcom[:, c] = (sub_mass[j] * Jw[:3, c] - np.cross(sub_first[j], Jw[3:, c])) / M
    return sub_first[0] / M, Jcom


class CentreOfMassTask:
    """ik::CentreOfMassTask (ik/ik/centre_of_mass.hpp:14-62): e = oMr^-1 com - target, J = R(oMr)^T Jcom."""

    def __init__(self, m, reference="universe", target=None, weights=None):
        self.reference = frame_id(m, reference)
        self.target = np.zeros(3) if target is None else np.array(target, float)
        self.w = np.ones(3) if weights is None else np.array(weights, float)


def frame_id(m, name):
    for i, f in enumerate(m.frames):
        if f["name"] == name:
            return i
    raise KeyError(name)


def neutral(m):
    q = np.zeros(m.nq)
    for j in range(1, m.njoints):
        if m.jtype[j] == J_FREEFLYER:
            q[m.idx_q[j] + 6] = 1.0
        if m.jtype[j] == J_REVOLUTE_UNBOUNDED:
            q[m.idx_q[j]] = 1.0
    return q


# ----------------------------------------------------------------------------------------------
# Kinematics
# ----------------------------------------------------------------------------------------------
def joint_transform(m, j, q):
    t = m.jtype[j]
    iq = m.idx_q[j]
    if t == J_REVOLUTE:
        return se3(axis_rotation(m.axis[j], math.cos(q[iq]), math.sin(q[iq])), np.zeros(3))
    if t == J_REVOLUTE_UNBOUNDED:
        return se3(axis_rotation(m.axis[j], q[iq], q[iq + 1]), np.zeros(3))
    if t == J_PRISMATIC:
        return se3(np.eye(3), m.axis[j] * q[iq])
    if t == J_FREEFLYER:
        x, y, z, w = q[iq + 3:iq + 7]
        return se3(quat_to_matrix(x, y, z, w), q[iq:iq + 3])
    raise ValueError


def fk(m, q):
    """framesForwardKinematics: all oMi, all oMf."""
    oMi = [np.eye(4)]
    for j in range(1, m.njoints):
        oMi.append(oMi[m.parent[j]] @ m.placement[j] @ joint_transform(m, j, q))
    oMf = [oMi[f["parent"]] @ f["placement"] for f in m.frames]
    return oMi, oMf


def joint_jacobians_world(m, oMi):
    """computeJointJacobians: 6 x nv, columns [v; w] expressed in the world frame."""
    J = np.zeros((6, m.nv))
    for j in range(1, m.njoints):
        R, p = oMi[j][:3, :3], oMi[j][:3, 3]
        iv = m.idx_v[j]
        if m.jtype[j] in (J_REVOLUTE, J_REVOLUTE_UNBOUNDED):
            w = R @ m.axis[j]
            J[:3, iv] = np.cross(p, w)
            J[3:, iv] = w
        elif m.jtype[j] == J_PRISMATIC:
            J[:3, iv] = R @ m.axis[j]
        elif m.jtype[j] == J_FREEFLYER:
            J[:3, iv:iv + 3] = R
            J[:3, iv + 3:iv + 6] = skew(p) @ R
            J[3:, iv + 3:iv + 6] = R
    return J


def support_columns(m, joint):
    cols = []
    j = joint
    while j > 0:
        n = 6 if m.jtype[j] == J_FREEFLYER else 1
        cols += list(range(m.idx_v[j], m.idx_v[j] + n))
        j = m.parent[j]
    return sorted(cols)


def frame_jacobian_local(m, Jw, oMf_f, joint):
    """getFrameJacobian(..., LOCAL): only the support columns are written."""
    J = np.zeros((6, m.nv))
    R, p = oMf_f[:3, :3], oMf_f[:3, 3]
    for c in support_columns(m, joint):
        v, w = Jw[:3, c], Jw[3:, c]
        J[:3, c] = R.T @ (v - np.cross(p, w))
        J[3:, c] = R.T @ w
    return J


def integrate(m, q, v):
    out = np.array(q, dtype=float)
    for j in range(1, m.njoints):
        iq, iv = m.idx_q[j], m.idx_v[j]
        if m.jtype[j] == J_FREEFLYER:
            quat = q[iq + 3:iq + 7]
            M0 = se3(quat_to_matrix(*quat), q[iq:iq + 3])
            M1 = M0 @ exp6(v[iv:iv + 6])
            out[iq:iq + 3] = M1[:3, 3]
            rq = np.array(matrix_to_quat(M1[:3, :3]))
            if float(np.dot(rq, quat)) < 0.0:
                rq = -rq
            rq = rq * ((3.0 - float(np.dot(rq, rq))) / 2.0)
            out[iq + 3:iq + 7] = rq
        elif m.jtype[j] == J_REVOLUTE_UNBOUNDED:
            cv, sv = math.cos(v[iv]), math.sin(v[iv])
            pair = np.array([cv * q[iq] - sv * q[iq + 1], sv * q[iq] + cv * q[iq + 1]])
            out[iq:iq + 2] = pair * ((3.0 - float(np.dot(pair, pair))) / 2.0)
        else:
            out[iq] = q[iq] + v[iv]
    return out


# ----------------------------------------------------------------------------------------------
# Tasks and the DLS loop
# ----------------------------------------------------------------------------------------------
class FrameTask:
    """ik::FrameTask (ik/ik/frame.hpp:78-200). target is a 4x4 w.r.t. the reference frame."""

    def __init__(self, m, frame, ktype=FULL, reference="universe", target=None, weights=None):
        self.frame = frame_id(m, frame)
        self.reference = frame_id(m, reference)
        self.type = ktype
        self.dim = 6 if ktype == FULL else (1 if ktype >= ALIGN_X else 3)
        self.target = np.eye(4) if target is None else np.array(target, float)
        self.w = np.ones(self.dim) if weights is None else np.array(weights, float)

    def rows(self):
        return {POSITION: slice(0, 3), ORIENTATION: slice(3, 6), FULL: slice(0, 6)}[self.type]


def evaluate(m, tasks, q):
    """evaluate_problem_data (ik/ik/data.cpp:25-58) for a single priority level."""
    oMi, oMf = fk(m, q)
    Jw = joint_jacobians_world(m, oMi)
    es, Js = [], []
    for t in tasks:
        if isinstance(t, CentreOfMassTask):
            com, Jcom = centre_of_mass(m, q)
            Mr = oMf[t.reference]
            es.append((Mr[:3, :3].T @ (com - Mr[:3, 3]) - t.target) * t.w)
            Js.append(t.w[:, None] * (Mr[:3, :3].T @ Jcom))
            continue
        if isinstance(t, PostureTask):
            J = np.zeros((t.nj, m.nv))
            J[:, m.nv - t.nj:] = np.eye(t.nj)
            es.append((q[m.nq - t.nj:] - t.target) * t.mask * t.w)
            Js.append(t.w[:, None] * J)
            continue
        if t.type >= ALIGN_X:  # AlignAxisTask (ik/ik/frame.hpp:257-301); target direction = t.target[:3, 3]
            rMf = se3_inv(oMf[t.reference]) @ oMf[t.frame]
            r = rMf[:3, t.type - ALIGN_X]
            tn = t.target[:3, 3] / np.linalg.norm(t.target[:3, 3])
            Jl = frame_jacobian_local(m, Jw, oMf[t.frame], m.frames[t.frame]["parent"])
            es.append(np.array([1.0 - r @ tn]) * t.w)
            Js.append(t.w[:, None] * (-(np.cross(r, tn)[None, :] @ rMf[:3, :3] @ Jl[3:, :])))
            continue
        oMt = oMf[t.reference] @ t.target
        fMt = se3_inv(oMf[t.frame]) @ oMt
        e = log6(fMt)[t.rows()]
        tMf = se3_inv(oMt) @ oMf[t.frame]
        Jl = frame_jacobian_local(m, Jw, oMf[t.frame], m.frames[t.frame]["parent"])
        J = (-Jlog6(tMf) @ Jl)[t.rows(), :]
        es.append(e * t.w)
        Js.append(t.w[:, None] * J)
    return np.concatenate(es), np.vstack(Js)


def clamp(m, q):
    return np.minimum(m.upper, np.maximum(q, m.lower))


def action_matrix(M):
    """pinocchio SE3::toActionMatrix: [[R, [p]x R], [0, R]] acting on [v; w]."""
    R, p = M[:3, :3], M[:3, 3]
    X = np.zeros((6, 6))
    X[:3, :3] = R
    X[:3, 3:] = skew(p) @ R
    X[3:, 3:] = R
    return X


def constraint_jacobian(m, constraints, q):
    """Stacked ik::FrameConstraint::compute_jacobian (ik/ik/frame.hpp:413-449; ik/ik/dls.cpp:26-34).  constraints are
    FrameTask-like records (frame, reference, type)."""
    oMi, oMf = fk(m, q)
    Jw = joint_jacobians_world(m, oMi)
    rows = []
    for c in constraints:
        rMf = se3_inv(oMf[c.reference]) @ oMf[c.frame]
        Jf = frame_jacobian_local(m, Jw, oMf[c.frame], m.frames[c.frame]["parent"])
        Jr = frame_jacobian_local(m, Jw, oMf[c.reference], m.frames[c.reference]["parent"])
        rows.append((Jf - action_matrix(se3_inv(rMf)) @ Jr)[c.rows(), :])
    return np.vstack(rows) if rows else np.zeros((0, m.nv))


def dls(m, tasks, q0, max_iterations=100, damping=1e-2, step_length=1.0, stop_sq_tol=1e-4, trace=None, constraints=None):
    """ik::dls (ik/ik/dls.cpp:5-78).  stop_sq_tol < 0 == a visitor that never stops.
    Returns (q, success, iterations-before-exit)."""
    q = np.array(q0, dtype=float)
    for i in range(max_iterations):
        e, J = evaluate(m, tasks, q)
        JJ = J @ J.T
        JJ[np.diag_indices_from(JJ)] += damping * damping
        dq = -(J.T @ np.linalg.solve(JJ, e))
        if constraints:  # ik/ik/dls.cpp:43-53: N = I - pinv(Jc) Jc
            Jc = constraint_jacobian(m, constraints, q)
            dq = dq - rowspace_projector(Jc) @ dq
        if trace is not None:
            trace.append(dict(q=q.copy(), e=e.copy(), J=J.copy(), JJ=JJ.copy(), dq=dq.copy()))
        if stop_sq_tol >= 0.0 and float(np.dot(e, e)) < stop_sq_tol:
            return q, True, i
        q = clamp(m, integrate(m, q, step_length * dq))
    return q, False, max_iterations


def damp_pseudoinverse(M, lam):
    """damp_pseudoinverse (ik/ik/pik.cpp:5-22)."""
    U, s, Vt = np.linalg.svd(M, full_matrices=False)
    res = np.zeros((M.shape[1], M.shape[0]))
    for i in range(s.size):
        res += (s[i] / (lam ** 2 + s[i] ** 2)) * np.outer(Vt[i], U[:, i])
    return res


def cod_rank(A):
    """Numerical rank as Eigen's CompleteOrthogonalDecomposition decides it: column-pivoted QR, pivots above
    eps * min(m, n) * max|pivot| (ColPivHouseholderQR::rank with the default threshold)."""
    import scipy.linalg
    if min(A.shape) == 0:
        return 0
    R = scipy.linalg.qr(A, mode="r", pivoting=True)[0]
    d = np.abs(np.diag(R))
    if d.max() == 0.0:
        return 0
    return int(np.sum(d > d.max() * np.finfo(float).eps * min(A.shape)))


def rowspace_projector(A):
    """Jbar.completeOrthogonalDecomposition().pseudoInverse() * Jbar (ik/ik/pik.cpp:59-61)."""
    r = cod_rank(A)
    Vt = np.linalg.svd(A, full_matrices=False)[2]
    return Vt[:r].T @ Vt[:r]


def pik(m, levels, q0, max_iterations=100, step_length=1.0, stop_sq_tol=1e-4, lam=None, da=None, trace=None):
    """ik::pik (ik/ik/pik.cpp:31-103).  levels: one task list per priority level; lam: damping factor per level
    (ik::pik_data::lambda, default 1.0 each, ik/ik/pik.hpp:24); da: ik::pik_data::da (default zero).
    Returns (q, success, iterations-before-exit)."""
    q = np.array(q0, dtype=float)
    lam = [1.0] * len(levels) if lam is None else list(lam)
    da = np.zeros(m.nv) if da is None else np.asarray(da, float)
    for it in range(max_iterations):
        eJ = [evaluate(m, lv, q) if lv else (np.zeros(0), np.zeros((0, m.nv))) for lv in levels]
        P = np.eye(m.nv)
        dq = np.zeros(m.nv)
        for (e, J), l in zip(eJ, lam):
            if e.size == 0:
                continue
            de_bar = e - J @ dq
            Jbar = J @ P
            dq = dq - damp_pseudoinverse(Jbar, l) @ de_bar
            P = P - rowspace_projector(Jbar)
        dq = dq + P @ da
        if trace is not None:
            trace.append(dict(q=q.copy(), e=np.concatenate([e for e, _ in eJ]), dq=dq.copy()))
        e0 = eJ[0][0]
        if stop_sq_tol >= 0.0 and float(np.dot(e0, e0)) < stop_sq_tol:
            return q, True, it
        q = clamp(m, integrate(m, q, step_length * dq))
    return q, False, max_iterations
