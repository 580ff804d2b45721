/*
 * ik_oracle.c -- plain-C CPU restatement of the reference IK hot path.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see ik_oracle.h for the full statement).
 *
 * "Faithful" variant: same loop order, whole-tree FK, every frame placed, dense 6 x nv joint
 * Jacobian, dense M x nv task Jacobian, dense M x M Gram, pivoted LDL^T, dense N = I mat-vec --
 * i.e. the work the reference does, minus its string look-ups and per-iteration heap traffic.
 *
 * Follows (reference root relative):
 *   ik/ik/dls.cpp:5-78, ik/ik/data.cpp:25-58, ik/ik/frame.hpp:37-62,152-182,
 *   ik/ik/common.hpp:47-56, ik/ik/visitor.hpp:15-21, ik/ik/dls.hpp:24-65, ik/ik/task.hpp:48-51.
 * Pinocchio / Eigen semantics: SURVEY.md Appendix A (third-party, not in the reference tree).
 */
#include "ik_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* TaylorSeriesExpansion<double>::precision<3>() = eps^(1/4) */
static double taylor_prec3(void) { return sqrt(sqrt(DBL_EPSILON)); }

/* ---------------------------------------------------------------- SE(3) as 12 doubles */
#define R_(M, i, j) ((M)[3 * (i) + (j)])
#define P_(M, i) ((M)[9 + (i)])

static void se3_identity(double *M) {
    memset(M, 0, 12 * sizeof(double));
    R_(M, 0, 0) = R_(M, 1, 1) = R_(M, 2, 2) = 1.0;
}

/* C = A * B  (SE3::act) */
static void se3_mul(const double *A, const double *B, double *C) {
    double T[12];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j)
            R_(T, i, j) = R_(A, i, 0) * R_(B, 0, j) + R_(A, i, 1) * R_(B, 1, j) + R_(A, i, 2) * R_(B, 2, j);
        P_(T, i) = P_(A, i) + R_(A, i, 0) * P_(B, 0) + R_(A, i, 1) * P_(B, 1) + R_(A, i, 2) * P_(B, 2);
    }
    memcpy(C, T, sizeof T);
}

/* C = A^-1 * B  (SE3::actInv) */
static void se3_inv_mul(const double *A, const double *B, double *C) {
    double T[12], d[3];
    for (int i = 0; i < 3; ++i) d[i] = P_(B, i) - P_(A, i);
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j)
            R_(T, i, j) = R_(A, 0, i) * R_(B, 0, j) + R_(A, 1, i) * R_(B, 1, j) + R_(A, 2, i) * R_(B, 2, j);
        P_(T, i) = R_(A, 0, i) * d[0] + R_(A, 1, i) * d[1] + R_(A, 2, i) * d[2];
    }
    memcpy(C, T, sizeof T);
}

static void cross3(const double *a, const double *b, double *c) {
    double x = a[1] * b[2] - a[2] * b[1];
    double y = a[2] * b[0] - a[0] * b[2];
    double z = a[0] * b[1] - a[1] * b[0];
    c[0] = x; c[1] = y; c[2] = z;
}

static double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

/* Eigen::Quaternion::toRotationMatrix, coefficients (x,y,z,w), not normalised */
static void quat_to_R(const double *q, double *M) {
    double x = q[0], y = q[1], z = q[2], w = q[3];
    double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    double twx = tx * w, twy = ty * w, twz = tz * w;
    double txx = tx * x, txy = ty * x, txz = tz * x;
    double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R_(M, 0, 0) = 1 - (tyy + tzz); R_(M, 0, 1) = txy - twz;       R_(M, 0, 2) = txz + twy;
    R_(M, 1, 0) = txy + twz;       R_(M, 1, 1) = 1 - (txx + tzz); R_(M, 1, 2) = tyz - twx;
    R_(M, 2, 0) = txz - twy;       R_(M, 2, 1) = tyz + twx;       R_(M, 2, 2) = 1 - (txx + tyy);
}

/* Eigen rotation matrix -> quaternion (x,y,z,w) */
static void R_to_quat(const double *M, double *q) {
    double t = R_(M, 0, 0) + R_(M, 1, 1) + R_(M, 2, 2);
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (R_(M, 2, 1) - R_(M, 1, 2)) * t;
        q[1] = (R_(M, 0, 2) - R_(M, 2, 0)) * t;
        q[2] = (R_(M, 1, 0) - R_(M, 0, 1)) * t;
    } else {
        int i = 0;
        if (R_(M, 1, 1) > R_(M, 0, 0)) i = 1;
        if (R_(M, 2, 2) > R_(M, i, i)) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(R_(M, i, i) - R_(M, j, j) - R_(M, k, k) + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (R_(M, k, j) - R_(M, j, k)) * t;
        q[j] = (R_(M, j, i) + R_(M, i, j)) * t;
        q[k] = (R_(M, k, i) + R_(M, i, k)) * t;
    }
}

/* ---------------------------------------------------------------- Lie-group maps (App. A.3, A.5) */
static void log3_(const double *M, double *w, double *theta_out) {
    const double tr = R_(M, 0, 0) + R_(M, 1, 1) + R_(M, 2, 2);
    double theta;
    if (tr >= 3.0) theta = 0.0;
    else if (tr <= -1.0) theta = M_PI;
    else theta = acos((tr - 1.0) / 2.0);
    if (theta >= M_PI - 1e-2) {
        const double cphi = -(tr - 1.0) / 2.0;
        const double beta = theta * theta / (1.0 + cphi);
        const double t0 = (R_(M, 0, 0) + cphi) * beta, t1 = (R_(M, 1, 1) + cphi) * beta,
                     t2 = (R_(M, 2, 2) + cphi) * beta;
        w[0] = (R_(M, 2, 1) > R_(M, 1, 2) ? 1.0 : -1.0) * (t0 > 0 ? sqrt(t0) : 0.0);
        w[1] = (R_(M, 0, 2) > R_(M, 2, 0) ? 1.0 : -1.0) * (t1 > 0 ? sqrt(t1) : 0.0);
        w[2] = (R_(M, 1, 0) > R_(M, 0, 1) ? 1.0 : -1.0) * (t2 > 0 ? sqrt(t2) : 0.0);
    } else {
        const double t = ((theta > taylor_prec3()) ? theta / sin(theta) : 1.0) / 2.0;
        w[0] = t * (R_(M, 2, 1) - R_(M, 1, 2));
        w[1] = t * (R_(M, 0, 2) - R_(M, 2, 0));
        w[2] = t * (R_(M, 1, 0) - R_(M, 0, 1));
    }
    *theta_out = theta;
}

void iko_log6(const double *M, double *out) {
    double w[3], t, wxp[3];
    const double *p = &P_(M, 0);
    log3_(M, w, &t);
    const double t2 = t * t;
    double alpha, beta;
    if (t < taylor_prec3()) {
        alpha = 1.0 - t2 / 12.0 - t2 * t2 / 720.0;
        beta = 1.0 / 12.0 + t2 / 720.0;
    } else {
        const double st = sin(t), ct = cos(t);
        alpha = t * st / (2.0 * (1.0 - ct));
        beta = 1.0 / t2 - st / (2.0 * t * (1.0 - ct));
    }
    cross3(w, p, wxp);
    const double bwp = beta * dot3(w, p);
    for (int i = 0; i < 3; ++i) {
        out[i] = alpha * p[i] - 0.5 * wxp[i] + bwp * w[i];
        out[3 + i] = w[i];
    }
}

static void add_skew(const double *v, double *A /*3x3 row-major*/) {
    A[1] -= v[2]; A[2] += v[1];
    A[3] += v[2]; A[5] -= v[0];
    A[6] -= v[1]; A[7] += v[0];
}

static void Jlog3_(double theta, const double *w, double *A) {
    double alpha, diag;
    if (theta < taylor_prec3()) {
        alpha = 1.0 / 12.0 + theta * theta / 720.0;
        diag = 0.5 * (2.0 - theta * theta / 6.0);
    } else {
        const double st = sin(theta), ct = cos(theta);
        const double st_1mct = st / (1.0 - ct);
        alpha = 1.0 / (theta * theta) - st_1mct / (2.0 * theta);
        diag = 0.5 * (theta * st_1mct);
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) A[3 * i + j] = alpha * w[i] * w[j];
    A[0] += diag; A[4] += diag; A[8] += diag;
    const double hw[3] = {0.5 * w[0], 0.5 * w[1], 0.5 * w[2]};
    add_skew(hw, A);
}

void iko_Jlog6(const double *M, double *J /*6x6 row-major*/) {
    double w[3], t, A[9], C[9], B[9];
    const double *p = &P_(M, 0);
    log3_(M, w, &t);
    Jlog3_(t, w, A);
    const double t2 = t * t;
    double beta, bdot;
    if (t < taylor_prec3()) {
        beta = 1.0 / 12.0 + t2 / 720.0;
        bdot = 1.0 / 360.0;
    } else {
        const double tinv = 1.0 / t, t2inv = tinv * tinv;
        const double st = sin(t), ct = cos(t);
        const double inv_2_2ct = 1.0 / (2.0 * (1.0 - ct));
        beta = t2inv - st * tinv * inv_2_2ct;
        bdot = -2.0 * t2inv * t2inv + (1.0 + st * tinv) * t2inv * inv_2_2ct;
    }
    const double wTp = dot3(w, p);
    double v3[3];
    for (int i = 0; i < 3; ++i) v3[i] = (bdot * wTp) * w[i] - (t2 * bdot + 2.0 * beta) * p[i];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C[3 * i + j] = v3[i] * w[j] + beta * w[i] * p[j];
    C[0] += wTp * beta; C[4] += wTp * beta; C[8] += wTp * beta;
    const double hp[3] = {0.5 * p[0], 0.5 * p[1], 0.5 * p[2]};
    add_skew(hp, C);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            B[3 * i + j] = C[3 * i] * A[j] + C[3 * i + 1] * A[3 + j] + C[3 * i + 2] * A[6 + j];
    memset(J, 0, 36 * sizeof(double));
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            J[6 * i + j] = A[3 * i + j];
            J[6 * i + 3 + j] = B[3 * i + j];
            J[6 * (3 + i) + 3 + j] = A[3 * i + j];
        }
}

void iko_exp6(const double *nu, double *M) {
    const double *v = nu, *w = nu + 3;
    const double t2 = dot3(w, w);
    const double t = sqrt(t2);
    double a_wxv, a_v, a_w, diag;
    if (t < taylor_prec3()) {
        a_wxv = 0.5 - t2 / 24.0;
        a_v = 1.0 - t2 / 6.0;
        a_w = 1.0 / 6.0 - t2 / 120.0;
        diag = 1.0 - t2 / 2.0;
    } else {
        const double st = sin(t), ct = cos(t);
        const double inv_t2 = 1.0 / t2;
        a_wxv = (1.0 - ct) * inv_t2;
        a_v = st / t;
        a_w = (1.0 - a_v) * inv_t2;
        diag = ct;
    }
    double wxv[3];
    cross3(w, v, wxv);
    const double awv = a_w * dot3(w, v);
    for (int i = 0; i < 3; ++i) P_(M, i) = a_v * v[i] + awv * w[i] + a_wxv * wxv[i];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R_(M, i, j) = a_wxv * w[i] * w[j];
    const double av[3] = {a_v * w[0], a_v * w[1], a_v * w[2]};
    add_skew(av, M);
    R_(M, 0, 0) += diag; R_(M, 1, 1) += diag; R_(M, 2, 2) += diag;
}

/* ---------------------------------------------------------------- kinematics (App. A.2) */
static void joint_transform(const iko_model *m, int j, const double *q, double *M) {
    const int iq = m->idx_q[j];
    const double *a = m->axis + 3 * j;
    se3_identity(M);
    switch (m->jtype[j]) {
        case IKO_JOINT_REVOLUTE:
        case IKO_JOINT_REVOLUTE_UNBOUNDED: {
            /* a continuous joint's configuration IS (cos, sin), used as given (JointModelRevoluteUnbounded::calc) */
            const int unb = m->jtype[j] == IKO_JOINT_REVOLUTE_UNBOUNDED;
            const double c = unb ? q[iq] : cos(q[iq]), s = unb ? q[iq + 1] : sin(q[iq]), k = 1.0 - c;
            /* Rodrigues; for an aligned axis every entry is exact */
            R_(M, 0, 0) = c + k * a[0] * a[0];
            R_(M, 0, 1) = k * a[0] * a[1] - s * a[2];
            R_(M, 0, 2) = k * a[0] * a[2] + s * a[1];
            R_(M, 1, 0) = k * a[1] * a[0] + s * a[2];
            R_(M, 1, 1) = c + k * a[1] * a[1];
            R_(M, 1, 2) = k * a[1] * a[2] - s * a[0];
            R_(M, 2, 0) = k * a[2] * a[0] - s * a[1];
            R_(M, 2, 1) = k * a[2] * a[1] + s * a[0];
            R_(M, 2, 2) = c + k * a[2] * a[2];
        } break;
        case IKO_JOINT_PRISMATIC:
            for (int i = 0; i < 3; ++i) P_(M, i) = a[i] * q[iq];
            break;
        case IKO_JOINT_FREEFLYER:
            quat_to_R(q + iq + 3, M);
            for (int i = 0; i < 3; ++i) P_(M, i) = q[iq + i];
            break;
        default: break;
    }
}

void iko_fk(const iko_model *m, const double *q, double *oMi_out, double *oMf_out) {
    double *oMi = oMi_out ? oMi_out : (double *)malloc(sizeof(double) * 12 * m->njoints);
    se3_identity(oMi);
    for (int j = 1; j < m->njoints; ++j) {
        double Mj[12], liMi[12];
        joint_transform(m, j, q, Mj);
        se3_mul(m->placement + 12 * j, Mj, liMi);
        se3_mul(oMi + 12 * m->parent[j], liMi, oMi + 12 * j);
    }
    if (oMf_out)
        for (int f = 0; f < m->nframes; ++f)
            se3_mul(oMi + 12 * m->frame_parent[f], m->frame_placement + 12 * f, oMf_out + 12 * f);
    if (!oMi_out) free(oMi);
}

/* computeJointJacobians: Jw[6][nv] row-major, columns [v; w] in the world frame */
static void joint_jacobians_world(const iko_model *m, const double *oMi, double *Jw) {
    const int nv = m->nv;
    memset(Jw, 0, sizeof(double) * 6 * nv);
    for (int j = 1; j < m->njoints; ++j) {
        const double *M = oMi + 12 * j;
        const double *a = m->axis + 3 * j;
        const int iv = m->idx_v[j];
        if (m->jtype[j] == IKO_JOINT_REVOLUTE || m->jtype[j] == IKO_JOINT_PRISMATIC || m->jtype[j] == IKO_JOINT_REVOLUTE_UNBOUNDED) {
            double Ra[3], v[3];
            for (int i = 0; i < 3; ++i) Ra[i] = R_(M, i, 0) * a[0] + R_(M, i, 1) * a[1] + R_(M, i, 2) * a[2];
            if (m->jtype[j] != IKO_JOINT_PRISMATIC) {
                cross3(&P_(M, 0), Ra, v);
                for (int i = 0; i < 3; ++i) { Jw[i * nv + iv] = v[i]; Jw[(3 + i) * nv + iv] = Ra[i]; }
            } else {
                for (int i = 0; i < 3; ++i) Jw[i * nv + iv] = Ra[i];
            }
        } else if (m->jtype[j] == IKO_JOINT_FREEFLYER) {
            /* Ad(oM1) = [[R, [p]x R], [0, R]] */
            for (int c = 0; c < 3; ++c) {
                double Rc[3] = {R_(M, 0, c), R_(M, 1, c), R_(M, 2, c)}, pxRc[3];
                cross3(&P_(M, 0), Rc, pxRc);
                for (int i = 0; i < 3; ++i) {
                    Jw[i * nv + iv + c] = Rc[i];
                    Jw[i * nv + iv + 3 + c] = pxRc[i];
                    Jw[(3 + i) * nv + iv + 3 + c] = Rc[i];
                }
            }
        }
    }
}

/* getFrameJacobian(LOCAL): writes only the columns in the support of `joint` into Jl[6][nv] */
static void frame_jacobian_local(const iko_model *m, const double *Jw, const double *oMf, int joint,
                                 double *Jl) {
    const int nv = m->nv;
    for (int j = joint; j > 0; j = m->parent[j]) {
        const int n = (m->jtype[j] == IKO_JOINT_FREEFLYER) ? 6 : 1;
        for (int c = m->idx_v[j]; c < m->idx_v[j] + n; ++c) {
            double v[3] = {Jw[c], Jw[nv + c], Jw[2 * nv + c]};
            double w[3] = {Jw[3 * nv + c], Jw[4 * nv + c], Jw[5 * nv + c]};
            double pxw[3];
            cross3(&P_(oMf, 0), w, pxw);
            for (int i = 0; i < 3; ++i) v[i] -= pxw[i];
            for (int i = 0; i < 3; ++i) {
                Jl[i * nv + c] = R_(oMf, 0, i) * v[0] + R_(oMf, 1, i) * v[1] + R_(oMf, 2, i) * v[2];
                Jl[(3 + i) * nv + c] = R_(oMf, 0, i) * w[0] + R_(oMf, 1, i) * w[1] + R_(oMf, 2, i) * w[2];
            }
        }
    }
}

static void rowspace_projector(const double *A, int m, int n, double *Pr);

static int task_dim(const iko_task *t) { /* align / posture rows: 1; frame position / orientation and centre of mass: 3 */
    return t->type == IKO_FULL ? 6 : (t->type >= IKO_ALIGN_X && t->type <= IKO_POSTURE_ROW ? 1 : 3);
}

/* pinocchio::centerOfMass + jacobianCenterOfMass(model, data, q, false) (data.cpp:31-34): data.com[0] and data.Jcom (3 x nv)
 * from the joint placements oMi and the world joint Jacobian Jw.  Backward pass over the tree: mass and first moment of
 * every subtree; column of joint j = (m_sub v_o - (m_sub c_sub) x w) / M with [v_o; w] the joint's world-frame spatial
 * motion.  Bodies welded to the universe (inertias[0]) do not count. */
static void centre_of_mass(const iko_model *m, const double *oMi, const double *Jw, double *com, double *Jcom) {
    const int nj = m->njoints, nv = m->nv;
    double *sm = (double *)calloc((size_t)nj, sizeof(double)), *sf = (double *)calloc((size_t)nj * 3, sizeof(double));
    for (int j = 1; j < nj; ++j) {
        const double *M = oMi + 12 * j, *c = m->lever + 3 * j;
        sm[j] = m->mass[j];
        for (int i = 0; i < 3; ++i)
            sf[3 * j + i] = m->mass[j] * (R_(M, i, 0) * c[0] + R_(M, i, 1) * c[1] + R_(M, i, 2) * c[2] + P_(M, i));
    }
    for (int j = nj - 1; j > 0; --j) {
        const int p = m->parent[j];
        sm[p] += sm[j];
        for (int i = 0; i < 3; ++i) sf[3 * p + i] += sf[3 * j + i];
    }
    const double Mt = sm[0];
    for (int i = 0; i < 3; ++i) com[i] = sf[i] / Mt;
    for (int i = 0; i < 3 * nv; ++i) Jcom[i] = 0.0;
    for (int j = 1; j < nj; ++j) {
        const int n = m->jtype[j] == IKO_JOINT_FREEFLYER ? 6 : 1;
        for (int c = m->idx_v[j]; c < m->idx_v[j] + n; ++c) {
            const double v[3] = {Jw[c], Jw[nv + c], Jw[2 * nv + c]}, w[3] = {Jw[3 * nv + c], Jw[4 * nv + c], Jw[5 * nv + c]};
            double fxw[3];
            cross3(sf + 3 * j, w, fxw);
            for (int i = 0; i < 3; ++i) Jcom[i * nv + c] = (sm[j] * v[i] - fxw[i]) / Mt;
        }
    }
    free(sm); free(sf);
}

int iko_task_rows(const iko_task *tasks, int ntasks) {
    int M = 0;
    for (int i = 0; i < ntasks; ++i) M += task_dim(&tasks[i]);
    return M;
}

typedef struct {
    double *oMi, *oMf, *Jw, *Jl, *et, *Jt, *JJ, *y, *dq, *tmp, *q, *qn;
    int *perm;
    int M;
} workspace;

static workspace *ws_new(const iko_model *m, int M) {
    workspace *w = (workspace *)calloc(1, sizeof *w);
    w->M = M;
    w->oMi = (double *)malloc(sizeof(double) * 12 * m->njoints);
    w->oMf = (double *)malloc(sizeof(double) * 12 * m->nframes);
    w->Jw = (double *)malloc(sizeof(double) * 6 * m->nv);
    w->Jl = (double *)calloc(6 * m->nv, sizeof(double)); /* zeroed once: frame.hpp:110 */
    w->et = (double *)malloc(sizeof(double) * M);
    w->Jt = (double *)malloc(sizeof(double) * M * m->nv);
    w->JJ = (double *)malloc(sizeof(double) * M * M);
    w->y = (double *)malloc(sizeof(double) * M);
    w->dq = (double *)malloc(sizeof(double) * m->nv);
    w->tmp = (double *)malloc(sizeof(double) * (m->nv > M ? m->nv : M));
    w->q = (double *)malloc(sizeof(double) * m->nq);
    w->qn = (double *)malloc(sizeof(double) * m->nq);
    w->perm = (int *)malloc(sizeof(int) * M);
    return w;
}

static void ws_free(workspace *w) {
    free(w->oMi); free(w->oMf); free(w->Jw); free(w->Jl); free(w->et); free(w->Jt); free(w->JJ);
    free(w->y); free(w->dq); free(w->tmp); free(w->q); free(w->qn); free(w->perm); free(w);
}

/* evaluate_problem_data + stacking. Each task owns a frame-Jacobian scratch in the reference
 * (frame.hpp:199, zeroed once); columns outside the support stay zero, so a shared scratch
 * re-zeroed per task is equivalent. Returns ||e[0]||^2 of the priority-0 rows via *e0sq. */
static void evaluate_ws(const iko_model *m, const iko_task *tasks, int ntasks, const double *targets,
                        const double *q, workspace *w, double *e0sq) {
    const int nv = m->nv;
    iko_fk(m, q, w->oMi, w->oMf);                 /* data.cpp:28-29 */
    joint_jacobians_world(m, w->oMi, w->Jw);      /* data.cpp:30 */
    int maxp = 0;
    for (int i = 0; i < ntasks; ++i) if (tasks[i].priority > maxp) maxp = tasks[i].priority;
    int row = 0;
    double acc0 = 0.0;
    for (int p = 0; p <= maxp; ++p) {             /* data.cpp:38; dls.cpp:20-24 */
        for (int ti = 0; ti < ntasks; ++ti) {
            const iko_task *t = &tasks[ti];
            if (t->priority != p) continue;
            if (t->type == IKO_COM) {
                /* ik::CentreOfMassTask (centre_of_mass.hpp:33-45): e = oMr.actInv(com) - target, J = R(oMr)^T Jcom; the
                 * target point rides in doubles 9..11 of the slot; then the task weights (data.cpp:49-50). */
                const double *oMr = w->oMf + 12 * t->reference;
                double com[3], d[3];
                double *Jcom = (double *)malloc(sizeof(double) * 3 * nv);
                centre_of_mass(m, w->oMi, w->Jw, com, Jcom);
                for (int i = 0; i < 3; ++i) d[i] = com[i] - P_(oMr, i);
                for (int r = 0; r < 3; ++r) {
                    const double wt = t->weight[r];
                    const double local = R_(oMr, 0, r) * d[0] + R_(oMr, 1, r) * d[1] + R_(oMr, 2, r) * d[2];
                    w->et[row + r] = (local - targets[12 * ti + 9 + r]) * wt;
                    if (p == 0) acc0 += w->et[row + r] * w->et[row + r];
                    for (int c = 0; c < nv; ++c)
                        w->Jt[(row + r) * nv + c] =
                            wt * (R_(oMr, 0, r) * Jcom[c] + R_(oMr, 1, r) * Jcom[nv + c] + R_(oMr, 2, r) * Jcom[2 * nv + c]);
                }
                free(Jcom);
                row += 3;
                continue;
            }
            if (t->type == IKO_POSTURE_ROW) {
                /* one row of ik::PostureTask (posture.hpp:51-68): e = (q_k - target_k) * mask_k, J = e_k^T (the mask is
                 * not applied to J: "todo - incorporate mask"), then the task weight on both (data.cpp:49-50).
                 * frame = column in the tangent vector, reference = index in q, weight[0] = w_k, weight[1] = mask_k,
                 * target_k rides in double 9 of the slot. */
                const double wt = t->weight[0];
                w->et[row] = (q[t->reference] - targets[12 * ti + 9]) * t->weight[1] * wt;
                if (p == 0) acc0 += w->et[row] * w->et[row];
                for (int c = 0; c < nv; ++c) w->Jt[row * nv + c] = 0.0;
                w->Jt[row * nv + t->frame] = wt;
                row += 1;
                continue;
            }
            const double *oMf = w->oMf + 12 * t->frame, *oMr = w->oMf + 12 * t->reference;
            double oMt[12], fMt[12], tMf[12], e6[6], Jlog[36];
            const int d = task_dim(t);
            memset(w->Jl, 0, sizeof(double) * 6 * nv);
            frame_jacobian_local(m, w->Jw, oMf, m->frame_parent[t->frame], w->Jl); /* frame.hpp:169 / :291 */
            if (t->type >= IKO_ALIGN_X) {
                /* AlignAxisTask (frame.hpp:257-301): e = 1 - r . t_hat, J = -(r x t_hat)^T R(rMf) J_local(angular) */
                double rMf[12], r[3], tn[3], rxt[3], g[3];
                const int ax = t->type - IKO_ALIGN_X;
                const double *tg = targets + 12 * ti + 9; /* the direction rides in the translation part */
                const double nrm = sqrt(tg[0] * tg[0] + tg[1] * tg[1] + tg[2] * tg[2]);
                se3_inv_mul(oMr, oMf, rMf);
                for (int i = 0; i < 3; ++i) { r[i] = R_(rMf, i, ax); tn[i] = tg[i] / nrm; }
                cross3(r, tn, rxt);
                for (int k = 0; k < 3; ++k) g[k] = rxt[0] * R_(rMf, 0, k) + rxt[1] * R_(rMf, 1, k) + rxt[2] * R_(rMf, 2, k);
                const double wt = t->weight[0];
                w->et[row] = (1.0 - dot3(r, tn)) * wt;
                if (p == 0) acc0 += w->et[row] * w->et[row];
                for (int c = 0; c < nv; ++c)
                    w->Jt[row * nv + c] = wt * -(g[0] * w->Jl[3 * nv + c] + g[1] * w->Jl[4 * nv + c] + g[2] * w->Jl[5 * nv + c]);
                row += d;
                continue;
            }
            se3_mul(oMr, targets + 12 * ti, oMt); /* frame.hpp:48 */
            se3_inv_mul(oMf, oMt, fMt);           /* frame.hpp:50 */
            iko_log6(fMt, e6);                    /* frame.hpp:53-61 */
            se3_inv_mul(oMt, oMf, tMf);           /* frame.hpp:162 */
            iko_Jlog6(tMf, Jlog);                 /* frame.hpp:165-166 */
            const int r0 = (t->type == IKO_ORIENTATION) ? 3 : 0;
            for (int r = 0; r < d; ++r) {         /* frame.hpp:173-181 */
                const double wt = t->weight[r];   /* data.cpp:49-50 */
                w->et[row + r] = e6[r0 + r] * wt;
                if (p == 0) acc0 += w->et[row + r] * w->et[row + r];
                for (int c = 0; c < nv; ++c) {
                    double s = 0.0;
                    for (int k = 0; k < 6; ++k) s += -Jlog[6 * (r0 + r) + k] * w->Jl[k * nv + c];
                    w->Jt[(row + r) * nv + c] = wt * s;
                }
            }
            row += d;
        }
    }
    if (e0sq) *e0sq = acc0;
}

void iko_evaluate(const iko_model *m, const iko_task *tasks, int ntasks, const double *targets,
                  const double *q, double *et, double *Jt) {
    const int M = iko_task_rows(tasks, ntasks);
    workspace *w = ws_new(m, M);
    evaluate_ws(m, tasks, ntasks, targets, q, w, NULL);
    memcpy(et, w->et, sizeof(double) * M);
    memcpy(Jt, w->Jt, sizeof(double) * M * m->nv);
    ws_free(w);
}

/* Eigen::LDLT restated: in-place lower LDL^T with symmetric pivoting on the largest |diagonal|,
 * then solve (dls.cpp:53 `JJ.ldlt().solve(et)`). A is M x M row-major, destroyed. */
static void ldlt_solve(double *A, int n, const double *b, double *x, int *perm, double *tmp) {
    for (int k = 0; k < n; ++k) {
        int piv = k;
        double big = fabs(A[k * n + k]);
        for (int i = k + 1; i < n; ++i)
            if (fabs(A[i * n + i]) > big) { big = fabs(A[i * n + i]); piv = i; }
        perm[k] = piv;
        if (piv != k) { /* symmetric swap of rows/cols k and piv in the lower triangle */
            for (int j = 0; j < k; ++j) { double t = A[k * n + j]; A[k * n + j] = A[piv * n + j]; A[piv * n + j] = t; }
            for (int i = piv + 1; i < n; ++i) { double t = A[i * n + k]; A[i * n + k] = A[i * n + piv]; A[i * n + piv] = t; }
            { double t = A[k * n + k]; A[k * n + k] = A[piv * n + piv]; A[piv * n + piv] = t; }
            for (int i = k + 1; i < piv; ++i) { double t = A[i * n + k]; A[i * n + k] = A[piv * n + i]; A[piv * n + i] = t; }
        }
        /* A[k][k] -= sum_j L[k][j]^2 D[j];  column k below the diagonal likewise, then / D[k] */
        for (int j = 0; j < k; ++j) tmp[j] = A[k * n + j] * A[j * n + j];
        double d = A[k * n + k];
        for (int j = 0; j < k; ++j) d -= A[k * n + j] * tmp[j];
        A[k * n + k] = d;
        for (int i = k + 1; i < n; ++i) {
            double s = A[i * n + k];
            for (int j = 0; j < k; ++j) s -= A[i * n + j] * tmp[j];
            A[i * n + k] = (fabs(d) > DBL_MIN) ? s / d : s;
        }
    }
    for (int i = 0; i < n; ++i) x[i] = b[i];
    for (int k = 0; k < n; ++k) if (perm[k] != k) { double t = x[k]; x[k] = x[perm[k]]; x[perm[k]] = t; }
    for (int i = 0; i < n; ++i) for (int j = 0; j < i; ++j) x[i] -= A[i * n + j] * x[j];
    for (int i = 0; i < n; ++i) x[i] = (fabs(A[i * n + i]) > DBL_MIN) ? x[i] / A[i * n + i] : 0.0;
    for (int i = n - 1; i >= 0; --i) for (int j = i + 1; j < n; ++j) x[i] -= A[j * n + i] * x[j];
    for (int k = n - 1; k >= 0; --k) if (perm[k] != k) { double t = x[k]; x[k] = x[perm[k]]; x[perm[k]] = t; }
}

/* pinocchio::integrate (App. A.5) */
void iko_integrate(const iko_model *m, const double *q, const double *v, double *out) {
    for (int j = 1; j < m->njoints; ++j) {
        const int iq = m->idx_q[j], iv = m->idx_v[j];
        if (m->jtype[j] == IKO_JOINT_FREEFLYER) {
            double M0[12], E[12], M1[12], rq[4];
            quat_to_R(q + iq + 3, M0);
            for (int i = 0; i < 3; ++i) P_(M0, i) = q[iq + i];
            iko_exp6(v + iv, E);
            se3_mul(M0, E, M1);
            for (int i = 0; i < 3; ++i) out[iq + i] = P_(M1, i);
            R_to_quat(M1, rq);
            double dot = 0.0;
            for (int i = 0; i < 4; ++i) dot += rq[i] * q[iq + 3 + i];
            if (dot < 0.0) for (int i = 0; i < 4; ++i) rq[i] = -rq[i];
            double n2 = 0.0;
            for (int i = 0; i < 4; ++i) n2 += rq[i] * rq[i];
            const double al = (3.0 - n2) / 2.0;
            for (int i = 0; i < 4; ++i) out[iq + 3 + i] = rq[i] * al;
        } else if (m->jtype[j] == IKO_JOINT_REVOLUTE_UNBOUNDED) {
            /* SpecialOrthogonalOperation<2>::integrate: rotate (cos, sin) by v, first-order renormalisation */
            const double cv = cos(v[iv]), sv = sin(v[iv]);
            const double c = cv * q[iq] - sv * q[iq + 1], s = sv * q[iq] + cv * q[iq + 1];
            const double k = (3.0 - (c * c + s * s)) / 2.0;
            out[iq] = c * k;
            out[iq + 1] = s * k;
        } else {
            out[iq] = q[iq] + v[iv];
        }
    }
}

/* ik::FrameConstraint::compute_jacobian (frame.hpp:413-449), stacked as dls.cpp:26-34 does: the velocity of the frame
 * relative to its reference frame, in the frame's local coordinates,
 *   Jc = J_frame(LOCAL) - rMf.toActionMatrixInverse() J_reference(LOCAL),  top / bottom / all rows by kinematic type.
 * Needs w->oMf and w->Jw of the current q (evaluate_ws). Jc is Mc x nv; Jf, Jr are 6 x nv scratch. */
static int constraint_rows(const iko_task *cons, int ncons) {
    int Mc = 0;
    for (int i = 0; i < ncons; ++i) Mc += task_dim(&cons[i]);
    return Mc;
}

static void constraint_jacobian_ws(const iko_model *m, const iko_task *cons, int ncons, const workspace *w, double *Jf, double *Jr,
                                   double *Jc) {
    const int nv = m->nv;
    int row = 0;
    for (int k = 0; k < ncons; ++k) {
        const iko_task *c = &cons[k];
        const double *oMf = w->oMf + 12 * c->frame, *oMr = w->oMf + 12 * c->reference;
        double fMr[12];
        memset(Jf, 0, sizeof(double) * 6 * nv);
        memset(Jr, 0, sizeof(double) * 6 * nv);
        frame_jacobian_local(m, w->Jw, oMf, m->frame_parent[c->frame], Jf);
        frame_jacobian_local(m, w->Jw, oMr, m->frame_parent[c->reference], Jr);
        se3_inv_mul(oMf, oMr, fMr); /* (rMf)^-1: its action matrix is [[R, [p]x R], [0, R]] */
        const int d = task_dim(c), r0 = (c->type == IKO_ORIENTATION) ? 3 : 0;
        for (int col = 0; col < nv; ++col) {
            double v[3], wv[3], Rv[3], Rw[3], pxRw[3], out[6];
            for (int i = 0; i < 3; ++i) { v[i] = Jr[i * nv + col]; wv[i] = Jr[(3 + i) * nv + col]; }
            for (int i = 0; i < 3; ++i) {
                Rv[i] = R_(fMr, i, 0) * v[0] + R_(fMr, i, 1) * v[1] + R_(fMr, i, 2) * v[2];
                Rw[i] = R_(fMr, i, 0) * wv[0] + R_(fMr, i, 1) * wv[1] + R_(fMr, i, 2) * wv[2];
            }
            cross3(fMr + 9, Rw, pxRw);
            for (int i = 0; i < 3; ++i) {
                out[i] = Jf[i * nv + col] - (Rv[i] + pxRw[i]);
                out[3 + i] = Jf[(3 + i) * nv + col] - Rw[i];
            }
            for (int r = 0; r < d; ++r) Jc[(row + r) * nv + col] = out[r0 + r];
        }
        row += d;
    }
}

void iko_constraint_jacobian(const iko_model *m, const iko_task *cons, int ncons, const double *q, double *Jc) {
    workspace *w = ws_new(m, 1);
    double *Jf = (double *)malloc(sizeof(double) * 6 * m->nv), *Jr = (double *)malloc(sizeof(double) * 6 * m->nv);
    iko_fk(m, q, w->oMi, w->oMf);
    joint_jacobians_world(m, w->oMi, w->Jw);
    constraint_jacobian_ws(m, cons, ncons, w, Jf, Jr, Jc);
    free(Jf); free(Jr);
    ws_free(w);
}

static iko_visitor g_visitor = {-1.0, 0, {0, 0, 0, 0, 0, 0, 0, 0}};
void iko_set_visitor(const iko_visitor *v) {
    if (v) g_visitor = *v;
    else { g_visitor.dq_sq_tol = -1.0; g_visitor.nlevels = 0; }
}

/* should_stop(ik, e, dq) of the visitor family above (visitor.hpp:15-21 is the member with nlevels == 0 and dq_sq_tol < 0) */
static int visitor_should_stop(const iko_task *tasks, int ntasks, const double *et, const double *dq, int nv, const iko_params *p, double e0sq) {
    int stop;
    if (g_visitor.nlevels > 0) {
        stop = 1;
        int row = 0, maxp = 0;
        for (int i = 0; i < ntasks; ++i) if (tasks[i].priority > maxp) maxp = tasks[i].priority;
        for (int l = 0; l <= maxp; ++l) {                      /* rows are stacked by priority, then insertion (dls.cpp:20-24) */
            double s = 0.0;
            for (int ti = 0; ti < ntasks; ++ti) {
                if (tasks[ti].priority != l) continue;
                const int d = task_dim(&tasks[ti]);
                for (int r = 0; r < d; ++r) s += et[row + r] * et[row + r];
                row += d;
            }
            if (l < g_visitor.nlevels && !(s < g_visitor.level_sq_tol[l])) stop = 0;
        }
    } else {
        stop = p->stop_sq_tol >= 0.0 && e0sq < p->stop_sq_tol;
    }
    if (!stop && g_visitor.dq_sq_tol >= 0.0) {
        double s = 0.0;
        for (int c = 0; c < nv; ++c) s += dq[c] * dq[c];
        stop = s < g_visitor.dq_sq_tol;
    }
    return stop;
}

static int dls_ws(const iko_model *m, const iko_task *tasks, int ntasks, const iko_task *cons, int ncons, const double *targets,
                  const double *q0, const iko_params *p, double *q_out, int *success, int *iters,
                  double *trace, workspace *w) {
    const int nq = m->nq, nv = m->nv, M = w->M;
    const int Mc = constraint_rows(cons, ncons);
    double *Jc = NULL, *N = NULL, *Jf = NULL, *Jr = NULL;
    if (Mc > 0) {
        Jc = (double *)malloc(sizeof(double) * Mc * nv);
        N = (double *)malloc(sizeof(double) * nv * nv);
        Jf = (double *)malloc(sizeof(double) * 6 * nv);
        Jr = (double *)malloc(sizeof(double) * 6 * nv);
    }
    int done = 0;
    memcpy(w->q, q0, sizeof(double) * nq);                     /* dls.cpp:8 */
    for (int it = 0; it < p->max_iterations; ++it) {           /* dls.cpp:14 */
        double e0sq;
        evaluate_ws(m, tasks, ntasks, targets, w->q, w, &e0sq); /* dls.cpp:16-24 */
        for (int i = 0; i < M; ++i)                             /* dls.cpp:39 */
            for (int j = 0; j < M; ++j) {
                double s = 0.0;
                for (int c = 0; c < nv; ++c) s += w->Jt[i * nv + c] * w->Jt[j * nv + c];
                w->JJ[i * M + j] = s;
            }
        for (int i = 0; i < M; ++i) w->JJ[i * M + i] += p->damping * p->damping; /* dls.cpp:41 */
        ldlt_solve(w->JJ, M, w->et, w->y, w->perm, w->tmp);    /* dls.cpp:53 */
        for (int c = 0; c < nv; ++c) {                          /* dls.cpp:52-53, N = I */
            double s = 0.0;
            for (int i = 0; i < M; ++i) s += w->Jt[i * nv + c] * w->y[i];
            w->dq[c] = -s;
        }
        if (Mc > 0) {                                           /* dls.cpp:26-34, 43-49: N = I - pinv(Jc) Jc, dq = -N (...) */
            constraint_jacobian_ws(m, cons, ncons, w, Jf, Jr, Jc);
            rowspace_projector(Jc, Mc, nv, N);
            for (int c = 0; c < nv; ++c) {
                double s = w->dq[c];
                for (int k = 0; k < nv; ++k) s -= N[c * nv + k] * w->dq[k];
                w->tmp[c] = s;
            }
            memcpy(w->dq, w->tmp, sizeof(double) * nv);
        }
        if (trace) {
            double *t = trace + (size_t)it * (nq + M + nv);
            memcpy(t, w->q, sizeof(double) * nq);
            memcpy(t + nq, w->et, sizeof(double) * M);
            memcpy(t + nq + M, w->dq, sizeof(double) * nv);
        }
        if (visitor_should_stop(tasks, ntasks, w->et, w->dq, nv, p, e0sq)) {  /* visitor.hpp:19; dls.cpp:61-64 */
            memcpy(q_out, w->q, sizeof(double) * nq);
            *success = 1;
            *iters = it;
            done = 1;
            break;
        }
        for (int c = 0; c < nv; ++c) w->tmp[c] = p->step_length * w->dq[c];
        iko_integrate(m, w->q, w->tmp, w->qn);                  /* dls.cpp:67-68 */
        for (int i = 0; i < nq; ++i) {                          /* common.hpp:53-56 */
            const double lo_clamped = w->qn[i] > m->lower[i] ? w->qn[i] : m->lower[i]; /* cwiseMax */
            w->q[i] = m->upper[i] < lo_clamped ? m->upper[i] : lo_clamped;             /* cwiseMin */
        }
    }
    if (!done) {
        memcpy(q_out, w->q, sizeof(double) * nq);               /* dls.cpp:76-77 */
        *success = 0;
        *iters = p->max_iterations;
    }
    free(Jc); free(N); free(Jf); free(Jr);
    return 0;
}

int iko_dls_constrained(const iko_model *m, const iko_task *tasks, int ntasks, const iko_task *cons, int ncons,
                        const double *targets, const double *q0, const iko_params *p, double *q_out, int *success, int *iters,
                        double *trace) {
    workspace *w = ws_new(m, iko_task_rows(tasks, ntasks));
    int rc = dls_ws(m, tasks, ntasks, cons, ncons, targets, q0, p, q_out, success, iters, trace, w);
    ws_free(w);
    return rc;
}

int iko_dls(const iko_model *m, const iko_task *tasks, int ntasks, const double *targets,
            const double *q0, const iko_params *p, double *q_out, int *success, int *iters,
            double *trace) {
    return iko_dls_constrained(m, tasks, ntasks, NULL, 0, targets, q0, p, q_out, success, iters, trace);
}

typedef struct {
    const iko_model *m; const iko_task *tasks; int ntasks; long b0, b1;
    const double *targets, *q0; const iko_params *p; double *q_out; unsigned char *success; int *iters;
    const iko_task *cons; int ncons;
} batch_job;

static void *batch_worker(void *arg) {
    batch_job *j = (batch_job *)arg;
    workspace *w = ws_new(j->m, iko_task_rows(j->tasks, j->ntasks));
    const int nq = j->m->nq;
    for (long b = j->b0; b < j->b1; ++b) {
        int ok = 0, it = 0;
        dls_ws(j->m, j->tasks, j->ntasks, j->cons, j->ncons, j->targets + (size_t)b * j->ntasks * 12, j->q0 + (size_t)b * nq,
               j->p, j->q_out + (size_t)b * nq, &ok, &it, NULL, w);
        if (j->success) j->success[b] = (unsigned char)ok;
        if (j->iters) j->iters[b] = it;
    }
    ws_free(w);
    return NULL;
}

int iko_dls_batch(const iko_model *m, const iko_task *tasks, int ntasks, long B, const double *targets,
                  const double *q0, const iko_params *p, double *q_out, unsigned char *success,
                  int *iters, int nthreads) {
    return iko_dls_batch_constrained(m, tasks, ntasks, NULL, 0, B, targets, q0, p, q_out, success, iters, nthreads);
}

int iko_dls_batch_constrained(const iko_model *m, const iko_task *tasks, int ntasks, const iko_task *cons, int ncons, long B,
                              const double *targets, const double *q0, const iko_params *p, double *q_out, unsigned char *success,
                              int *iters, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256];
    batch_job jobs[256];
    for (int t = 0; t < nthreads; ++t) {
        batch_job j = {m, tasks, ntasks, B * t / nthreads, B * (t + 1) / nthreads, targets, q0, p, q_out, success, iters, cons, ncons};
        jobs[t] = j;
    }
    if (nthreads == 1) { batch_worker(&jobs[0]); return 0; }
    for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, batch_worker, &jobs[t]);
    for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    return 0;
}

/* ---------------------------------------------------------------------------------------------------
 * ik::pik -- prioritised IK (ik/ik/pik.cpp:5-103)
 * ------------------------------------------------------------------------------------------------- */

/* Thin SVD by one-sided Jacobi (Hestenes): A (m x n, row-major) = U diag(s) V^T with k = min(m, n) columns.
 * Stands in for Eigen::JacobiSVD(ComputeThinU | ComputeThinV) at pik.cpp:8-9: singular values are unique and the
 * sum pik.cpp:12-19 forms from the triplets is invariant under the freedom left in the vectors. */
static void svd_thin(const double *A, int m, int n, double *U, double *s, double *V) {
    const int big = m >= n ? m : n, k = m >= n ? n : m;
    double *W = (double *)malloc(sizeof(double) * big * k); /* column j = W[j*big ..] */
    double *R = (double *)calloc((size_t)k * k, sizeof(double));
    for (int j = 0; j < k; ++j) {
        for (int i = 0; i < big; ++i) W[j * big + i] = (m >= n) ? A[i * n + j] : A[j * n + i];
        R[j * k + j] = 1.0; /* column j of the accumulated right rotations = R[j*k ..] */
    }
    for (int sweep = 0; sweep < 60; ++sweep) {
        int rotated = 0;
        for (int a = 0; a < k - 1; ++a)
            for (int b = a + 1; b < k; ++b) {
                double al = 0, be = 0, ga = 0;
                for (int i = 0; i < big; ++i) {
                    al += W[a * big + i] * W[a * big + i];
                    be += W[b * big + i] * W[b * big + i];
                    ga += W[a * big + i] * W[b * big + i];
                }
                if (ga == 0.0 || fabs(ga) <= DBL_EPSILON * sqrt(al * be)) continue;
                rotated = 1;
                const double zeta = (be - al) / (2.0 * ga);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                for (int i = 0; i < big; ++i) {
                    const double x = W[a * big + i], y = W[b * big + i];
                    W[a * big + i] = c * x - sn * y;
                    W[b * big + i] = sn * x + c * y;
                }
                for (int i = 0; i < k; ++i) {
                    const double x = R[a * k + i], y = R[b * k + i];
                    R[a * k + i] = c * x - sn * y;
                    R[b * k + i] = sn * x + c * y;
                }
            }
        if (!rotated) break;
    }
    for (int j = 0; j < k; ++j) {
        double nn = 0;
        for (int i = 0; i < big; ++i) nn += W[j * big + i] * W[j * big + i];
        s[j] = sqrt(nn);
        for (int i = 0; i < big; ++i) {
            const double w = s[j] > 0 ? W[j * big + i] / s[j] : 0.0;
            if (m >= n) U[i * k + j] = w; else V[i * k + j] = w;
        }
        for (int i = 0; i < k; ++i) {
            if (m >= n) V[i * k + j] = R[j * k + i]; else U[i * k + j] = R[j * k + i];
        }
    }
    free(W); free(R);
}

/* damp_pseudoinverse (pik.cpp:5-22): res (n x m) = sum_i sigma_i / (lambda^2 + sigma_i^2) v_i u_i^T */
static void damp_pseudoinverse(const double *A, int m, int n, double lambda, double *res) {
    const int k = m >= n ? n : m;
    double *U = (double *)malloc(sizeof(double) * m * k), *V = (double *)malloc(sizeof(double) * n * k);
    double *s = (double *)malloc(sizeof(double) * k);
    svd_thin(A, m, n, U, s, V);
    for (int i = 0; i < n * m; ++i) res[i] = 0.0;
    for (int i = 0; i < k; ++i) {
        const double f = s[i] / (lambda * lambda + s[i] * s[i]);
        for (int r = 0; r < n; ++r)
            for (int c = 0; c < m; ++c) res[r * m + c] += f * V[r * k + i] * U[c * k + i];
    }
    free(U); free(V); free(s);
}

/* Jbar.completeOrthogonalDecomposition().pseudoInverse() * Jbar (pik.cpp:59-61): the orthogonal projector onto the
 * row space of A, with the numerical rank Eigen's COD would find -- column-pivoted Householder QR
 * (ColPivHouseholderQR), pivots with |R_kk| > eps * min(m, n) * max|R_kk| counted, and Eigen's early "the rest is
 * exactly zero" cut on the remaining column norms.  Pr is n x n. */
static void rowspace_projector(const double *A, int m, int n, double *Pr) {
    const int k = m >= n ? n : m;
    double *Q = (double *)malloc(sizeof(double) * m * n);
    int *perm = (int *)malloc(sizeof(int) * n);
    double *diag = (double *)malloc(sizeof(double) * (k > 0 ? k : 1));
    memcpy(Q, A, sizeof(double) * m * n);
    for (int j = 0; j < n; ++j) perm[j] = j;
    double maxcol = 0.0;
    for (int j = 0; j < n; ++j) {
        double nn = 0;
        for (int i = 0; i < m; ++i) nn += Q[i * n + j] * Q[i * n + j];
        if (sqrt(nn) > maxcol) maxcol = sqrt(nn);
    }
    const double helper = (maxcol * DBL_EPSILON) * (maxcol * DBL_EPSILON) / (double)(m > 0 ? m : 1);
    int nonzero = k;
    double maxpivot = 0.0;
    for (int c = 0; c < k; ++c) {
        int piv = c;
        double best = -1.0;
        for (int j = c; j < n; ++j) {
            double nn = 0;
            for (int i = c; i < m; ++i) nn += Q[i * n + j] * Q[i * n + j];
            if (nn > best) { best = nn; piv = j; }
        }
        if (nonzero == k && best < helper * (double)(m - c)) nonzero = c;
        if (piv != c) {
            for (int i = 0; i < m; ++i) { const double t = Q[i * n + c]; Q[i * n + c] = Q[i * n + piv]; Q[i * n + piv] = t; }
            const int t = perm[c]; perm[c] = perm[piv]; perm[piv] = t;
        }
        /* Householder reflector on rows c.. of column c */
        double nn = 0;
        for (int i = c; i < m; ++i) nn += Q[i * n + c] * Q[i * n + c];
        const double nrm = sqrt(nn), x0 = Q[c * n + c];
        const double beta = x0 >= 0 ? -nrm : nrm;
        diag[c] = beta;
        if (fabs(beta) > maxpivot) maxpivot = fabs(beta);
        if (nrm > 0) {
            const double v0 = x0 - beta;
            double vv = v0 * v0;
            for (int i = c + 1; i < m; ++i) vv += Q[i * n + c] * Q[i * n + c];
            if (vv > 0)
                for (int j = c + 1; j < n; ++j) {
                    double d = v0 * Q[c * n + j];
                    for (int i = c + 1; i < m; ++i) d += Q[i * n + c] * Q[i * n + j];
                    const double f = 2.0 * d / vv;
                    Q[c * n + j] -= f * v0;
                    for (int i = c + 1; i < m; ++i) Q[i * n + j] -= f * Q[i * n + c];
                }
        }
        Q[c * n + c] = beta;
        for (int i = c + 1; i < m; ++i) Q[i * n + c] = 0.0; /* (the reflector is not needed afterwards) */
    }
    int rank = 0;
    for (int c = 0; c < nonzero; ++c)
        if (fabs(diag[c]) > maxpivot * DBL_EPSILON * (double)k) ++rank;
    /* rows 0..rank-1 of R, columns un-permuted, orthonormalised (modified Gram-Schmidt, applied twice) */
    double *Z = (double *)calloc((size_t)(rank > 0 ? rank : 1) * n, sizeof(double));
    for (int r = 0; r < rank; ++r)
        for (int j = r; j < n; ++j) Z[r * n + perm[j]] = Q[r * n + j];
    for (int r = 0; r < rank; ++r) {
        for (int pass = 0; pass < 2; ++pass)
            for (int q = 0; q < r; ++q) {
                double d = 0;
                for (int j = 0; j < n; ++j) d += Z[r * n + j] * Z[q * n + j];
                for (int j = 0; j < n; ++j) Z[r * n + j] -= d * Z[q * n + j];
            }
        double nn = 0;
        for (int j = 0; j < n; ++j) nn += Z[r * n + j] * Z[r * n + j];
        nn = sqrt(nn);
        for (int j = 0; j < n; ++j) Z[r * n + j] /= nn;
    }
    for (int i = 0; i < n * n; ++i) Pr[i] = 0.0;
    for (int r = 0; r < rank; ++r)
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) Pr[i * n + j] += Z[r * n + i] * Z[r * n + j];
    free(Q); free(perm); free(diag); free(Z);
}

void iko_damp_pseudoinverse(const double *A, int m, int n, double lambda, double *res) { damp_pseudoinverse(A, m, n, lambda, res); }
void iko_rowspace_projector(const double *A, int m, int n, double *Pr) { rowspace_projector(A, m, n, Pr); }

static int pik_ws(const iko_model *m, const iko_task *tasks, int ntasks, const double *targets, const double *q0,
                  const iko_pik_params *p, double *q_out, int *success, int *iters, double *trace, workspace *w) {
    const int nq = m->nq, nv = m->nv, M = w->M;
    int maxp = 0;
    for (int i = 0; i < ntasks; ++i) if (tasks[i].priority > maxp) maxp = tasks[i].priority;
    if (p->nlevels < maxp + 1) return -1;                       /* trailing levels without tasks are no-ops (pik.cpp:47) */
    int *row0 = (int *)calloc((size_t)maxp + 2, sizeof(int));
    for (int i = 0; i < ntasks; ++i) row0[tasks[i].priority + 1] += task_dim(&tasks[i]);
    for (int l = 0; l <= maxp; ++l) row0[l + 1] += row0[l];
    double *P = (double *)malloc(sizeof(double) * nv * nv), *Pr = (double *)malloc(sizeof(double) * nv * nv);
    double *Jbar = (double *)malloc(sizeof(double) * (M > 0 ? M : 1) * nv), *pinv = (double *)malloc(sizeof(double) * (M > 0 ? M : 1) * nv);
    double *de = (double *)malloc(sizeof(double) * (M > 0 ? M : 1));
    int rc = 0, done = 0;
    memcpy(w->q, q0, sizeof(double) * nq);                      /* pik.cpp:34 */
    for (int it = 0; it < p->max_iterations && !done; ++it) {   /* pik.cpp:39 */
        double e0sq;
        evaluate_ws(m, tasks, ntasks, targets, w->q, w, &e0sq); /* pik.cpp:41 */
        for (int i = 0; i < nv * nv; ++i) P[i] = 0.0;           /* pik.cpp:44-45 */
        for (int i = 0; i < nv; ++i) { P[i * nv + i] = 1.0; w->dq[i] = 0.0; }
        for (int l = 0; l <= maxp; ++l) {                       /* pik.cpp:47 */
            const int r0 = row0[l], ml = row0[l + 1] - row0[l];
            if (ml == 0) continue;
            const double *Jl = w->Jt + (size_t)r0 * nv;
            for (int r = 0; r < ml; ++r) {                      /* pik.cpp:49-51 */
                double s = w->et[r0 + r];
                for (int c = 0; c < nv; ++c) s -= Jl[r * nv + c] * w->dq[c];
                de[r] = s;
                for (int c = 0; c < nv; ++c) {
                    double a = 0.0;
                    for (int k = 0; k < nv; ++k) a += Jl[r * nv + k] * P[k * nv + c];
                    Jbar[r * nv + c] = a;
                }
            }
            damp_pseudoinverse(Jbar, ml, nv, p->lambda[l], pinv); /* pik.cpp:54-55 */
            for (int c = 0; c < nv; ++c) {
                double s = 0.0;
                for (int r = 0; r < ml; ++r) s += pinv[c * ml + r] * de[r];
                w->dq[c] -= s;
            }
            rowspace_projector(Jbar, ml, nv, Pr);               /* pik.cpp:58-61 */
            for (int i = 0; i < nv * nv; ++i) P[i] -= Pr[i];
        }
        if (p->da)                                              /* pik.cpp:65 */
            for (int c = 0; c < nv; ++c) {
                double s = 0.0;
                for (int k = 0; k < nv; ++k) s += P[c * nv + k] * p->da[k];
                w->dq[c] += s;
            }
        if (trace) {
            double *t = trace + (size_t)it * (nq + M + nv);
            memcpy(t, w->q, sizeof(double) * nq);
            memcpy(t + nq, w->et, sizeof(double) * M);
            memcpy(t + nq + M, w->dq, sizeof(double) * nv);
        }
        if (p->stop_sq_tol >= 0.0 && e0sq < p->stop_sq_tol) {   /* visitor.hpp:19; pik.cpp:67-70 */
            memcpy(q_out, w->q, sizeof(double) * nq);
            *success = 1;
            *iters = it;
            done = 1;
            break;
        }
        for (int c = 0; c < nv; ++c) w->tmp[c] = p->step_length * w->dq[c];
        iko_integrate(m, w->q, w->tmp, w->qn);                   /* pik.cpp:73-74 */
        for (int i = 0; i < nq; ++i) {                           /* pik.cpp:77 */
            const double lo_clamped = w->qn[i] > m->lower[i] ? w->qn[i] : m->lower[i];
            w->q[i] = m->upper[i] < lo_clamped ? m->upper[i] : lo_clamped;
        }
    }
    if (!done) {
        memcpy(q_out, w->q, sizeof(double) * nq);                /* pik.cpp:99-100 */
        *success = 0;
        *iters = p->max_iterations;
    }
    free(row0); free(P); free(Pr); free(Jbar); free(pinv); free(de);
    return rc;
}

int iko_pik(const iko_model *m, const iko_task *tasks, int ntasks, const double *targets, const double *q0,
            const iko_pik_params *p, double *q_out, int *success, int *iters, double *trace) {
    const int M = iko_task_rows(tasks, ntasks);
    workspace *w = ws_new(m, M);
    int rc = pik_ws(m, tasks, ntasks, targets, q0, p, q_out, success, iters, trace, w);
    ws_free(w);
    return rc;
}

typedef struct {
    const iko_model *m; const iko_task *tasks; int ntasks; long b0, b1;
    const double *targets, *q0; const iko_pik_params *p; double *q_out; unsigned char *success; int *iters; int rc;
} pik_job;

static void *pik_worker(void *arg) {
    pik_job *j = (pik_job *)arg;
    workspace *w = ws_new(j->m, iko_task_rows(j->tasks, j->ntasks));
    const int nq = j->m->nq;
    for (long b = j->b0; b < j->b1; ++b) {
        int ok = 0, it = 0;
        if (pik_ws(j->m, j->tasks, j->ntasks, j->targets + (size_t)b * j->ntasks * 12, j->q0 + (size_t)b * nq, j->p,
                   j->q_out + (size_t)b * nq, &ok, &it, NULL, w) != 0) { j->rc = -1; break; }
        if (j->success) j->success[b] = (unsigned char)ok;
        if (j->iters) j->iters[b] = it;
    }
    ws_free(w);
    return NULL;
}

int iko_pik_batch(const iko_model *m, const iko_task *tasks, int ntasks, long B, const double *targets, const double *q0,
                  const iko_pik_params *p, double *q_out, unsigned char *success, int *iters, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256];
    pik_job jobs[256];
    for (int t = 0; t < nthreads; ++t) {
        pik_job j = {m, tasks, ntasks, B * t / nthreads, B * (t + 1) / nthreads, targets, q0, p, q_out, success, iters, 0};
        jobs[t] = j;
    }
    if (nthreads == 1) { pik_worker(&jobs[0]); return jobs[0].rc; }
    for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, pik_worker, &jobs[t]);
    int rc = 0;
    for (int t = 0; t < nthreads; ++t) { pthread_join(th[t], NULL); if (jobs[t].rc) rc = jobs[t].rc; }
    return rc;
}

void iko_fk_batch(const iko_model *m, long B, const double *q, const int *frames, int nsel, double *out) {
    double *oMi = (double *)malloc(sizeof(double) * 12 * m->njoints);
    double *oMf = (double *)malloc(sizeof(double) * 12 * m->nframes);
    for (long b = 0; b < B; ++b) {
        iko_fk(m, q + (size_t)b * m->nq, oMi, oMf);
        for (int s = 0; s < nsel; ++s)
            memcpy(out + ((size_t)b * nsel + s) * 12, oMf + 12 * frames[s], 12 * sizeof(double));
    }
    free(oMi); free(oMf);
}
