// generic_solver.hpp -- ik::dls() for ANY model tree and task list the path can express, one problem per
// lane, as a straight SIMT rendering of the reference loop (reference ik/ik/dls.cpp:5-78, ik/ik/data.cpp:25-58,
// ik/ik/frame.hpp:37-62,152-182 and :257-301): whole-tree FK, world joint Jacobian, per-task error and dense
// M x nv Jacobian rows, dense Gram, Cholesky, step, SE(3)/vector integrate, clamp.
//
// It is the FALLBACK behind the register-resident specialisations (chain_solver.hpp, tree_solver.hpp): those
// cover the benchmark shapes; this one covers everything else -- fixed-base multi-task problems, chains that
// share joints, reference frames that move with q (the reference's Jacobian ignores that motion,
// ik/ik/frame.hpp:152-182, and so does this), prismatic joints, AlignAxisTask rows -- at memory-bound speed.
// Every lane runs the same control flow (model and task table are shared), so there is no divergence; the
// per-lane workspace lives in HBM as [word][lane], which makes every access of a wave a coalesced 512-byte
// transaction.  Sizes are runtime values; nothing here is unrolled.
#pragma once
#if !defined(__HIPCC_RTC__)
#include <cstdint>
#endif

#include "chain_solver.hpp"
#include "tree_solver.hpp"

namespace ikdev {

enum : int { GJ_UNIVERSE = 0, GJ_REVOLUTE = 1, GJ_PRISMATIC = 2, GJ_FREEFLYER = 3, GJ_REVOLUTE_UNBOUNDED = 4 };  // == ikgpu_joint_type

// pinocchio::integrate on a "continuous" joint (SpecialOrthogonalOperation<2>): the (cos, sin) pair rotated by v, then
// renormalised to first order -- the same (3 - |.|^2) / 2 factor as the free-flyer's quaternion.
IKD_FN void unbounded_integrate(double c0, double s0, double v, double &c1, double &s1) {
    double sv, cv;
    dsincos(v, sv, cv);
    const double c = dfma(cv, c0, -(sv * s0)), s = dfma(sv, c0, cv * s0);
    const double k = (3.0 - dfma(c, c, s * s)) * 0.5;
    c1 = c * k;
    s1 = s * k;
}
enum : int { GT_POSITION = 0, GT_ORIENTATION = 1, GT_FULL = 2, GT_ALIGN_X = 3, GT_POSTURE_ROW = 6, GT_COM = 7 };  // == ikgpu_kinematic_type

// Read-only tables shared by all lanes (device global memory; host memory in the lane emulator).  DP / IP: the pointer types --
// plain pointers (host, and the cooperative kernels, which re-point them at LDS copies), or pointers into the constant address
// space (GenericTablesK, the per-lane kernels): every table read there has a wave-uniform index, so it becomes a scalar load
// (s_load into SGPRs, ~100 cycles from the scalar cache) instead of 64 lanes fetching one address through the vector memory
// path -- measured at ~1 us per dependent load with one wave per SIMD, which was 3/4 of the per-lane program's time.
template <class DP, class IP>
struct GenericTablesT {
    int njoints, nq, nv, ntasks, M;
    IP jtype, parent, idx_q, idx_v;  // [njoints]
    DP placement;                    // [njoints][12]
    DP axis;                         // [njoints][3]
    DP lower, upper;                // [nq]
    IP t_type, t_fjoint, t_rjoint, t_row, t_dim, t_prio;  // [ntasks]
    DP t_fpl, t_rpl;                // [ntasks][12] frame / reference placement on their joints
    DP t_w;                          // [ntasks][6]
    // workspace layout, in doubles per lane
    int off_q, off_oMi, off_Jw, off_e, off_J, off_G, off_y, off_dq, ws_words;
    // prioritised IK (pik_solver.hpp): rows of level l are [lvl_row0[l], lvl_row0[l + 1]); its workspace extends the one above
    int nlevels;
    IP lvl_row0;                        // [nlevels + 1]
    int off_P, off_Jb, off_de, ws_words_pik;
    // ik::FrameConstraint rows (ik/ik/frame.hpp:325-449), projected out of the DLS step (ik/ik/dls.cpp:26-34,43-53)
    int ncons, Mc;
    IP c_type, c_fjoint, c_rjoint, c_row, c_dim;  // [ncons]
    DP c_fpl, c_rpl;                               // [ncons][12]
    int off_Jc;                                                // Mc x nv, inside the first ws_words words
    // ik::CentreOfMassTask (ik/ik/centre_of_mass.hpp:14-62): mass and lever of the bodies on each joint, subtree masses
    int has_com;
    DP j_mass, j_lever, j_submass;                // [njoints], [njoints][3], [njoints]
    double inv_total_mass;
    int off_sf;                                                // first moments of the subtrees, 3 x njoints
};
using GenericTables = GenericTablesT<const double *, const int *>;
using GenericTablesK = GenericTablesT<const IKD_CONST_AS double *, const IKD_CONST_AS int *>;  // (the same layout)


struct Ws {  // word w of this lane
    double *base;
    int64_t stride;
    IKD_FN double &operator[](int w) const { return base[static_cast<int64_t>(w) * stride]; }
};

// The same lane program specialised at run time (rtc_generic.cpp): the tables are `static constexpr` members of a generated type, the
// workspace is a LOCAL array -- with every loop fully unrolled all its indices are constants, so it is promoted to registers, the
// structural zeros of the dense M x nv Jacobian fold away (the translation unit is compiled with -fno-signed-zeros -fno-honor-nans
// -fno-honor-infinities, as kernels_hot.hip) and everything the task list never reads (joints outside every support, unused
// Jacobian columns) is dead code.  IKD_UNROLL marks the loops that must unroll for that; it expands to nothing in the ordinary builds.
struct WsReg {
    double *w;
    IKD_FN double &operator[](int i) const { return w[i]; }
};
#ifndef IKD_UNROLL
#ifdef IKD_STATIC_TABLES
#define IKD_UNROLL _Pragma("unroll")
#else
#define IKD_UNROLL
#endif
#endif

template <class PA, class PB>
IKD_FN void g_se3_mul(PA A, PB B, double *C) {  // C = A * B, C may not alias
    IKD_UNROLL
    for (int i = 0; i < 3; ++i) {
        IKD_UNROLL
        for (int j = 0; j < 3; ++j) C[3 * i + j] = dfma(A[3 * i], B[j], dfma(A[3 * i + 1], B[3 + j], A[3 * i + 2] * B[6 + j]));
        C[9 + i] = dfma(A[3 * i], B[9], dfma(A[3 * i + 1], B[10], dfma(A[3 * i + 2], B[11], A[9 + i])));
    }
}

template <class PA, class PB>
IKD_FN void g_se3_inv_mul(PA A, PB B, double *C) {  // C = A^-1 * B
    const double d[3] = {B[9] - A[9], B[10] - A[10], B[11] - A[11]};
    IKD_UNROLL
    for (int i = 0; i < 3; ++i) {
        IKD_UNROLL
        for (int j = 0; j < 3; ++j) C[3 * i + j] = dfma(A[i], B[j], dfma(A[3 + i], B[3 + j], A[6 + i] * B[6 + j]));
        C[9 + i] = dfma(A[i], d[0], dfma(A[3 + i], d[1], A[6 + i] * d[2]));
    }
}

// framesForwardKinematics (joints) + computeJointJacobians (ik/ik/data.cpp:28-30) into the workspace: q -> oMi, Jw.
template <class TB, class WS>
IKD_FN void generic_fk(const TB &T, const WS &ws) {
    {
        const double I[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
        IKD_UNROLL
        for (int k = 0; k < 12; ++k) ws[T.off_oMi + k] = I[k];
    }
    IKD_UNROLL
    for (int j = 1; j < T.njoints; ++j) {
        double Mj[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0}, li[12], oP[12], oM[12];
        const int iq = T.idx_q[j], iv = T.idx_v[j], jt = T.jtype[j];
        const auto a = T.axis + 3 * j;
        if (jt == GJ_REVOLUTE || jt == GJ_REVOLUTE_UNBOUNDED) {
            double s, c;
            if (jt == GJ_REVOLUTE) dsincos(ws[T.off_q + iq], s, c);
            else { c = ws[T.off_q + iq]; s = ws[T.off_q + iq + 1]; }   // a continuous joint's configuration IS (cos, sin), used as given
            const double k = 1.0 - c;  // Rodrigues; exact entries for an aligned axis
            Mj[0] = c + k * a[0] * a[0];        Mj[1] = k * a[0] * a[1] - s * a[2]; Mj[2] = k * a[0] * a[2] + s * a[1];
            Mj[3] = k * a[1] * a[0] + s * a[2]; Mj[4] = c + k * a[1] * a[1];        Mj[5] = k * a[1] * a[2] - s * a[0];
            Mj[6] = k * a[2] * a[0] - s * a[1]; Mj[7] = k * a[2] * a[1] + s * a[0]; Mj[8] = c + k * a[2] * a[2];
        } else if (jt == GJ_PRISMATIC) {
            const double v = ws[T.off_q + iq];
            Mj[9] = a[0] * v; Mj[10] = a[1] * v; Mj[11] = a[2] * v;
        } else if (jt == GJ_FREEFLYER) {
            double qb[7];
            IKD_UNROLL
            for (int k = 0; k < 7; ++k) qb[k] = ws[T.off_q + iq + k];
            double R[9];
            quat_to_R(qb, R);
            IKD_UNROLL
            for (int k = 0; k < 9; ++k) Mj[k] = R[k];
            Mj[9] = qb[0]; Mj[10] = qb[1]; Mj[11] = qb[2];
        }
        g_se3_mul(T.placement + 12 * j, Mj, li);
        IKD_UNROLL
        for (int k = 0; k < 12; ++k) oP[k] = ws[T.off_oMi + 12 * T.parent[j] + k];
        g_se3_mul(oP, li, oM);
        IKD_UNROLL
        for (int k = 0; k < 12; ++k) ws[T.off_oMi + 12 * j + k] = oM[k];
        // world Jacobian columns [v; w]
        if (jt == GJ_REVOLUTE || jt == GJ_PRISMATIC || jt == GJ_REVOLUTE_UNBOUNDED) {
            const double Ra[3] = {dfma(oM[0], a[0], dfma(oM[1], a[1], oM[2] * a[2])), dfma(oM[3], a[0], dfma(oM[4], a[1], oM[5] * a[2])),
                                  dfma(oM[6], a[0], dfma(oM[7], a[1], oM[8] * a[2]))};
            const double p[3] = {oM[9], oM[10], oM[11]};
            double v[3] = {Ra[0], Ra[1], Ra[2]}, w[3] = {0, 0, 0};
            if (jt != GJ_PRISMATIC) {
                cross(p, Ra, v);
                w[0] = Ra[0]; w[1] = Ra[1]; w[2] = Ra[2];
            }
            IKD_UNROLL
            for (int r = 0; r < 3; ++r) { ws[T.off_Jw + r * T.nv + iv] = v[r]; ws[T.off_Jw + (3 + r) * T.nv + iv] = w[r]; }
        } else if (jt == GJ_FREEFLYER) {  // Ad(oM1) = [[R, [p]x R], [0, R]]
            const double p[3] = {oM[9], oM[10], oM[11]};
            IKD_UNROLL
            for (int c = 0; c < 3; ++c) {
                const double Rc[3] = {oM[c], oM[3 + c], oM[6 + c]};
                double pxR[3];
                cross(p, Rc, pxR);
                IKD_UNROLL
                for (int r = 0; r < 3; ++r) {
                    ws[T.off_Jw + r * T.nv + iv + c] = Rc[r];
                    ws[T.off_Jw + (3 + r) * T.nv + iv + c] = 0.0;
                    ws[T.off_Jw + r * T.nv + iv + 3 + c] = pxR[r];
                    ws[T.off_Jw + (3 + r) * T.nv + iv + 3 + c] = Rc[r];
                }
            }
        }
    }
}

// evaluate_problem_data (ik/ik/data.cpp:25-58) into the workspace: q -> oMi, Jw, et, Jt.  Returns ||e[0]||^2.
template <class TB, class WS>
IKD_FN double generic_evaluate(const TB &T, const WS &ws, const LaneRows &targets) {
    generic_fk(T, ws);
    if (T.has_com) {  // pinocchio::centerOfMass, backward pass: first moment of every subtree (the subtree masses are constants)
        IKD_UNROLL
        for (int j = 1; j < T.njoints; ++j) {
            const auto c = T.j_lever + 3 * j;
            const double mj = T.j_mass[j];
            IKD_UNROLL
            for (int i = 0; i < 3; ++i)
                ws[T.off_sf + 3 * j + i] = mj * dfma(ws[T.off_oMi + 12 * j + 3 * i], c[0], dfma(ws[T.off_oMi + 12 * j + 3 * i + 1], c[1],
                                                dfma(ws[T.off_oMi + 12 * j + 3 * i + 2], c[2], ws[T.off_oMi + 12 * j + 9 + i])));
        }
        IKD_UNROLL
        for (int i = 0; i < 3; ++i) ws[T.off_sf + i] = 0.0;
        IKD_UNROLL
        for (int j = T.njoints - 1; j > 0; --j)
            IKD_UNROLL
            for (int i = 0; i < 3; ++i) ws[T.off_sf + 3 * T.parent[j] + i] += ws[T.off_sf + 3 * j + i];
    }
    double e0sq = 0.0;
    IKD_UNROLL
    for (int t = 0; t < T.ntasks; ++t) {
        const int fj = T.t_fjoint[t], rj = T.t_rjoint[t], type = T.t_type[t], row = T.t_row[t], dim = T.t_dim[t];
        const auto w6 = T.t_w + 6 * t;
        if (type == GT_COM) {  // ik::CentreOfMassTask, ik/ik/centre_of_mass.hpp:33-45; jacobianCenterOfMass, ik/ik/data.cpp:31-34
            double oJ[12], oMr[12];
            IKD_UNROLL
            for (int k = 0; k < 12; ++k) oJ[k] = ws[T.off_oMi + 12 * rj + k];
            g_se3_mul(oJ, T.t_rpl + 12 * t, oMr);
            const double d[3] = {dfma(ws[T.off_sf], T.inv_total_mass, -oMr[9]), dfma(ws[T.off_sf + 1], T.inv_total_mass, -oMr[10]),
                                 dfma(ws[T.off_sf + 2], T.inv_total_mass, -oMr[11])};
            double tcom[3];
            targets.template run<3>(t * 12 + 9, tcom);
            IKD_UNROLL
            for (int r = 0; r < 3; ++r) {
                const double e = (dfma(oMr[r], d[0], dfma(oMr[3 + r], d[1], oMr[6 + r] * d[2])) - tcom[r]) * w6[r];
                ws[T.off_e + row + r] = e;
                if (T.t_prio[t] == 0) e0sq = dfma(e, e, e0sq);
            }
            IKD_UNROLL
            for (int j = 1; j < T.njoints; ++j) {
                const int n = T.jtype[j] == GJ_FREEFLYER ? 6 : 1;
                const double ms = T.j_submass[j];
                const double f[3] = {ws[T.off_sf + 3 * j], ws[T.off_sf + 3 * j + 1], ws[T.off_sf + 3 * j + 2]};
                IKD_UNROLL
                for (int c = T.idx_v[j]; c < T.idx_v[j] + n; ++c) {
                    const double v[3] = {ws[T.off_Jw + c], ws[T.off_Jw + T.nv + c], ws[T.off_Jw + 2 * T.nv + c]};
                    const double w[3] = {ws[T.off_Jw + 3 * T.nv + c], ws[T.off_Jw + 4 * T.nv + c], ws[T.off_Jw + 5 * T.nv + c]};
                    double fxw[3];
                    cross(f, w, fxw);
                    const double col[3] = {(ms * v[0] - fxw[0]) * T.inv_total_mass, (ms * v[1] - fxw[1]) * T.inv_total_mass,
                                           (ms * v[2] - fxw[2]) * T.inv_total_mass};
                    IKD_UNROLL
                    for (int r = 0; r < 3; ++r)
                        ws[T.off_J + (row + r) * T.nv + c] = w6[r] * dfma(oMr[r], col[0], dfma(oMr[3 + r], col[1], oMr[6 + r] * col[2]));
                }
            }
            continue;
        }
        if (type == GT_POSTURE_ROW) {  // one row of ik::PostureTask, ik/ik/posture.hpp:51-68 (fjoint = tangent column, rjoint = q index)
            const double e = (ws[T.off_q + rj] - targets(t * 12 + 9)) * w6[1] * w6[0];
            ws[T.off_e + row] = e;
            if (T.t_prio[t] == 0) e0sq = dfma(e, e, e0sq);
            IKD_UNROLL
            for (int c = 0; c < T.nv; ++c) ws[T.off_J + row * T.nv + c] = (c == fj) ? w6[0] : 0.0;
            continue;
        }
        double oJ[12], oMf[12], oMr[12], tg[12];
        IKD_UNROLL
        for (int k = 0; k < 12; ++k) oJ[k] = ws[T.off_oMi + 12 * fj + k];
        g_se3_mul(oJ, T.t_fpl + 12 * t, oMf);
        IKD_UNROLL
        for (int k = 0; k < 12; ++k) oJ[k] = ws[T.off_oMi + 12 * rj + k];
        g_se3_mul(oJ, T.t_rpl + 12 * t, oMr);
        targets.template run<12>(t * 12, tg);
        IKD_UNROLL
        for (int r = 0; r < dim; ++r)
            IKD_UNROLL
            for (int c = 0; c < T.nv; ++c) ws[T.off_J + (row + r) * T.nv + c] = 0.0;  // frame.hpp:110: zero outside the support
        const double Rf[9] = {oMf[0], oMf[1], oMf[2], oMf[3], oMf[4], oMf[5], oMf[6], oMf[7], oMf[8]};
        const double pf[3] = {oMf[9], oMf[10], oMf[11]};
        if (type >= GT_ALIGN_X) {  // AlignAxisTask, ik/ik/frame.hpp:257-301
            double rMf[12];
            g_se3_inv_mul(oMr, oMf, rMf);
            const int ax = type - GT_ALIGN_X;
            const double r[3] = {rMf[ax], rMf[3 + ax], rMf[6 + ax]};
            const double inv = drsqrt(dfma(tg[9], tg[9], dfma(tg[10], tg[10], tg[11] * tg[11])));
            const double tn[3] = {tg[9] * inv, tg[10] * inv, tg[11] * inv};
            double rxt[3];
            cross(r, tn, rxt);
            const double g[3] = {dfma(rxt[0], rMf[0], dfma(rxt[1], rMf[3], rxt[2] * rMf[6])),
                                 dfma(rxt[0], rMf[1], dfma(rxt[1], rMf[4], rxt[2] * rMf[7])),
                                 dfma(rxt[0], rMf[2], dfma(rxt[1], rMf[5], rxt[2] * rMf[8]))};
            const double e = (1.0 - dot(r, tn)) * w6[0];
            ws[T.off_e + row] = e;
            if (T.t_prio[t] == 0) e0sq = dfma(e, e, e0sq);
            IKD_UNROLL
            for (int j = fj; j > 0; j = T.parent[j]) {
                const int n = T.jtype[j] == GJ_FREEFLYER ? 6 : 1;
                IKD_UNROLL
                for (int c = T.idx_v[j]; c < T.idx_v[j] + n; ++c) {
                    const double w[3] = {ws[T.off_Jw + 3 * T.nv + c], ws[T.off_Jw + 4 * T.nv + c], ws[T.off_Jw + 5 * T.nv + c]};
                    double wl[3];
                    rotT_vec(Rf, w, wl);
                    ws[T.off_J + row * T.nv + c] = -w6[0] * dot(g, wl);
                }
            }
            continue;
        }
        double oMt[12], Re[9], pe[3];
        g_se3_mul(oMr, tg, oMt);                        // frame.hpp:48
        IKD_UNROLL
        for (int i = 0; i < 3; ++i)
            IKD_UNROLL
            for (int j = 0; j < 3; ++j) Re[3 * i + j] = dfma(Rf[i], oMt[j], dfma(Rf[3 + i], oMt[3 + j], Rf[6 + i] * oMt[6 + j]));
        {
            const double dp[3] = {oMt[9] - pf[0], oMt[10] - pf[1], oMt[11] - pf[2]};
            rotT_vec(Rf, dp, pe);
        }
        LogAndJlog lj;
        log6_and_jlog6_inv(Re, pe, lj);                 // frame.hpp:50-61, :162-166
        const int r0 = (type == GT_ORIENTATION) ? 3 : 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) {   // (static register indices: a runtime index would put lj.e into scratch memory)
            if (k < r0 || k >= r0 + dim) continue;
            const double e = lj.e[k] * w6[k - r0];
            ws[T.off_e + row + k - r0] = e;
            if (T.t_prio[t] == 0) e0sq = dfma(e, e, e0sq);
        }
        IKD_UNROLL
        for (int j = fj; j > 0; j = T.parent[j]) {      // support of the frame's joint (getFrameJacobian, LOCAL)
            const int n = T.jtype[j] == GJ_FREEFLYER ? 6 : 1;
            IKD_UNROLL
            for (int c = T.idx_v[j]; c < T.idx_v[j] + n; ++c) {
                double v[3] = {ws[T.off_Jw + c], ws[T.off_Jw + T.nv + c], ws[T.off_Jw + 2 * T.nv + c]};
                const double w[3] = {ws[T.off_Jw + 3 * T.nv + c], ws[T.off_Jw + 4 * T.nv + c], ws[T.off_Jw + 5 * T.nv + c]};
                double pxw[3], vl[3], wl[3];
                cross(pf, w, pxw);
                v[0] -= pxw[0]; v[1] -= pxw[1]; v[2] -= pxw[2];
                rotT_vec(Rf, v, vl);
                rotT_vec(Rf, w, wl);
                double out[6];
                IKD_UNROLL
                for (int i = 0; i < 3; ++i) {
                    out[i] = -dfma(lj.A[3 * i], vl[0], dfma(lj.A[3 * i + 1], vl[1], dfma(lj.A[3 * i + 2], vl[2],
                              dfma(lj.Bm[3 * i], wl[0], dfma(lj.Bm[3 * i + 1], wl[1], lj.Bm[3 * i + 2] * wl[2])))));
                    out[3 + i] = -dfma(lj.A[3 * i], wl[0], dfma(lj.A[3 * i + 1], wl[1], lj.A[3 * i + 2] * wl[2]));
                }
#pragma unroll
                for (int k = 0; k < 6; ++k)
                    if (k >= r0 && k < r0 + dim) ws[T.off_J + (row + k - r0) * T.nv + c] = w6[k - r0] * out[k];
            }
        }
    }
    return e0sq;
}

// Column c of pinocchio::getFrameJacobian(..., LOCAL) for a frame placed at (Rf, pf) in the world, from the world joint
// Jacobian in the workspace: linear part vl, angular part wl.
// (per-lane pointer + element stride: the interpreter forms and ik::pik)
template <class TB, class WS>
IKD_FN double generic_evaluate(const TB &T, const WS &ws, const double *targets_lane, int64_t tstride) {
    return generic_evaluate(T, ws, LaneRows{reinterpret_cast<const char *>(targets_lane), 0u, tstride * 8, false});
}

template <class TB, class WS>
IKD_FN void local_column(const TB &T, const WS &ws, int c, const double (&Rf)[9], const double (&pf)[3], double (&vl)[3],
                         double (&wl)[3]) {
    double v[3] = {ws[T.off_Jw + c], ws[T.off_Jw + T.nv + c], ws[T.off_Jw + 2 * T.nv + c]};
    const double w[3] = {ws[T.off_Jw + 3 * T.nv + c], ws[T.off_Jw + 4 * T.nv + c], ws[T.off_Jw + 5 * T.nv + c]};
    double pxw[3];
    cross(pf, w, pxw);
    v[0] -= pxw[0]; v[1] -= pxw[1]; v[2] -= pxw[2];
    rotT_vec(Rf, v, vl);
    rotT_vec(Rf, w, wl);
}

// Cyclic one-sided Jacobi on the m rows (length nv) stored at off_rows: plane rotations from the left until the rows are
// mutually orthogonal, U^T A = diag(sigma) V^T, so that row i ends as sigma_i v_i^T.  A companion column at off_col
// (length m; pass a negative offset for none) is rotated along and ends as U^T col.  Every lane of a wave runs the same
// sweeps: a lane whose pair is already orthogonal applies the identity rotation, and the wave leaves on a uniform vote.
template <class WS, class AnyFn>
IKD_FN void jacobi_rows(const WS &ws, int off_rows, int off_col, int m, int nv, AnyFn any_lane) {
    const double eps = 2.220446049250313e-16;
    for (int sweep = 0; sweep < 40; ++sweep) {
        bool rotated = false;
        for (int a = 0; a < m - 1; ++a)
            for (int b = a + 1; b < m; ++b) {
                double al = 0.0, be = 0.0, ga = 0.0;
                for (int c = 0; c < nv; ++c) {
                    const double x = ws[off_rows + a * nv + c], y = ws[off_rows + b * nv + c];
                    al = dfma(x, x, al);
                    be = dfma(y, y, be);
                    ga = dfma(x, y, ga);
                }
                const bool rot = __builtin_fabs(ga) > eps * __builtin_sqrt(al * be);
                if (!any_lane(rot)) continue;
                const double zeta = (be - al) / (2.0 * ga);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (__builtin_fabs(zeta) + __builtin_sqrt(dfma(zeta, zeta, 1.0)));
                const double c0 = 1.0 / __builtin_sqrt(dfma(t, t, 1.0));
                const double cs = dsel(rot, c0, 1.0), sn = dsel(rot, c0 * t, 0.0);
                for (int c = 0; c < nv; ++c) {
                    const double x = ws[off_rows + a * nv + c], y = ws[off_rows + b * nv + c];
                    ws[off_rows + a * nv + c] = dfma(cs, x, -sn * y);
                    ws[off_rows + b * nv + c] = dfma(sn, x, cs * y);
                }
                if (off_col >= 0) {
                    const double x = ws[off_col + a], y = ws[off_col + b];
                    ws[off_col + a] = dfma(cs, x, -sn * y);
                    ws[off_col + b] = dfma(sn, x, cs * y);
                }
                rotated = rotated || rot;
            }
        if (!any_lane(rotated)) break;
    }
}

// ik::FrameConstraint::compute_jacobian (ik/ik/frame.hpp:413-449) for every constraint, into the workspace (Jc, Mc x nv):
// the velocity of the frame relative to its reference frame, in the frame's local coordinates:
//   Jc = J_frame(LOCAL) - Ad(fMr) J_reference(LOCAL),  rows by kinematic type.   Needs oMi and Jw (generic_fk).
template <class TB, class WS>
IKD_FN void generic_constraint_jacobian(const TB &T, const WS &ws) {
    IKD_UNROLL
    for (int k = 0; k < T.ncons; ++k) {
        const int fj = T.c_fjoint[k], rj = T.c_rjoint[k], row = T.c_row[k], dim = T.c_dim[k];
        const int r0 = (T.c_type[k] == GT_ORIENTATION) ? 3 : 0;
        double oJ[12], oMf[12], oMr[12], fMr[12];
        IKD_UNROLL
        for (int i = 0; i < 12; ++i) oJ[i] = ws[T.off_oMi + 12 * fj + i];
        g_se3_mul(oJ, T.c_fpl + 12 * k, oMf);
        IKD_UNROLL
        for (int i = 0; i < 12; ++i) oJ[i] = ws[T.off_oMi + 12 * rj + i];
        g_se3_mul(oJ, T.c_rpl + 12 * k, oMr);
        g_se3_inv_mul(oMf, oMr, fMr);
        IKD_UNROLL
        for (int r = 0; r < dim; ++r)
            IKD_UNROLL
            for (int c = 0; c < T.nv; ++c) ws[T.off_Jc + (row + r) * T.nv + c] = 0.0;
        const double Rf[9] = {oMf[0], oMf[1], oMf[2], oMf[3], oMf[4], oMf[5], oMf[6], oMf[7], oMf[8]};
        const double pf[3] = {oMf[9], oMf[10], oMf[11]};
        IKD_UNROLL
        for (int j = fj; j > 0; j = T.parent[j]) {
            const int n = T.jtype[j] == GJ_FREEFLYER ? 6 : 1;
            IKD_UNROLL
            for (int c = T.idx_v[j]; c < T.idx_v[j] + n; ++c) {
                double vl[3], wl[3];
                local_column(T, ws, c, Rf, pf, vl, wl);
                const double out[6] = {vl[0], vl[1], vl[2], wl[0], wl[1], wl[2]};
#pragma unroll
                for (int k = 0; k < 6; ++k)
                    if (k >= r0 && k < r0 + dim) ws[T.off_Jc + (row + k - r0) * T.nv + c] = out[k];
            }
        }
        const double Rr[9] = {oMr[0], oMr[1], oMr[2], oMr[3], oMr[4], oMr[5], oMr[6], oMr[7], oMr[8]};
        const double pr[3] = {oMr[9], oMr[10], oMr[11]};
        const double Rx[9] = {fMr[0], fMr[1], fMr[2], fMr[3], fMr[4], fMr[5], fMr[6], fMr[7], fMr[8]};
        const double px[3] = {fMr[9], fMr[10], fMr[11]};
        IKD_UNROLL
        for (int j = rj; j > 0; j = T.parent[j]) {
            const int n = T.jtype[j] == GJ_FREEFLYER ? 6 : 1;
            IKD_UNROLL
            for (int c = T.idx_v[j]; c < T.idx_v[j] + n; ++c) {
                double vr[3], wr[3], Rv[3], Rw[3], pxRw[3];
                local_column(T, ws, c, Rr, pr, vr, wr);
                IKD_UNROLL
                for (int i = 0; i < 3; ++i) {   // Ad(fMr) [v; w] = [R v + p x (R w); R w]
                    Rv[i] = dfma(Rx[3 * i], vr[0], dfma(Rx[3 * i + 1], vr[1], Rx[3 * i + 2] * vr[2]));
                    Rw[i] = dfma(Rx[3 * i], wr[0], dfma(Rx[3 * i + 1], wr[1], Rx[3 * i + 2] * wr[2]));
                }
                cross(px, Rw, pxRw);
                const double out[6] = {Rv[0] + pxRw[0], Rv[1] + pxRw[1], Rv[2] + pxRw[2], Rw[0], Rw[1], Rw[2]};
#pragma unroll
                for (int k = 0; k < 6; ++k)
                    if (k >= r0 && k < r0 + dim) ws[T.off_Jc + (row + k - r0) * T.nv + c] -= out[k];
            }
        }
    }
}

// v <- (I - pinv(A) A) v for the m x nv matrix stored at off_rows (destroyed): ik/ik/dls.cpp:43-53's N applied to a vector.
// Rank as Eigen's COD counts it, relative to the largest value: sigma_i > eps * min(m, nv) * sigma_max (see pik_solver.hpp).
template <class WS, class AnyFn>
IKD_FN void project_out_rowspace(const WS &ws, int off_rows, int m, int nv, int off_vec, AnyFn any_lane) {
    jacobi_rows(ws, off_rows, -1, m, nv, any_lane);
    double smax2 = 0.0;
    for (int i = 0; i < m; ++i) {
        double s2 = 0.0;
        for (int c = 0; c < nv; ++c) { const double x = ws[off_rows + i * nv + c]; s2 = dfma(x, x, s2); }
        smax2 = dmax(smax2, s2);
    }
    const double kk = 2.220446049250313e-16 * static_cast<double>(m < nv ? m : nv);
    const double thr2 = kk * kk * smax2;
    for (int i = 0; i < m; ++i) {
        double s2 = 0.0, d = 0.0;
        for (int c = 0; c < nv; ++c) {
            const double x = ws[off_rows + i * nv + c];
            s2 = dfma(x, x, s2);
            d = dfma(x, ws[off_vec + c], d);
        }
        const double f = dsel(s2 > thr2, d / s2, 0.0);
        for (int c = 0; c < nv; ++c) ws[off_vec + c] = dfma(-f, ws[off_rows + i * nv + c], ws[off_vec + c]);
    }
}

// The same projection, v <- (I - pinv(A) A) v, for the run-time specialised program (every loop unrolls: no data-dependent sweep
// count): N = I - V^T V with V an orthonormal basis of the row space of A from Gram-Schmidt with every projection applied twice; a
// row whose remainder falls below the rank threshold of the complete orthogonal decomposition behind the reference's pseudo-inverse
// (epsilon x rows x the largest row norm) is dropped by a select -- what the tree kernel's constraint_project does (tree_solver.hpp).
template <class WS>
IKD_FN void project_out_rowspace_gs(const WS &ws, int off_rows, int m, int nv, int off_vec) {
    double maxn2 = 0.0;
    IKD_UNROLL
    for (int i = 0; i < m; ++i) {
        double n2 = 0.0;
        IKD_UNROLL
        for (int c = 0; c < nv; ++c) { const double x = ws[off_rows + i * nv + c]; n2 = dfma(x, x, n2); }
        maxn2 = dmax(maxn2, n2);
    }
    const double thr = 2.220446049250313e-16 * static_cast<double>(m);
    const double thr2 = thr * thr * maxn2;
    IKD_UNROLL
    for (int k = 0; k < m; ++k) {
        IKD_UNROLL
        for (int pass = 0; pass < 2; ++pass) {
            IKD_UNROLL
            for (int i = 0; i < k; ++i) {
                double d = 0.0;
                IKD_UNROLL
                for (int c = 0; c < nv; ++c) d = dfma(ws[off_rows + i * nv + c], ws[off_rows + k * nv + c], d);
                IKD_UNROLL
                for (int c = 0; c < nv; ++c) ws[off_rows + k * nv + c] = dfma(-d, ws[off_rows + i * nv + c], ws[off_rows + k * nv + c]);
            }
        }
        double n2 = 0.0;
        IKD_UNROLL
        for (int c = 0; c < nv; ++c) { const double x = ws[off_rows + k * nv + c]; n2 = dfma(x, x, n2); }
        const double inv = dsel(n2 > thr2, drsqrt(dmax(n2, 1e-300)), 0.0);
        double d = 0.0;
        IKD_UNROLL
        for (int c = 0; c < nv; ++c) {
            const double x = ws[off_rows + k * nv + c] * inv;
            ws[off_rows + k * nv + c] = x;
            d = dfma(x, ws[off_vec + c], d);
        }
        IKD_UNROLL
        for (int c = 0; c < nv; ++c) ws[off_vec + c] = dfma(-d, ws[off_rows + k * nv + c], ws[off_vec + c]);
    }
}

// q <- clip(integrate(q, step * dq)) on the workspace (pinocchio::integrate + apply_joint_clipping,
// ik/ik/common.hpp:53-56); a lane that is no longer active keeps its q.
template <class TB, class WS>
IKD_FN void generic_integrate_clip(const TB &T, const WS &ws, double step_length, bool active) {
    IKD_UNROLL
    for (int j = 1; j < T.njoints; ++j) {
        const int iq = T.idx_q[j], iv = T.idx_v[j];
        if (T.jtype[j] == GJ_FREEFLYER) {
            double qb[7], v[6], qn[7], R1[9];
            IKD_UNROLL
            for (int k = 0; k < 7; ++k) qb[k] = ws[T.off_q + iq + k];
            IKD_UNROLL
            for (int k = 0; k < 6; ++k) v[k] = step_length * ws[T.off_dq + iv + k];
            quat_to_R(qb, R1);
            freeflyer_integrate(qb, R1, v, qn);
            IKD_UNROLL
            for (int k = 0; k < 7; ++k) {
                const double c = dmin(T.upper[iq + k], dmax(qn[k], T.lower[iq + k]));
                ws[T.off_q + iq + k] = active ? c : qb[k];
            }
        } else if (T.jtype[j] == GJ_REVOLUTE_UNBOUNDED) {
            const double c0 = ws[T.off_q + iq], s0 = ws[T.off_q + iq + 1];
            double c1, s1;
            unbounded_integrate(c0, s0, step_length * ws[T.off_dq + iv], c1, s1);
            c1 = dmin(T.upper[iq], dmax(c1, T.lower[iq]));
            s1 = dmin(T.upper[iq + 1], dmax(s1, T.lower[iq + 1]));
            ws[T.off_q + iq] = active ? c1 : c0;
            ws[T.off_q + iq + 1] = active ? s1 : s0;
        } else {
            const double qo = ws[T.off_q + iq];
            const double lo_ = T.lower[iq], hi_ = T.upper[iq];
            const double c = dmin(hi_, dmax(dfma(step_length, ws[T.off_dq + iv], qo), lo_));
            ws[T.off_q + iq] = active ? c : qo;
        }
    }
}

// One full solve on the workspace (q already stored at off_q).
// R: the lane-refill hook (GenericRefill below, the static lane programs' stop-rule mode on batches larger than the machine);
// NoRefill (tree_solver.hpp) is the lock-step loop -- every `if constexpr (R::on)` compiles to nothing there.
template <class TB, class WS, class AnyFn, class R = NoRefill>
IKD_FN void generic_dls(const TB &T, const LoopParams &prm, const WS &ws, const LaneRows &targets_in,
                        int &iters_out, bool &success_out, AnyFn any_active, R refill = R{}) {
    bool active = true, success = false;
    if constexpr (R::on) active = refill.start;   // (a tail lane of the first round holds no problem)
    LaneRows targets = targets_in;                 // (refill: re-pointed when the lane takes its next problem)
    int iters = prm.max_iterations;
    int lit = 0;                                   // (refill) this lane's own iteration count
    if constexpr (R::on) lit = refill.it0();       // (second phase of a two-phase solve: the iterations the first phase took)
    const int M = T.M, nv = T.nv;
    for (int it = 0; R::on || it < prm.max_iterations; ++it) {
        const double e0sq = generic_evaluate(T, ws, targets);
#ifdef IKD_STATIC_TABLES
        if constexpr (TB::elim != 0) {
            // PostureTask rows eliminated from the linear system (static lane programs of problems with many of them; the cooperative
            // kernel does the same, coop_solver.hpp).  Each posture row has ONE non-zero entry, so with J_f the other Mf rows and
            // D = lambda^2 I + (the posture rows' squares on their columns), JtJ + lambda^2 I = D + J_f^T J_f and by Woodbury
            //   dq = -(D + J_f^T J_f)^-1 J^T e = -(u - D^-1 J_f^T z),   u = D^-1 J^T e,   (I + J_f D^-1 J_f^T) z = J_f u
            // -- an Mf x Mf system instead of M x M (the reference demo with every line switched on: 13 instead of 29).  J_f is
            // scaled by D^-1/2 in place so that the system matrix is I + Jt Jt^T.
            constexpr int Mf = TB::Mf, Mp = TB::Mp, NV = TB::nv;
            double dh[NV], u[NV], A[Mf * (Mf + 1) / 2], z[Mf];
            IKD_UNROLL
            for (int c = 0; c < NV; ++c) {
                double d = prm.lam2;
                IKD_UNROLL
                for (int k = 0; k < Mp; ++k)
                    if (TB::p_col[k] == c) { const double w = ws[T.off_J + TB::p_row[k] * NV + c]; d = dfma(w, w, d); }
                dh[c] = drsqrt(d);
            }
            IKD_UNROLL
            for (int c = 0; c < NV; ++c) {
                double sj = 0.0;
                IKD_UNROLL
                for (int r = 0; r < TB::M; ++r) sj = dfma(ws[T.off_J + r * NV + c], ws[T.off_e + r], sj);
                u[c] = sj * (dh[c] * dh[c]);
            }
            IKD_UNROLL
            for (int i = 0; i < Mf; ++i) {
                double sb = 0.0;
                IKD_UNROLL
                for (int c = 0; c < NV; ++c) sb = dfma(ws[T.off_J + TB::f_row[i] * NV + c], u[c], sb);
                z[i] = sb;
                IKD_UNROLL
                for (int c = 0; c < NV; ++c) ws[T.off_J + TB::f_row[i] * NV + c] = ws[T.off_J + TB::f_row[i] * NV + c] * dh[c];
            }
            IKD_UNROLL
            for (int i = 0; i < Mf; ++i)
                IKD_UNROLL
                for (int j = 0; j <= i; ++j) {
                    double sa = (i == j) ? 1.0 : 0.0;
                    IKD_UNROLL
                    for (int c = 0; c < NV; ++c) sa = dfma(ws[T.off_J + TB::f_row[i] * NV + c], ws[T.off_J + TB::f_row[j] * NV + c], sa);
                    A[tri(i, j)] = sa;
                }
            IKD_UNROLL
            for (int k = 0; k < Mf; ++k) {   // Cholesky in place (diagonal holds 1 / L_kk), as the dense path below
                double d = A[tri(k, k)];
                IKD_UNROLL
                for (int m = 0; m < k; ++m) { const double l = A[tri(k, m)]; d = dfma(-l, l, d); }
                const double inv = drsqrt(d);
                A[tri(k, k)] = inv;
                IKD_UNROLL
                for (int i = k + 1; i < Mf; ++i) {
                    double sl = A[tri(i, k)];
                    IKD_UNROLL
                    for (int m = 0; m < k; ++m) sl = dfma(-A[tri(i, m)], A[tri(k, m)], sl);
                    A[tri(i, k)] = sl * inv;
                }
            }
            IKD_UNROLL
            for (int k = 0; k < Mf; ++k) {
                double sf = z[k];
                IKD_UNROLL
                for (int m = 0; m < k; ++m) sf = dfma(-A[tri(k, m)], z[m], sf);
                z[k] = sf * A[tri(k, k)];
            }
            IKD_UNROLL
            for (int k = Mf - 1; k >= 0; --k) {
                double sb = z[k];
                IKD_UNROLL
                for (int m = Mf - 1; m > k; --m) sb = dfma(-A[tri(m, k)], z[m], sb);
                z[k] = sb * A[tri(k, k)];
            }
            IKD_UNROLL
            for (int c = 0; c < NV; ++c) {
                double sd = 0.0;
                IKD_UNROLL
                for (int i = 0; i < Mf; ++i) sd = dfma(ws[T.off_J + TB::f_row[i] * NV + c], z[i], sd);
                ws[T.off_dq + c] = -(u[c] - dh[c] * sd);
            }
        } else
#endif
        {
            // JJ = Jt Jt^T + damping^2 I (lower triangle, packed), ik/ik/dls.cpp:39-41
            IKD_UNROLL
            for (int i = 0; i < M; ++i)
                IKD_UNROLL
                for (int j = 0; j <= i; ++j) {
                    double s = (i == j) ? prm.lam2 : 0.0;
                    IKD_UNROLL
                    for (int c = 0; c < nv; ++c) s = dfma(ws[T.off_J + i * nv + c], ws[T.off_J + j * nv + c], s);
                    ws[T.off_G + tri(i, j)] = s;
                }
            // Cholesky in place (diagonal holds 1/L_ii), forward and backward substitution: y = JJ^-1 et
            IKD_UNROLL
            for (int k = 0; k < M; ++k) {
                double d = ws[T.off_G + tri(k, k)];
                IKD_UNROLL
                for (int m = 0; m < k; ++m) { const double l = ws[T.off_G + tri(k, m)]; d = dfma(-l, l, d); }
                const double inv = drsqrt(d);
                ws[T.off_G + tri(k, k)] = inv;
                IKD_UNROLL
                for (int i = k + 1; i < M; ++i) {
                    double s = ws[T.off_G + tri(i, k)];
                    IKD_UNROLL
                    for (int m = 0; m < k; ++m) s = dfma(-ws[T.off_G + tri(i, m)], ws[T.off_G + tri(k, m)], s);
                    ws[T.off_G + tri(i, k)] = s * inv;
                }
            }
            IKD_UNROLL
            for (int k = 0; k < M; ++k) {
                double s = ws[T.off_e + k];
                IKD_UNROLL
                for (int m = 0; m < k; ++m) s = dfma(-ws[T.off_G + tri(k, m)], ws[T.off_y + m], s);
                ws[T.off_y + k] = s * ws[T.off_G + tri(k, k)];
            }
            IKD_UNROLL
            for (int k = M - 1; k >= 0; --k) {
                double s = ws[T.off_y + k];
                IKD_UNROLL
                for (int m = M - 1; m > k; --m) s = dfma(-ws[T.off_G + tri(m, k)], ws[T.off_y + m], s);  // (the order coop_solver.hpp takes)
                ws[T.off_y + k] = s * ws[T.off_G + tri(k, k)];
            }
            IKD_UNROLL
            for (int c = 0; c < nv; ++c) {  // dq = -Jt^T y, ik/ik/dls.cpp:52-53 (N = I)
                double s = 0.0;
                IKD_UNROLL
                for (int r = 0; r < M; ++r) s = dfma(ws[T.off_J + r * nv + c], ws[T.off_y + r], s);
                ws[T.off_dq + c] = -s;
            }
        }
        if (T.Mc > 0) {  // dq <- N dq, N = I - pinv(Jc) Jc: the step stays in the null space of the constraints (dls.cpp:26-34,43-53)
            generic_constraint_jacobian(T, ws);
#ifdef IKD_STATIC_TABLES
            project_out_rowspace_gs(ws, T.off_Jc, T.Mc, nv, T.off_dq);
#else
            project_out_rowspace(ws, T.off_Jc, T.Mc, nv, T.off_dq, any_active);
#endif
        }
        // inverse_kinematics_visitor::should_stop(ik, e, dq) (ik/ik/visitor.hpp:15-21; dls.cpp:61-64) and its derived family
        bool err_ok = (prm.stop_sq_tol >= 0.0) && (e0sq < prm.stop_sq_tol);
        if (prm.nlt > 0) {   // (wave-uniform) every listed level under its own tolerance
            err_ok = true;
            IKD_UNROLL
            for (int l = 0; l < T.nlevels; ++l) {
                double s = 0.0;
                IKD_UNROLL
                for (int r = T.lvl_row0[l]; r < T.lvl_row0[l + 1]; ++r) s = dfma(ws[T.off_e + r], ws[T.off_e + r], s);
                if (l < prm.nlt) err_ok = err_ok && (s < prm.lvl_tol[l < 8 ? l : 7]);
            }
        }
        bool step_small = false;
        if (prm.dq_sq_tol > 0.0) {   // (wave-uniform)
            double s = 0.0;
            IKD_UNROLL
            for (int c = 0; c < nv; ++c) s = dfma(ws[T.off_dq + c], ws[T.off_dq + c], s);
            step_small = s < prm.dq_sq_tol;
        }
        const bool stop_now = active && (err_ok || step_small);
        if (stop_now) { success = true; iters = R::on ? lit : it; }
        const bool had = active;   // (refill) the lane held a problem during this iteration
        active = active && !stop_now;
        generic_integrate_clip(T, ws, prm.step_length, active);  // ik/ik/dls.cpp:67-71
        if constexpr (R::on) {
            // a lane whose visitor fired (q is the configuration the error was evaluated at, dls.cpp:61-63) or whose count reached
            // max_iterations (the stepped q, dls.cpp:76-77) stores its result and takes the next unsolved problem
            ++lit;
            const bool done = had && (stop_now || lit >= prm.max_iterations);
            active = had && !done;   // (a lane that ran out of iterations is as finished as one whose visitor fired)
            refill.took = false;
            const bool any_left = refill.step(T, done, stop_now, stop_now ? lit - 1 : prm.max_iterations, ws, targets, active);
            if (refill.took) { lit = refill.it0(); success = false; }
            if (!any_left) break;
        } else {
            if (!any_active(active)) break;
        }
    }
    iters_out = (R::on || success) ? iters : iterations_taken(any_active, prm.max_iterations);   // (chain_solver.hpp chain_dls)
    success_out = success;
}

// The tables seen through the constant address space (device kernels: a.T holds plain pointers, as the host wrote them).
IKD_FN GenericTablesK const_tables(const GenericTables &t) {
    static_assert(sizeof(GenericTablesK) == sizeof(GenericTables), "same layout");
    return __builtin_bit_cast(GenericTablesK, t);
}

// What one lane of the generic kernels does (shared by kernels.hip and the lane emulator).
struct GenericKernelArgs {
    GenericTables T;
    LoopParams prm;
    int layout;
    int64_t B;
    const double *q0, *targets;
    double *q_out;
    uint8_t *success;
    int32_t *iters;
    double *ws;          // [ws_words][ws_stride]
    int64_t ws_stride;   // >= B
    double *e_out, *J_out, *oMf_out;  // stage kernel
    // the two-phase stop-rule solve of the static lane programs (as ChainKernelArgs: chain_kernel_body.hpp; rtc.cpp rtc_launch_generic_static)
    const int32_t *worklist;
    const unsigned long long *count;
    int leave_active, leave_after;
    int32_t *append_list;
    unsigned long long *append_count;
};

// ws: the lane's workspace column -- of the HBM workspace (dls_generic_body below) or of the workgroup's LDS (the on-chip form,
// kernels.hip: word w of lane l at lds[w * 64 + l], conflict-free).
// group0: the first problem of the lane's workgroup (wave-uniform: the targets are then addressed as SGPR row + lane offset, see
// LaneRows in tree_solver.hpp; the static lane programs), or -1: per-lane pointers.
template <class TB, class WS, class AnyFn>
IKD_FN void dls_generic_body_ws(const GenericKernelArgs &a, const TB &T, int64_t gid, const WS &ws, AnyFn any_active, int64_t group0 = -1) {
    const bool valid = gid < a.B;
    const int64_t b = valid ? gid : a.B - 1;
    IKD_UNROLL
    for (int i = 0; i < T.nq; ++i) ws[T.off_q + i] = a.q0[at(a.layout, a.B, T.nq, i, b)];
    const int64_t g0 = group0 >= 0 ? group0 : b;
    const int64_t tcol = a.layout == LAYOUT_SOA ? 1 : static_cast<int64_t>(T.ntasks) * 12;   // doubles between consecutive problems
    const LaneRows tl{reinterpret_cast<const char *>(a.targets + g0 * tcol), static_cast<uint32_t>((b - g0) * tcol * 8),
                      static_cast<int64_t>(a.layout == LAYOUT_SOA ? a.B * 8 : 8), group0 >= 0};
    int iters;
    bool success;
    generic_dls(T, a.prm, ws, tl, iters, success, any_active);
    if (a.append_count) append_unfinished(a.append_list, a.append_count, valid && !success && iters < a.prm.max_iterations, b);   // (wave-uniform test)
    if (!valid) return;
    IKD_UNROLL
    for (int i = 0; i < T.nq; ++i) a.q_out[at(a.layout, a.B, T.nq, i, b)] = ws[T.off_q + i];
    if (a.success) a.success[b] = success ? 1 : 0;
    if (a.iters) a.iters[b] = iters;
}

#if IKD_HIP_LANG
// Lane refill for the static lane programs (the scheme of chain_kernel_body.hpp chain_refill_loop / tree_kernel_body.hpp TreeRefill):
// persistent one-wave workgroups; the first nwaves * 64 problems are dealt statically, the rest through `queue` in chunks a wave
// reserves for itself ([pool_lo, pool_hi)); a lane that is done stores (q, success, iters) and loads the next problem's q0 into its
// workspace.  The arithmetic of a problem's iterations is the lock-step program's: results are bit-identical.
struct GenericRefill {
    static constexpr bool on = true;
    const GenericKernelArgs *a;
    unsigned long long *queue;
    int chunk, batch;   // problems pulled from the head at a time; idle lanes a refill event waits for
    int64_t b, first_round, pool_lo, pool_hi;
    bool exhausted;
    bool start;
    bool took;          // (set by step) this lane has just taken a new problem
    int64_t nwork;      // work items: the batch's problems, or the entries of a->worklist (`b` is always a PROBLEM index)
    __device__ __forceinline__ int64_t problem(int64_t w) const { return a->worklist ? static_cast<int64_t>(a->worklist[w]) : w; }
    __device__ __forceinline__ int it0() const { return a->worklist ? a->iters[b] : 0; }   // (of the lane's CURRENT problem)

    template <class TB>
    __device__ __forceinline__ LaneRows target_rows(const TB &T, int64_t bb) const {
        return LaneRows{reinterpret_cast<const char *>(a->layout == LAYOUT_SOA ? a->targets + bb : a->targets + bb * T.ntasks * 12), 0u,
                        static_cast<int64_t>(a->layout == LAYOUT_SOA ? a->B * 8 : 8), false};
    }

    template <class TB, class WS>
    __device__ __forceinline__ bool step(const TB &T, bool done, bool stopped, int iters, const WS &ws, LaneRows &tl, bool &active) {
        const GenericKernelArgs &A = *a;
        const int lane = static_cast<int>(threadIdx.x) & 63;
        if (done) {   // the result leaves at once; the lane then idles until the wave's next refill event (batched: chain_kernel_body.hpp)
            IKD_UNROLL
            for (int i = 0; i < T.nq; ++i) A.q_out[at(A.layout, A.B, T.nq, i, b)] = ws[T.off_q + i];
            if (A.success) A.success[b] = stopped ? 1 : 0;
            if (A.iters) A.iters[b] = iters;
            active = false;
        }
        const unsigned long long mask = __ballot(!active);
        const int need = __popcll(mask);
        const bool supply = pool_hi > pool_lo || !exhausted;          // (wave-uniform)
        if (supply && (need >= batch || need == 64)) {
            const int rank = __popcll(mask & ((1ull << lane) - 1ull));
            const int64_t avail = pool_hi - pool_lo;
            int64_t nb = pool_lo + rank;
            bool got = rank < avail;
            if (avail < need && !exhausted) {                             // wave-uniform: pull the next chunk
                unsigned long long v = 0;
                if (lane == 0) v = atomicAdd(queue, static_cast<unsigned long long>(chunk));
                const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v)), hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v >> 32));
                const int64_t nlo = first_round + static_cast<int64_t>((static_cast<unsigned long long>(hi) << 32) | lo);
                const int64_t nhi = nlo + chunk < nwork ? nlo + chunk : nwork;
                exhausted = nlo + chunk >= nwork;
                if (rank >= avail) { nb = nlo + (rank - avail); got = nb < nhi; }
                pool_lo = nlo + (need - avail);
                pool_hi = nhi > pool_lo ? nhi : pool_lo;
            } else {
                pool_lo += need < avail ? need : avail;
            }
            if (!active && got) {
                active = true;
                took = true;
                b = problem(nb);
                const double *src = A.worklist ? A.q_out : A.q0;   // (second phase: the first phase's iterate)
                IKD_UNROLL
                for (int i = 0; i < T.nq; ++i) ws[T.off_q + i] = src[at(A.layout, A.B, T.nq, i, b)];
                tl = target_rows(T, b);
            }
        }
        return __any(active) != 0;
    }
};

template <class TB, class WS>
__device__ __forceinline__ void dls_generic_refill_body(const GenericKernelArgs &a, const TB &T, int64_t wave, int64_t nwaves, const WS &ws,
                                                        unsigned long long *queue, int chunk) {
    const int lane = static_cast<int>(threadIdx.x) & 63;
    const int64_t nwork = a.worklist ? static_cast<int64_t>(*a.count) : a.B;
    // waves without a share of the first round leave at once and are not counted below (chain_kernel_body.hpp chain_refill_loop)
    const int64_t working = (nwork + 63) / 64 < nwaves ? (nwork + 63) / 64 : nwaves;
    if (wave >= working) return;
    GenericRefill rf{&a, queue, chunk & 0xffff, chunk >> 16, 0, nwaves * 64, 0, 0, nwaves * 64 >= nwork, wave * 64 + lane < nwork, false, nwork};
    rf.b = rf.problem(rf.start ? wave * 64 + lane : 0);   // a tail lane of the first round shadows a valid problem
    const double *src = a.worklist ? a.q_out : a.q0;
    IKD_UNROLL
    for (int i = 0; i < T.nq; ++i) ws[T.off_q + i] = src[at(a.layout, a.B, T.nq, i, rf.b)];
    int iters;
    bool success;
    generic_dls(T, a.prm, ws, rf.target_rows(T, rf.b), iters, success, [](bool act) { return __any(act) != 0; }, rf);
    if (lane == 0) {   // the last wave out resets the queue slot for the stream's next launch (kernels.hpp QueuePool)
        __threadfence();
        if (atomicAdd(queue + 1, 1ull) == static_cast<unsigned long long>(working) - 1ull) {
            queue[0] = 0ull;
            queue[1] = 0ull;
            queue[2] = 0ull;   // (the two-phase worklist's length)
            __threadfence();
        }
    }
}
#endif

template <class AnyFn>
IKD_FN void dls_generic_body(const GenericKernelArgs &a, int64_t gid, AnyFn any_active) {
    dls_generic_body_ws(a, a.T, gid, Ws{a.ws + gid, a.ws_stride}, any_active);  // tail lanes own (padding) workspace columns too: ws_stride is a multiple of 64
}

// Stage kernel: e, dense J and the world placement of each task frame.
IKD_FN void eval_generic_body(const GenericKernelArgs &a, int64_t gid) {
    if (gid >= a.B) return;
    const int64_t b = gid;
    const Ws ws{a.ws + gid, a.ws_stride};
    for (int i = 0; i < a.T.nq; ++i) ws[a.T.off_q + i] = a.q0[at(a.layout, a.B, a.T.nq, i, b)];
    const int M = a.T.M, nv = a.T.nv;
    if (a.e_out || a.J_out) {
        const double *tl = a.layout == LAYOUT_SOA ? a.targets + b : a.targets + b * a.T.ntasks * 12;
        const int64_t ts = a.layout == LAYOUT_SOA ? a.B : 1;
        generic_evaluate(a.T, ws, tl, ts);
    } else {
        generic_fk(a.T, ws);  // task-frame FK only: no targets are read
    }
    if (a.e_out)
        for (int r = 0; r < M; ++r) a.e_out[at(a.layout, a.B, M, r, b)] = ws[a.T.off_e + r];
    if (a.J_out)
        for (int r = 0; r < M; ++r)
            for (int c = 0; c < nv; ++c) a.J_out[at(a.layout, a.B, M * nv, r * nv + c, b)] = ws[a.T.off_J + r * nv + c];
    if (a.oMf_out)
        for (int t = 0; t < a.T.ntasks; ++t) {
            double oJ[12], oMf[12];
            if (a.T.t_type[t] == GT_POSTURE_ROW) {  // no frame: report the identity
                for (int k = 0; k < 12; ++k) a.oMf_out[at(a.layout, a.B, a.T.ntasks * 12, t * 12 + k, b)] = (k < 9 && k % 4 == 0) ? 1.0 : 0.0;
                continue;
            }
            for (int k = 0; k < 12; ++k) oJ[k] = ws[a.T.off_oMi + 12 * a.T.t_fjoint[t] + k];
            g_se3_mul(oJ, a.T.t_fpl + 12 * t, oMf);
            for (int k = 0; k < 12; ++k) a.oMf_out[at(a.layout, a.B, a.T.ntasks * 12, t * 12 + k, b)] = oMf[k];
        }
}

}  // namespace ikdev
