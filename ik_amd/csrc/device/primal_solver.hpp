// primal_solver.hpp -- ik::dls() as a compiled lane program in PRIMAL, TREE-SPARSE form: what a static lane program (rtc.cpp
// generic_static_source) runs when the dense dual program of generic_solver.hpp would not fit the register file (more than 12
// solved rows).  Same iteration as reference ik/ik/dls.cpp:5-78 -- evaluate, solve, stop test, integrate, clip -- with the solve
// turned around:
//
//     dq = -J^T (J J^T + lambda^2 I)^-1 e  =  -(J^T J + lambda^2 I)^-1 J^T e              (dls.cpp:39-53; the two are the same vector)
//
// and the nv x nv normal matrix H = lambda^2 I + sum_t J_t^T J_t never formed densely.  A frame task's Jacobian block is supported on
// the path from its frame's joint to the root (getFrameJacobian, ik/ik/frame.hpp:152-182), so H couples two tangent directions only
// when one's joint is an ancestor of the other's: eliminating the directions deepest-first (descending index: the model numbers its
// joints depth-first) is a Cholesky factorisation WITHOUT FILL -- column c of the factor lives on c and its ancestors.
// For Cassie with both feet and the pelvis: 70 + 70 + 21 = 161 entries against 253 dense, or the 171-entry dual Gram matrix of the
// same problem plus its 192 Jacobian entries.  That is the arrow structure the hand-written tree kernel (tree_solver.hpp) exploits
// for ONE shape; here the generator emits it for whatever tree and task list the problem has:
//
//   * one sweep c = nv-1 ... 0: the tasks anchored at c (deepest direction of their path) are evaluated -- forward kinematics along
//     the task's own path, error, Jlog6, the block's columns (no whole-tree FK, no world Jacobian, no dense J: a block lives only
//     until it is accumulated into H and g = J^T e) -- then direction c is eliminated (pivot, column, rank-one update of its
//     ancestors, forward substitution of g).  PostureTask rows (ik/ik/posture.hpp:51-68) are a diagonal entry and an entry of g.
//   * finished columns of the factor are PARKED in an LDS slab [word][lane] (conflict-free: lane l reads word w at w * 64 + l) when
//     the generator says so (TB::park_off): they are next needed in the back substitution, one sweep later.  40 KB per wave: four
//     one-wave workgroups per CU.
//   * every index is a compile-time constant after unrolling, so H, g and the block are registers, a structural zero of a placement
//     or of the coupling pattern is never computed, and a direction no task moves is dead code (its dq is zero).
//
// Accuracy: the primal solve loses kappa_2(H) u = (sigma_1^2 / lambda^2) u against the dual's u kappa_2(J J^T + lambda^2 I) when J has a
// null space (tests/test_gpu_full_size.py, rule S3'): ~1e-12 rad per step at the default damping instead of ~1e-13 -- six orders
// below the 1e-6 rad bar.  Not taken: CentreOfMassTask rows (dense over every direction) and FrameConstraint rows.
#pragma once
#include "generic_solver.hpp"

namespace ikdev {

#ifdef IKD_STATIC_TABLES

// Word w of this lane in the LDS slab [word][lane] (device), or of a plain array (host emulation: stride 1).
struct LdsColumn {
    double *base;
    int stride;
    IKD_FN double &operator[](int w) const { return base[w * stride]; }
};

// The lane's workspace as the primal program sees it: q lives in the LDS slab (words [0, nq): read once and written once per iteration
// -- what the register allocator would spill to scratch memory first), everything else in the caller's local array (registers).
template <class TB>
struct WsPrimal {
    double *w;
    LdsColumn lds;
    IKD_FN double &operator[](int i) const { return (i >= TB::off_q && i < TB::off_q + TB::nq) ? lds[i - TB::off_q] : w[i]; }
};

// One frame / alignment task on the path of its frame's joint: FK along the path, the error, the block's columns, accumulated into
// H (packed by TB::hidx) and g.  The arithmetic of a column is generic_evaluate's (generic_solver.hpp), which cites the reference.
template <class TB, class WS>
IKD_FN void primal_frame_task(const TB &T, const WS &ws, const double (&tgt)[12], const int t, double (&H)[TB::nnz], double (&g)[TB::nv],
                              double &e0sq, double (&Jt)[6][TB::max_pdofs], double (&er)[6]) {
    constexpr int NV = TB::nv, MP = TB::max_path, MD = TB::max_pdofs;
    const int fj = TB::t_fjoint[t], rj = TB::t_rjoint[t], type = TB::t_type[t], row = TB::t_row[t], dim = TB::t_dim[t];
    const auto w6 = TB::t_w + 6 * t;
    // forward kinematics along the path root -> frame joint (ik/ik/data.cpp:28-29 restricted to what this task reads).  Only the frame's
    // placement and each joint's (sin, cos) are kept: the block's columns come from a walk BACK along the path (below), which needs
    // neither the world axes nor the origins of the joints -- 2 doubles per joint alive between the passes instead of 6
    double oM[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
    double sn[MP], cs[MP], qbase[7];
    IKD_UNROLL
    for (int k = 0; k < TB::t_npath[t]; ++k) {
        const int j = TB::t_path[t * MP + k];
        const int iq = TB::idx_q[j], jt = TB::jtype[j];
        const auto a = TB::axis + 3 * j;
        double Mj[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0}, li[12], nx[12];
        if (jt == GJ_REVOLUTE || jt == GJ_REVOLUTE_UNBOUNDED) {
            double s, c;
            if (jt == GJ_REVOLUTE) {
                // (each task walks its own copy of the angle: left to itself the compiler merges the FK of two tasks that share a
                // path -- and then keeps the first task's whole walk alive until the second is done: 2 KB of scratch per lane on
                // Cassie with two frames on one foot)
                double qj = ws[T.off_q + iq];
                IKD_PIN(qj);
                dsincos(qj, s, c);
            } else { c = ws[T.off_q + iq]; s = ws[T.off_q + iq + 1]; }
            sn[k] = s; cs[k] = c;
            const double kk = 1.0 - c;
            Mj[0] = c + kk * a[0] * a[0];        Mj[1] = kk * a[0] * a[1] - s * a[2]; Mj[2] = kk * a[0] * a[2] + s * a[1];
            Mj[3] = kk * a[1] * a[0] + s * a[2]; Mj[4] = c + kk * a[1] * a[1];        Mj[5] = kk * a[1] * a[2] - s * a[0];
            Mj[6] = kk * a[2] * a[0] - s * a[1]; Mj[7] = kk * a[2] * a[1] + s * a[0]; Mj[8] = c + kk * a[2] * a[2];
        } else if (jt == GJ_PRISMATIC) {
            const double v = ws[T.off_q + iq];
            sn[k] = v; cs[k] = 0.0;
            Mj[9] = a[0] * v; Mj[10] = a[1] * v; Mj[11] = a[2] * v;
        } else if (jt == GJ_FREEFLYER) {
            double R[9];
            IKD_UNROLL
            for (int i = 0; i < 7; ++i) qbase[i] = ws[T.off_q + iq + i];
            quat_to_R(qbase, R);
            IKD_UNROLL
            for (int i = 0; i < 9; ++i) Mj[i] = R[i];
            Mj[9] = qbase[0]; Mj[10] = qbase[1]; Mj[11] = qbase[2];
        }
        g_se3_mul(TB::placement + 12 * j, Mj, li);
        g_se3_mul(oM, li, nx);
        IKD_UNROLL
        for (int i = 0; i < 12; ++i) oM[i] = nx[i];
    }
    double oMf[12], oMr[12], tg[12];
    g_se3_mul(oM, TB::t_fpl + 12 * t, oMf);
    if (rj == 0) {   // the universe: the reference placement is a constant
        IKD_UNROLL
        for (int i = 0; i < 12; ++i) oMr[i] = TB::t_rpl[12 * t + i];
    } else {         // a moving reference frame: FK along ITS path (placement only -- the reference's Jacobian ignores its motion, frame.hpp:152-182)
        double oR[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
        IKD_UNROLL
        for (int k = 0; k < TB::t_nrpath[t]; ++k) {
            const int j = TB::t_rpath[t * MP + k];
            const int iq = TB::idx_q[j], jt = TB::jtype[j];
            const auto a = TB::axis + 3 * j;
            double Mj[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0}, li[12], nx[12];
            if (jt == GJ_REVOLUTE || jt == GJ_REVOLUTE_UNBOUNDED) {
                double s, c;
                if (jt == GJ_REVOLUTE) dsincos(ws[T.off_q + iq], s, c);
                else { c = ws[T.off_q + iq]; s = ws[T.off_q + iq + 1]; }
                const double kk = 1.0 - c;
                Mj[0] = c + kk * a[0] * a[0];        Mj[1] = kk * a[0] * a[1] - s * a[2]; Mj[2] = kk * a[0] * a[2] + s * a[1];
                Mj[3] = kk * a[1] * a[0] + s * a[2]; Mj[4] = c + kk * a[1] * a[1];        Mj[5] = kk * a[1] * a[2] - s * a[0];
                Mj[6] = kk * a[2] * a[0] - s * a[1]; Mj[7] = kk * a[2] * a[1] + s * a[0]; Mj[8] = c + kk * a[2] * a[2];
            } else if (jt == GJ_PRISMATIC) {
                const double v = ws[T.off_q + iq];
                Mj[9] = a[0] * v; Mj[10] = a[1] * v; Mj[11] = a[2] * v;
            } else if (jt == GJ_FREEFLYER) {
                double qb[7], R[9];
                IKD_UNROLL
                for (int i = 0; i < 7; ++i) qb[i] = ws[T.off_q + iq + i];
                quat_to_R(qb, R);
                IKD_UNROLL
                for (int i = 0; i < 9; ++i) Mj[i] = R[i];
                Mj[9] = qb[0]; Mj[10] = qb[1]; Mj[11] = qb[2];
            }
            g_se3_mul(TB::placement + 12 * j, Mj, li);
            g_se3_mul(oR, li, nx);
            IKD_UNROLL
            for (int i = 0; i < 12; ++i) oR[i] = nx[i];
        }
        g_se3_mul(oR, TB::t_rpl + 12 * t, oMr);
    }
    const double Rf[9] = {oMf[0], oMf[1], oMf[2], oMf[3], oMf[4], oMf[5], oMf[6], oMf[7], oMf[8]};
    const double pf[3] = {oMf[9], oMf[10], oMf[11]};

    // the block: rows [row, row + dim), one column per tangent direction of the path; Jt[r][kc], kc counting the path's directions
    LogAndJlog lj;
    double galign[3] = {0.0, 0.0, 0.0};
    const bool align = type >= GT_ALIGN_X;
    const int r0 = (type == GT_ORIENTATION) ? 3 : 0;
    if (align) {   // AlignAxisTask, ik/ik/frame.hpp:257-301
        galign[0] = tgt[9]; galign[1] = tgt[10]; galign[2] = tgt[11];
        double rMf[12];
        g_se3_inv_mul(oMr, oMf, rMf);
        const int axn = type - GT_ALIGN_X;
        const double r[3] = {rMf[axn], rMf[3 + axn], rMf[6 + axn]};
        const double inv = drsqrt(dfma(galign[0], galign[0], dfma(galign[1], galign[1], galign[2] * galign[2])));
        const double tn[3] = {galign[0] * inv, galign[1] * inv, galign[2] * inv};
        double rxt[3];
        cross(r, tn, rxt);
        galign[0] = dfma(rxt[0], rMf[0], dfma(rxt[1], rMf[3], rxt[2] * rMf[6]));
        galign[1] = dfma(rxt[0], rMf[1], dfma(rxt[1], rMf[4], rxt[2] * rMf[7]));
        galign[2] = dfma(rxt[0], rMf[2], dfma(rxt[1], rMf[5], rxt[2] * rMf[8]));
        er[0] = (1.0 - dot(r, tn)) * w6[0];
    } else {
        IKD_UNROLL
        for (int i = 0; i < 12; ++i) tg[i] = tgt[i];
        double oMt[12], Re[9], pe[3];
        g_se3_mul(oMr, tg, oMt);                        // frame.hpp:48
        IKD_UNROLL
        for (int i = 0; i < 3; ++i)
            IKD_UNROLL
            for (int j = 0; j < 3; ++j) Re[3 * i + j] = dfma(Rf[i], oMt[j], dfma(Rf[3 + i], oMt[3 + j], Rf[6 + i] * oMt[6 + j]));
        const double dp[3] = {oMt[9] - pf[0], oMt[10] - pf[1], oMt[11] - pf[2]};
        rotT_vec(Rf, dp, pe);
        log6_and_jlog6_inv(Re, pe, lj);                 // frame.hpp:50-61, :162-166
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (k >= r0 && k < r0 + dim) er[k - r0] = lj.e[k] * w6[k - r0];
    }
    IKD_UNROLL
    for (int r = 0; r < dim; ++r) {
        ws[T.off_e + row + r] = er[r];
        if (TB::t_prio[t] == 0) e0sq = dfma(er[r], er[r], e0sq);
    }
    // The block's columns, walking BACK from the frame to the root with fM = oMf^-1 oMj, the placement of joint j (after its motion)
    // seen from the frame: a revolute joint's column of getFrameJacobian(LOCAL) is Ad(fM) [0; a] = [p x (R a); R a], a prismatic
    // joint's [R a; 0], the free-flyer's six are Ad(fM) itself -- no world Jacobian, no rotation into the frame.
    // fM_parent = fM_j (placement_j M_j(q_j))^-1.  Column slots run root-first (kc), as the accumulation below reads them.
    double fM[12];
    {   // fM of the frame's joint: the inverse of the frame's placement on it (constants: folds)
        const auto F = TB::t_fpl + 12 * t;
        IKD_UNROLL
        for (int i = 0; i < 3; ++i) {
            IKD_UNROLL
            for (int j = 0; j < 3; ++j) fM[3 * i + j] = F[3 * j + i];
            fM[9 + i] = -(F[i] * F[9] + F[3 + i] * F[10] + F[6 + i] * F[11]);
        }
    }
    int kend = 0;   // number of directions on the path (a compile-time value after unrolling)
    IKD_UNROLL
    for (int k = 0; k < TB::t_npath[t]; ++k) kend += TB::jtype[TB::t_path[t * MP + k]] == GJ_FREEFLYER ? 6 : 1;
    int kc = kend;
    IKD_UNROLL
    for (int k = TB::t_npath[t] - 1; k >= 0; --k) {
        const int j = TB::t_path[t * MP + k];
        const int jt = TB::jtype[j];
        const int n = jt == GJ_FREEFLYER ? 6 : 1;
        const auto a = TB::axis + 3 * j;
        kc -= n;
        const double Ra[3] = {dfma(fM[0], a[0], dfma(fM[1], a[1], fM[2] * a[2])), dfma(fM[3], a[0], dfma(fM[4], a[1], fM[5] * a[2])),
                              dfma(fM[6], a[0], dfma(fM[7], a[1], fM[8] * a[2]))};
        const double pM[3] = {fM[9], fM[10], fM[11]};
        IKD_UNROLL
        for (int u = 0; u < n; ++u) {
            double vl[3], wl[3];
            if (jt == GJ_FREEFLYER) {
                const int cc = u < 3 ? u : u - 3;
                const double Rc[3] = {fM[cc], fM[3 + cc], fM[6 + cc]};
                if (u < 3) { vl[0] = Rc[0]; vl[1] = Rc[1]; vl[2] = Rc[2]; wl[0] = 0.0; wl[1] = 0.0; wl[2] = 0.0; }
                else { cross(pM, Rc, vl); wl[0] = Rc[0]; wl[1] = Rc[1]; wl[2] = Rc[2]; }
            } else if (jt == GJ_PRISMATIC) {
                vl[0] = Ra[0]; vl[1] = Ra[1]; vl[2] = Ra[2]; wl[0] = 0.0; wl[1] = 0.0; wl[2] = 0.0;
            } else {
                cross(pM, Ra, vl);
                wl[0] = Ra[0]; wl[1] = Ra[1]; wl[2] = Ra[2];
            }
            if (align) {
                Jt[0][kc + u] = -w6[0] * dot(galign, wl);
            } else {
                double out[6];
                IKD_UNROLL
                for (int i = 0; i < 3; ++i) {
                    out[i] = -dfma(lj.A[3 * i], vl[0], dfma(lj.A[3 * i + 1], vl[1], dfma(lj.A[3 * i + 2], vl[2],
                              dfma(lj.Bm[3 * i], wl[0], dfma(lj.Bm[3 * i + 1], wl[1], lj.Bm[3 * i + 2] * wl[2])))));
                    out[3 + i] = -dfma(lj.A[3 * i], wl[0], dfma(lj.A[3 * i + 1], wl[1], lj.A[3 * i + 2] * wl[2]));
                }
#pragma unroll
                for (int q = 0; q < 6; ++q)
                    if (q >= r0 && q < r0 + dim) Jt[q - r0][kc + u] = w6[q - r0] * out[q];
            }
        }
        if (k > 0) {   // fM <- fM (placement_j M_j)^-1 = (fM M_j^-1) placement_j^-1
            double X[12];
            if (jt == GJ_REVOLUTE || jt == GJ_REVOLUTE_UNBOUNDED) {   // M_j^-1: the rotation by -q about a, no translation
                const double s = -sn[k], c = cs[k], kk = 1.0 - c;
                const double Mi[9] = {c + kk * a[0] * a[0],        kk * a[0] * a[1] - s * a[2], kk * a[0] * a[2] + s * a[1],
                                      kk * a[1] * a[0] + s * a[2], c + kk * a[1] * a[1],        kk * a[1] * a[2] - s * a[0],
                                      kk * a[2] * a[0] - s * a[1], kk * a[2] * a[1] + s * a[0], c + kk * a[2] * a[2]};
                IKD_UNROLL
                for (int i = 0; i < 3; ++i) {
                    IKD_UNROLL
                    for (int jj = 0; jj < 3; ++jj) X[3 * i + jj] = dfma(fM[3 * i], Mi[jj], dfma(fM[3 * i + 1], Mi[3 + jj], fM[3 * i + 2] * Mi[6 + jj]));
                    X[9 + i] = fM[9 + i];
                }
            } else if (jt == GJ_PRISMATIC) {                          // M_j^-1: the translation by -q a
                IKD_UNROLL
                for (int i = 0; i < 9; ++i) X[i] = fM[i];
                IKD_UNROLL
                for (int i = 0; i < 3; ++i) X[9 + i] = dfma(-sn[k], Ra[i], fM[9 + i]);
            } else {                                                   // (a free-flyer below the root does not occur: it IS the root joint)
                IKD_UNROLL
                for (int i = 0; i < 12; ++i) X[i] = fM[i];
            }
            const auto P = TB::placement + 12 * j;                    // X placement^-1 = (R_X R_P^T, p_X - R_X R_P^T p_P)
            IKD_UNROLL
            for (int i = 0; i < 3; ++i) {
                IKD_UNROLL
                for (int jj = 0; jj < 3; ++jj) fM[3 * i + jj] = dfma(X[3 * i], P[3 * jj], dfma(X[3 * i + 1], P[3 * jj + 1], X[3 * i + 2] * P[3 * jj + 2]));
            }
            IKD_UNROLL
            for (int i = 0; i < 3; ++i) fM[9 + i] = X[9 + i] - dfma(fM[3 * i], P[9], dfma(fM[3 * i + 1], P[10], fM[3 * i + 2] * P[11]));
        }
    }
    // H += J_t^T J_t, g += J_t^T e_t on the path's SHARED directions (those other anchors' tasks move too: the base of a humanoid):
    // their rows collect contributions from several blocks and are accumulated here.  A direction private to this anchor's tasks
    // forms its row when its turn comes in the sweep (primal_dls), straight from the blocks -- no entry of H is held for it.
    int ka = 0;
    IKD_UNROLL
    for (int k = 0; k < TB::t_npath[t]; ++k) {
        const int j = TB::t_path[t * MP + k];
        const int n = TB::jtype[j] == GJ_FREEFLYER ? 6 : 1;
        IKD_UNROLL
        for (int u = 0; u < n; ++u) {
            const int ca = TB::idx_v[j] + u;
            if (TB::shared[ca]) {
                double sg = g[ca];
                IKD_UNROLL
                for (int r = 0; r < dim; ++r) sg = dfma(Jt[r][ka], er[r], sg);
                g[ca] = sg;
                int kb = 0;
                IKD_UNROLL
                for (int k2 = 0; k2 <= k; ++k2) {
                    const int j2 = TB::t_path[t * MP + k2];
                    const int n2 = TB::jtype[j2] == GJ_FREEFLYER ? 6 : 1;
                    IKD_UNROLL
                    for (int u2 = 0; u2 < n2; ++u2) {
                        const int cb = TB::idx_v[j2] + u2;
                        if (cb <= ca && TB::shared[cb]) {
                            double s = H[TB::hidx[ca * NV + cb]];
                            IKD_UNROLL
                            for (int r = 0; r < dim; ++r) s = dfma(Jt[r][ka], Jt[r][kb], s);
                            H[TB::hidx[ca * NV + cb]] = s;
                        }
                        ++kb;
                    }
                }
            }
            ++ka;
        }
    }
}

// One full solve on the workspace (q already stored at off_q): the loop of generic_dls with the primal tree-sparse step.
template <class TB, class WS, class AnyFn>
IKD_FN void primal_dls(const TB &T, const LoopParams &prm, const WS &ws, const LdsColumn &lds, const LaneRows &targets,
                       int &iters_out, bool &success_out, AnyFn any_active) {
    constexpr int NV = TB::nv;
    bool active = true, success = false;
    int iters = prm.max_iterations;
    for (int it = 0; it < prm.max_iterations; ++it) {
        double H[TB::nnz], g[NV], x[NV], e0sq = 0.0;
        double Jt[TB::ntasks][6][TB::max_pdofs], er[TB::ntasks][6];   // the blocks of the tasks whose private rows are still to come
        // every task's target, loaded in ONE go at the top of the iteration (one exposed round trip to HBM / L2 per iteration instead of
        // one per task; the words wait in registers -- 12 per pose -- until their task's turn, the first tasks of the sweep are the
        // lightest phase).  Re-read every iteration: kept across the loop they would cost their registers for the whole solve.
        double tgt[TB::ntasks][12];
        IKD_UNROLL
        for (int t = 0; t < TB::ntasks; ++t) {
            if (TB::t_type[t] == GT_POSTURE_ROW) {
                double tv[1];
                targets.template run_fresh<1>(t * 12 + 9, tv);
                tgt[t][9] = tv[0];
            } else if (TB::t_type[t] >= GT_ALIGN_X) {
                double tv[3];
                targets.template run_fresh<3>(t * 12 + 9, tv);
                tgt[t][9] = tv[0]; tgt[t][10] = tv[1]; tgt[t][11] = tv[2];
            } else {
                targets.template run_fresh<12>(t * 12, tgt[t]);
            }
        }
        IKD_SCHED_FENCE();
        IKD_UNROLL
        for (int c = 0; c < NV; ++c) {
            g[c] = 0.0;
            if (TB::hidx[c * NV + c] < 0) continue;
            H[TB::hidx[c * NV + c]] = prm.lam2;
            IKD_UNROLL
            for (int k = 0; k < TB::anc_n[c]; ++k) H[TB::hidx[c * NV + TB::anc[c * TB::max_anc + k]]] = 0.0;
        }
        IKD_UNROLL
        for (int c = NV - 1; c >= 0; --c) {
            IKD_UNROLL
            for (int i = 0; i < TB::anch_n[c]; ++i) {   // the tasks anchored at c (listed by the generator: nt task bodies, not nv x nt)
                const int t = TB::anch_t[c * TB::ntasks + i];
                if (TB::t_type[t] == GT_POSTURE_ROW) {   // one row of ik::PostureTask, ik/ik/posture.hpp:51-68: a diagonal entry
                    const auto w6 = TB::t_w + 6 * t;
                    const double e = (ws[T.off_q + TB::t_rjoint[t]] - tgt[t][9]) * w6[1] * w6[0];
                    ws[T.off_e + TB::t_row[t]] = e;
                    if (TB::t_prio[t] == 0) e0sq = dfma(e, e, e0sq);
                    H[TB::hidx[c * NV + c]] = dfma(w6[0], w6[0], H[TB::hidx[c * NV + c]]);
                    g[c] = dfma(w6[0], e, g[c]);
                } else {
                    IKD_SCHED_FENCE();   // one task at a time: nothing of the next task's FK may start while this block is live
                    primal_frame_task(T, ws, tgt[t], t, H, g, e0sq, Jt[t], er[t]);
                    IKD_SCHED_FENCE();
                }
            }
            if (TB::hidx[c * NV + c] < 0) continue;      // a direction no task moves
            constexpr int MA = TB::max_anc;
            if (TB::shared[c]) {
                // a SHARED direction (all of its ancestors are shared too): its row sits complete in H -- the blocks' contributions and
                // the rank-one updates of every deeper direction.  Pivot, column, rank-one update of the ancestors, forward substitution.
                const double inv = drsqrt(H[TB::hidx[c * NV + c]]);
                const double y = g[c] * inv;
                IKD_UNROLL
                for (int k = 0; k < TB::anc_n[c]; ++k) {
                    const int d = TB::anc[c * MA + k];
                    const double l = H[TB::hidx[c * NV + d]] * inv;
                    H[TB::hidx[c * NV + d]] = l;
                    g[d] = dfma(-l, y, g[d]);
                }
                IKD_UNROLL
                for (int k = 0; k < TB::anc_n[c]; ++k) {
                    const int d = TB::anc[c * MA + k];
                    IKD_UNROLL
                    for (int k2 = 0; k2 <= k; ++k2) {   // (anc is ascending: e <= d)
                        const int e = TB::anc[c * MA + k2];
                        H[TB::hidx[d * NV + e]] = dfma(-H[TB::hidx[c * NV + d]], H[TB::hidx[c * NV + e]], H[TB::hidx[d * NV + e]]);
                    }
                }
                H[TB::hidx[c * NV + c]] = inv;
                g[c] = y;
            } else {
                // a direction PRIVATE to one anchor's tasks (a leg joint): its row is formed NOW, left-looking -- from the blocks of that
                // anchor's tasks (still alive: they die with the anchor's last private direction) minus the columns of the deeper private
                // directions -- so no entry of H was ever held for it: while a 6 x 13 block is alive the leg's 70 entries are not.
                double diag = H[TB::hidx[c * NV + c]], gc = g[c], row[MA];   // (lambda^2 and the direction's own PostureTask row)
                IKD_UNROLL
                for (int k = 0; k < TB::anc_n[c]; ++k) row[k] = 0.0;
                IKD_UNROLL
                for (int t = 0; t < TB::ntasks; ++t) {
                    if (TB::t_type[t] == GT_POSTURE_ROW || TB::t_anchor[t] != TB::grp[c] || TB::t_kc[t * NV + c] < 0) continue;
                    const int kc = TB::t_kc[t * NV + c];
                    IKD_UNROLL
                    for (int r = 0; r < TB::t_dim[t]; ++r) {
                        diag = dfma(Jt[t][r][kc], Jt[t][r][kc], diag);
                        gc = dfma(Jt[t][r][kc], er[t][r], gc);
                    }
                    IKD_UNROLL
                    for (int k = 0; k < TB::anc_n[c]; ++k) {
                        const int kd = TB::t_kc[t * NV + TB::anc[c * MA + k]];
                        IKD_UNROLL
                        for (int r = 0; r < TB::t_dim[t]; ++r) row[k] = dfma(Jt[t][r][kc], Jt[t][r][kd], row[k]);
                    }
                }
                IKD_UNROLL
                for (int a = NV - 1; a > c; --a) {   // deeper private directions of the same anchor that have c as an ancestor
                    if (TB::shared[a] || TB::grp[a] != TB::grp[c] || TB::hidx[a * NV + c] < 0) continue;
                    const double lca = H[TB::hidx[a * NV + c]];
                    diag = dfma(-lca, lca, diag);
                    gc = dfma(-lca, g[a], gc);
                    IKD_UNROLL
                    for (int k = 0; k < TB::anc_n[c]; ++k) row[k] = dfma(-lca, H[TB::hidx[a * NV + TB::anc[c * MA + k]]], row[k]);
                }
                const double inv = drsqrt(diag);
                const double y = gc * inv;
                IKD_UNROLL
                for (int k = 0; k < TB::anc_n[c]; ++k) H[TB::hidx[c * NV + TB::anc[c * MA + k]]] = row[k] * inv;
                H[TB::hidx[c * NV + c]] = inv;
                g[c] = y;
                IKD_UNROLL
                for (int k = 0; k < TB::anc_n[c]; ++k) {   // right-looking onto the SHARED ancestors (their rows live in H)
                    const int d = TB::anc[c * MA + k];
                    if (!TB::shared[d]) continue;
                    g[d] = dfma(-H[TB::hidx[c * NV + d]], y, g[d]);
                    IKD_UNROLL
                    for (int k2 = 0; k2 <= k; ++k2) {
                        const int e = TB::anc[c * MA + k2];
                        if (!TB::shared[e]) continue;
                        H[TB::hidx[d * NV + e]] = dfma(-H[TB::hidx[c * NV + d]], H[TB::hidx[c * NV + e]], H[TB::hidx[d * NV + e]]);
                    }
                }
            }
            // finished columns wait in LDS for the back substitution -- a private direction's once its anchor's last private row is
            // formed (the rows above read the deeper columns from registers)
            IKD_UNROLL
            for (int a = NV - 1; a >= c; --a) {
                if (TB::park_off[a] < 0 || TB::park_at[a] != c) continue;
                const int o = TB::park_off[a];
                lds[o] = H[TB::hidx[a * NV + a]];
                lds[o + 1] = g[a];
                IKD_UNROLL
                for (int k = 0; k < TB::anc_n[a]; ++k) lds[o + 2 + k] = H[TB::hidx[a * NV + TB::anc[a * MA + k]]];
            }
        }
        // back substitution, root first: x_c = (y_c - sum_{d in anc(c)} L_dc x_d) / l_cc;  dq = -x
        IKD_UNROLL
        for (int c = 0; c < NV; ++c) {
            constexpr int MA = TB::max_anc;
            if (TB::hidx[c * NV + c] < 0) { x[c] = 0.0; continue; }
            double inv, s;
            if (TB::park_off[c] >= 0) {
                const int o = TB::park_off[c];
                inv = lds[o];
                s = lds[o + 1];
                IKD_UNROLL
                for (int k = 0; k < TB::anc_n[c]; ++k) s = dfma(-lds[o + 2 + k], x[TB::anc[c * MA + k]], s);
            } else {
                inv = H[TB::hidx[c * NV + c]];
                s = g[c];
                IKD_UNROLL
                for (int k = 0; k < TB::anc_n[c]; ++k) s = dfma(-H[TB::hidx[c * NV + TB::anc[c * MA + k]]], x[TB::anc[c * MA + k]], s);
            }
            x[c] = s * inv;
        }
        IKD_UNROLL
        for (int c = 0; c < NV; ++c) ws[T.off_dq + c] = -x[c];
        // inverse_kinematics_visitor::should_stop(ik, e, dq) (ik/ik/visitor.hpp:15-21; dls.cpp:61-64) and its derived family, as generic_dls
        bool err_ok = (prm.stop_sq_tol >= 0.0) && (e0sq < prm.stop_sq_tol);
        if (prm.nlt > 0) {
            err_ok = true;
            IKD_UNROLL
            for (int l = 0; l < TB::nlevels; ++l) {
                double s = 0.0;
                IKD_UNROLL
                for (int r = TB::lvl_row0[l]; r < TB::lvl_row0[l + 1]; ++r) s = dfma(ws[T.off_e + r], ws[T.off_e + r], s);
                if (l < prm.nlt) err_ok = err_ok && (s < prm.lvl_tol[l < 8 ? l : 7]);
            }
        }
        bool step_small = false;
        if (prm.dq_sq_tol > 0.0) {
            double s = 0.0;
            IKD_UNROLL
            for (int c = 0; c < NV; ++c) s = dfma(x[c], x[c], s);
            step_small = s < prm.dq_sq_tol;
        }
        const bool stop_now = active && (err_ok || step_small);
        if (stop_now) { success = true; iters = it; }
        active = active && !stop_now;
        generic_integrate_clip(T, ws, prm.step_length, active);  // ik/ik/dls.cpp:67-71
        if (!any_active(active)) break;
    }
    iters_out = iters;
    success_out = success;
}

// What one lane of the primal program does (kernel ikgpu_lane_dls of a source generated with T::primal = 1).
template <class TB, class AnyFn>
IKD_FN void dls_primal_body_ws(const GenericKernelArgs &a, const TB &T, int64_t gid, const WsReg &regs, const LdsColumn &lds, AnyFn any_active,
                               int64_t group0 = -1) {
    const WsPrimal<TB> ws{regs.w, lds};
    const bool valid = gid < a.B;
    const int64_t b = valid ? gid : a.B - 1;
    IKD_UNROLL
    for (int i = 0; i < T.nq; ++i) ws[T.off_q + i] = a.q0[at(a.layout, a.B, T.nq, i, b)];
    const int64_t g0 = group0 >= 0 ? group0 : b;
    const int64_t tcol = a.layout == LAYOUT_SOA ? 1 : static_cast<int64_t>(T.ntasks) * 12;
    const LaneRows tl{reinterpret_cast<const char *>(a.targets + g0 * tcol), static_cast<uint32_t>((b - g0) * tcol * 8),
                      static_cast<int64_t>(a.layout == LAYOUT_SOA ? a.B * 8 : 8), group0 >= 0};
    int iters;
    bool success;
    primal_dls(T, a.prm, ws, lds, tl, iters, success, any_active);
    if (!valid) return;
    // (the store addresses are formed HERE: computed from the same b as the loads above, the compiler keeps all nq 64-bit indices --
    // 46 VGPRs on Cassie -- alive across the whole iteration loop)
    int64_t bs = b;
    IKD_PIN(bs);
    IKD_UNROLL
    for (int i = 0; i < T.nq; ++i) a.q_out[at(a.layout, a.B, T.nq, i, bs)] = ws[T.off_q + i];
    if (a.success) a.success[bs] = success ? 1 : 0;
    if (a.iters) a.iters[bs] = iters;
}

#endif  // IKD_STATIC_TABLES

}  // namespace ikdev
