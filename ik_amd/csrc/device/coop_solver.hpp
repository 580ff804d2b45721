// coop_solver.hpp -- ik::dls() for any model tree and task list (what generic_solver.hpp covers), with the per-problem
// workspace ON CHIP: sixteen lanes of a wave work on one problem whose state lives in LDS, four problems per 64-lane
// workgroup.  The per-lane generic program streams ~1.5 GB of workspace through HBM per iteration at B = 65536
// (profiles/r01_pmc/demo_*.csv: 77 GB per 50-iteration launch against 43 MB of algorithmic bytes); here HBM sees
// q0 and the targets once on entry and q once on exit.
//
// Same arithmetic as generic_solver.hpp (reference ik/ik/dls.cpp:5-78, ik/ik/data.cpp:25-58, ik/ik/frame.hpp:37-62,152-182,
// 257-301, ik/ik/posture.hpp:51-68), reorganised into phases whose items are independent and are dealt to the lanes of
// the group (IKC_FOR), with a workgroup barrier between phases (IKC_SYNC):
//   forward kinematics   one lane per joint builds its local transform; then, tree level by tree level, one lane per
//                        ELEMENT of oM_j = oM_parent * li_j for the joints of the level
//   joint Jacobian       one lane per joint
//   task error, blocks   one lane per task (log6, Jlog6 -> two 3 x 3 blocks; posture rows complete their row here)
//   task Jacobian        one lane per tangent column, all tasks
//   Gram matrix          one lane per entry of the lower triangle, augmented by the right-hand side as row M
//   Cholesky             right-looking, one phase per pivot, one lane per entry of the trailing triangle; row M rides
//                        along = forward substitution
//   back substitution    column oriented, one lane per remaining row
//   step, integrate      one lane per tangent column / joint
// Every expression and every order of accumulation is the one generic_solver.hpp uses, so the two forms of the generic
// kernel agree to rounding of the fused operations the compiler picks (measured: < 1e-9 rad after 200 iterations).
// On the host (tests/lane_emu) the same source runs with ONE lane that takes every item of a phase in turn.
//
// Measured (MI355X, demo task set M = 10 / nv = 22, B = 65536, 50 iterations): 18.8 ms per launch against 33.8 ms for the
// per-lane program, HBM traffic 0.14 GB against 77 GB.  The phases are LDS-latency bound (tools/coop_profile.py: tree
// levels 24 %, Cholesky 25 %, Gram 11 %, Jacobian columns 11 % of 65 k cycles per iteration) and the number of problems in
// flight is LDS-capacity bound (6 KB per problem + 7 KB of tables per workgroup: five workgroups per CU).
#pragma once
#include "generic_solver.hpp"

namespace ikdev {

constexpr int kCoopGroup = 16;      // lanes per problem
constexpr int kCoopPerBlock = 4;    // problems per 64-lane workgroup

#if IKD_ON_DEVICE
#define IKC_FOR(i, n) for (int i = g; i < (n); i += kCoopGroup)
#define IKC_SYNC() __syncthreads()
#else
#define IKC_FOR(i, n) for (int i = 0; i < (n); ++i)
#define IKC_SYNC() ((void)0)
#endif

// Per-phase cycle counters of workgroup 0 (debug builds only: make KERNEL_EXTRA=-DIKGPU_COOP_PROFILE)
#if defined(__HIPCC__) && defined(IKGPU_COOP_PROFILE)
__device__ long long g_coop_prof[16];
#endif
#if IKD_ON_DEVICE && defined(IKGPU_COOP_PROFILE)
#define IKC_TICK_INIT long long tick = clock64()
#define IKC_TICK(n) do { if (threadIdx.x == 0 && blockIdx.x == 0) { const long long now = clock64(); g_coop_prof[n] += now - tick; tick = now; } } while (0)
#define IKC_TICK_ARG , long long &tick
#define IKC_TICK_PASS , tick
#else
#define IKC_TICK_INIT ((void)0)
#define IKC_TICK(n) ((void)0)
#define IKC_TICK_ARG
#define IKC_TICK_PASS
#endif

struct CoopLayout {
    int q, tg, A0, A1, Jw, tb, e, J, G, dinv, x, dq, sf, words;  // offsets into the group's workspace, in doubles
    int cb, Jc, cnrm;            // FrameConstraint projection: per-constraint placements (36 each), Jc (Mc x nv), Mc scratch words
    int cholqr_c;                // 1: Jc Jc^T and its pivots fit the G region (the Cholesky-QR basis may be tried)
    const int *cpair_i, *cpair_j;  // [tri(Mc, 0)] lower triangle of Jc Jc^T by rows
    const int *csupp_f, *csupp_r;  // [ncons][nv] 1 when tangent column c moves the constrained frame / its reference frame
    int rounds, npairs;
    const int *support;          // [ntasks][nv] 1 when the task's rows touch tangent column c
    const int *pair_i, *pair_j;  // [npairs]
    const int *order, *chain_start, *lvl_start;  // joints 1.. chain by chain (problem.cpp: build_coop); first entry of each chain (+ end);
                                                 // first chain of each of the `rounds` levels (+ end)
    const int *tb_index;         // [ntasks] slot of the task's block in tb (-1: posture row, no block)
    const int *btask;            // [nblocks] the task of each block
    int nblocks;
    const int *col_joint;        // [nv] the joint a tangent column belongs to
    // PostureTask rows eliminated from the linear system (coop_dls): the Mf remaining rows, the posture tasks by tangent column
    // (CSR: pstart[nv + 1], ptask), the diagonal metric's inverse at Dd (nv words)
    int post_elim, Mf, Dd;
    const int *frow, *pstart, *ptask;
    const int *jrow;             // [ntasks] first row of the task's block in J (the task's row; posture rows eliminated: its row among the rest)
    // targets: block of task t at tg + tg_off[t] (twelve words; a posture row: one word, at + 9); LDS word i comes from slot tg_src[i]
    int ntg;
    const int *tg_off, *tg_src;
};

// C = A * B on 12-double SE(3) values held in the workspace (A, B) -> registers (C)
IKD_FN void coop_se3_mul_ws(const double *A, const double *B, double (&C)[12]) {
    double a[12], b[12];
    for (int k = 0; k < 12; ++k) { a[k] = A[k]; b[k] = B[k]; }
    g_se3_mul(a, b, C);
}

// evaluate_problem_data (ik/ik/data.cpp:25-58) on the group's workspace: q -> oMi, Jw, et, Jt.  Returns ||e[0]||^2 (the same in
// every lane).  Ends on a barrier.
IKD_FN double coop_evaluate(const GenericTables &T, const CoopLayout &L, const int g, double *ws IKC_TICK_ARG) {
    (void)g;
    const int nv = T.nv, nj = T.njoints, nt = T.ntasks;
    // ---- forward kinematics: every joint's local transform li_j = placement_j * motion_j(q), one lane per joint, parked
    // in the joint's own slot ...
    const int oMi = L.A1;
    IKC_FOR(j, nj) {
        double Mj[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0}, li[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
        if (j > 0) {
            const int iq = T.idx_q[j], jt = T.jtype[j];
            const double *a = T.axis + 3 * j;
            if (jt == GJ_REVOLUTE || jt == GJ_REVOLUTE_UNBOUNDED) {
                double s, c;
                if (jt == GJ_REVOLUTE) dsincos(ws[L.q + iq], s, c);
                else { c = ws[L.q + iq]; s = ws[L.q + iq + 1]; }   // a continuous joint's configuration IS (cos, sin)
                const double k = 1.0 - c;
                Mj[0] = c + k * a[0] * a[0];        Mj[1] = k * a[0] * a[1] - s * a[2]; Mj[2] = k * a[0] * a[2] + s * a[1];
                Mj[3] = k * a[1] * a[0] + s * a[2]; Mj[4] = c + k * a[1] * a[1];        Mj[5] = k * a[1] * a[2] - s * a[0];
                Mj[6] = k * a[2] * a[0] - s * a[1]; Mj[7] = k * a[2] * a[1] + s * a[0]; Mj[8] = c + k * a[2] * a[2];
            } else if (jt == GJ_PRISMATIC) {
                const double v = ws[L.q + iq];
                Mj[9] = a[0] * v; Mj[10] = a[1] * v; Mj[11] = a[2] * v;
            } else if (jt == GJ_FREEFLYER) {
                double qb[7], R[9];
                for (int k = 0; k < 7; ++k) qb[k] = ws[L.q + iq + k];
                quat_to_R(qb, R);
                for (int k = 0; k < 9; ++k) Mj[k] = R[k];
                Mj[9] = qb[0]; Mj[10] = qb[1]; Mj[11] = qb[2];
            }
            g_se3_mul(T.placement + 12 * j, Mj, li);
        }
        for (int k = 0; k < 12; ++k) ws[(j == 0 ? oMi : L.A0 + 12 * j) + k] = li[k];
    }
    IKC_SYNC();
    IKC_TICK(0);
    // ... then oM_j = oM_parent * li_j down every chain of the tree, one lane per chain with the running product in registers (the
    // same expression per element as g_se3_mul, so the same bits as the sequential pass of generic_fk); the chains of one level
    // hang off joints of earlier levels ...
    for (int lv = 0; lv < L.rounds; ++lv) {
        const int first = L.lvl_start[lv], count = L.lvl_start[lv + 1] - first;
        IKC_FOR(ci, count) {
            const int c0 = L.chain_start[first + ci], c1 = L.chain_start[first + ci + 1];
            double oM[12];
            {
                const double *P = ws + oMi + 12 * T.parent[L.order[c0]];
                for (int k = 0; k < 12; ++k) oM[k] = P[k];
            }
            for (int n = c0; n < c1; ++n) {
                const int j = L.order[n];
                double li[12], r[12];
                for (int k = 0; k < 12; ++k) li[k] = ws[L.A0 + 12 * j + k];
                g_se3_mul(oM, li, r);
                for (int k = 0; k < 12; ++k) { ws[oMi + 12 * j + k] = r[k]; oM[k] = r[k]; }
            }
        }
        IKC_SYNC();
    }
    IKC_TICK(1);
    // ... and the world joint Jacobian columns [v; w] (computeJointJacobians), one lane per joint
    {
        IKC_FOR(jj, nj - 1) {
            const int j = jj + 1;
            double oM[12];
            const int iv = T.idx_v[j], jt = T.jtype[j];
            const double *a = T.axis + 3 * j;
            for (int k = 0; k < 12; ++k) oM[k] = ws[oMi + 12 * j + k];
            if (T.has_com) {  // pinocchio::centerOfMass: first moment of the bodies on this joint
                const double *cl = T.j_lever + 3 * j;
                const double mj = T.j_mass[j];
                for (int i = 0; i < 3; ++i)
                    ws[L.sf + 3 * j + i] = mj * dfma(oM[3 * i], cl[0], dfma(oM[3 * i + 1], cl[1], dfma(oM[3 * i + 2], cl[2], oM[9 + i])));
            }
            if (jt == GJ_REVOLUTE || jt == GJ_PRISMATIC || jt == GJ_REVOLUTE_UNBOUNDED) {
                const double Ra[3] = {dfma(oM[0], a[0], dfma(oM[1], a[1], oM[2] * a[2])), dfma(oM[3], a[0], dfma(oM[4], a[1], oM[5] * a[2])),
                                      dfma(oM[6], a[0], dfma(oM[7], a[1], oM[8] * a[2]))};
                const double p[3] = {oM[9], oM[10], oM[11]};
                double v[3] = {Ra[0], Ra[1], Ra[2]}, w[3] = {0, 0, 0};
                if (jt != GJ_PRISMATIC) {
                    cross(p, Ra, v);
                    w[0] = Ra[0]; w[1] = Ra[1]; w[2] = Ra[2];
                }
                for (int r = 0; r < 3; ++r) { ws[L.Jw + r * nv + iv] = v[r]; ws[L.Jw + (3 + r) * nv + iv] = w[r]; }
            } else if (jt == GJ_FREEFLYER) {
                const double p[3] = {oM[9], oM[10], oM[11]};
                for (int c = 0; c < 3; ++c) {
                    const double Rc[3] = {oM[c], oM[3 + c], oM[6 + c]};
                    double pxR[3];
                    cross(p, Rc, pxR);
                    for (int r = 0; r < 3; ++r) {
                        ws[L.Jw + r * nv + iv + c] = Rc[r];
                        ws[L.Jw + (3 + r) * nv + iv + c] = 0.0;
                        ws[L.Jw + r * nv + iv + 3 + c] = pxR[r];
                        ws[L.Jw + (3 + r) * nv + iv + 3 + c] = Rc[r];
                    }
                }
            }
        }
        IKC_SYNC();
    }
    if (T.has_com) {  // ... its backward pass: first moment of every subtree, one lane per component, joints in generic_evaluate's order
        IKC_FOR(i, 3) {
            ws[L.sf + i] = 0.0;
            for (int j = nj - 1; j > 0; --j) ws[L.sf + 3 * T.parent[j] + i] += ws[L.sf + 3 * j + i];
        }
        IKC_SYNC();
    }
    IKC_TICK(13);
    // ---- per task: frame placement, error, the blocks its Jacobian columns need (tb: Rf 9 | pf 3 | A 9 | B 9 | spare 6)
    IKC_FOR(t, nt) {
        const int fj = T.t_fjoint[t], rj = T.t_rjoint[t], type = T.t_type[t], row = T.t_row[t], dim = T.t_dim[t];
        const double *w6 = T.t_w + 6 * t;
        double *tb = ws + L.tb + 36 * (L.tb_index[t] < 0 ? 0 : L.tb_index[t]);
        if (type == GT_POSTURE_ROW) {  // ik/ik/posture.hpp:51-68: the whole row is written here
            ws[L.e + row] = (ws[L.q + rj] - ws[L.tg + L.tg_off[t] + 9]) * w6[1] * w6[0];
            if (!L.post_elim)   // (eliminated rows: coop_dls never reads their Jacobian)
                for (int c = 0; c < nv; ++c) ws[L.J + row * nv + c] = (c == fj) ? w6[0] : 0.0;
            continue;
        }
        if (type == GT_COM) {  // ik::CentreOfMassTask, ik/ik/centre_of_mass.hpp:33-45: tb = placement of the reference frame
            double oMr[12];
            coop_se3_mul_ws(ws + oMi + 12 * rj, T.t_rpl + 12 * t, oMr);
            for (int k = 0; k < 12; ++k) tb[k] = oMr[k];
            const double d[3] = {dfma(ws[L.sf], T.inv_total_mass, -oMr[9]), dfma(ws[L.sf + 1], T.inv_total_mass, -oMr[10]),
                                 dfma(ws[L.sf + 2], T.inv_total_mass, -oMr[11])};
            for (int r = 0; r < 3; ++r)
                ws[L.e + row + r] = (dfma(oMr[r], d[0], dfma(oMr[3 + r], d[1], oMr[6 + r] * d[2])) - ws[L.tg + L.tg_off[t] + 9 + r]) * w6[r];
            continue;
        }
        double oMf[12], oMr[12], tg[12];
        coop_se3_mul_ws(ws + oMi + 12 * fj, T.t_fpl + 12 * t, oMf);
        coop_se3_mul_ws(ws + oMi + 12 * rj, T.t_rpl + 12 * t, oMr);
        { const int to = L.tg + L.tg_off[t]; for (int k = 0; k < 12; ++k) { tg[k] = ws[to + k]; tb[k] = oMf[k]; } }
        const double Rf[9] = {oMf[0], oMf[1], oMf[2], oMf[3], oMf[4], oMf[5], oMf[6], oMf[7], oMf[8]};
        const double pf[3] = {oMf[9], oMf[10], oMf[11]};
        if (type >= GT_ALIGN_X) {  // AlignAxisTask, ik/ik/frame.hpp:257-301: tb[12..14] = the row's direction in the frame
            double rMf[12];
            g_se3_inv_mul(oMr, oMf, rMf);
            const int ax = type - GT_ALIGN_X;
            const double r[3] = {rMf[ax], rMf[3 + ax], rMf[6 + ax]};
            const double inv = drsqrt(dfma(tg[9], tg[9], dfma(tg[10], tg[10], tg[11] * tg[11])));
            const double tn[3] = {tg[9] * inv, tg[10] * inv, tg[11] * inv};
            double rxt[3];
            cross(r, tn, rxt);
            tb[12] = dfma(rxt[0], rMf[0], dfma(rxt[1], rMf[3], rxt[2] * rMf[6]));
            tb[13] = dfma(rxt[0], rMf[1], dfma(rxt[1], rMf[4], rxt[2] * rMf[7]));
            tb[14] = dfma(rxt[0], rMf[2], dfma(rxt[1], rMf[5], rxt[2] * rMf[8]));
            ws[L.e + row] = (1.0 - dot(r, tn)) * w6[0];
            continue;
        }
        double oMt[12], Re[9], pe[3];
        g_se3_mul(oMr, tg, oMt);                        // frame.hpp:48
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) Re[3 * i + j] = dfma(Rf[i], oMt[j], dfma(Rf[3 + i], oMt[3 + j], Rf[6 + i] * oMt[6 + j]));
        {
            const double dp[3] = {oMt[9] - pf[0], oMt[10] - pf[1], oMt[11] - pf[2]};
            rotT_vec(Rf, dp, pe);
        }
        LogAndJlog lj;
        log6_and_jlog6_inv(Re, pe, lj);                 // frame.hpp:50-61, :162-166
        for (int k = 0; k < 9; ++k) { tb[12 + k] = lj.A[k]; tb[21 + k] = lj.Bm[k]; }
        const int r0 = (type == GT_ORIENTATION) ? 3 : 0;
#pragma unroll
        for (int r = 0; r < 6; ++r)
            if (r >= r0 && r < r0 + dim) ws[L.e + row + r - r0] = lj.e[r] * w6[r - r0];
    }
    IKC_SYNC();
    IKC_TICK(2);
    // ---- task Jacobian, one tangent column per lane (zero outside the support of the task's frame, frame.hpp:110)
    IKC_FOR(c, nv) {
        const double vw[6] = {ws[L.Jw + c], ws[L.Jw + nv + c], ws[L.Jw + 2 * nv + c],
                              ws[L.Jw + 3 * nv + c], ws[L.Jw + 4 * nv + c], ws[L.Jw + 5 * nv + c]};
        for (int kb = 0; kb < L.nblocks; ++kb) {   // (the tasks with a block: posture rows have no column work)
            const int t = L.btask[kb];
            const int type = T.t_type[t], row = L.jrow[t], dim = T.t_dim[t];   // (row: of J)
            if (type == GT_COM) {  // jacobianCenterOfMass, ik/ik/data.cpp:31-34: every column, scaled by the mass of its joint's subtree
                const double *w6 = T.t_w + 6 * t;
                const double *tb = ws + L.tb + 36 * L.tb_index[t];
                const int j = L.col_joint[c];
                const double ms = T.j_submass[j];
                const double f[3] = {ws[L.sf + 3 * j], ws[L.sf + 3 * j + 1], ws[L.sf + 3 * j + 2]};
                const double w[3] = {vw[3], vw[4], vw[5]};
                double fxw[3];
                cross(f, w, fxw);
                const double col[3] = {(ms * vw[0] - fxw[0]) * T.inv_total_mass, (ms * vw[1] - fxw[1]) * T.inv_total_mass,
                                       (ms * vw[2] - fxw[2]) * T.inv_total_mass};
                for (int r = 0; r < 3; ++r)
                    ws[L.J + (row + r) * nv + c] = w6[r] * dfma(tb[r], col[0], dfma(tb[3 + r], col[1], tb[6 + r] * col[2]));
                continue;
            }
            if (!L.support[t * nv + c]) {
                for (int r = 0; r < dim; ++r) ws[L.J + (row + r) * nv + c] = 0.0;
                continue;
            }
            const double *w6 = T.t_w + 6 * t;
            const double *tb = ws + L.tb + 36 * L.tb_index[t];
            const double Rf[9] = {tb[0], tb[1], tb[2], tb[3], tb[4], tb[5], tb[6], tb[7], tb[8]};
            const double pf[3] = {tb[9], tb[10], tb[11]};
            double v[3] = {vw[0], vw[1], vw[2]};
            const double w[3] = {vw[3], vw[4], vw[5]};
            double pxw[3], vl[3], wl[3];
            rotT_vec(Rf, w, wl);
            if (type >= GT_ALIGN_X) {
                const double gv[3] = {tb[12], tb[13], tb[14]};
                ws[L.J + row * nv + c] = -w6[0] * dot(gv, wl);
                continue;
            }
            cross(pf, w, pxw);
            v[0] -= pxw[0]; v[1] -= pxw[1]; v[2] -= pxw[2];
            rotT_vec(Rf, v, vl);
            const int r0 = (type == GT_ORIENTATION) ? 3 : 0;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double lin = -dfma(tb[12 + 3 * i], vl[0], dfma(tb[12 + 3 * i + 1], vl[1], dfma(tb[12 + 3 * i + 2], vl[2],
                                    dfma(tb[21 + 3 * i], wl[0], dfma(tb[21 + 3 * i + 1], wl[1], tb[21 + 3 * i + 2] * wl[2])))));
                const double ang = -dfma(tb[12 + 3 * i], wl[0], dfma(tb[12 + 3 * i + 1], wl[1], tb[12 + 3 * i + 2] * wl[2]));
                if (i >= r0 && i < r0 + dim) ws[L.J + (row + i - r0) * nv + c] = w6[i - r0] * lin;
                if (3 + i >= r0 && 3 + i < r0 + dim) ws[L.J + (row + 3 + i - r0) * nv + c] = w6[3 + i - r0] * ang;
            }
        }
    }
    IKC_SYNC();
    IKC_TICK(3);
    double e0sq = 0.0;  // ||e[0]||^2 over the rows of priority level 0, the same in every lane
#pragma unroll 8
    for (int r = 0; r < T.lvl_row0[1]; ++r) { const double e = ws[L.e + r]; e0sq = dfma(e, e, e0sq); }
    return e0sq;
}

// Solve the m x m SPD system whose lower triangle sits at offG (packed by rows) with the right-hand side as row m: right-looking
// Cholesky, ONE phase per pivot -- column k stays unscaled while the trailing triangle takes its update
// G(i,j) -= (G(i,k) inv)(G(j,k) inv); every entry receives its updates in the order m = 0, 1, ... of the left-looking loop of
// generic_solver.hpp, with the same roundings -- one phase that turns the strict lower triangle into L by the pivots'
// reciprocals (row m comes out as y = L^-1 rhs), then the column-oriented back substitution.  x is left at offx.  The pair
// tables list the lower triangle by rows; row m contributes its first m entries.
#if IKD_ON_DEVICE
// The value x holds in lane N of this lane's group (a group is one DPP row of sixteen lanes: v_mov_b64_dpp row_newbcast)
template <int N>
IKD_FN double group_bcast(double x) { return __builtin_amdgcn_update_dpp(x, x, 0x150 + N, 0xf, 0xf, false); }
// acc += (x in lane N of the group) * y, ONE instruction (v_fmac_f64 with a DPP row broadcast on its first source; the broadcast as a
// separate v_mov_b64_dpp costs two more issue slots, profiles/r02_issue_probe_dpp.csv).
// HAZARD (a VALU write needs two wait states before a DPP read of the same VGPR): LLVM's hazard recogniser sees neither the writes
// nor the DPP reads of inline asm, so every such pair that involves an asm statement is fenced IN THE CODE by an s_nop that is
// data-tied to the register in question -- the producer is ordered before the nop, every later reader after it:
//   * a compiler-generated VALU write read by an asm DPP (the pivot column `lik`):  dpp_settle(lik) before the asm block;
//   * an asm fmac's write read by a compiler-generated DPP (group_bcast of row[K] at the next pivot; the right-hand-side broadcasts):
//     dpp_settle(row[K]) / dpp_settle_all(row) before the group_bcast.
template <int N>
IKD_FN void fmac_bcast(double &acc, const double x_lane_n, const double y) {
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x_lane_n), "v"(y), "n"(N));
}
IKD_FN void dpp_settle(double &x) { asm volatile("s_nop 1" : "+v"(x)); }
IKD_FN void dpp_settle(double &x, double &y) { asm volatile("s_nop 1" : "+v"(x), "+v"(y)); }
template <int N>
IKD_FN void dpp_settle_all(double (&r)[N]) {   // one s_nop ordered after the last write of every element
    static_assert(N == 16 || N == 32, "register rows of the cooperative solver");
    asm volatile("s_nop 1" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]),
                 "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]));
    if constexpr (N == 32)
        asm volatile("s_nop 1" : "+v"(r[16]), "+v"(r[17]), "+v"(r[18]), "+v"(r[19]), "+v"(r[20]), "+v"(r[21]), "+v"(r[22]), "+v"(r[23]), "+v"(r[24]),
                     "+v"(r[25]), "+v"(r[26]), "+v"(r[27]), "+v"(r[28]), "+v"(r[29]), "+v"(r[30]), "+v"(r[31]));
}
// Eight columns of two dot products in ONE asm statement: s0 += x_c(lane J0) x_c, s1 += x_c(lane J1) x_c, c = 0 .. 7, the two
// accumulators alternating (a v_fmac_f64_dpp that depends on the one before takes 8 cycles, independent ones 4.8), and nothing of the
// compiler's between them (it puts an s_nop after every single-instruction asm statement whose result the next one reads).
#define IKC_F2(A, B, X, Y, L) "v_fmac_f64_dpp %" #A ", %" #X ", %" #Y " row_newbcast:%" #L " row_mask:0xf bank_mask:0xf\n"
template <int J0, int J1>
IKD_FN void fmac2_bcast8(double &s0, double &s1, const double c0, const double c1, const double c2, const double c3, const double c4,
                         const double c5, const double c6, const double c7) {
    asm volatile(IKC_F2(0, 0, 2, 2, 10) IKC_F2(1, 1, 2, 2, 11) IKC_F2(0, 0, 3, 3, 10) IKC_F2(1, 1, 3, 3, 11) IKC_F2(0, 0, 4, 4, 10)
                 IKC_F2(1, 1, 4, 4, 11) IKC_F2(0, 0, 5, 5, 10) IKC_F2(1, 1, 5, 5, 11) IKC_F2(0, 0, 6, 6, 10) IKC_F2(1, 1, 6, 6, 11)
                 IKC_F2(0, 0, 7, 7, 10) IKC_F2(1, 1, 7, 7, 11) IKC_F2(0, 0, 8, 8, 10) IKC_F2(1, 1, 8, 8, 11) IKC_F2(0, 0, 9, 9, 10)
                 IKC_F2(1, 1, 9, 9, 11)
                 : "+&v"(s0), "+&v"(s1)   // (written before the last inputs are read: never in a register an input lives in)
                 : "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(c4), "v"(c5), "v"(c6), "v"(c7), "n"(J0), "n"(J1));
}
template <int J0>
IKD_FN void fmac1_bcast8(double &s0, const double c0, const double c1, const double c2, const double c3, const double c4, const double c5,
                         const double c6, const double c7) {
    asm volatile(IKC_F2(0, 0, 1, 1, 9) IKC_F2(0, 0, 2, 2, 9) IKC_F2(0, 0, 3, 3, 9) IKC_F2(0, 0, 4, 4, 9) IKC_F2(0, 0, 5, 5, 9) IKC_F2(0, 0, 6, 6, 9)
                 IKC_F2(0, 0, 7, 7, 9) IKC_F2(0, 0, 8, 8, 9)
                 : "+&v"(s0)
                 : "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(c4), "v"(c5), "v"(c6), "v"(c7), "n"(J0));
}
#undef IKC_F2

// The same solve with the matrix in REGISTERS, one row per lane of the group (lane i: row i of the lower triangle; lane 15: the
// right-hand side, riding along as row M), for M <= 15: the entries another lane needs come through DPP broadcasts, so the
// factorisation has no LDS round trip and no barrier between its pivots (they were 30 % of the kernel: ~21 k cycles per
// iteration at M = 10, bound by LDS latency; here ~400 issue slots).  Same operations in the same order on every entry as the
// LDS form below -- L(i,k) = G(i,k) inv_k, G(i,j) -= L(i,k) L(j,k), x_k = (y_k - sum_{m > k, descending} L(m,k) x_m) inv_k -- so
// the two agree bit for bit.  The back substitution runs redundantly in every lane (each broadcast reaches the whole group), so
// x ends up replicated; lane 0 parks it at offx.  Ends on a barrier.
template <int MMAX, int K, int J>
IKD_FN void chol_regs_trail(double (&row)[16], const double lik, const double nlik) {   // G(i,j) -= L(i,k) L(j,k), j = K+1 .. MMAX-1 (rows past M hold zeros)
    if constexpr (J < MMAX) {
        fmac_bcast<J>(row[J], lik, nlik);
        chol_regs_trail<MMAX, K, J + 1>(row, lik, nlik);
    }
}
template <int MMAX, int K>
IKD_FN void chol_regs_pivots(double (&row)[16], const int g, const int M) {
    if constexpr (K < MMAX) {
        if (K < M) {   // (wave-uniform)
            if constexpr (K > 0) dpp_settle(row[K]);   // row[K]'s last writer is the asm fmac of pivot K - 1
            const double inv = drsqrt(group_bcast<K>(row[K]));
            double lik = row[K] * inv;
            const double nlik = -lik;
            row[K] = g == K ? inv : lik;       // lane K keeps 1 / L(K,K) where its diagonal entry was (nobody reads L(K,K) itself)
            dpp_settle(lik);
            chol_regs_trail<MMAX, K, K + 1>(row, lik, nlik);
        }
        chol_regs_pivots<MMAX, K + 1>(row, g, M);
    }
}
template <int Mm, int C>
IKD_FN void chol_regs_sub(const double (&row)[16], double (&x)[16], const double nxm) {   // x_c -= L(m,c) x_m, c = 0 .. m-1
    if constexpr (C < Mm) {
        fmac_bcast<Mm>(x[C], row[C], nxm);
        chol_regs_sub<Mm, C + 1>(row, x, nxm);
    }
}
template <int Mm>
IKD_FN void chol_regs_back(const double (&row)[16], double (&x)[16], const int M) {
    if constexpr (Mm >= 0) {
        if (Mm < M) {
            x[Mm] = x[Mm] * group_bcast<Mm>(row[Mm]);
            chol_regs_sub<Mm, 0>(row, x, -x[Mm]);
        }
        chol_regs_back<Mm - 1>(row, x, M);
    }
}
template <int MMAX, int C>
IKD_FN void chol_regs_rhs(const double (&row)[16], double (&x)[16]) {
    if constexpr (C < MMAX) {
        x[C] = group_bcast<15>(row[C]);
        chol_regs_rhs<MMAX, C + 1>(row, x);
    }
}

// row: this lane's row of the augmented matrix (see above); leaves x at offx.  Ends on a barrier.
template <int MMAX>
IKD_FN void chol_regs_solve(double (&row)[16], const int g, double *ws, const int offx, const int M) {
    double x[16];
    chol_regs_pivots<MMAX, 0>(row, g, M);         // after pivot k, row[k] holds L(i,k) (lane 15: y_k; lane k: 1 / L(k,k))
    dpp_settle_all(row);                          // the broadcasts below read entries the asm fmacs wrote
    chol_regs_rhs<MMAX, 0>(row, x);               // back substitution, in every lane: x[c] starts as y_c ...
    chol_regs_back<MMAX - 1>(row, x, M);          // ... takes -L(m,c) x_m for m = M-1 .. c+1, then inv_c
    if (g == 0) {
#pragma unroll
        for (int k = 0; k < MMAX; ++k)
            if (k < M) ws[offx + k] = x[k];
    }
    IKC_SYNC();
}

template <int MMAX>
IKD_FN void coop_chol_solve_regs(const int g, double *ws, const int offG, const int offx, const int M) {
    const int mine = g == 15 ? M : g;             // the row of the packed triangle this lane holds (lanes M .. 14: none, zeros)
    const bool holds = g < M || g == 15;
    double row[16];
#pragma unroll
    for (int j = 0; j < MMAX; ++j) row[j] = (holds && j <= mine && j < M) ? ws[offG + tri(mine, j)] : 0.0;
    chol_regs_solve<MMAX>(row, g, ws, offx, M);
}

// 16 <= M <= 31: TWO rows per lane -- lane g holds row g (A: entries 0 .. g) and row g + 16 (B: entries 0 .. g + 16); lane 15's B is
// the right-hand side (row M of the packed triangle).  Same operations, same order; an entry of row j comes from lane j mod 16.
template <int MMAX, int K, int J>
IKD_FN void chol_regs2_trail(double (&ra)[16], double (&rb)[32], const double lika, const double likb, const double nlika, const double nlikb, const int M) {
    if constexpr (J < MMAX) {
        // (no guard on J: rows past M hold zeros, and a wave-uniform test per update costs more than the update -- M sits in a
        // spilled SGPR here: v_readlane + s_and + s_cbranch against one v_fmac)
        if constexpr (J < 16) {
            fmac_bcast<J>(ra[J], lika, nlika);
            fmac_bcast<J>(rb[J], lika, nlikb);
        } else {
            fmac_bcast<J - 16>(rb[J], likb, nlikb);
        }
        chol_regs2_trail<MMAX, K, J + 1>(ra, rb, lika, likb, nlika, nlikb, M);
    }
}
template <int MMAX, int K>
IKD_FN void chol_regs2_pivots(double (&ra)[16], double (&rb)[32], const int g, const int M) {
    if constexpr (K < MMAX) {
        if (K < M) {   // (wave-uniform)
            double d;
            if constexpr (K > 0) {   // the pivot entry's last writer is the asm fmac of pivot K - 1
                if constexpr (K < 16) dpp_settle(ra[K]);
                else dpp_settle(rb[K]);
            }
            if constexpr (K < 16) d = group_bcast<K>(ra[K]);
            else d = group_bcast<K - 16>(rb[K]);
            const double inv = drsqrt(d);
            double lika = 0.0;
            if constexpr (K < 16) { lika = ra[K] * inv; ra[K] = g == K ? inv : lika; }
            double likb = rb[K] * inv;
            if constexpr (K < 16) rb[K] = likb;
            else rb[K] = g == K - 16 ? inv : likb;
            dpp_settle(lika, likb);
            chol_regs2_trail<MMAX, K, K + 1>(ra, rb, lika, likb, -lika, -likb, M);
            IKD_SCHED_FENCE();   // (pivot by pivot: the scheduler otherwise pulls later pivots' broadcasts forward and the rows spill)
        }
        chol_regs2_pivots<MMAX, K + 1>(ra, rb, g, M);
    }
}
template <int Mm, int C>
IKD_FN void chol_regs2_sub(const double (&ra)[16], const double (&rb)[32], double (&x)[32], const double nxm) {
    if constexpr (C < Mm) {
        if constexpr (Mm < 16) fmac_bcast<Mm>(x[C], ra[C], nxm);
        else fmac_bcast<Mm - 16>(x[C], rb[C], nxm);
        chol_regs2_sub<Mm, C + 1>(ra, rb, x, nxm);
    }
}
template <int Mm>
IKD_FN void chol_regs2_back(const double (&ra)[16], const double (&rb)[32], double (&x)[32], const int M) {
    if constexpr (Mm >= 0) {
        if (Mm < M) {
            double dinv;
            if constexpr (Mm < 16) dinv = group_bcast<Mm>(ra[Mm]);
            else dinv = group_bcast<Mm - 16>(rb[Mm]);
            x[Mm] = x[Mm] * dinv;
            chol_regs2_sub<Mm, 0>(ra, rb, x, -x[Mm]);
            IKD_SCHED_FENCE();
        }
        chol_regs2_back<Mm - 1>(ra, rb, x, M);
    }
}
template <int MMAX, int C>
IKD_FN void chol_regs2_rhs(const double (&rb)[32], double (&x)[32]) {
    if constexpr (C < MMAX) {
        x[C] = group_bcast<15>(rb[C]);
        chol_regs2_rhs<MMAX, C + 1>(rb, x);
    }
}
// ... and its Gram matrix in registers: lane g holds rows g and g + 16 of J; row j comes from lane j mod 16 (A or B half)
template <int MMAX, int NVMAX, int J, int C>
IKD_FN void gram_regs2_dot(const double (&ja)[NVMAX], const double (&jb)[NVMAX], double &sa, double &sb) {
    if constexpr (C < NVMAX) {
        if constexpr (J < 16) {
            fmac_bcast<J>(sa, ja[C], ja[C]);
            fmac_bcast<J>(sb, ja[C], jb[C]);
        } else {
            fmac_bcast<J - 16>(sb, jb[C], jb[C]);   // (row A of a lane is above row j >= 16: not in the lower triangle)
        }
        gram_regs2_dot<MMAX, NVMAX, J, C + 1>(ja, jb, sa, sb);
    }
}
template <int MMAX, int NVMAX, int J>
IKD_FN void gram_regs2_rows(const double (&ja)[NVMAX], const double (&jb)[NVMAX], double (&ra)[16], double (&rb)[32], const int g,
                            const double lam2, const int M) {
    if constexpr (J < MMAX) {
        if (J >= M) return;   // (wave-uniform)
        double sa = g == J ? lam2 : 0.0, sb = g + 16 == J ? lam2 : 0.0;
        gram_regs2_dot<MMAX, NVMAX, J, 0>(ja, jb, sa, sb);
        if constexpr (J < 16) ra[J] = sa;
        rb[J] = sb;
        gram_regs2_rows<MMAX, NVMAX, J + 1>(ja, jb, ra, rb, g, lam2, M);
    }
}

// offJ >= 0: the matrix is J J^T + lam2 I of the M x nv rows at offJ (nv <= NVMAX), formed in registers, right-hand side at offe;
// offJ < 0: the packed lower triangle (right-hand side as row M) is read from offG.
template <int MMAX, int NVMAX>
IKD_FN void coop_chol_solve_regs2(const int g, double *ws, const int offG, const int offx, const int M, const int offJ, const int offe,
                                  const int nv, const double lam2) {
    const int mineb = g == 15 ? M : g + 16;
    const bool holdsb = g == 15 || g + 16 < M;
    double ra[16], rb[32], x[32];
    if (offJ >= 0) {   // (wave-uniform)
        double ja[NVMAX], jb[NVMAX];
#pragma unroll
        for (int c = 0; c < NVMAX; ++c) {
            ja[c] = c < nv ? ws[offJ + g * nv + c] : 0.0;
            jb[c] = (g + 16 < M && c < nv) ? ws[offJ + (g + 16) * nv + c] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) ra[j] = 0.0;
#pragma unroll
        for (int j = 0; j < 32; ++j) rb[j] = 0.0;
        gram_regs2_rows<MMAX, NVMAX, 0>(ja, jb, ra, rb, g, lam2, M);
        if (g == 15) {
#pragma unroll
            for (int j = 0; j < MMAX; ++j) rb[j] = j < M ? ws[offe + j] : 0.0;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) ra[j] = j <= g ? ws[offG + tri(g, j)] : 0.0;
#pragma unroll
        for (int j = 0; j < MMAX; ++j) rb[j] = (holdsb && j <= mineb && j < M) ? ws[offG + tri(mineb, j)] : 0.0;
    }
    chol_regs2_pivots<MMAX, 0>(ra, rb, g, M);
    dpp_settle_all(ra);                           // the broadcasts below read entries the asm fmacs wrote
    dpp_settle_all(rb);
    chol_regs2_rhs<MMAX, 0>(rb, x);
    chol_regs2_back<MMAX - 1>(ra, rb, x, M);
    if (g == 0) {
#pragma unroll
        for (int k = 0; k < MMAX; ++k)
            if (k < M) ws[offx + k] = x[k];
    }
    IKC_SYNC();
}

// The Gram matrix straight into that layout: lane i loads row i of the M x nv matrix at offJ into registers once, and
// G(i,j) = [i == j] lam2 + sum_c J(i,c) J(j,c) (c ascending, as the LDS form) takes row j from lane j by DPP broadcast -- no LDS
// traffic in the inner loop, where the pair-per-lane form reads two operands per product (the phase was bound by the CU's LDS
// bandwidth: 18 % of the kernel).  Lane 15 holds the right-hand side (offe).  Then the solve above.
template <int NVMAX, int J0, int J1, int C>
IKD_FN void gram_regs_dot2(const double (&jr)[NVMAX], double &s0, double &s1) {   // rows J0 and J1 (J1 < 0: row J0 alone), eight columns at a time
    if constexpr (C < NVMAX) {
        if constexpr (J1 >= 0) fmac2_bcast8<J0, J1>(s0, s1, jr[C], jr[C + 1], jr[C + 2], jr[C + 3], jr[C + 4], jr[C + 5], jr[C + 6], jr[C + 7]);
        else fmac1_bcast8<J0>(s0, jr[C], jr[C + 1], jr[C + 2], jr[C + 3], jr[C + 4], jr[C + 5], jr[C + 6], jr[C + 7]);
        gram_regs_dot2<NVMAX, J0, J1, C + 8>(jr, s0, s1);
    }
}
template <int MMAX, int NVMAX, int J>
IKD_FN void gram_regs_rows(const double (&jr)[NVMAX], double (&row)[16], const int g, const double lam2, const int M) {
    static_assert(NVMAX % 8 == 0, "eight columns per asm block");
    if constexpr (J < MMAX) {
        double s0 = g == J ? lam2 : 0.0, s1 = g == J + 1 ? lam2 : 0.0;
        if (J + 1 < M) {          // (wave-uniform) two rows at a time: their accumulators alternate
            gram_regs_dot2<NVMAX, J, J + 1, 0>(jr, s0, s1);
        } else if (J < M) {
            gram_regs_dot2<NVMAX, J, -1, 0>(jr, s0, s1);
            s1 = 0.0;
        } else {
            s0 = 0.0; s1 = 0.0;
        }
        row[J] = s0;
        if constexpr (J + 1 < 16) row[J + 1] = s1;
        gram_regs_rows<MMAX, NVMAX, J + 2>(jr, row, g, lam2, M);
    }
}
template <int MMAX, int NVMAX>
IKD_FN void coop_gram_solve_regs(const int g, double *ws, const int offJ, const int offe, const int offx, const int M, const int nv,
                                 const double lam2) {
    double jr[NVMAX], row[16];
#pragma unroll
    for (int c = 0; c < NVMAX; ++c) jr[c] = (g < M && c < nv) ? ws[offJ + g * nv + c] : 0.0;
    gram_regs_rows<MMAX, NVMAX, 0>(jr, row, g, lam2, M);
    if (g == 15) {
#pragma unroll
        for (int j = 0; j < MMAX; ++j) row[j] = j < M ? ws[offe + j] : 0.0;
    }
    chol_regs_solve<MMAX>(row, g, ws, offx, M);
}
#endif

// BIG: the kernel build for 16 <= M <= 31 (two rows per lane: 160 registers for the matrix alone, so it is kept out of the builds that
// run two waves per SIMD under a 256-register cap -- problems of that size leave fewer than five workgroups per CU anyway)
template <bool BIG = false>
IKD_FN void coop_chol_solve(const CoopLayout &L, const int g, double *ws, const int offG, const int offdinv, const int offx, const int M,
                            const int offJ = -1, const int offe = 0, const int nv = 0, const double lam2 = 0.0) {
    (void)g; (void)offJ; (void)offe; (void)nv; (void)lam2;
#if IKD_ON_DEVICE
    if constexpr (BIG) {   // (the launch picks this build for 16 <= M <= 31 only; ONE instantiation: see coop_dls)
        if (M <= 31) { coop_chol_solve_regs2<31, 24>(g, ws, offG, offx, M, offJ, offe, nv, lam2); return; }   // (wave-uniform)
    } else {
        if (M <= 10) { coop_chol_solve_regs<10>(g, ws, offG, offx, M); return; }
        if (M <= 15) { coop_chol_solve_regs<15>(g, ws, offG, offx, M); return; }
    }
#endif
    const int npairs = tri(M, 0) + M;
    // ---- Cholesky, right-looking, ONE phase per pivot: column k stays unscaled while the trailing triangle takes its
    // update G(i,j) -= (G(i,k) inv)(G(j,k) inv) -- every entry receives its updates in the order m = 0, 1, ... of the
    // left-looking loop of generic_solver.hpp, with the same roundings -- and one last phase turns the strict lower triangle
    // into L by the pivots' reciprocals.  Row M comes out as y = L^-1 et.
    for (int k = 0; k < M; ++k) {
        const double inv = drsqrt(ws[offG + tri(k, k)]);
        ws[offdinv + k] = inv;  // every lane holds the same value
        const int p0 = tri(k + 1, 0);  // the pairs are sorted by row: rows k + 1 .. M start here
#pragma unroll 2
        IKC_FOR(pp, npairs - p0) {
            const int i = L.pair_i[p0 + pp], j = L.pair_j[p0 + pp];
            if (j > k) {
                const double lik = ws[offG + tri(i, k)] * inv, ljk = ws[offG + tri(j, k)] * inv;
                ws[offG + tri(i, j)] = dfma(-lik, ljk, ws[offG + tri(i, j)]);
            }
        }
        IKC_SYNC();
    }
#pragma unroll 2
    IKC_FOR(p, npairs) {
        const int i = L.pair_i[p], j = L.pair_j[p];
        if (j < i) ws[offG + tri(i, j)] *= ws[offdinv + j];
    }
    IKC_SYNC();
    // ---- back substitution x = L^-T y, column oriented: x_k is final once the rows above it were eliminated
    // (one lane running all M columns back to back, without the barriers, was measured SLOWER: 23.1 vs 18.6 ms on the demo task
    // set -- its read-modify-writes of row M serialise on LDS latency, where a phase spreads a column over the lanes)
    for (int k = M - 1; k >= 0; --k) {
        const double xk = ws[offG + tri(M, k)] * ws[offdinv + k];
        IKC_FOR(i, k) ws[offG + tri(M, i)] = dfma(-ws[offG + tri(k, i)], xk, ws[offG + tri(M, i)]);
        ws[offx + k] = xk;
        IKC_SYNC();
    }
}

// Orthonormal basis of the row space of the m x nv matrix at offRows, in place (v_1 .. v_rank end up in its first rows):
// Gram-Schmidt with row pivoting on the largest remaining norm, every projection applied twice.  The pivot norms are the
// |R_kk| of Eigen's column-pivoted QR of the transpose -- what its complete orthogonal decomposition starts from -- and its
// rank rule is applied to them: |R_kk| > eps min(m, nv) max|R| (reference: Jbar.completeOrthogonalDecomposition().pseudoInverse()
// * Jbar, ik/ik/pik.cpp:59-61; Jc likewise, ik/ik/dls.cpp:45-48).  offNrm: m scratch words.  All trip counts are the same in
// every group of the wave (the barriers sit outside anything that depends on the data); a group whose rank is exhausted keeps
// walking with `live` false.  Ends on a barrier.
IKD_FN int coop_rowspace_basis(const int g, double *ws, const int offRows, const int offNrm, const int m, const int nv) {
    (void)g;
    const double kk = 2.220446049250313e-16 * static_cast<double>(m < nv ? m : nv), thr2 = kk * kk;
    int rank = 0;
    bool live = true;
    double maxpiv2 = 0.0;
    for (int k = 0; k < m; ++k) {
        IKC_FOR(ii, m - k) {
            const int row = k + ii;
            double n2 = 0.0;
#pragma unroll 8
            for (int c = 0; c < nv; ++c) { const double x = ws[offRows + row * nv + c]; n2 = dfma(x, x, n2); }
            ws[offNrm + row] = n2;
        }
        IKC_SYNC();
        int piv = k;
        double best = ws[offNrm + k];
#pragma unroll 8
        for (int i = k + 1; i < m; ++i) {
            const double n2 = ws[offNrm + i];
            if (n2 > best) { best = n2; piv = i; }
        }
        if (k == 0) maxpiv2 = best;
        live = live && best > thr2 * maxpiv2 && best > 0.0;   // Eigen's rank rule on |R_kk| = sqrt(best)
        const double inv = live ? 1.0 / __builtin_sqrt(best) : 0.0;
        IKC_FOR(c, nv) {
            if (live) {
                const double a = ws[offRows + piv * nv + c], b = ws[offRows + k * nv + c];
                ws[offRows + piv * nv + c] = b;
                ws[offRows + k * nv + c] = a * inv;
            }
        }
        IKC_SYNC();
        IKC_FOR(ii, m - k - 1) {
            if (live) {
                const int row = k + 1 + ii;
                for (int pass = 0; pass < 2; ++pass) {
                    double d = 0.0;
#pragma unroll 8
                    for (int c = 0; c < nv; ++c) d = dfma(ws[offRows + row * nv + c], ws[offRows + k * nv + c], d);
#pragma unroll 8
                    for (int c = 0; c < nv; ++c) ws[offRows + row * nv + c] = dfma(-d, ws[offRows + k * nv + c], ws[offRows + row * nv + c]);
                }
            }
        }
        IKC_SYNC();
        if (live) ++rank;
    }
    return rank;
}

// Orthonormal basis of the row space of a FULL-ROW-RANK, well-conditioned m x nv matrix by Cholesky QR: its Gram matrix
// A A^T = L L^T (lower triangle packed by rows at offG0, destroyed), then V = L^-1 A in place, one lane per column with no
// barrier inside -- m + 2 phases against the ~3 m of the pivoted Gram-Schmidt (coop_rowspace_basis), whose projections also run
// on only m - k of the sixteen lanes.  Any orthonormal basis gives the same projector I - V^T V.  Orthogonality is lost like
// eps cond(A)^2, and the Gram matrix cannot resolve the reference's rank rule (|R_kk| against eps max|R|), so the shortcut is
// taken only when every pivot L_kk^2 stays above 1e-4 of the largest diagonal entry (cond^2 < ~1e5: error < 1e-10) in ALL
// groups of the wave; otherwise it returns false with A untouched and the caller runs the rank-revealing routine.
// pair_i / pair_j: the (i, j) of the lower triangle of an (at least) m-row matrix listed by rows.
#if IKD_ON_DEVICE
// The factorisation of coop_rowspace_basis_cholqr in registers (one row per lane, see coop_chol_solve_regs): same pivots, same
// test on them; L (strict lower triangle, scaled) and the pivots' reciprocals are written back for the forward substitution.
template <int MMAX, int K>
IKD_FN void cholqr_regs_pivots(double (&row)[16], const int g, const int m, const double thr, bool &ok) {
    if constexpr (K < MMAX) {
        if (K < m) {   // (wave-uniform)
            if constexpr (K > 0) dpp_settle(row[K]);
            const double d = group_bcast<K>(row[K]);
            ok = ok && d > thr;                    // false for NaN as well
            const double inv = drsqrt(d);
            double lik = row[K] * inv;
            const double nlik = -lik;
            row[K] = g == K ? inv : lik;
            dpp_settle(lik);
            chol_regs_trail<MMAX, K, K + 1>(row, lik, nlik);
        }
        cholqr_regs_pivots<MMAX, K + 1>(row, g, m, thr, ok);
    }
}
template <int MMAX, int K>
IKD_FN double cholqr_regs_maxdiag(const double (&row)[16], const int m, double acc) {
    if constexpr (K < MMAX) {
        if (K < m) acc = dmax(acc, group_bcast<K>(row[K]));
        return cholqr_regs_maxdiag<MMAX, K + 1>(row, m, acc);
    } else {
        return acc;
    }
}
template <int MMAX>
IKD_FN bool cholqr_regs_factor(const int g, double *ws, const int offG0, const int offdinv, const int m) {
    double row[16];
#pragma unroll
    for (int j = 0; j < MMAX; ++j) row[j] = (g < m && j <= g) ? ws[offG0 + tri(g, j)] : 0.0;
    const double maxdiag = cholqr_regs_maxdiag<MMAX, 0>(row, m, 0.0);
    bool ok = maxdiag > 0.0;
    cholqr_regs_pivots<MMAX, 0>(row, g, m, 1e-4 * maxdiag, ok);
    if (g < m) {
#pragma unroll
        for (int j = 0; j < MMAX; ++j) {
            if (j < g) ws[offG0 + tri(g, j)] = row[j];
            if (j == g) ws[offdinv + g] = row[j];
        }
    }
    IKC_SYNC();
    return ok;
}
#endif

template <class AnyFn>
IKD_FN bool coop_rowspace_basis_cholqr(const int *pair_i, const int *pair_j, const int g, double *ws, const int offRows, const int offG0,
                                       const int offdinv, const int m, const int nv, AnyFn any_fn) {
    (void)g;
    const int npairs = tri(m, 0);
#if IKD_ON_DEVICE
    if (m <= 16) {   // (wave-uniform)
        const bool okr = m <= 10 ? cholqr_regs_factor<10>(g, ws, offG0, offdinv, m) : cholqr_regs_factor<16>(g, ws, offG0, offdinv, m);
        if (any_fn(!okr)) return false;
        IKC_FOR(c, nv) {                        // forward substitution, as below
            for (int k = 0; k < m; ++k) {
                double v = ws[offRows + k * nv + c];
                for (int j = 0; j < k; ++j) v = dfma(-ws[offG0 + tri(k, j)], ws[offRows + j * nv + c], v);
                ws[offRows + k * nv + c] = v * ws[offdinv + k];
            }
        }
        IKC_SYNC();
        return true;
    }
#endif
    double maxdiag = 0.0;
    for (int k = 0; k < m; ++k) maxdiag = dmax(maxdiag, ws[offG0 + tri(k, k)]);
    bool ok = maxdiag > 0.0;
    for (int k = 0; k < m; ++k) {
        const double d = ws[offG0 + tri(k, k)];
        ok = ok && d > 1e-4 * maxdiag;          // false for NaN as well
        const double inv = drsqrt(d);
        ws[offdinv + k] = inv;
        const int p0 = tri(k + 1, 0);
        IKC_FOR(pp, npairs - p0) {
            const int i = pair_i[p0 + pp], j = pair_j[p0 + pp];
            if (j > k) {
                const double lik = ws[offG0 + tri(i, k)] * inv, ljk = ws[offG0 + tri(j, k)] * inv;
                ws[offG0 + tri(i, j)] = dfma(-lik, ljk, ws[offG0 + tri(i, j)]);
            }
        }
        IKC_SYNC();
    }
    if (any_fn(!ok)) return false;              // wave-uniform: every group of the workgroup takes the same path
    IKC_FOR(p, npairs) {
        const int i = pair_i[p], j = pair_j[p];
        if (j < i) ws[offG0 + tri(i, j)] *= ws[offdinv + j];
    }
    IKC_SYNC();
    IKC_FOR(c, nv) {                            // forward substitution, column c: V(k, c) = (A(k, c) - sum_{j<k} L(k, j) V(j, c)) / L(k, k)
        for (int k = 0; k < m; ++k) {
            double v = ws[offRows + k * nv + c];
            for (int j = 0; j < k; ++j) v = dfma(-ws[offG0 + tri(k, j)], ws[offRows + j * nv + c], v);
            ws[offRows + k * nv + c] = v * ws[offdinv + k];
        }
    }
    IKC_SYNC();
    return true;
}

// q <- clip(integrate(q, step * dq)) (ik/ik/dls.cpp:67-71), one lane per joint; a group that is no longer active keeps its q.
// Ends on a barrier.
IKD_FN void coop_integrate(const GenericTables &T, const CoopLayout &L, const int g, double *ws, const double step_length, const bool active) {
    (void)g;
    const int nj = T.njoints;
    // ---- integrate + clip (ik/ik/dls.cpp:67-71), one lane per joint
    IKC_FOR(j, nj) {
        if (j == 0) continue;
        const int iq = T.idx_q[j], iv = T.idx_v[j];
        if (T.jtype[j] == GJ_FREEFLYER) {
            double qb[7], v[6], qn[7], R1[9];
            for (int k = 0; k < 7; ++k) qb[k] = ws[L.q + iq + k];
            for (int k = 0; k < 6; ++k) v[k] = step_length * ws[L.dq + iv + k];
            quat_to_R(qb, R1);
            freeflyer_integrate(qb, R1, v, qn);
            for (int k = 0; k < 7; ++k) {
                const double c = dmin(T.upper[iq + k], dmax(qn[k], T.lower[iq + k]));
                ws[L.q + iq + k] = active ? c : qb[k];
            }
        } else if (T.jtype[j] == GJ_REVOLUTE_UNBOUNDED) {
            const double c0 = ws[L.q + iq], s0 = ws[L.q + iq + 1];
            double c1, s1;
            unbounded_integrate(c0, s0, step_length * ws[L.dq + iv], c1, s1);
            c1 = dmin(T.upper[iq], dmax(c1, T.lower[iq]));
            s1 = dmin(T.upper[iq + 1], dmax(s1, T.lower[iq + 1]));
            ws[L.q + iq] = active ? c1 : c0;
            ws[L.q + iq + 1] = active ? s1 : s0;
        } else {
            const double qo = ws[L.q + iq];
            const double c = dmin(T.upper[iq], dmax(dfma(step_length, ws[L.dq + iv], qo), T.lower[iq]));
            ws[L.q + iq] = active ? c : qo;
        }
    }
    IKC_SYNC();
}

template <bool BIG = false, class AnyFn>
IKD_FN void coop_dls(const GenericTables &T, const CoopLayout &L, const LoopParams &prm, const int g, double *ws, int &iters_out,
                     bool &success_out, AnyFn any_active) {
    (void)g;
    const int nv = T.nv, M = T.M;
    bool active = true, success = false;
    int iters = prm.max_iterations;
    IKC_TICK_INIT;
    // PostureTask rows (ik/ik/posture.hpp:51-68) have ONE non-zero entry each, so they are taken out of the linear system: with J_f the
    // other Mf rows and D = lambda^2 I + sum of the posture rows' w^2 on their columns (diagonal),
    //   dq = -J^T (J J^T + lambda^2 I)^-1 e = -(J_f^T J_f + D)^-1 J^T e = -(u - D^-1 J_f^T z),   u = D^-1 J^T e,  (I + J_f D^-1 J_f^T) z = J_f u
    // -- an Mf x Mf system instead of M x M (the demo with its posture regulariser: 10 instead of 26; two feet + pelvis + sixteen
    // posture rows: 12 instead of 28, which is the difference between the register Cholesky and the LDS one, 199 ms per launch).
    const bool elim = L.post_elim != 0;
    if (elim) {
        IKC_FOR(c, nv) {
            double d = prm.lam2;
            for (int k = L.pstart[c]; k < L.pstart[c + 1]; ++k) { const double w = T.t_w[6 * L.ptask[k]]; d = dfma(w, w, d); }
            ws[L.Dd + c] = 1.0 / d;
        }
        IKC_SYNC();
    }
    for (int it = 0; it < prm.max_iterations; ++it) {
        const double e0sq = coop_evaluate(T, L, g, ws IKC_TICK_PASS);
        bool solved = false;   // (device, M <= 15: the register Gram form solves as well)
        bool gram_in_regs = false;
        if (elim) {
            const int Mf = L.Mf;
            IKC_FOR(c, nv) {   // u = D^-1 J^T e (parked where dq goes)
                double s = 0.0;
#pragma unroll 4
                for (int k = 0; k < Mf; ++k) s = dfma(ws[L.J + k * nv + c], ws[L.e + L.frow[k]], s);
                for (int k = L.pstart[c]; k < L.pstart[c + 1]; ++k) { const int t = L.ptask[k]; s = dfma(T.t_w[6 * t], ws[L.e + T.t_row[t]], s); }
                ws[L.dq + c] = s * ws[L.Dd + c];
            }
            IKC_SYNC();
            IKC_FOR(p, tri(Mf, 0) + Mf) {   // I + J_f D^-1 J_f^T, lower triangle, with J_f u as row Mf
                const int i = L.pair_i[p], j = L.pair_j[p];
                double s;
                if (i == Mf) {
                    s = 0.0;
#pragma unroll 8
                    for (int c = 0; c < nv; ++c) s = dfma(ws[L.J + j * nv + c], ws[L.dq + c], s);
                } else {
                    s = (i == j) ? 1.0 : 0.0;
#pragma unroll 8
                    for (int c = 0; c < nv; ++c) s = dfma(ws[L.J + i * nv + c] * ws[L.Dd + c], ws[L.J + j * nv + c], s);
                }
                ws[L.G + tri(i, j)] = s;
            }
            IKC_SYNC();
        } else {
        // ---- JJ = Jt Jt^T + damping^2 I (ik/ik/dls.cpp:39-41), lower triangle, with the right-hand side et as row M
#if IKD_ON_DEVICE
        // (wave-uniform choices; the register form covers M <= 15, nv <= 40)
#define IKC_GRAM_REGS(MM, NN) if (!BIG && M <= MM && nv <= NN) { coop_gram_solve_regs<MM, NN>(g, ws, L.J, L.e, L.x, M, nv, prm.lam2); solved = true; } else
        IKC_GRAM_REGS(10, 8) IKC_GRAM_REGS(10, 24) IKC_GRAM_REGS(15, 24) IKC_GRAM_REGS(10, 32) IKC_GRAM_REGS(15, 32) IKC_GRAM_REGS(15, 40)
#undef IKC_GRAM_REGS
#endif
        if (BIG && nv <= 24) {
            gram_in_regs = true;   // (the two-row register form builds J J^T itself)
        } else {
        IKC_FOR(p, L.npairs) {
            const int i = L.pair_i[p], j = L.pair_j[p];
            double s;
            if (i == M) {
                s = ws[L.e + j];
            } else {
                s = (i == j) ? prm.lam2 : 0.0;
#pragma unroll 8
                for (int c = 0; c < nv; ++c) s = dfma(ws[L.J + i * nv + c], ws[L.J + j * nv + c], s);
            }
            ws[L.G + tri(i, j)] = s;
        }
        IKC_SYNC();
        }
        }
        IKC_TICK(4);
        // (ONE call site for the solve: two inlined copies of the two-row register form in sibling branches send hipcc's register
        // allocation past 512 registers)
        if (!solved) coop_chol_solve<BIG>(L, g, ws, L.G, L.dinv, L.x, elim ? L.Mf : M, gram_in_regs ? L.J : -1, L.e, nv, prm.lam2);
        IKC_TICK(6);
        if (elim) {
            const int Mf = L.Mf;
            IKC_FOR(c, nv) {   // dq = -(u - D^-1 J_f^T z)
                double s = 0.0;
#pragma unroll 4
                for (int k = 0; k < Mf; ++k) s = dfma(ws[L.J + k * nv + c], ws[L.x + k], s);
                ws[L.dq + c] = dfma(ws[L.Dd + c], s, -ws[L.dq + c]);
            }
        } else {
            IKC_FOR(c, nv) {  // dq = -Jt^T x, ik/ik/dls.cpp:52-53 (N = I)
                double s = 0.0;
#pragma unroll 8
                for (int r = 0; r < M; ++r) s = dfma(ws[L.J + r * nv + c], ws[L.x + r], s);
                ws[L.dq + c] = -s;
            }
        }
        IKC_SYNC();
        IKC_TICK(7);
        if (T.Mc > 0) {  // dq <- N dq, N = I - pinv(Jc) Jc: the step stays in the null space of the constraints (dls.cpp:26-34,43-53)
            IKC_FOR(k, T.ncons) {  // placements of the constrained frame, its reference frame, and the one in the other
                double oMf[12], oMr[12], fMr[12];
                coop_se3_mul_ws(ws + L.A1 + 12 * T.c_fjoint[k], T.c_fpl + 12 * k, oMf);
                coop_se3_mul_ws(ws + L.A1 + 12 * T.c_rjoint[k], T.c_rpl + 12 * k, oMr);
                g_se3_inv_mul(oMf, oMr, fMr);
                for (int i = 0; i < 12; ++i) { ws[L.cb + 36 * k + i] = oMf[i]; ws[L.cb + 36 * k + 12 + i] = oMr[i]; ws[L.cb + 36 * k + 24 + i] = fMr[i]; }
            }
            IKC_SYNC();
            IKC_FOR(c, nv) {  // Jc = J_frame(LOCAL) - Ad(fMr) J_reference(LOCAL) (ik/ik/frame.hpp:413-449), one tangent column per lane
                const double vw[6] = {ws[L.Jw + c], ws[L.Jw + nv + c], ws[L.Jw + 2 * nv + c],
                                      ws[L.Jw + 3 * nv + c], ws[L.Jw + 4 * nv + c], ws[L.Jw + 5 * nv + c]};
                const double w[3] = {vw[3], vw[4], vw[5]};
                for (int k = 0; k < T.ncons; ++k) {
                    const int row = T.c_row[k], dim = T.c_dim[k], r0 = (T.c_type[k] == GT_ORIENTATION) ? 3 : 0;
                    const double *cb = ws + L.cb + 36 * k;
                    double out[6] = {0, 0, 0, 0, 0, 0};
                    if (L.csupp_f[k * nv + c]) {
                        const double Rf[9] = {cb[0], cb[1], cb[2], cb[3], cb[4], cb[5], cb[6], cb[7], cb[8]};
                        const double pf[3] = {cb[9], cb[10], cb[11]};
                        double v[3] = {vw[0], vw[1], vw[2]}, pxw[3], vl[3], wl[3];
                        cross(pf, w, pxw);
                        v[0] -= pxw[0]; v[1] -= pxw[1]; v[2] -= pxw[2];
                        rotT_vec(Rf, v, vl);
                        rotT_vec(Rf, w, wl);
                        out[0] = vl[0]; out[1] = vl[1]; out[2] = vl[2]; out[3] = wl[0]; out[4] = wl[1]; out[5] = wl[2];
                    }
                    if (L.csupp_r[k * nv + c]) {
                        const double Rr[9] = {cb[12], cb[13], cb[14], cb[15], cb[16], cb[17], cb[18], cb[19], cb[20]};
                        const double pr[3] = {cb[21], cb[22], cb[23]};
                        const double px[3] = {cb[33], cb[34], cb[35]};
                        double v[3] = {vw[0], vw[1], vw[2]}, pxw[3], vr[3], wr[3], Rv[3], Rw[3], pxRw[3];
                        cross(pr, w, pxw);
                        v[0] -= pxw[0]; v[1] -= pxw[1]; v[2] -= pxw[2];
                        rotT_vec(Rr, v, vr);
                        rotT_vec(Rr, w, wr);
                        for (int i = 0; i < 3; ++i) {   // Ad(fMr) [v; w] = [R v + p x (R w); R w]
                            Rv[i] = dfma(cb[24 + 3 * i], vr[0], dfma(cb[24 + 3 * i + 1], vr[1], cb[24 + 3 * i + 2] * vr[2]));
                            Rw[i] = dfma(cb[24 + 3 * i], wr[0], dfma(cb[24 + 3 * i + 1], wr[1], cb[24 + 3 * i + 2] * wr[2]));
                        }
                        cross(px, Rw, pxRw);
                        out[0] -= Rv[0] + pxRw[0]; out[1] -= Rv[1] + pxRw[1]; out[2] -= Rv[2] + pxRw[2];
                        out[3] -= Rw[0]; out[4] -= Rw[1]; out[5] -= Rw[2];
                    }
#pragma unroll
                    for (int r = 0; r < 6; ++r)
                        if (r >= r0 && r < r0 + dim) ws[L.Jc + (row + r - r0) * nv + c] = out[r];
                }
            }
            IKC_SYNC();
            bool by_cholqr = false;   // full-rank, well-conditioned Jc (the usual case): Cholesky QR, see coop_rowspace_basis_cholqr
            if (L.cholqr_c) {
                IKC_FOR(p, tri(T.Mc, 0)) {   // Jc Jc^T where the per-constraint placements lived (dead now)
                    const int i = L.cpair_i[p], j = L.cpair_j[p];
                    double s = 0.0;
#pragma unroll 8
                    for (int c = 0; c < nv; ++c) s = dfma(ws[L.Jc + i * nv + c], ws[L.Jc + j * nv + c], s);
                    ws[L.G + tri(i, j)] = s;
                }
                IKC_SYNC();
                by_cholqr = coop_rowspace_basis_cholqr(L.cpair_i, L.cpair_j, g, ws, L.Jc, L.G, L.G + tri(T.Mc, 0), T.Mc, nv, any_active);
            }
            const int rank = by_cholqr ? T.Mc : coop_rowspace_basis(g, ws, L.Jc, L.cnrm, T.Mc, nv);   // v_1 .. v_rank in the first rows of Jc
            IKC_FOR(k, T.Mc) {
                double d = 0.0;
                if (k < rank)
                    for (int c = 0; c < nv; ++c) d = dfma(ws[L.Jc + k * nv + c], ws[L.dq + c], d);
                ws[L.cnrm + k] = d;
            }
            IKC_SYNC();
            IKC_FOR(c, nv) {   // the v_k are orthonormal: dq -= sum_k v_k (v_k . dq)
                double s = ws[L.dq + c];
                for (int k = 0; k < T.Mc; ++k) s = dfma(-ws[L.cnrm + k], ws[L.Jc + k * nv + c], s);
                ws[L.dq + c] = s;
            }
            IKC_SYNC();
        }
        const bool stop_now = active && (prm.stop_sq_tol >= 0.0) && (e0sq < prm.stop_sq_tol);
        if (stop_now) { success = true; iters = it; }
        active = active && !stop_now;
        coop_integrate(T, L, g, ws, prm.step_length, active);
        IKC_TICK(8);
        if (!any_active(active)) break;
    }
    iters_out = iters;
    success_out = success;
}

struct CoopKernelArgs {
    GenericTables T;
    CoopLayout L;
    LoopParams prm;
    int layout;
    int64_t B;
    const double *q0, *targets;
    double *q_out;
    uint8_t *success;
    int32_t *iters;
};

// One problem (index `problem`) on the group's workspace `ws`; g = lane within the group.
template <bool BIG = false, class AnyFn>
IKD_FN void dls_coop_body(const CoopKernelArgs &a, int64_t problem, const int g, double *ws, AnyFn any_active) {
    (void)g;
    const bool valid = problem < a.B;
    const int64_t b = valid ? problem : a.B - 1;
    const int nq = a.T.nq, nslots = a.T.ntasks * 12;
    IKC_FOR(i, nq) ws[a.L.q + i] = a.q0[at(a.layout, a.B, nq, i, b)];
    IKC_FOR(i, a.L.ntg) {
        const int slot = a.L.tg_src[i];
        ws[a.L.tg + i] = a.layout == LAYOUT_SOA ? a.targets[static_cast<int64_t>(slot) * a.B + b] : a.targets[b * nslots + slot];
    }
    IKC_SYNC();
    int iters;
    bool success;
    coop_dls<BIG>(a.T, a.L, a.prm, g, ws, iters, success, any_active);
    if (valid) {   // (no early return: the workgroup goes on to its next group of problems, see dls_coop_kernel)
        IKC_FOR(i, nq) a.q_out[at(a.layout, a.B, nq, i, b)] = ws[a.L.q + i];
        IKC_FOR(one, 1) {
            if (a.success) a.success[b] = success ? 1 : 0;
            if (a.iters) a.iters[b] = iters;
        }
    }
    IKC_SYNC();    // the workspace is free for the next problem
}

}  // namespace ikdev
