// chain_solver.hpp -- the whole ik::dls() loop for ONE problem whose task support is a serial
// chain of NJ revolute joints hanging off a fixed base (shapes "S" and "U" of SURVEY.md section 8:
// Cassie single leg NJ = 7, UR5 NJ = 6), executed by one wavefront lane entirely in registers.
//
// Reference path restated per iteration (file:line relative to the reference root):
//   ik/ik/data.cpp:28-30    FK + joint Jacobians            -> chain FK, axis / origin per joint
//   ik/ik/frame.hpp:37-62   e = log6(oMf^-1 oMr target)     -> log6_and_jlog6_inv
//   ik/ik/frame.hpp:152-182 J = -Jlog6(tMf) J_local         -> per-column transform
//   ik/ik/data.cpp:49-50    task weighting                  -> folded into the 6x6 blocks
//   ik/ik/dls.cpp:39-41     JJ = Jt Jt^T + damping^2 I      -> symmetric rank-1 accumulation
//   ik/ik/dls.cpp:52-53     dq = -Jt^T JJ^-1 et (N = I)     -> per-lane Cholesky
//   ik/ik/dls.cpp:61-64     stop test before the step       -> lane freezes, wave goes on
//   ik/ik/dls.cpp:67-71     integrate + joint clipping      -> q + step*dq, clamp
//
// Every joint is "revolute about local z": an arbitrary unit axis a is folded into the constant
// placements on the host (pl' = P_prev^T pl P, P e_z = a), which leaves oMi * a and the joint
// origin -- all the Jacobian needs -- unchanged.
#pragma once
#if !defined(__HIPCC_RTC__)
#include <type_traits>
#endif

#include "lane_math.hpp"

namespace ikdev {

enum : int { KT_POSITION = 0, KT_ORIENTATION = 1, KT_FULL = 2 };  // ik::KinematicType order

template <int NJ>
struct ChainDesc {
    double pl[NJ][12];    // pl[0]: world -> joint-0 frame (all fixed transforms folded); pl[j]: joint j-1 -> j
    double frame_pl[12];  // last joint frame -> task frame
    double lo[NJ], hi[NJ];
    double wgt[6];        // Task::weighting(), first `dimension` entries
};  // all doubles: the kernels stage it from HBM into LDS as a flat table

struct LoopParams {
    int max_iterations;
    double lam2;         // damping^2
    double step_length;
    double stop_sq_tol;  // < 0: never stop
    int priority;        // priority level of the task (the stop test reads priority-0 rows only)
    int idmask;          // bit j: placement pl[j] has an exactly-identity rotation; bit NJ: frame_pl has
    int unit_weights;    // every Task::weighting() entry is exactly 1
    // derived-visitor family (include/ikgpu.h ikgpu_dls_params; read by the generic lane program only -- a solve that uses them is
    // routed there): dq_sq_tol > 0: also stop when ||dq||^2 < dq_sq_tol; nlt > 0: the error test is ||e[l]||^2 < lvl_tol[l], all l < nlt
    double dq_sq_tol;
    int nlt;
    double lvl_tol[8];
};

constexpr int kSpecUnit = 30;

template <int KT>
struct TaskDim {
    static constexpr int value = (KT == KT_FULL) ? 6 : 3;
};

// Evaluate e (M) and the NEGATED M x NJ task Jacobian columns (col = -J_task(:, j) = +W Jlog6(tMf) J_local(:, j):
// the minus sign of reference ik/ik/frame.hpp:173-181 is folded into the step, dq = +col^T y) at configuration q.
// oMt: target placement in the world (reference frame fixed in the world), 12 doubles.
// SMASK >= 0: specialisation known at compile time -- bits 0..NJ: identity-rotation placement mask (must be a subset
// of the problem's mask), bit kSpecUnit: every weight is exactly 1.  SMASK = -1: take the wave-uniform runtime values
// `idmask` / `unit_weights`.  A compile-time value keeps the whole iteration one basic block, so the scheduler can
// hoist the LDS reads of the constant table far ahead of their use -- with one wave per SIMD nothing else hides that
// latency (measured: s_waitcnt stalls were 17 % of wave cycles with runtime branches; +18 % solves/s without them).
// Desc: ChainDesc<NJ> (staged in LDS, or plain memory on the host) or IKD_CONST_AS ChainDesc<NJ> (HBM through scalar loads).
template <int NJ, int KT, int SMASK = -1, class Desc = ChainDesc<NJ>>
IKD_FN void chain_evaluate(const Desc &d, const double (&q)[NJ], const double (&oMt)[12], int idmask,
                           bool unit_weights, double (&e)[TaskDim<KT>::value], double (&col)[NJ][TaskDim<KT>::value],
                           double (&Rf)[9], double (&pf)[3]) {
    constexpr int M = TaskDim<KT>::value;
    double zax[NJ][3], org[NJ][3];
    double R[9], p[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = d.pl[0][k];
#pragma unroll
    for (int k = 0; k < 3; ++k) p[k] = d.pl[0][9 + k];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        if (j > 0) se3_compose_const(R, p, d.pl[j], SMASK >= 0 ? ((SMASK >> j) & 1) != 0 : ((idmask >> j) & 1) != 0);
        double s, c;
        if constexpr (SMASK < 0) dsincos(q[j], s, c);  // stage kernels and the runtime-parameter build
        else dsincos_fast(q[j], s, c);                 // the device's iteration loops
        rot_z_right(R, s, c);
        zax[j][0] = R[2]; zax[j][1] = R[5]; zax[j][2] = R[8];
        org[j][0] = p[0]; org[j][1] = p[1]; org[j][2] = p[2];
    }
    se3_compose_const(R, p, d.frame_pl, SMASK >= 0 ? ((SMASK >> NJ) & 1) != 0 : ((idmask >> NJ) & 1) != 0);
#pragma unroll
    for (int k = 0; k < 9; ++k) Rf[k] = R[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) pf[k] = p[k];

    // fMt = oMf^-1 oMt
    double Re[9], pe[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) Re[3 * i + j] = dfma(R[i], oMt[j], dfma(R[3 + i], oMt[3 + j], R[6 + i] * oMt[6 + j]));
    {
        const double dp[3] = {oMt[9] - p[0], oMt[10] - p[1], oMt[11] - p[2]};
        rotT_vec(R, dp, pe);
    }
    LogAndJlog lj;
    log6_and_jlog6_inv(Re, pe, lj);

    // K' = +diag(w) * Jlog6(tMf) restricted to the task rows:  top rows [At | Bt], bottom rows [0 | Ab]
    double At[9], Bt[9], Ab[9];
    constexpr int w0 = (KT == KT_FULL) ? 3 : 0;
    const bool unit = SMASK >= 0 ? ((SMASK >> kSpecUnit) & 1) != 0 : unit_weights;
    if (unit) {  // compile-time, or wave-uniform
#pragma unroll
        for (int k = 0; k < 9; ++k) { At[k] = lj.A[k]; Bt[k] = lj.Bm[k]; Ab[k] = lj.A[k]; }
        if (KT == KT_FULL) {
#pragma unroll
            for (int i = 0; i < 6; ++i) e[i] = lj.e[i];
        } else {
#pragma unroll
            for (int i = 0; i < 3; ++i) e[i] = lj.e[(KT == KT_ORIENTATION ? 3 : 0) + i];
        }
    } else {
        if (KT == KT_FULL || KT == KT_POSITION) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    At[3 * i + j] = d.wgt[i] * lj.A[3 * i + j];
                    Bt[3 * i + j] = d.wgt[i] * lj.Bm[3 * i + j];
                }
        }
        if (KT == KT_FULL || KT == KT_ORIENTATION) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) Ab[3 * i + j] = d.wgt[w0 + i] * lj.A[3 * i + j];
        }
        if (KT == KT_FULL) {
#pragma unroll
            for (int i = 0; i < 6; ++i) e[i] = lj.e[i] * d.wgt[i];
        } else if (KT == KT_POSITION) {
#pragma unroll
            for (int i = 0; i < 3; ++i) e[i] = lj.e[i] * d.wgt[i];
        } else {
#pragma unroll
            for (int i = 0; i < 3; ++i) e[i] = lj.e[3 + i] * d.wgt[i];
        }
    }

    // Columns.  J_local(:, j) = [r x w' ; w'] with w' = Rf^T z_j, r = Rf^T (o_j - p_f), and r x w' = Rf^T ((o_j - p_f) x z_j):
    // the rotation into the frame is folded once into the 3x3 blocks (K' Rf^T), so a column costs one world-frame
    // cross product and three 3x3 products instead of two extra rotations (-18 FMA-class instructions per joint).
    double AtR[9], BtR[9], AbR[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (KT == KT_FULL || KT == KT_POSITION) {
                AtR[3 * i + k] = dfma(At[3 * i], R[3 * k], dfma(At[3 * i + 1], R[3 * k + 1], At[3 * i + 2] * R[3 * k + 2]));
                BtR[3 * i + k] = dfma(Bt[3 * i], R[3 * k], dfma(Bt[3 * i + 1], R[3 * k + 1], Bt[3 * i + 2] * R[3 * k + 2]));
            }
        }
    if (KT == KT_FULL && unit) {  // Ab == At
#pragma unroll
        for (int k = 0; k < 9; ++k) AbR[k] = AtR[k];
    } else if (KT == KT_FULL || KT == KT_ORIENTATION) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int k = 0; k < 3; ++k)
                AbR[3 * i + k] = dfma(Ab[3 * i], R[3 * k], dfma(Ab[3 * i + 1], R[3 * k + 1], Ab[3 * i + 2] * R[3 * k + 2]));
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const double dj[3] = {org[j][0] - p[0], org[j][1] - p[1], org[j][2] - p[2]};
        double vw[3];
        cross(dj, zax[j], vw);
        if (KT == KT_FULL || KT == KT_POSITION) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
                col[j][i] = dfma(AtR[3 * i], vw[0], dfma(AtR[3 * i + 1], vw[1], dfma(AtR[3 * i + 2], vw[2],
                            dfma(BtR[3 * i], zax[j][0], dfma(BtR[3 * i + 1], zax[j][1], BtR[3 * i + 2] * zax[j][2])))));
        }
        if (KT == KT_FULL || KT == KT_ORIENTATION) {
            constexpr int r0 = (KT == KT_FULL) ? 3 : 0;
#pragma unroll
            for (int i = 0; i < 3; ++i)
                col[j][r0 + i] = dfma(AbR[3 * i], zax[j][0], dfma(AbR[3 * i + 1], zax[j][1], AbR[3 * i + 2] * zax[j][2]));
        }
    }
    (void)M;
}

// ONE iteration at q (reference ik/ik/dls.cpp:14-71): evaluate, Gram, solve, the visitor's test BEFORE the step (dls.cpp:61-64;
// ik/ik/visitor.hpp:19: ||e[0]||^2 < stop_sq_tol on the priority-0 rows), then q <- clip(q + step_length dq) where the lane is
// `active` and the visitor did not fire.  Returns "the visitor fired" (false for an inactive lane).  Shared by the lock-step loop
// (chain_dls) and the lane-refill loop (chain_kernel_body.hpp): one source of the arithmetic, bit-identical results.
template <int NJ, int KT, int SMASK = -1, class Desc>
IKD_FN bool chain_iteration(const Desc &d, const LoopParams &prm, double (&q)[NJ], const double (&oMt)[12], bool active) {
    constexpr int M = TaskDim<KT>::value;
    double e[M], col[NJ][M], Rf[9], pf[3];
    chain_evaluate<NJ, KT, SMASK>(d, q, oMt, prm.idmask, prm.unit_weights != 0, e, col, Rf, pf);

    double G[M * M];
#pragma unroll
    for (int a = 0; a < M; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b) {
            double s = (a == b) ? prm.lam2 : 0.0;
#pragma unroll
            for (int j = 0; j < NJ; ++j) s = dfma(col[j][a], col[j][b], s);
            G[a * M + b] = s;
        }
    double y[M];
    chol_solve<M>(G, e, y);

    double e0sq = 0.0;
    if (prm.priority == 0) {
#pragma unroll
        for (int a = 0; a < M; ++a) e0sq = dfma(e[a], e[a], e0sq);
    }
    const bool stop_now = active && (prm.stop_sq_tol >= 0.0) && (e0sq < prm.stop_sq_tol);
    const bool step = active && !stop_now;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        double s = 0.0;
#pragma unroll
        for (int a = 0; a < M; ++a) s = dfma(col[j][a], y[a], s);
        const double qn = dfma(prm.step_length, s, q[j]);  // dq_j = -J_task(:, j)^T y = +col_j^T y
        const double qc = dmin(d.hi[j], dmax(qn, d.lo[j]));
        q[j] = step ? qc : q[j];
    }
    return stop_now;
}

// The wave-uniform "keep iterating" test of the lock-step loops (chain_dls, hot_dls, tree_dls call it once per iteration): some lane is
// still active -- and, in the first phase of a two-phase solve, more than `leave_active` lanes are once `leave_after` iterations are
// done: the stragglers of a wave are cheaper to finish in compacted waves than with three quarters of this one idle.
struct KeepGoing {
    int leave_active, leave_after, n;
    IKD_FN bool operator()(bool act) {
        ++n;
#if IKD_ON_DEVICE
        const int live = static_cast<int>(__popcll(__ballot(act)));
        return live > (n >= leave_after ? leave_active : 0);
#else
        return act;
#endif
    }
};
// Iterations a lock-step loop has taken when it ends with this lane unfinished: with a plain "any lane active" test the loop only ends
// early when every lane is done, so an unfinished lane saw all of them; KeepGoing counts its calls (one per iteration).
template <class F>
IKD_FN int iterations_taken(const F &, int max_iterations) { return max_iterations; }
IKD_FN int iterations_taken(const KeepGoing &k, int) { return k.n; }


// One full solve. q: in = q0 (chain joints only), out = result. Returns iterations / success.
// any_active(bool) must return a wave-uniform "some lane still iterating" (identity on the host).
template <int NJ, int KT, int SMASK = -1, class Desc, class AnyFn>
IKD_FN void chain_dls(const Desc &d_in, const LoopParams &prm, double (&q)[NJ], const double (&oMt)[12],
                      int &iters_out, bool &success_out, AnyFn any_active) {
    bool active = true;
    bool success = false;
    int iters = prm.max_iterations;
    const Desc *dp = &d_in;
#pragma unroll 1
    for (int it = 0; it < prm.max_iterations; ++it) {
        // The constant table is re-read every iteration (scalar loads from the constant address space, or broadcast
        // ds_read from LDS: both off the VALU) instead of letting the compiler hoist ~150 doubles into registers.
        asm volatile("" ::: "memory");
        if constexpr (!std::is_same<Desc, ChainDesc<NJ>>::value) IKD_LAUNDER(dp);  // constant address space: see lane_math.hpp
        const bool stop_now = chain_iteration<NJ, KT, SMASK>(*dp, prm, q, oMt, active);
        if (stop_now) { success = true; iters = it; }
        active = active && !stop_now;
        if (!any_active(active)) break;
    }
    // (a lane the loop left unfinished reports the iterations it took: max_iterations, or fewer when the wave left early -- the first
    // phase of a two-phase solve, chain_kernel_body.hpp KeepGoing)
    iters_out = success ? iters : iterations_taken(any_active, prm.max_iterations);
    success_out = success;
}

}  // namespace ikdev
