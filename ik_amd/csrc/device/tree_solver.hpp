// tree_solver.hpp -- the whole ik::dls() loop for ONE problem on a free-flyer model whose tasks
// hang off the floating base: up to two serial chains of revolute joints, each ending in a frame
// task, plus an optional frame task on the base link itself (shape "F" of SURVEY.md section 8:
// Cassie full body, nq = 23 / nv = 22, LeftFootFront + RightFootFront + pelvis, M = 18).
//
// Reference path restated per iteration (file:line relative to the reference root):
//   ik/ik/data.cpp:28-30     FK + joint Jacobians (free-flyer columns = Ad(oM1))
//   ik/ik/frame.hpp:37-62    e_t = log6(oMf^-1 oMt)            per task
//   ik/ik/frame.hpp:152-182  J_t = -Jlog6(tMf) J_local          per task
//   ik/ik/data.cpp:49-50     weighting;  ik/ik/dls.cpp:18-24 stacking of all priority levels
//   ik/ik/dls.cpp:39-53      dq = -Jt^T (Jt Jt^T + damping^2 I)^-1 et
//   ik/ik/dls.cpp:61-71      stop test, pinocchio::integrate (SE(3) for the base), joint clipping
//
// The linear solve uses the identity  J^T (J J^T + l I)^-1 = (J^T J + l I)^-1 J^T : the 20 x 20
// normal matrix H = J^T J + l I has an arrow structure (the two chains couple only through the six
// base columns), so each chain is eliminated by a 7 x 7 Cholesky and the base by a 6 x 6 Schur
// complement -- the same dq as the reference's dense 18 x 18 LDL^T up to rounding (cond(H) ~ 1e5),
// with a third of its registers.  A Position / Orientation task is a Full task whose unused rows
// carry weight zero: zero rows change neither H nor J^T e.
#pragma once
#include "chain_solver.hpp"
#include "chain_hot.hpp"   // ChainStruct: the placement-structure codes

namespace ikdev {

// Constant table of one chain (axis-folded, see chain_solver.hpp) and of the whole problem.
template <int NJ>
struct ChainTable {
    double pl[NJ][12];  // pl[0]: base joint frame -> first joint of the chain; pl[j]: joint j-1 -> j
    double fr[12];      // last joint -> task frame
    double lo[NJ], hi[NJ];
    double w[6];        // six-row weights (zeros on rows the task's kinematic type drops)
};

template <int NJ, int NCH>
struct TreeDesc {
    ChainTable<NJ> chain[NCH];
    double frP[12];  // base joint frame -> frame of the base task
    double wP[6];
    double refpl[NCH][12];  // base joint frame -> reference frame of chain c's task (used when TreeParams::ref_base[c])
};  // all doubles: staged HBM -> LDS as a flat table

constexpr int kMaxPostOut = 32;

struct TreeParams {
    int max_iterations;
    double lam2, step_length, stop_sq_tol;
    int prio[2], prioP;  // priority level per task; the stop test sums priority-0 rows
    int hasP;            // a base task is present
    int idmask[2], idmaskP;  // identity-rotation placement masks (see chain_solver.hpp LoopParams::idmask)
    int unit[2], unitP;      // the task is Full with all-ones weights
    // general builds only (SPEC <= 0), what the reference's demo adds to two pose tasks (ik_ros/src/cassie.cpp:45-81):
    int ref_base[2];         // chain c's target is given in a frame that rides on the floating base (TreeDesc::refpl[c]);
                             // the task Jacobian leaves that frame's motion out, as the reference does (frame.hpp:152-182)
    int align_chain;         // -1, or the chain whose task frame also carries an AlignAxisTask row (frame.hpp:257-301),
    int align_axis;          //   the frame axis 0 / 1 / 2 it aligns,
    int align_slot;          //   the target slot whose doubles 9..11 hold the direction (in the world),
    int align_prio;          //   its priority level
    double align_w;          //   and its weight
    int fixed_base;          // general builds: the base does not move (a fixed-base model): dq_base is dropped, the base pose stays the world
    // constraint builds only (SPEC bit kSpecCons, or -1): ONE ik::FrameConstraint (reference ik/ik/frame.hpp:325-449) whose reference
    // frame is the universe, on a frame at the end of a second chain that carries no task (the pinned stance foot): chain 1 of
    // the table is that chain, and the step is projected onto the null space of the constraint Jacobian (dls.cpp:26-34,43-53)
    int cons_on;
    int cons_type;           // KT_POSITION / KT_ORIENTATION / KT_FULL: the rows of the frame's LOCAL Jacobian that are held
    // general builds: ik::pik with TWO priority levels in the one shape the tree structure makes cheap (reference
    // ik/ik/pik.cpp:47-61) -- level 0: the frame tasks, among them a Full task on the base link; level 1: the AlignAxisTask row.
    // lam2 is then lambda[0]^2 and pik_lam2_1 = lambda[1]^2.  See PikRow.
    int pik_on;
    double pik_lam2_1;
    // posture builds only (SPEC bit kSpecPost, or -1): PostureTask rows (reference ik/ik/posture.hpp:51-68), one per joint,
    // e = (q - target) mask w, J = w in the joint's own tangent column; all on one priority level.  A row on a chain joint
    // adds w^2 to that joint's diagonal of the normal matrix and -w e to its right-hand side; a joint outside the chains is
    // its own 1x1 system, dq = -w e / (w^2 + lambda^2), its q kept in the caller's q_out column between iterations.
    int post_on;                        // the problem has posture rows
    int post_prio;                      // their priority level
    int post_n;                         // rows on joints outside the chains
    int post_q[kMaxPostOut];            //   index in q,
    int post_slot[kMaxPostOut];         //   target slot (the value is double 9 of the slot),
    double post_w[kMaxPostOut], post_m[kMaxPostOut];  // weight and mask entry
    int postc_slot[2][8];               // rows on chain joints: target slot, -1 where the joint carries none
    double postc_w[2][8], postc_m[2][8];
};

// One AlignAxisTask row on a chain's task frame, reference frame = the world: e = w (1 - r . t), r = the frame's axis in the
// world, and (negated convention) J'(:, c) = w (r x t) . omega_c with omega_c the world angular velocity of column c.
struct AlignRow {
    bool on;
    int ax;
    double w, tn[3];
    bool prio0;
    bool pik;   // the row is level 1 of ik::pik: it stays out of the level-0 system, leg_eval_factor returns a PikRow instead
};

// Level 1 of the two-level ik::pik the tree kernel takes (reference ik/ik/pik.cpp:47-61).  Level 0 holds a Full task on the
// base link whose 6 x 6 block is invertible, so the row space of J_0 is (all six base directions) + (the row space of the
// chain task's chain columns C): the projector after level 0 is P = diag(0_6, I - V^T V, I) with V an orthonormal basis of
// rowspace(C).  The level-1 row is the AlignAxisTask row j = [0 | b (base angular) | a (chain)], so Jbar = j P = [0 | 0 | abar],
// abar = a - V^T V a, its damped pseudo-inverse is Jbar^T / (|abar|^2 + lambda_1^2), and the level's step changes the chain's
// joints only:  dq_chain -= abar (e_1 - j dq_0) / (|abar|^2 + lambda_1^2).  Negated-Jacobian convention: a, b hold -j.
template <int NJ>
struct PikRow {
    bool on;
    double a[NJ], abar[NJ], b[3], e;
};

// Packed lower-triangular index
IKD_FN constexpr int tri(int i, int j) { return i * (i + 1) / 2 + j; }

// Eigen::Quaternion::toRotationMatrix, coefficients (x, y, z, w), not normalised (App. A.2)
IKD_FN void quat_to_R(const double (&qb)[7], double (&R)[9]) {
    const double x = qb[3], y = qb[4], z = qb[5], w = qb[6];
    const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
    R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}

// What one frame task contributes, given its frame placement (Rf, pf) in the world: weighted error e (6), and
// K' = +diag(w) Jlog6(tMf) as blocks -- top rows [At | Bt], bottom rows [0 | Ab] -- each already multiplied by Rf^T
// on the right, so that they act on WORLD-frame twists taken about the frame origin:
//   -J_task(:, c) = [At (d x w) + Bt w ; Ab w],  (v, w) the world Jacobian column, d = origin_of_the_joint - p_f.
struct TaskTerms {
    double e[6];
    double At[9], Bt[9], Ab[9];
};

// out = X * R^T
IKD_FN void mul_RT(const double (&X)[9], const double (&R)[9], double (&out)[9]) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) out[3 * i + k] = dfma(X[3 * i], R[3 * k], dfma(X[3 * i + 1], R[3 * k + 1], X[3 * i + 2] * R[3 * k + 2]));
}

template <class WPtr>  // const double *, in LDS / host memory or in the constant address space
IKD_FN void task_terms(const double (&Rf)[9], const double (&pf)[3], const double (&oMt)[12], WPtr w6, bool unit,
                       TaskTerms &t) {
    double Re[9], pe[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) Re[3 * i + j] = dfma(Rf[i], oMt[j], dfma(Rf[3 + i], oMt[3 + j], Rf[6 + i] * oMt[6 + j]));
    const double dp[3] = {oMt[9] - pf[0], oMt[10] - pf[1], oMt[11] - pf[2]};
    rotT_vec(Rf, dp, pe);
    LogAndJlog lj;
    double Cm[9];
    log6_and_jlog6_hot<false>(Re, pe, lj, &Cm);   // the branch-free, one-reciprocal front end (lane_math.hpp): full body 0.84 -> 0.81 ms
    // K' = +diag(w) Jlog6(tMf): the task Jacobian is carried NEGATED (J' = -J_task); H = J'^T J' is unchanged and
    // the right-hand side becomes g' = J'^T e = -J_task^T e, so the system solved is H dq = g'.
    // Jlog6's top-right block is C A: (C A) Rf^T is formed as C (A Rf^T), the product C A itself never (27 multiply-adds less), and
    // the weights scale the rows of the two products afterwards instead of the factors before.
    double AR[9], CAR[9];
    mul_RT(lj.A, Rf, AR);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) CAR[3 * i + k] = dfma(Cm[3 * i], AR[k], dfma(Cm[3 * i + 1], AR[3 + k], Cm[3 * i + 2] * AR[6 + k]));
    if (unit) {  // wave-uniform: Full task, all weights exactly 1
#pragma unroll
        for (int k = 0; k < 6; ++k) t.e[k] = lj.e[k];
#pragma unroll
        for (int k = 0; k < 9; ++k) { t.At[k] = AR[k]; t.Bt[k] = CAR[k]; t.Ab[k] = AR[k]; }
        return;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double wt = w6[i], wb = w6[3 + i];
        t.e[i] = lj.e[i] * wt;
        t.e[3 + i] = lj.e[3 + i] * wb;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            t.At[3 * i + j] = wt * AR[3 * i + j];
            t.Bt[3 * i + j] = wt * CAR[3 * i + j];
            t.Ab[3 * i + j] = wb * AR[3 * i + j];
        }
    }
}

// (Negated) task Jacobian columns of the six free-flyer DoFs: J_local = Ad(oMf^-1 oM1).
// JL[c] (c = 0..2, linear DoFs): only the top three rows are non-zero.  JA[c]: all six rows.
IKD_FN void base_columns(const TaskTerms &t, const double (&pf)[3], const double (&R1)[9], const double (&p1)[3],
                         double (&JL)[3][3], double (&JA)[3][6]) {
    // world columns of the free-flyer: linear DoF c: (R1 e_c, 0); angular DoF c: (p1 x R1 e_c, R1 e_c) about the world
    // origin, i.e. ((p1 - pf) x R1 e_c, R1 e_c) about the frame origin
    const double dp[3] = {p1[0] - pf[0], p1[1] - pf[1], p1[2] - pf[2]};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double rc[3] = {R1[c], R1[3 + c], R1[6 + c]};
        double pxr[3];
        cross(dp, rc, pxr);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            JL[c][i] = dfma(t.At[3 * i], rc[0], dfma(t.At[3 * i + 1], rc[1], t.At[3 * i + 2] * rc[2]));
            JA[c][i] = dfma(t.At[3 * i], pxr[0], dfma(t.At[3 * i + 1], pxr[1], dfma(t.At[3 * i + 2], pxr[2],
                       dfma(t.Bt[3 * i], rc[0], dfma(t.Bt[3 * i + 1], rc[1], t.Bt[3 * i + 2] * rc[2])))));
            JA[c][3 + i] = dfma(t.Ab[3 * i], rc[0], dfma(t.Ab[3 * i + 1], rc[1], t.Ab[3 * i + 2] * rc[2]));
        }
    }
}

// Hbb += Jb^T Jb, gb += Jb^T e  for the six base columns (Jb = [JL | JA])
IKD_FN void accumulate_base(const double (&JL)[3][3], const double (&JA)[3][6], const double (&e)[6], double (&Hbb)[21],
                            double (&gb)[6]) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int b = 0; b <= a; ++b)
            Hbb[tri(a, b)] = dfma(JL[a][0], JL[b][0], dfma(JL[a][1], JL[b][1], dfma(JL[a][2], JL[b][2], Hbb[tri(a, b)])));
        gb[a] = dfma(JL[a][0], e[0], dfma(JL[a][1], e[1], dfma(JL[a][2], e[2], gb[a])));
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int b = 0; b < 3; ++b)
            Hbb[tri(3 + a, b)] = dfma(JA[a][0], JL[b][0], dfma(JA[a][1], JL[b][1], dfma(JA[a][2], JL[b][2], Hbb[tri(3 + a, b)])));
#pragma unroll
        for (int b = 0; b <= a; ++b) {
            double s = Hbb[tri(3 + a, 3 + b)];
#pragma unroll
            for (int r = 0; r < 6; ++r) s = dfma(JA[a][r], JA[b][r], s);
            Hbb[tri(3 + a, 3 + b)] = s;
        }
        double s = gb[3 + a];
#pragma unroll
        for (int r = 0; r < 6; ++r) s = dfma(JA[a][r], e[r], s);
        gb[3 + a] = s;
    }
}

// Per-chain factorisation kept for the back-substitution: H_ll = L L^T (packed, diagonal holds
// 1/L_ii), W = L^-1 H_lb (NJ x 6), u = L^-1 g_l.
template <int NJ>
struct LegFactor {
    double L[NJ * (NJ + 1) / 2];
    double W[NJ][6];
    double u[NJ];
};

// Evaluate one chain task at the current configuration, add its base-block contributions to
// (Hbb, gb), eliminate its NJ joint unknowns (Schur complement onto the base) and return the factor.
// FAST: sin / cos by dsincos_fast (the device loops) instead of dsincos (runtime-parameter build), lane_math.hpp.
// This lane's element of row r of a caller's array of doubles laid out [rows][B] (component-major) or [B][rows] (problem-major):
//   *(base + r * stride_bytes + off).
// Lock-step kernels: `base` is the array at the workgroup's first problem (wave-uniform) and `off` the lane's byte offset from
// there, 32 bits -- a row's address is scalar arithmetic and the load takes the SGPR-base + VGPR-offset form.  (With a 64-bit
// per-lane pointer per row hipcc hoisted up to 36 row addresses out of the iteration loop, two VGPRs each, and spilled them to
// scratch memory: every target word then cost a scratch reload AND the load, back to back.)  Refill kernels, whose lanes change
// problems: `base` is the lane's own pointer, off = 0.
struct LaneRows {
    const char *base;
    uint32_t off;
    int64_t stride_bytes;    // bytes between consecutive rows: B * 8 or 8 (64 bits: 2^29 problems of a small task set fit the device)
    bool uniform;   // `base` is wave-uniform (a literal at every construction site: folds after inlining)
    IKD_FN double operator()(int r) const {
        const char *row = base + static_cast<int64_t>(r) * stride_bytes;
#if IKD_ON_DEVICE
        // pin the row address in an SGPR pair: left to itself hipcc adds `off` to `base` first and is back to one 64-bit VGPR address
        // per row (the asm hides the pointer's provenance, so its address space is stated: a generic pointer would be a flat_load)
        if (uniform) {
            typedef const char __attribute__((address_space(1))) *GlobalBytes;
            typedef const double __attribute__((address_space(1))) *GlobalDouble;
            GlobalBytes g = (GlobalBytes)row;
            asm("" : "+s"(g));
            // (and the zero-extension of `off` has to sit in the load's own basic block for the SGPR-base + 32-bit-VGPR-offset form
            // to be selected: hoisted out of the loop as a 64-bit pair it turns every load into a 64-bit vector add + load)
            uint32_t o = off;
            asm volatile("" : "+v"(o));
            return *reinterpret_cast<GlobalDouble>(g + o);
        }
#endif
        return *reinterpret_cast<const double *>(row + off);
    }
    // N consecutive rows from row r0: one scalar multiply for the first row, one 64-bit scalar add per further row
    template <int N>
    IKD_FN void run(int r0, double (&out)[N]) const {
#if IKD_ON_DEVICE
        if (uniform) {
            typedef const char __attribute__((address_space(1))) *GlobalBytes;
            typedef const double __attribute__((address_space(1))) *GlobalDouble;
            GlobalBytes g = (GlobalBytes)(base + static_cast<int64_t>(r0) * stride_bytes);
#pragma unroll
            for (int k = 0; k < N; ++k) {
                asm("" : "+s"(g));
                uint32_t o = off;
                asm volatile("" : "+v"(o));
                out[k] = *reinterpret_cast<GlobalDouble>(g + o);
                g += stride_bytes;
            }
            return;
        }
#endif
#pragma unroll
        for (int k = 0; k < N; ++k) out[k] = (*this)(r0 + k);
    }
    // The same, with the row addresses formed where they are used: `run` leaves the N row pointers loop-invariant, and the compiler
    // hoists them out of the iteration loop -- 2 SGPRs per row, free while they fit, but a program that reads 45 target words (four
    // tasks) then spills scalar registers into VGPR lanes (primal_solver.hpp: 80 v_writelane / 64 v_readlane per iteration).
    template <int N>
    IKD_FN void run_fresh(int r0, double (&out)[N]) const {
#if IKD_ON_DEVICE
        if (uniform) {
            typedef const char __attribute__((address_space(1))) *GlobalBytes;
            typedef const double __attribute__((address_space(1))) *GlobalDouble;
            GlobalBytes g = (GlobalBytes)(base + static_cast<int64_t>(r0) * stride_bytes);
            asm volatile("" : "+s"(g));
#pragma unroll
            for (int k = 0; k < N; ++k) {
                asm("" : "+s"(g));
                uint32_t o = off;
                asm volatile("" : "+v"(o));
                out[k] = *reinterpret_cast<GlobalDouble>(g + o);
                g += stride_bytes;
            }
            return;
        }
#endif
        run<N>(r0, out);
    }
};

// (R, p) <- (R, p) * placement i of a chain whose placement STRUCTURE is the compile-time code S (chain_hot.hpp ChainStruct: every
// rotation entry exactly 0 / +1 / -1 or general, every translation component zero or not).  The same expressions as
// se3_compose_const, with the structural terms written out -- a zero term is not formed, a +-1 factor is an add / subtract -- so no
// compiler flag is needed for the folding and the result has the bits of the full product on finite data (fma(x, 1, t) = x + t;
// fma(x, 0, t) = t).  Entries that are structural are never READ (the table sits in LDS: a Cassie leg reads 22 words instead of 96).
// `i` is the index of an unrolled loop: S::ent / S::tnz fold to constants there.
IKD_FN void struct_term(bool &has, double &v, double x, unsigned cls, double m) {
    if (cls == kEntZero) return;
    if (cls == kEntOne) v = has ? x + v : x;
    else if (cls == kEntMinusOne) v = has ? v - x : -x;
    else v = has ? dfma(x, m, v) : x * m;
    has = true;
}
template <class S, class ConstPtr>
IKD_FN void se3_compose_struct(double (&R)[9], double (&p)[3], ConstPtr c, int i) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const double a = R[3 * r], b = R[3 * r + 1], d = R[3 * r + 2];
        double t = p[r];
        if (S::tnz(i, 2)) t = dfma(d, c[11], t);
        if (S::tnz(i, 1)) t = dfma(b, c[10], t);
        if (S::tnz(i, 0)) t = dfma(a, c[9], t);
        p[r] = t;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            bool has = false;
            double v = 0.0;
            struct_term(has, v, d, S::ent(i, 6 + k), S::ent(i, 6 + k) == kEntGeneral ? c[6 + k] : 0.0);
            struct_term(has, v, b, S::ent(i, 3 + k), S::ent(i, 3 + k) == kEntGeneral ? c[3 + k] : 0.0);
            struct_term(has, v, a, S::ent(i, k), S::ent(i, k) == kEntGeneral ? c[k] : 0.0);
            R[3 * r + k] = v;
        }
    }
}

// The structure code the mask-folded tree builds are compiled for (NJ = 7: a Cassie leg -- both legs share it, and it is the hot chain
// kernel's code for the same leg, kernels_hot.hip); the launcher takes the hot build only for chains that carry it.
constexpr uint64_t kTreeHotCode7[3] = {0x04f0208cce8c7664ull, 0x395959cacad65656ull, 0x000001cacace5656ull};
template <int NJ> struct TreeHotStruct { typedef void type; static constexpr int mask = 0; };
template <> struct TreeHotStruct<7> { typedef ChainStruct<kTreeHotCode7[0], kTreeHotCode7[1], kTreeHotCode7[2]> type; static constexpr int mask = 0xf8; };   // (mask: the code's identity-rotation placements, kernels.hip HotMask)

// Posture rows on the joints of one chain (wave-uniform description; targets per lane).
struct ChainPosture {
    bool on, prio0;
    const int *slot;        // [NJ] target slot or -1
    const double *w, *m;    // [NJ]
    LaneRows targets;
    const double *t_chain;  // when not null: the lane's LDS column of the chain rows' target values, row a at t_chain[a * t_stride]
    int64_t t_stride;
};

template <int NJ, bool FAST = false, bool POST = false, bool PIK = false, class S = void, class PlPtr, class FrPtr, class WPtr>
IKD_FN void leg_eval_factor(const double (&R1)[9], const double (&p1)[3], PlPtr pl, FrPtr frame_pl,
                            WPtr w6, int idmask, bool unit, const double (&q)[NJ], const double (&oMt)[12],
                            double lam2, bool prio0, const AlignRow &al, const ChainPosture &po, double (&Hbb)[21],
                            double (&gb)[6], double &e0sq, LegFactor<NJ> &F, PikRow<NJ> &pr) {
    double zax[NJ][3], org[NJ][3];
    double R[9], p[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = R1[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) p[k] = p1[k];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        if constexpr (std::is_same<S, void>::value) se3_compose_const(R, p, pl[j], (idmask >> j) & 1);
        else se3_compose_struct<S>(R, p, pl[j], j);
        double s, c;
        if constexpr (FAST) dsincos_fast(q[j], s, c);
        else dsincos(q[j], s, c);
        rot_z_right(R, s, c);
        zax[j][0] = R[2]; zax[j][1] = R[5]; zax[j][2] = R[8];
        org[j][0] = p[0]; org[j][1] = p[1]; org[j][2] = p[2];
    }
    if constexpr (std::is_same<S, void>::value) se3_compose_const(R, p, frame_pl, (idmask >> NJ) & 1);
    else se3_compose_struct<S>(R, p, frame_pl, NJ);

    IKD_SCHED_FENCE();
    TaskTerms t;
    task_terms(R, p, oMt, w6, unit, t);
    if (prio0) {
#pragma unroll
        for (int r = 0; r < 6; ++r) e0sq = dfma(t.e[r], t.e[r], e0sq);
    }
    double JL[3][3], JA[3][6];
    base_columns(t, p, R1, p1, JL, JA);
    accumulate_base(JL, JA, t.e, Hbb, gb);

    IKD_SCHED_FENCE();
    // chain columns (world-frame twist about the frame origin, see TaskTerms)
    double col[NJ][6];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const double dj[3] = {org[j][0] - p[0], org[j][1] - p[1], org[j][2] - p[2]};
        double vw[3];
        cross(dj, zax[j], vw);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            col[j][i] = dfma(t.At[3 * i], vw[0], dfma(t.At[3 * i + 1], vw[1], dfma(t.At[3 * i + 2], vw[2],
                        dfma(t.Bt[3 * i], zax[j][0], dfma(t.Bt[3 * i + 1], zax[j][1], t.Bt[3 * i + 2] * zax[j][2])))));
            col[j][3 + i] = dfma(t.Ab[3 * i], zax[j][0], dfma(t.Ab[3 * i + 1], zax[j][1], t.Ab[3 * i + 2] * zax[j][2]));
        }
    }
    if (PIK && al.on && al.pik) {   // level 1 of ik::pik (wave-uniform; PIK builds only), see PikRow.  Done here, while the chain
                                    // columns are live and the factor is not yet: the basis V costs 42 registers
        const double r[3] = {al.ax == 0 ? R[0] : (al.ax == 1 ? R[1] : R[2]), al.ax == 0 ? R[3] : (al.ax == 1 ? R[4] : R[5]),
                             al.ax == 0 ? R[6] : (al.ax == 1 ? R[7] : R[8])};
        double rxt[3], aj[NJ], ab[3];
        cross(r, al.tn, rxt);
        const double ea = (1.0 - dot(r, al.tn)) * al.w;
        if (al.prio0) e0sq = dfma(ea, ea, e0sq);
#pragma unroll
        for (int j = 0; j < NJ; ++j) aj[j] = al.w * dot(rxt, zax[j]);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double rc[3] = {R1[c], R1[3 + c], R1[6 + c]};
            ab[c] = al.w * dot(rxt, rc);
        }
        // orthonormal basis of the row space of the chain task's chain columns (rows with weight zero -- a Position /
        // Orientation task -- are skipped, wave-uniform), Gram-Schmidt with every projection applied twice, rank rule as in
        // constraint_project; then abar = a - V^T V a, twice
        double V[6][NJ];
        double maxn2 = 0.0;
        int rows = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            if (w6[r] == 0.0) continue;
            ++rows;
            double n2 = 0.0;
#pragma unroll
            for (int j = 0; j < NJ; ++j) { V[r][j] = col[j][r]; n2 = dfma(V[r][j], V[r][j], n2); }
            maxn2 = dmax(maxn2, n2);
        }
        const double thr = 2.220446049250313e-16 * static_cast<double>(rows < NJ ? rows : NJ);
        const double thr2 = thr * thr * maxn2;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            if (w6[k] == 0.0) continue;
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
                for (int i = 0; i < k; ++i) {
                    if (w6[i] == 0.0) continue;
                    double d = 0.0;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) d = dfma(V[i][j], V[k][j], d);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) V[k][j] = dfma(-d, V[i][j], V[k][j]);
                }
            }
            double n2 = 0.0;
#pragma unroll
            for (int j = 0; j < NJ; ++j) n2 = dfma(V[k][j], V[k][j], n2);
            const double inv = dsel(n2 > thr2, drsqrt(dmax(n2, 1e-300)), 0.0);
#pragma unroll
            for (int j = 0; j < NJ; ++j) V[k][j] = V[k][j] * inv;
        }
        pr.on = true;
        pr.e = ea;
#pragma unroll
        for (int c = 0; c < 3; ++c) pr.b[c] = ab[c];
#pragma unroll
        for (int j = 0; j < NJ; ++j) { pr.a[j] = aj[j]; pr.abar[j] = aj[j]; }
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                if (w6[k] == 0.0) continue;
                double d = 0.0;
#pragma unroll
                for (int j = 0; j < NJ; ++j) d = dfma(V[k][j], pr.abar[j], d);
#pragma unroll
                for (int j = 0; j < NJ; ++j) pr.abar[j] = dfma(-d, V[k][j], pr.abar[j]);
            }
        }
    }
    IKD_SCHED_FENCE();
    // H_ll (packed), H_lb -> W, g_l -> u
#pragma unroll
    for (int a = 0; a < NJ; ++a) {
#pragma unroll
        for (int b = 0; b <= a; ++b) {
            double s = (a == b) ? lam2 : 0.0;
#pragma unroll
            for (int r = 0; r < 6; ++r) s = dfma(col[a][r], col[b][r], s);
            F.L[tri(a, b)] = s;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            F.W[a][c] = dfma(col[a][0], JL[c][0], dfma(col[a][1], JL[c][1], col[a][2] * JL[c][2]));
            double s = 0.0;
#pragma unroll
            for (int r = 0; r < 6; ++r) s = dfma(col[a][r], JA[c][r], s);
            F.W[a][3 + c] = s;
        }
        double s = 0.0;
#pragma unroll
        for (int r = 0; r < 6; ++r) s = dfma(col[a][r], t.e[r], s);
        F.u[a] = s;
    }
    if (al.on && !(PIK && al.pik)) {  // rank-one terms of the alignment row (wave-uniform; general builds only).  A level-1 row of
                                      // ik::pik stays out of the level-0 system: it became the PikRow above
        const double r[3] = {al.ax == 0 ? R[0] : (al.ax == 1 ? R[1] : R[2]), al.ax == 0 ? R[3] : (al.ax == 1 ? R[4] : R[5]),
                             al.ax == 0 ? R[6] : (al.ax == 1 ? R[7] : R[8])};
        double rxt[3], aj[NJ], ab[3];
        cross(r, al.tn, rxt);
        const double ea = (1.0 - dot(r, al.tn)) * al.w;
        if (al.prio0) e0sq = dfma(ea, ea, e0sq);
#pragma unroll
        for (int j = 0; j < NJ; ++j) aj[j] = al.w * dot(rxt, zax[j]);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double rc[3] = {R1[c], R1[3 + c], R1[6 + c]};
            ab[c] = al.w * dot(rxt, rc);
        }
#pragma unroll
        for (int a = 0; a < NJ; ++a) {
#pragma unroll
            for (int b = 0; b <= a; ++b) F.L[tri(a, b)] = dfma(aj[a], aj[b], F.L[tri(a, b)]);
#pragma unroll
            for (int c = 0; c < 3; ++c) F.W[a][3 + c] = dfma(aj[a], ab[c], F.W[a][3 + c]);
            F.u[a] = dfma(aj[a], ea, F.u[a]);
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
#pragma unroll
            for (int b = 0; b <= a; ++b) Hbb[tri(3 + a, 3 + b)] = dfma(ab[a], ab[b], Hbb[tri(3 + a, 3 + b)]);
            gb[3 + a] = dfma(ab[a], ea, gb[3 + a]);
        }
    }
    if (POST && po.on) {  // posture rows on this chain's joints (wave-uniform; posture builds only)
#pragma unroll
        for (int a = 0; a < NJ; ++a) {
            const int slot = po.slot[a];
            if (slot >= 0) {
                const double w = po.w[a];
                const double tgt = po.t_chain ? po.t_chain[a * po.t_stride] : po.targets(slot * 12 + 9);
                const double ea = (q[a] - tgt) * po.m[a] * w;
                if (po.prio0) e0sq = dfma(ea, ea, e0sq);
                F.L[tri(a, a)] = dfma(w, w, F.L[tri(a, a)]);
                F.u[a] = dfma(-w, ea, F.u[a]);  // negated-Jacobian convention: the row's column entry is -w
            }
        }
    }
    IKD_SCHED_FENCE();
    // Cholesky of H_ll, right-looking, with the 6 + 1 right-hand sides (W, u) carried as extra rows: every
    // trailing update is an independent FMA (see chol_solve in lane_math.hpp)
#pragma unroll
    for (int k = 0; k < NJ; ++k) {
        const double inv = drsqrt(F.L[tri(k, k)]);
        F.L[tri(k, k)] = inv;
#pragma unroll
        for (int c = 0; c < 6; ++c) F.W[k][c] = F.W[k][c] * inv;
        F.u[k] = F.u[k] * inv;
#pragma unroll
        for (int i = k + 1; i < NJ; ++i) F.L[tri(i, k)] = F.L[tri(i, k)] * inv;
#pragma unroll
        for (int i = k + 1; i < NJ; ++i) {
            const double lik = F.L[tri(i, k)];
#pragma unroll
            for (int jj = k + 1; jj <= i; ++jj) F.L[tri(i, jj)] = dfma(-lik, F.L[tri(jj, k)], F.L[tri(i, jj)]);
#pragma unroll
            for (int c = 0; c < 6; ++c) F.W[i][c] = dfma(-lik, F.W[k][c], F.W[i][c]);
            F.u[i] = dfma(-lik, F.u[k], F.u[i]);
        }
    }
    // Schur complement onto the base: Hbb -= W^T W, gb -= W^T u
#pragma unroll
    for (int a = 0; a < 6; ++a) {
#pragma unroll
        for (int b = 0; b <= a; ++b) {
            double s = Hbb[tri(a, b)];
#pragma unroll
            for (int j = 0; j < NJ; ++j) s = dfma(-F.W[j][a], F.W[j][b], s);
            Hbb[tri(a, b)] = s;
        }
        double s = gb[a];
#pragma unroll
        for (int j = 0; j < NJ; ++j) s = dfma(-F.W[j][a], F.u[j], s);
        gb[a] = s;
    }
}

// dq_l = L^-T (u' - W dq_b)   (u' = L^-1 g'_l with the negated Jacobian convention of task_terms)
template <int NJ>
IKD_FN void leg_back_substitute(const LegFactor<NJ> &F, const double (&dqb)[6], double (&dql)[NJ]) {
    double t[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        double s = F.u[j];
#pragma unroll
        for (int c = 0; c < 6; ++c) s = dfma(-F.W[j][c], dqb[c], s);
        t[j] = s;
    }
#pragma unroll
    for (int k = NJ - 1; k >= 0; --k) {
        dql[k] = t[k] * F.L[tri(k, k)];
#pragma unroll
        for (int m = 0; m < k; ++m) t[m] = dfma(-F.L[tri(k, m)], dql[k], t[m]);
    }
}

// pinocchio::integrate for the free-flyer: q_b <- q_b (+) v, v = [v_lin; omega]
IKD_FN void freeflyer_integrate(const double (&qb)[7], const double (&R1)[9], const double (&v)[6], double (&out)[7]) {
    const double w[3] = {v[3], v[4], v[5]};
    const double vl[3] = {v[0], v[1], v[2]};
    const double t2 = dot(w, w);
    const double t = dsqrt(t2);
    double st, ct;
    dsincos(t, st, ct);
    const bool small = t < kTaylorPrec3;
    const double inv_t2 = drcp(t2);
    const double a_wxv = dsel(small, 0.5 - t2 * (1.0 / 24.0), (1.0 - ct) * inv_t2);
    const double a_v = dsel(small, 1.0 - t2 * (1.0 / 6.0), st * drcp(t));
    const double a_w = dsel(small, 1.0 / 6.0 - t2 * (1.0 / 120.0), (1.0 - a_v) * inv_t2);
    const double diag = dsel(small, 1.0 - t2 * 0.5, ct);
    double wxv[3];
    cross(w, vl, wxv);
    const double awv = a_w * dot(w, vl);
    double tr[3], E[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) tr[i] = dfma(a_v, vl[i], dfma(awv, w[i], a_wxv * wxv[i]));
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) E[3 * i + j] = a_wxv * w[i] * w[j];
    E[1] -= a_v * w[2]; E[2] += a_v * w[1];
    E[3] += a_v * w[2]; E[5] -= a_v * w[0];
    E[6] -= a_v * w[1]; E[7] += a_v * w[0];
    E[0] += diag; E[4] += diag; E[8] += diag;
    // M1 = (R1, p) * (E, tr)
    double M[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        out[i] = dfma(R1[3 * i], tr[0], dfma(R1[3 * i + 1], tr[1], dfma(R1[3 * i + 2], tr[2], qb[i])));
#pragma unroll
        for (int j = 0; j < 3; ++j) M[3 * i + j] = dfma(R1[3 * i], E[j], dfma(R1[3 * i + 1], E[3 + j], R1[3 * i + 2] * E[6 + j]));
    }
    // rotation matrix -> quaternion, Eigen's branch selection done with selects
    const double trace = M[0] + M[4] + M[8];
    const bool cw = trace > 0.0;
    int i0 = 0;
    i0 = (M[4] > M[0]) ? 1 : 0;
    const double mii = (i0 == 1) ? M[4] : M[0];
    i0 = (M[8] > mii) ? 2 : i0;
    const bool cx = !cw && i0 == 0, cy = !cw && i0 == 1;
    // t = sqrt(1 + trace)  or  sqrt(1 + M_ii - M_jj - M_kk)
    const double arg = IKD_CHOOSE(cw, trace + 1.0, IKD_CHOOSE(cx, M[0] - M[4] - M[8] + 1.0, IKD_CHOOSE(cy, M[4] - M[8] - M[0] + 1.0, M[8] - M[0] - M[4] + 1.0)));
    const double tq = dsqrt(arg);
    const double half_t = 0.5 * tq;
    const double s = 0.5 * drcp(tq);
    const double d21 = M[7] - M[5], d02 = M[2] - M[6], d10 = M[3] - M[1];  // (R21-R12), (R02-R20), (R10-R01)
    const double s01 = M[3] + M[1], s02 = M[6] + M[2], s12 = M[7] + M[5];
    double rq[4];  // x y z w
    rq[0] = IKD_CHOOSE(cw, d21 * s, IKD_CHOOSE(cx, half_t, IKD_CHOOSE(cy, s01 * s, s02 * s)));
    rq[1] = IKD_CHOOSE(cw, d02 * s, IKD_CHOOSE(cx, s01 * s, IKD_CHOOSE(cy, half_t, s12 * s)));
    rq[2] = IKD_CHOOSE(cw, d10 * s, IKD_CHOOSE(cx, s02 * s, IKD_CHOOSE(cy, s12 * s, half_t)));
    rq[3] = IKD_CHOOSE(cw, half_t, IKD_CHOOSE(cx, d21 * s, IKD_CHOOSE(cy, d02 * s, d10 * s)));
    const double dp = dfma(rq[0], qb[3], dfma(rq[1], qb[4], dfma(rq[2], qb[5], rq[3] * qb[6])));
    const double sg = (dp < 0.0) ? -1.0 : 1.0;
    const double n2 = dfma(rq[0], rq[0], dfma(rq[1], rq[1], dfma(rq[2], rq[2], rq[3] * rq[3])));
    const double al = sg * ((3.0 - n2) * 0.5);
#pragma unroll
    for (int k = 0; k < 4; ++k) out[3 + k] = rq[k] * al;
}

// The step projected onto the null space of ONE frame constraint (reference ik/ik/dls.cpp:26-34,43-53: dq <- N dq,
// N = I - pinv(Jc) Jc; constraint Jacobian ik/ik/frame.hpp:413-449 with the universe as reference frame: the frame's LOCAL
// Jacobian, rows by kinematic type).  The constrained frame ends a chain that carries no task, so before the projection
// dq is zero on that chain; Jc is supported on the six base columns and the chain's NJ columns, and N is the identity on
// every other column.  N = I - V^T V with V an orthonormal basis of the row space of Jc -- any basis of that space gives the
// same projector, so the rows are taken in WORLD coordinates about the frame origin (the local rows are an invertible 3 x 3 /
// block-diagonal mix of them) and orthogonalised by Gram-Schmidt with every projection applied twice.  A row whose remainder
// falls below Eigen's rank threshold (epsilon x rows x the largest row norm, as the complete orthogonal decomposition behind
// the reference's pseudo-inverse) is dropped, per lane, by a select.
template <int NJ, bool FAST, class S = void, class PlPtr, class FrPtr>
IKD_FN void constraint_project(const double (&R1)[9], const double (&p1)[3], PlPtr pl, FrPtr frame_pl, int idmask,
                               const double (&q)[NJ], int type, double (&dqb)[6], double (&dql)[NJ]) {
    constexpr int W = 6 + NJ;
    double zax[NJ][3], org[NJ][3];
    double R[9], p[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = R1[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) p[k] = p1[k];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        if constexpr (std::is_same<S, void>::value) se3_compose_const(R, p, pl[j], (idmask >> j) & 1);
        else se3_compose_struct<S>(R, p, pl[j], j);
        double s, c;
        if constexpr (FAST) dsincos_fast(q[j], s, c);
        else dsincos(q[j], s, c);
        rot_z_right(R, s, c);
        zax[j][0] = R[2]; zax[j][1] = R[5]; zax[j][2] = R[8];
        org[j][0] = p[0]; org[j][1] = p[1]; org[j][2] = p[2];
    }
    if constexpr (std::is_same<S, void>::value) se3_compose_const(R, p, frame_pl, (idmask >> NJ) & 1);
    else se3_compose_struct<S>(R, p, frame_pl, NJ);
    // world rows about the frame origin: rows 0..2 linear velocity, 3..5 angular velocity; columns: base linear 0..2 (body
    // axes R1 e_c), base angular 3..5, chain joints 6..
    double V[6][W];
    const double dp[3] = {p1[0] - p[0], p1[1] - p[1], p1[2] - p[2]};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double rc[3] = {R1[c], R1[3 + c], R1[6 + c]};
        double pxr[3];
        cross(dp, rc, pxr);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            V[i][c] = rc[i];
            V[i][3 + c] = pxr[i];
            V[3 + i][c] = 0.0;
            V[3 + i][3 + c] = rc[i];
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const double dj[3] = {org[j][0] - p[0], org[j][1] - p[1], org[j][2] - p[2]};
        double vw[3];
        cross(dj, zax[j], vw);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            V[i][6 + j] = vw[i];
            V[3 + i][6 + j] = zax[j][i];
        }
    }
    const bool use_lin = type != KT_ORIENTATION, use_ang = type != KT_POSITION;  // wave-uniform
    const int rows = (use_lin ? 3 : 0) + (use_ang ? 3 : 0);
    double maxn2 = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        if (k < 3 ? use_lin : use_ang) {
            double n2 = 0.0;
#pragma unroll
            for (int c = 0; c < W; ++c) n2 = dfma(V[k][c], V[k][c], n2);
            maxn2 = dmax(maxn2, n2);
        }
    }
    const double thr = 2.220446049250313e-16 * static_cast<double>(rows);
    const double thr2 = thr * thr * maxn2;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        if (!(k < 3 ? use_lin : use_ang)) continue;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int i = 0; i < k; ++i) {
                if (!(i < 3 ? use_lin : use_ang)) continue;
                double d = 0.0;
#pragma unroll
                for (int c = 0; c < W; ++c) d = dfma(V[i][c], V[k][c], d);
#pragma unroll
                for (int c = 0; c < W; ++c) V[k][c] = dfma(-d, V[i][c], V[k][c]);
            }
        }
        double n2 = 0.0;
#pragma unroll
        for (int c = 0; c < W; ++c) n2 = dfma(V[k][c], V[k][c], n2);
        const double inv = dsel(n2 > thr2, drsqrt(dmax(n2, 1e-300)), 0.0);
#pragma unroll
        for (int c = 0; c < W; ++c) V[k][c] = V[k][c] * inv;
    }
    // dq <- dq - V^T (V dq) on the base and chain columns
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        if (!(k < 3 ? use_lin : use_ang)) continue;
        double d = 0.0;
#pragma unroll
        for (int c = 0; c < 6; ++c) d = dfma(V[k][c], dqb[c], d);
#pragma unroll
        for (int j = 0; j < NJ; ++j) d = dfma(V[k][6 + j], dql[j], d);
#pragma unroll
        for (int c = 0; c < 6; ++c) dqb[c] = dfma(-d, V[k][c], dqb[c]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) dql[j] = dfma(-d, V[k][6 + j], dql[j]);
    }
}

// One full solve.  The chains share ONE copy of the evaluation / factorisation code (a rolled 2-trip
// loop: the unrolled body of one chain is ~3000 instructions, two inlined copies overflow the
// instruction cache).  Chain 0's factor (L, W, u) is parked while chain 1 is processed -- `park` stores /
// reloads it: LDS [entry][lane] on the device (conflict-free ds_write/read_b64), a plain copy on the host.
// SPEC >= 0: compile-time specialisation (see chain_solver.hpp) -- bits 0..NJ: identity-rotation placement mask shared
// by the chains, bit kSpecUnit: the chain tasks are Full with unit weights, bit kSpecUnitP / kSpecIdP: the base task is
// Full with unit weights / its frame placement is a pure translation.  SPEC = -1: runtime (wave-uniform) values.
// Bit kSpecPost (alone: otherwise a general build): the posture code above is compiled in.
// Bit kSpecCons / kSpecPik (alone): a general build with the FrameConstraint code / with level 1 of a two-level ik::pik.
constexpr int kSpecUnitP = 29, kSpecIdP = 28, kSpecPost = 26, kSpecCons = 25, kSpecPik = 24;
// Bit kSpecGen: the general extras next to a folded placement mask (bits 0..NJ) -- "general" builds without a mask carry none of it
// (SPEC = 0, or the pure posture / constraint / pik flags).
constexpr int kSpecGen = 23;
// Bit kSpecNever (hot builds): the never-stop visitor -- no stop test, no `active` selects (every lane takes every step)
constexpr int kSpecNever = 27;
constexpr int kSpecPostCons = (1 << kSpecPost) | (1 << kSpecCons);   // posture rows next to the constraint (the demo with the stance foot pinned)
constexpr int kSpecExtras = (1 << kSpecGen) | (1 << kSpecPost) | (1 << kSpecCons) | (1 << kSpecPik);
constexpr bool spec_has_posture(int spec) { return spec < 0 || (spec & (1 << kSpecPost)) != 0; }
constexpr bool spec_has_constraint(int spec) { return spec < 0 || (spec & (1 << kSpecCons)) != 0; }
constexpr bool spec_has_pik(int spec) { return spec < 0 || (spec & (1 << kSpecPik)) != 0; }
constexpr bool spec_is_general(int spec) { return spec <= 0 || (spec & kSpecExtras) != 0; }

// Where the lane keeps the joints outside the chains that carry a posture row: its own column of the caller's q_out.
struct PostureState {
    double *q_lane;     // by_row: outside row k at q_lane[k * stride] (an LDS column of the lane); else element i of this
    int64_t stride;     // lane's q at q_lane[i * stride] (the lane's column of the caller's q_out)
    bool by_row;
    const double *lower, *upper;
    bool store;         // false for the tail lanes that shadow the last problem
    // by_row only: the target values of the posture rows, staged in LDS once (outside row k at t_out[k * stride], chain joint a at
    // t_chain[a * stride]) -- read from the caller's buffer every iteration they cost 4x the launch's algorithmic HBM bytes
    const double *t_out, *t_chain;
};

// Posture rows on joints outside the chains (posture builds): each is its own 1x1 system,
//   e = (q - target) mask w,   dq = -w e / (w^2 + lambda^2),   q <- clamp(q + step dq).
// One pass over them, three at a time with the loads issued together (A/B on one box, demo + posture workload: 0.72 ms;
// four at a time 0.76 ms; state in the q_out column instead of LDS 0.75 ms; without the fence below 0.77 ms): `apply` takes the step of the previous iteration
// (deferred to the top of the next one, where next to nothing is live in registers), and the error at the resulting q is
// added to `e0sq` when the rows sit on priority level 0.
IKD_FN void posture_outside_pass(const TreeParams &prm, const PostureState &ps, const LaneRows &targets,
                                 bool apply, double &e0sq) {
    constexpr int kChunk = 3;
    const bool prio0 = prm.post_prio == 0;
#pragma unroll 1
    for (int k0 = 0; k0 < prm.post_n; k0 += kChunk) {
        double qv[kChunk], tv[kChunk];
#pragma unroll
        for (int u = 0; u < kChunk; ++u) {
            const int k = k0 + u < prm.post_n ? k0 + u : k0;  // a short last chunk re-reads its first row; nothing of the repeats is stored or summed
            qv[u] = ps.q_lane[(ps.by_row ? k : prm.post_q[k]) * ps.stride];
            tv[u] = ps.t_out ? ps.t_out[k * ps.stride] : targets(prm.post_slot[k] * 12 + 9);
        }
#pragma unroll
        for (int u = 0; u < kChunk; ++u) {
            const bool live = k0 + u < prm.post_n;
            const int k = live ? k0 + u : k0;
            const int qi = prm.post_q[k];
            const double w = prm.post_w[k], mw = prm.post_m[k] * w;
            const double ea = (qv[u] - tv[u]) * mw;
            const double dq = -(w * ea) * drcp(dfma(w, w, prm.lam2));
            const double qc = dmin(ps.upper[qi], dmax(dfma(prm.step_length, dq, qv[u]), ps.lower[qi]));
            const double qn = apply ? qc : qv[u];
            if (apply && ps.store && live) ps.q_lane[(ps.by_row ? k : qi) * ps.stride] = qn;
            if (prio0 && live) {
                const double en = (qn - tv[u]) * mw;
                e0sq = dfma(en, en, e0sq);
            }
        }
    }
}

// Desc: TreeDesc<NJ, NCH> (LDS / host memory) or IKD_CONST_AS TreeDesc<NJ, NCH> (HBM through scalar loads, see chain_solver.hpp).
// Lane refill hook of tree_dls (the stop-rule mode on batches larger than the machine: tree_kernel_body.hpp TreeRefill): NoRefill is
// the lock-step loop -- every `if constexpr (R::on)` below compiles to nothing there.
struct NoRefill {
    static constexpr bool on = false;
};

template <int NJ, int NCH, int SPEC = -1, class Desc, class Park, class AnyFn, class R = NoRefill>
IKD_FN void tree_dls(const Desc &d_in, const TreeParams &prm, double (&qb)[7], double (&qj0)[NJ], double (&qj1)[NJ],
                     const LaneRows &targets_in, const int (&tslot)[3], const PostureState &ps, int &iters_out,
                     bool &success_out, Park park, AnyFn any_active, R refill = R{}) {
    constexpr bool kGeneral = spec_is_general(SPEC);  // the demo's extras exist in the general builds only
    // builds with the shape's placement mask folded (the hot builds and the general builds next to bit kSpecGen): the launcher takes
    // them only for chains that carry the shape's whole placement-structure code, which is then compile-time too
    typedef typename std::conditional<(SPEC > 0 && TreeHotStruct<NJ>::mask != 0 && (SPEC & ((2 << NJ) - 1)) == TreeHotStruct<NJ>::mask),
                                      typename TreeHotStruct<NJ>::type, void>::type HotS;
    static_assert(!R::on || !(spec_has_posture(SPEC) || spec_has_pik(SPEC)), "lane refill: builds without per-lane state outside q");
    constexpr bool kPik = spec_has_pik(SPEC);  // the orthogonalisation behind PikRow costs the other builds registers
    constexpr bool kPost = spec_has_posture(SPEC);
    constexpr bool kCons = NCH > 1 && spec_has_constraint(SPEC);
    // device constraint builds (SPEC > 0 with bit kSpecCons: launched only for problems with the constraint on): chain 1 carries the
    // constrained frame and no task, so the park buffer is free -- chain 0's factor (77 doubles for NJ = 7) waits there while the base
    // task and the 6 x 6 solve run instead of occupying 154 VGPRs (the posture + constraint build spilled 108 B per lane without it)
    constexpr bool kConsAlways = false && SPEC > 0 && kCons;
    constexpr bool kConsRuntimeOff = false;
    // (posture builds: a tail lane shadowing the last problem would re-read that problem's outside joints while their owner
    // updates them -- it sits the loop out instead; nothing of it is stored anyway)
    bool active = kPost ? ps.store : true, success = false;
    LaneRows targets = targets_in;   // (refill: re-pointed when the lane takes its next problem)
    if constexpr (R::on) active = refill.start;   // (a tail lane of the first round holds no problem)
    const Desc *dp = &d_in;
    int iters = prm.max_iterations;
    // hot builds: the three pose targets stay in registers (the general builds re-read them every iteration -- their register budget
    // is spent; here 36 doubles fit, and the loads' L2 round trips leave the loop: full body 0.810 -> 0.800 ms, same-box A/B)
#ifndef IKGPU_TREE_TG_REGS
#define IKGPU_TREE_TG_REGS 1
#endif
    constexpr bool kTgRegs = IKGPU_TREE_TG_REGS && SPEC > 0 && ((SPEC >> kSpecUnit) & 1) != 0;
    constexpr bool kNever = !R::on && SPEC > 0 && ((SPEC >> kSpecNever) & 1) != 0;
    double tg0[12], tg1[12], tgP[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        tg0[k] = kTgRegs ? targets(tslot[0] * 12 + k) : 0.0;
        tg1[k] = (kTgRegs && NCH > 1) ? targets(tslot[1] * 12 + k) : 0.0;
        tgP[k] = (kTgRegs && prm.hasP) ? targets(tslot[2] * 12 + k) : 0.0;
    }
    int lit = 0;   // (refill) this lane's own iteration count
    if constexpr (R::on) lit = refill.it0();   // (second phase of a two-phase solve: the iterations the first phase took)
#pragma unroll 1
    for (int it = 0; R::on || it < prm.max_iterations; ++it) {
        asm volatile("" ::: "memory");  // re-read the table and the targets every iteration (see chain_solver.hpp)
        if constexpr (!std::is_same<Desc, TreeDesc<NJ, NCH>>::value) IKD_LAUNDER(dp);
        const Desc &d = *dp;
        // R1 = R(quaternion) is recomputed where it is needed (24 instructions) instead of being kept live
        // across the chain bodies: nine doubles less at the register-pressure peak.
        const double p1[3] = {qb[0], qb[1], qb[2]};
        double Hbb[21], gb[6], e0sq = 0.0;
        if (kPost && prm.post_on) {
            posture_outside_pass(prm, ps, targets, it > 0 && active, e0sq);
            IKD_SCHED_FENCE();  // keep the chain bodies' loads out of this pass: they would only lengthen live ranges
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
#pragma unroll
            for (int j = 0; j <= i; ++j) Hbb[tri(i, j)] = (i == j) ? prm.lam2 : 0.0;
            gb[i] = 0.0;
        }
        LegFactor<NJ> F;
        PikRow<NJ> pr;
        pr.on = false;
        // (constraint builds: chain 1 carries the constrained frame and no task -- it is walked after the solve)
        const bool cons_on = kCons && (prm.cons_on != 0 || kConsRuntimeOff);
        const int ntask_chains = cons_on ? 1 : NCH;
#pragma unroll 1
        for (int c = 0; c < ntask_chains; ++c) {
            const auto &ct = d.chain[c];
            double q[NJ], oMt[12], R1[9];
            quat_to_R(qb, R1);
#pragma unroll
            for (int j = 0; j < NJ; ++j) q[j] = (NCH > 1 && c == 1) ? qj1[j] : qj0[j];
#pragma unroll
            for (int k = 0; k < 12; ++k) oMt[k] = (NCH > 1 && c == 1) ? tg1[k] : tg0[k];
            if (!kTgRegs) targets.run(tslot[c] * 12, oMt);
            AlignRow al{false, 0, 0.0, {0.0, 0.0, 0.0}, false, false};
            if (kGeneral) {  // the demo's extras exist in the general builds only (SPEC = 0 [+ posture] on the device, -1 = all runtime
                              // in the emulator); hot builds compile none of this
                if (prm.ref_base[c]) {  // target given in a frame on the floating base: oMt = (oM1 * refpl) * target (frame.hpp:48)
                    double Rr[9], pr[3], tg[12];
#pragma unroll
                    for (int k = 0; k < 9; ++k) Rr[k] = R1[k];
#pragma unroll
                    for (int k = 0; k < 3; ++k) pr[k] = p1[k];
                    se3_compose_const(Rr, pr, d.refpl[c], false);
#pragma unroll
                    for (int k = 0; k < 12; ++k) tg[k] = oMt[k];
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
#pragma unroll
                        for (int j = 0; j < 3; ++j) oMt[3 * i + j] = dfma(Rr[3 * i], tg[j], dfma(Rr[3 * i + 1], tg[3 + j], Rr[3 * i + 2] * tg[6 + j]));
                        oMt[9 + i] = dfma(Rr[3 * i], tg[9], dfma(Rr[3 * i + 1], tg[10], dfma(Rr[3 * i + 2], tg[11], pr[i])));
                    }
                }
                if (prm.align_chain == c) {
                    double td[3];
                    targets.run(prm.align_slot * 12 + 9, td);
                    const double tx = td[0], ty = td[1], tz = td[2];
                    const double inv = drsqrt(dfma(tx, tx, dfma(ty, ty, tz * tz)));
                    al.on = true; al.ax = prm.align_axis; al.w = prm.align_w; al.prio0 = prm.align_prio == 0;
                    al.pik = kPik && prm.pik_on != 0;
                    al.tn[0] = tx * inv; al.tn[1] = ty * inv; al.tn[2] = tz * inv;
                }
            }
            const ChainPosture po{kPost && prm.post_on != 0, prm.post_prio == 0, prm.postc_slot[c], prm.postc_w[c], prm.postc_m[c],
                                  targets, (ps.by_row && c == 0) ? ps.t_chain : nullptr, ps.stride};
            leg_eval_factor<NJ, (SPEC >= 0), kPost, kPik, HotS>(R1, p1, ct.pl, ct.fr, ct.w, SPEC >= 0 ? (SPEC & ((2 << NJ) - 1)) : prm.idmask[c],
                                SPEC >= 0 ? ((SPEC >> kSpecUnit) & 1) != 0 : prm.unit[c] != 0, q, oMt, prm.lam2, prm.prio[c] == 0, al,
                                po, Hbb, gb, e0sq, F, pr);
            if (NCH > 1 && c == 0 && (kConsAlways || ntask_chains > 1)) park.store(F);
        }
        if (prm.hasP) {
            double oMt[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) oMt[k] = tgP[k];
            if (!kTgRegs) targets.run(tslot[2] * 12, oMt);
            double Rf[9], pf[3], R1[9];
            quat_to_R(qb, R1);
#pragma unroll
            for (int k = 0; k < 9; ++k) Rf[k] = R1[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) pf[k] = p1[k];
            se3_compose_const(Rf, pf, d.frP, SPEC >= 0 ? ((SPEC >> kSpecIdP) & 1) != 0 : (prm.idmaskP & 1) != 0);
            TaskTerms t;
            task_terms(Rf, pf, oMt, d.wP, SPEC >= 0 ? ((SPEC >> kSpecUnitP) & 1) != 0 : prm.unitP != 0, t);
            if (prm.prioP == 0) {
#pragma unroll
                for (int r = 0; r < 6; ++r) e0sq = dfma(t.e[r], t.e[r], e0sq);
            }
            double JL[3][3], JA[3][6];
            base_columns(t, pf, R1, p1, JL, JA);
            accumulate_base(JL, JA, t.e, Hbb, gb);
        }
        // constraint builds with posture rows: a joint of the constrained chain carries no task, so its posture row is its own
        // 1 x 1 system (as the joints outside the chains, posture_outside_pass); the projection below then acts on that step.  Here
        // only the rows' share of the level-0 error (the stop test comes first); the steps are formed next to the projection.
        if (kCons && kPost && cons_on && prm.post_on && prm.post_prio == 0) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int slot = prm.postc_slot[NCH - 1][j];
                if (slot >= 0) {   // (wave-uniform)
                    const double ea = (qj1[j] - targets(slot * 12 + 9)) * (prm.postc_m[NCH - 1][j] * prm.postc_w[NCH - 1][j]);
                    e0sq = dfma(ea, ea, e0sq);
                }
            }
        }
        // base: S dq_b = g'
        double S[36], dqb[6];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) S[i * 6 + j] = Hbb[tri(i, j)];
        chol_solve<6>(S, gb, dqb);
        const bool fixed = kGeneral && prm.fixed_base != 0;  // wave-uniform
        if (fixed) {  // the chains of a fixed-base model: independent systems, the base block is solved and dropped
#pragma unroll
            for (int i = 0; i < 6; ++i) dqb[i] = 0.0;
        }

        const bool stop_now = !kNever && active && (prm.stop_sq_tol >= 0.0) && (e0sq < prm.stop_sq_tol);
        if (stop_now) { success = true; iters = R::on ? lit : it; }
        const bool had = active;   // (refill) the lane held a problem during this iteration
        active = active && !stop_now;

#pragma unroll 1
        for (int c = ntask_chains - 1; c >= 0; --c) {
            const auto &ct = d.chain[c];
            if (NCH > 1 && c == 0 && (kConsAlways || ntask_chains > 1)) park.load(F);
            double dql[NJ];
            leg_back_substitute<NJ>(F, dqb, dql);
            if (kPik && prm.pik_on && pr.on && c == prm.align_chain) {   // level 1 of ik::pik, see PikRow (wave-uniform)
                // negated convention: a, b, abar hold -j, so  e_1 - j dq_0 = e + a . dq_chain + b . dq_base_angular  and
                // dq_chain -= (-abar) (...) / (|abar|^2 + lambda_1^2)
                double de = pr.e, n2 = prm.pik_lam2_1;
#pragma unroll
                for (int k = 0; k < 3; ++k) de = dfma(pr.b[k], dqb[3 + k], de);
#pragma unroll
                for (int j = 0; j < NJ; ++j) { de = dfma(pr.a[j], dql[j], de); n2 = dfma(pr.abar[j], pr.abar[j], n2); }
                const double sc = de * drcp(n2);
#pragma unroll
                for (int j = 0; j < NJ; ++j) dql[j] = dfma(pr.abar[j], sc, dql[j]);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const double qold = (NCH > 1 && c == 1) ? qj1[j] : qj0[j];
                const double qc = dmin(ct.hi[j], dmax(dfma(prm.step_length, dql[j], qold), ct.lo[j]));
                const double qn = active ? qc : qold;
                // two separate arrays updated through selects: anything indexed by the runtime chain
                // number `c` would be demoted from registers to scratch memory
                qj1[j] = (NCH > 1 && c == 1) ? qn : qj1[j];
                qj0[j] = (NCH > 1 && c == 1) ? qj0[j] : qn;
            }
        }
        if (cons_on) {
            // the constrained chain: no task moves it, so dq is zero there (or its posture rows' own steps) before the projection
            // N = I - pinv(Jc) Jc, which touches the base columns and this chain's only
            double R1[9], dq1[NJ];
            quat_to_R(qb, R1);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                dq1[j] = 0.0;
                if (kPost && prm.post_on) {
                    const int slot = prm.postc_slot[NCH - 1][j];
                    if (slot >= 0) {   // (wave-uniform)
                        const double w = prm.postc_w[NCH - 1][j], mw = prm.postc_m[NCH - 1][j] * w;
                        const double ea = (qj1[j] - targets(slot * 12 + 9)) * mw;
                        dq1[j] = -(w * ea) * drcp(dfma(w, w, prm.lam2));
                    }
                }
            }
            const auto &cc = d.chain[NCH - 1];
            constraint_project<NJ, (SPEC >= 0), HotS>(R1, p1, cc.pl, cc.fr, prm.idmask[NCH - 1], qj1, prm.cons_type, dqb, dq1);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const double qc = dmin(cc.hi[j], dmax(dfma(prm.step_length, dq1[j], qj1[j]), cc.lo[j]));
                qj1[j] = active ? qc : qj1[j];
            }
        }
        if (!fixed) {
            double v[6], qn[7], R1[9];
            quat_to_R(qb, R1);
#pragma unroll
            for (int i = 0; i < 6; ++i) v[i] = prm.step_length * dqb[i];
            freeflyer_integrate(qb, R1, v, qn);
#pragma unroll
            for (int i = 0; i < 7; ++i) qb[i] = active ? qn[i] : qb[i];
        }
        if constexpr (R::on) {
            // a lane whose visitor fired (q is the configuration the error was evaluated at, dls.cpp:61-63) or whose count reached
            // max_iterations (the stepped q, dls.cpp:76-77) stores its result and takes the next unsolved problem
            ++lit;
            const bool done = had && (stop_now || lit >= prm.max_iterations);
            active = had && !done;   // (a lane that ran out of iterations is as finished as one whose visitor fired)
            refill.took = false;
            const bool any_left = refill.step(done, stop_now, stop_now ? lit - 1 : prm.max_iterations, qb, qj0, qj1, targets, active, [&](const LaneRows &tl) {
#pragma unroll
                for (int k = 0; k < 12; ++k) {
                    tg0[k] = kTgRegs ? tl(tslot[0] * 12 + k) : 0.0;
                    tg1[k] = (kTgRegs && NCH > 1) ? tl(tslot[1] * 12 + k) : 0.0;
                    tgP[k] = (kTgRegs && prm.hasP) ? tl(tslot[2] * 12 + k) : 0.0;
                }
            });
            if (refill.took) { lit = refill.it0(); success = false; }
            if (!any_left) break;
        } else {
            if (!kNever && !any_active(active)) break;
        }
    }
    if (kPost && prm.post_on && prm.max_iterations > 0) {  // the step of the last iteration, for the lanes that never stopped
        double unused = 0.0;
        posture_outside_pass(prm, ps, targets, active, unused);
    }
    iters_out = (R::on || kNever || success) ? iters : iterations_taken(any_active, prm.max_iterations);   // (chain_solver.hpp chain_dls)
    success_out = success;
}

}  // namespace ikdev
