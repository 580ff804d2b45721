// chain_hot.hpp -- the headline lane program: ik::dls() for ONE problem with one Full FrameTask (unit weights, reference
// frame fixed in the world) on a serial chain of NJ <= 7 revolute joints, specialised at compile time on the STRUCTURE of the
// chain's constant placements.
//
// Same algorithm and the same arithmetic as device/chain_solver.hpp (which stays the program of every other chain problem;
// the reference path it restates is cited there: ik/ik/dls.cpp:5-78, data.cpp:25-58, frame.hpp:37-62,152-182,
// common.hpp:53-56, visitor.hpp:15-21).  What differs is only what a lone wave pays for: measured on gfx950
// (tools/issue_probe.hip, profiles/r02_issue_probe.csv) a wave that has its SIMD to itself -- the situation at the metric's
// batch, 65536 problems = 1024 waves on 1024 SIMDs -- issues ONE instruction of any kind every 4 cycles (VALU, SALU, s_nop,
// s_waitcnt alike; 16 for an FP64 transcendental, 8 for v_mov_b64) and a dependent FP64 instruction can follow its producer
// in the next slot.  The iteration therefore costs 4 cycles x (number of instructions) + the scalar-load waits, and nothing
// else: instruction-level parallelism buys nothing, every instruction removed buys 4 cycles.  Hence:
//
//  * Placement structure as a template argument.  Joint placements in URDFs are mostly axis permutations (rpy multiples of
//    pi/2) and translations along one or two axes.  The host classifies every entry of every placement of the chain --
//    rotation entries: exactly 0, exactly +1, exactly -1, or general; translation: which components are exactly zero --
//    into a 21-bit code (ChainStruct).  For a structural entry the lane program uses the LITERAL 0.0 / +-1.0, and the translation
//    unit is compiled with -fno-signed-zeros -fno-honor-nans -fno-honor-infinities (no reassociation, no reciprocal maths:
//    results stay bit-identical on finite data) so that x * 0.0, x * 1.0, x + 0.0 fold away: a near-permutation placement costs
//    6-15 instructions instead of 36.  No robot constant is compiled in: the non-structural values (22 doubles for a
//    Cassie leg instead of 96) arrive in the kernel-argument segment and stay in registers for the whole loop -- no table
//    loads, no s_waitcnt inside the iteration.  A model whose code has no instantiation runs on chain_solver.hpp.
//  * The visitor that never stops (the metric's fixed-iteration mode) is its own instantiation: no `active` selects, no
//    stop-test arithmetic.
#pragma once
#if !defined(__HIPCC_RTC__)
#include <cstdint>
#endif

#include "chain_kernel_body.hpp"

namespace ikdev {

// 21 bits per placement i = 0 .. NJ (i = NJ: last joint frame -> task frame), three placements per 64-bit word:
//   bits 0-17   nine 2-bit classes of the rotation entries, row-major: 0 general (a table value), 1 exactly 0, 2 exactly +1,
//               3 exactly -1   (URDF rpy values are decimal approximations of pi/2, so a "permutation" placement usually
//               keeps a few entries of 1e-12 .. 1e-16: those stay general and exact -- nothing is snapped)
//   bits 18-20  translation component k is non-zero
constexpr int kStructBits = 21;
constexpr int kStructPerWord = 3;
enum : unsigned { kEntGeneral = 0, kEntZero = 1, kEntOne = 2, kEntMinusOne = 3 };

template <uint64_t C0, uint64_t C1, uint64_t C2>
struct ChainStruct {
    static constexpr uint64_t word(int w) { return w == 0 ? C0 : (w == 1 ? C1 : C2); }
    static constexpr unsigned code(int i) { return static_cast<unsigned>((word(i / kStructPerWord) >> (kStructBits * (i % kStructPerWord))) & 0x1fffffu); }
    static constexpr unsigned ent(int i, int e) { return (code(i) >> (2 * e)) & 3u; }           // e = 3 m + k
    static constexpr bool tnz(int i, int k) { return ((code(i) >> (18 + k)) & 1u) != 0; }
    static constexpr int ngen_before(int i, int e) { return e == 0 ? 0 : ngen_before(i, e - 1) + (ent(i, e - 1) == kEntGeneral ? 1 : 0); }
    static constexpr int count(int i) { return ngen_before(i, 9) + (tnz(i, 0) ? 1 : 0) + (tnz(i, 1) ? 1 : 0) + (tnz(i, 2) ? 1 : 0); }
    // position of placement i's values in the compact table: [general rotation entries, row-major][non-zero translation entries]
    static constexpr int offset(int i) { return i == 0 ? 0 : offset(i - 1) + count(i - 1); }
    static constexpr int rot_at(int i, int e) { return offset(i) + ngen_before(i, e); }
    static constexpr int trans_at(int i, int k) {
        return offset(i) + ngen_before(i, 9) + (k > 0 && tnz(i, 0) ? 1 : 0) + (k > 1 && tnz(i, 1) ? 1 : 0);
    }
    static constexpr double literal(int i, int e) { return ent(i, e) == kEntOne ? 1.0 : (ent(i, e) == kEntMinusOne ? -1.0 : 0.0); }
    // Joint j (>= 1) turns about the SAME world axis as joint j - 1 when its placement has an exactly-identity rotation (R * Rz leaves
    // the third column of R alone).  leader(j): the first joint of j's run of parallel axes; members(L): how many joints L leads.
    static constexpr bool identity_rotation(int i) {
        return ent(i, 0) == kEntOne && ent(i, 4) == kEntOne && ent(i, 8) == kEntOne && ent(i, 1) == kEntZero && ent(i, 2) == kEntZero &&
               ent(i, 3) == kEntZero && ent(i, 5) == kEntZero && ent(i, 6) == kEntZero && ent(i, 7) == kEntZero;
    }
    static constexpr int leader(int j) { return (j > 0 && identity_rotation(j)) ? leader(j - 1) : j; }
    static constexpr int members(int L, int nj) { return nj <= 0 ? 0 : members(L, nj - 1) + (leader(nj - 1) == L ? 1 : 0); }
};

// leader / members as compile-time tables (indexed by the unrolled joint loops: a constant index into a constant array folds;
// the recursive constexpr functions above, called with a loop variable, would be emitted as real calls)
template <class S, int NJ>
struct ChainRuns {
    struct Table { int leader[8], members[8]; };
    static constexpr Table make() {
        Table t{};
        for (int j = 0; j < 8; ++j) { t.leader[j] = j < NJ ? S::leader(j) : j; t.members[j] = 0; }
        for (int j = 0; j < NJ; ++j) ++t.members[t.leader[j]];
        return t;
    }
    static constexpr Table value = make();
};

constexpr int kHotTableMax = 112;  // doubles in the kernel-argument copy of the compact table (values, then lo[NJ], hi[NJ]): 8 x 12 + 2 x 7 when nothing is structural

struct HotTable {
    double v[kHotTableMax];
};

// entry e = 3 m + k of placement I's rotation: the literal 0.0 / +-1.0 when structural, else its table value
template <class S, int I, class Tab>
IKD_FN double hot_rot(const Tab &t, int e) {
    return S::ent(I, e) == kEntGeneral ? t.v[S::rot_at(I, e)] : S::literal(I, e);
}

// (R, p) <- (R, p) * placement I.  Written as the general product; with literal zeros and ones in it the compiler folds the
// multiplications by 1 and (under -fno-signed-zeros -fno-honor-nans) by 0 away: same bits as the full product on finite data.
template <class S, int I, class Tab>
IKD_FN void hot_compose(double (&R)[9], double (&p)[3], const Tab &t) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (S::tnz(I, k)) {
            const double tk = t.v[S::trans_at(I, k)];
#pragma unroll
            for (int r = 0; r < 3; ++r) p[r] = dfma(R[3 * r + k], tk, p[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const double a = R[3 * r], b = R[3 * r + 1], d = R[3 * r + 2];
#pragma unroll
        for (int k = 0; k < 3; ++k)
            R[3 * r + k] = dfma(a, hot_rot<S, I>(t, k), dfma(b, hot_rot<S, I>(t, 3 + k), d * hot_rot<S, I>(t, 6 + k)));
    }
}

// e (6) and the NEGATED task Jacobian columns at q (see chain_evaluate in chain_solver.hpp: same expressions, KT_FULL,
// unit weights).  oMt: target placement in the world.
template <int NJ, class S, class Tab>
IKD_FN void hot_evaluate(const Tab &t, const double (&q)[NJ], const double (&oMt)[12], double (&e)[6], double (&col)[NJ][6]) {
    double zax[NJ][3], org[NJ][3];
    double R[9], p[3];
    // placement 0: world -> joint-0 frame
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = hot_rot<S, 0>(t, k);
#pragma unroll
    for (int k = 0; k < 3; ++k) p[k] = S::tnz(0, k) ? t.v[S::trans_at(0, k)] : 0.0;

    // the NJ sin / cos first: independent of the chain, the constants are live once
    double sn[NJ], cs[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) dsincos_hot(q[j], sn[j], cs[j]);

#define IKD_HOT_JOINT(J)                                                   \
    if (J < NJ) {                                                          \
        if (J > 0) hot_compose<S, (J < NJ ? J : 0)>(R, p, t);              \
        rot_z_right(R, sn[J < NJ ? J : 0], cs[J < NJ ? J : 0]);            \
        zax[J < NJ ? J : 0][0] = R[2]; zax[J < NJ ? J : 0][1] = R[5]; zax[J < NJ ? J : 0][2] = R[8]; \
        org[J < NJ ? J : 0][0] = p[0]; org[J < NJ ? J : 0][1] = p[1]; org[J < NJ ? J : 0][2] = p[2]; \
    }
    IKD_HOT_JOINT(0) IKD_HOT_JOINT(1) IKD_HOT_JOINT(2) IKD_HOT_JOINT(3) IKD_HOT_JOINT(4) IKD_HOT_JOINT(5) IKD_HOT_JOINT(6)
#undef IKD_HOT_JOINT
    hot_compose<S, NJ>(R, p, t);

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
    double Cm[9];
    log6_and_jlog6_hot<false>(Re, pe, lj, &Cm);
#pragma unroll
    for (int i = 0; i < 6; ++i) e[i] = lj.e[i];

    // K' = Jlog6(tMf) with the frame rotation folded in: top rows [A Rf^T | Bm Rf^T], bottom rows [0 | A Rf^T]
    double AR[9], BR[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            AR[3 * i + k] = dfma(lj.A[3 * i], R[3 * k], dfma(lj.A[3 * i + 1], R[3 * k + 1], lj.A[3 * i + 2] * R[3 * k + 2]));
        }
    // (C A) Rf^T = C (A Rf^T): the product C A itself is never formed
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k)
            BR[3 * i + k] = dfma(Cm[3 * i], AR[k], dfma(Cm[3 * i + 1], AR[3 + k], Cm[3 * i + 2] * AR[6 + k]));
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const double dj[3] = {org[j][0] - p[0], org[j][1] - p[1], org[j][2] - p[2]};
        double vw[3];
        cross(dj, zax[j], vw);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            // the z-axis terms first: consecutive joints with parallel axes (identity-rotation placements) share them
            const double bz = dfma(BR[3 * i], zax[j][0], dfma(BR[3 * i + 1], zax[j][1], BR[3 * i + 2] * zax[j][2]));
            col[j][i] = dfma(AR[3 * i], vw[0], dfma(AR[3 * i + 1], vw[1], dfma(AR[3 * i + 2], vw[2], bz)));
            col[j][3 + i] = dfma(AR[3 * i], zax[j][0], dfma(AR[3 * i + 1], zax[j][1], AR[3 * i + 2] * zax[j][2]));
        }
    }
}

// Gram matrix G = J J^T + lam2 I (lower triangle) from the negated task Jacobian columns.
template <int NJ, class S>
IKD_FN void hot_gram(const double (&col)[NJ][6], double lam2, double (&G)[36]) {
    constexpr int M = 6;
    constexpr typename ChainRuns<S, NJ>::Table kRuns = ChainRuns<S, NJ>::value;
            // Gram matrix.  Joints of one run of parallel axes (S::leader) share their bottom (angular) rows: col[j][3..5] = A Rf^T z_j is
            // the same vector for all of them, so  sum_j col[j][3+a] col[j][3+b] = n c_a c_b  and  sum_j col[j][3+a] col[j][b] =
            // c_a (sum_j col[j][b])  within a run -- 102 instead of 147 multiply-adds for a Cassie leg (runs of 1, 1 and 5 joints).
                    double top_sum[NJ][3], nbot[NJ][3];
    #pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (kRuns.leader[j] == j) {
    #pragma unroll
                    for (int b = 0; b < 3; ++b) {
                        top_sum[j][b] = col[j][b];
                        nbot[j][b] = kRuns.members[j] > 1 ? static_cast<double>(kRuns.members[j]) * col[j][3 + b] : col[j][3 + b];
                    }
                } else {
    #pragma unroll
                    for (int b = 0; b < 3; ++b) top_sum[kRuns.leader[j]][b] += col[j][b];
                }
            }
    #pragma unroll
            for (int a = 0; a < 3; ++a)
    #pragma unroll
                for (int b = 0; b <= a; ++b) {
                    double s = (a == b) ? lam2 : 0.0;
    #pragma unroll
                    for (int j = 0; j < NJ; ++j) s = dfma(col[j][a], col[j][b], s);
                    G[a * M + b] = s;
                }
    #pragma unroll
            for (int a = 3; a < 6; ++a) {
    #pragma unroll
                for (int b = 0; b < 3; ++b) {
                    double s = 0.0;
    #pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        if (kRuns.leader[j] == j) s = dfma(col[j][a], top_sum[j][b], s);
                    G[a * M + b] = s;
                }
    #pragma unroll
                for (int b = 3; b <= a; ++b) {
                    double s = (a == b) ? lam2 : 0.0;
    #pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        if (kRuns.leader[j] == j) s = dfma(nbot[j][a - 3], col[j][b], s);
                    G[a * M + b] = s;
                }
            }
}

// q <- clip(q + step_length dq), dq = -J^T y = +col^T y (reference ik/ik/dls.cpp:52-53,67-71) where `take`; q stays elsewhere.
template <int NJ, class S, class Tab>
IKD_FN void hot_step(const Tab &t, const LoopParams &prm, const double (&col)[NJ][6], const double (&y)[6], double (&q)[NJ], bool take) {
    constexpr int kLim = S::offset(NJ + 1);  // lo[NJ], hi[NJ] follow the placement values
    constexpr typename ChainRuns<S, NJ>::Table kRuns = ChainRuns<S, NJ>::value;
    double ang[NJ];   // the angular part of col_j^T y, shared by a run of parallel axes
#pragma unroll
    for (int j = 0; j < NJ; ++j)
        if (kRuns.leader[j] == j) ang[j] = dfma(col[j][3], y[3], dfma(col[j][4], y[4], col[j][5] * y[5]));
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        double s = ang[kRuns.leader[j]];
#pragma unroll
        for (int a = 0; a < 3; ++a) s = dfma(col[j][a], y[a], s);
        const double qn = dfma(prm.step_length, s, q[j]);  // dq_j = -J_task(:, j)^T y = +col_j^T y
        const double qc = dmin(t.v[kLim + NJ + j], dmax(qn, t.v[kLim + j]));
        q[j] = take ? qc : q[j];
    }
}

// One full solve.  q: in = q0 (chain joints), out = result.  NEVERSTOP: the visitor never stops (stop_sq_tol < 0): every
// lane takes exactly max_iterations steps.
template <int NJ, class S, bool NEVERSTOP, class Tab, class AnyFn>
IKD_FN void hot_dls(const Tab &t, const LoopParams &prm, double (&q)[NJ], const double (&oMt)[12], int &iters_out,
                    bool &success_out, AnyFn any_active) {
    constexpr int M = 6;
    bool active = true;
    bool success = false;
    int iters = prm.max_iterations;
#pragma unroll 1
    for (int it = 0; it < prm.max_iterations; ++it) {
        double e[M], col[NJ][M];
        hot_evaluate<NJ, S>(t, q, oMt, e, col);

        double G[M * M];
        hot_gram<NJ, S>(col, prm.lam2, G);
        double y[M];
        chol_solve<M>(G, e, y);

        if (!NEVERSTOP) {
            double e0sq = 0.0;
            if (prm.priority == 0) {
#pragma unroll
                for (int a = 0; a < M; ++a) e0sq = dfma(e[a], e[a], e0sq);
            }
            const bool stop_now = active && (prm.stop_sq_tol >= 0.0) && (e0sq < prm.stop_sq_tol);
            if (stop_now) { success = true; iters = it; }
            active = active && !stop_now;
        }
        hot_step<NJ, S>(t, prm, col, y, q, NEVERSTOP || active);
        if (!NEVERSTOP && !any_active(active)) break;
    }
    iters_out = (NEVERSTOP || success) ? iters : iterations_taken(any_active, prm.max_iterations);   // (chain_solver.hpp chain_dls)
    success_out = success;
}

// Entries of q outside the task support: dq = 0 there, so the loop only ever clips them to the limits (reference
// ik/ik/dls.cpp:71 clips the whole q after each step; no step is taken when the solve stops at iteration 0).  Eight at a time,
// every load of a group issued before its first store: one entry per pass (load, wait, store; the next load cannot move above a
// store that might alias it) cost one HBM round trip per entry -- nine for a Cassie leg, ~10 us of a 150 us launch.
template <int NJ>
IKD_FN void hot_pass_through(const ChainKernelArgs<NJ> &a, int64_t b, bool stepped) {
    constexpr int kGroup = 16;   // (a Cassie model's sixteen entries in ONE pass: with groups of eight the second group's loads waited
                                 // for the first group's stores -- a second HBM round trip in the prologue)
    for (int i0 = 0; i0 < a.nq; i0 += kGroup) {
        double v[kGroup], lo[kGroup], hi[kGroup];
        bool out[kGroup];
#pragma unroll
        for (int k = 0; k < kGroup; ++k) {
            const int i = i0 + k < a.nq ? i0 + k : a.nq - 1;
            out[k] = i0 + k < a.nq && !a.q_in_chain[i];
            v[k] = a.q0[at(a.layout, a.B, a.nq, i, b)];
            lo[k] = a.lower[i];
            hi[k] = a.upper[i];
        }
#pragma unroll
        for (int k = 0; k < kGroup; ++k) {
            const double c = dmin(hi[k], dmax(v[k], lo[k]));
            if (out[k]) a.q_out[at(a.layout, a.B, a.nq, i0 + k, b)] = stepped ? c : v[k];
        }
    }
}

#if IKD_HIP_LANG
// The hot program under lane refill (chain_kernel_body.hpp chain_refill_loop): the stop-rule mode for batches larger than the machine.
template <int NJ, class S, class Tab>
__device__ __forceinline__ void hot_refill_body(const ChainKernelArgs<NJ> &a, const Tab &t, unsigned long long *queue, int chunk) {
    chain_refill_loop<NJ>(a, queue, chunk, [&](double (&q)[NJ], const double (&oMt)[12], bool have) {
        constexpr int M = 6;
        double e[M], col[NJ][M];
        hot_evaluate<NJ, S>(t, q, oMt, e, col);
        double G[M * M];
        hot_gram<NJ, S>(col, a.prm.lam2, G);
        double y[M];
        chol_solve<M>(G, e, y);
        double e0sq = 0.0;
        if (a.prm.priority == 0) {
#pragma unroll
            for (int k = 0; k < M; ++k) e0sq = dfma(e[k], e[k], e0sq);
        }
        const bool stop_now = have && (a.prm.stop_sq_tol >= 0.0) && (e0sq < a.prm.stop_sq_tol);
        hot_step<NJ, S>(t, a.prm, col, y, q, !stop_now);
        return stop_now;
    });
}
#endif

// B independent ik::dls() calls, lane `gid`: load, solve, store -- dls_chain_body (chain_kernel_body.hpp) with the hot program.
template <int NJ, class S, bool NEVERSTOP, class Tab, class AnyFn>
IKD_FN void hot_chain_body(const ChainKernelArgs<NJ> &a, const Tab &t, int64_t gid, AnyFn any_active) {
#if defined(IKGPU_HOT_STAMP) && IKD_ON_DEVICE
    const long long rs = wall_clock64();   // wave start, 100 MHz ticks
#endif
    const bool valid = gid < a.B;
    const int64_t b = valid ? gid : a.B - 1;  // tail lanes shadow the last problem and store nothing
    double q[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) q[j] = a.q0[at(a.layout, a.B, a.nq, a.qidx[j], b)];
    double oMt[12];
    load_target(a, b, oMt);
    // The visitor never stops: every lane takes max_iterations steps, so what happens to the entries outside the support is
    // known before the loop -- copy them now (the stores drain while the loop runs) instead of after it (4 us of epilogue).
    if (NEVERSTOP && valid) hot_pass_through(a, b, a.prm.max_iterations > 0);
    int iters;
    bool success;
#if defined(IKGPU_HOT_STAMP) && IKD_ON_DEVICE
    // measurement build (tools/build_variant.sh stamp -DIKGPU_HOT_STAMP): lanes 0 / 1 of every wave return the iteration loop's
    // duration in shader clocks (s_memtime) and in 100 MHz ticks (s_memrealtime) through `iters`
    const long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
#endif
    hot_dls<NJ, S, NEVERSTOP>(t, a.prm, q, oMt, iters, success, any_active);
#if defined(IKGPU_HOT_STAMP) && IKD_ON_DEVICE
    const long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    const int stamp_loop_cycles = static_cast<int>(c1 - c0), stamp_loop_ticks = static_cast<int>(r1 - r0);
    const int stamp_pre_ticks = static_cast<int>(r0 - rs);
    const int stamp_true_iters = iters;
#endif
    if (!NEVERSTOP && a.append_count)   // (first phase of a two-phase solve)
        append_unfinished(a.append_list, a.append_count, valid && !success && iters < a.prm.max_iterations, b);
    if (!valid) return;
#pragma unroll
    for (int j = 0; j < NJ; ++j) a.q_out[at(a.layout, a.B, a.nq, a.qidx[j], b)] = q[j];
    if (!NEVERSTOP) hot_pass_through(a, b, iters > 0);
    if (a.success) a.success[b] = success ? 1 : 0;
#if defined(IKGPU_HOT_STAMP) && IKD_ON_DEVICE
    // lanes 0..5 of every wave: loop cycles, loop ticks, prologue ticks (wave start -> loop), epilogue ticks (loop end -> here,
    // the stores issued), absolute start and end ticks (low 31 bits)
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long re = wall_clock64();
        const int lane = threadIdx.x % 64;
        iters = lane == 0 ? stamp_loop_cycles : lane == 1 ? stamp_loop_ticks : lane == 2 ? stamp_pre_ticks
              : lane == 3 ? static_cast<int>(re - r1) : lane == 4 ? static_cast<int>(rs & 0x7fffffff)
              : lane == 5 ? static_cast<int>(re & 0x7fffffff) : stamp_true_iters;
    }
#endif
    if (a.iters) a.iters[b] = iters;
}

#if IKD_HIP_LANG
// ---- kernel entries, shared by the instantiations compiled into the library (kernels_hot.hip) and the ones compiled at run time for
// a chain's own structure code (rtc.cpp) -----------------------------------------------------------------------------------------
#ifndef IKGPU_HOT_PIN_LO
// the placement values are parked in vector registers, the joint limits stay in scalar registers (A/B on one box, B = 65536:
// everything in VGPRs 0.1440 ms -- 22 v_accvgpr_read per iteration --, limits in SGPRs 0.1417 ms)
#define IKGPU_HOT_PIN_LO 0
#define IKGPU_HOT_PIN_HI (S::offset(NJ + 1))
#endif

// The compact table (<= 36 doubles for the fixture shapes) arrives in the kernel-argument segment and is parked in vector
// registers for the whole loop: in scalar registers it competes with the ~40 polynomial constants for the 100 SGPRs (31 v_readlane
// + 39 s_mov of spill code per iteration in the round-1 kernel); a lone wave has 512 VGPRs to itself.
template <int NJ, class S>
__device__ __forceinline__ void hot_park_table(const HotTable &t, HotTable &tv) {
    constexpr int kUsed = S::offset(NJ + 1) + 2 * NJ;
    static_assert(kUsed <= kHotTableMax, "compact table too long");
#pragma unroll
    for (int k = 0; k < kHotTableMax; ++k) {
        tv.v[k] = k < kUsed ? t.v[k] : 0.0;
        if (k >= IKGPU_HOT_PIN_LO && k < kUsed && k < IKGPU_HOT_PIN_HI) IKD_PIN(tv.v[k]);
    }
}

template <int NJ, class S, bool NEVERSTOP>
__device__ __forceinline__ void hot_kernel_entry(const ChainKernelArgs<NJ> &a, const HotTable &t) {
    const int64_t gid = static_cast<int64_t>(blockIdx.x) * 64 + threadIdx.x;   // one wave64 per workgroup
    HotTable tv;
    hot_park_table<NJ, S>(t, tv);
    hot_chain_body<NJ, S, NEVERSTOP>(a, tv, gid, KeepGoing{a.leave_active, a.leave_after, 0});
}

template <int NJ, class S>
__device__ __forceinline__ void hot_refill_entry(const ChainKernelArgs<NJ> &a, const HotTable &t, unsigned long long *queue, int chunk) {
    HotTable tv;
    hot_park_table<NJ, S>(t, tv);
    hot_refill_body<NJ, S>(a, tv, queue, chunk);
}
#endif

}  // namespace ikdev
