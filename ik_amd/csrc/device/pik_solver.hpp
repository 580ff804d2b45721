// pik_solver.hpp -- ik::pik(), the reference's prioritised IK (reference ik/ik/pik.cpp:31-103), one problem per lane
// on the generic lane program's memory-resident workspace (generic_solver.hpp).
//
// Per iteration the reference walks the priority levels with a projector P (P = I, dq = 0):
//     de   = e_l - J_l dq                                   pik.cpp:49
//     Jbar = J_l P                                          pik.cpp:51
//     dq  -= damp_pseudoinverse(Jbar, lambda_l) de          pik.cpp:54-55   (JacobiSVD: sum sigma/(lambda^2+sigma^2) v u^T)
//     P   -= pinv(Jbar) Jbar                                pik.cpp:58-61   (CompleteOrthogonalDecomposition)
// and ends with dq += P da (pik.cpp:65), the stop test on e[0] and integrate + clip, as ik::dls does.
//
// Both matrix functions come out of ONE one-sided Jacobi pass on the rows of [Jbar | de]: plane rotations from the
// left make the rows of Jbar mutually orthogonal, U^T Jbar = diag(sigma) V^T, so row i becomes sigma_i v_i^T and the
// appended column becomes U^T de.  Then
//     damp_pseudoinverse(Jbar) de = sum_i row_i (U^T de)_i / (lambda^2 + sigma_i^2)         -- no division by sigma
//     pinv(Jbar) Jbar             = sum_{i < rank} row_i row_i^T / sigma_i^2
// with the numerical rank counted as Eigen's COD counts it on its pivots, relative to the largest one:
// sigma_i > eps * min(rows, cols) * sigma_max.  (Pivot magnitudes and singular values differ by modest factors, so the two
// counts agree except for matrices within such a factor of the threshold -- where the reference's own result is noise.)
// A Jacobi SVD is what the reference asks Eigen for, it needs no pivoting (no per-lane data-dependent addressing), and every
// lane of a wave runs the same sweeps: a lane whose rows are already orthogonal applies identity rotations.
#pragma once
#include "generic_solver.hpp"

namespace ikdev {

constexpr int kMaxPikLevels = 8;    // == IKGPU_MAX_PIK_LEVELS
constexpr int kMaxPikDa = 128;      // == IKGPU_MAX_PIK_DA: longest da carried by value in the kernel arguments

struct PikParams {
    int max_iterations;
    double step_length, stop_sq_tol;
    double lam2[kMaxPikLevels];  // lambda_l^2
    int has_da;
    double da[kMaxPikDa];
};

// One priority level on the workspace: rows [r0, r0 + ml) of (e, J).  Updates dq and, when asked, P.
template <class AnyFn>
IKD_FN void pik_level(const GenericTables &T, const Ws &ws, int r0, int ml, double lam2, bool update_P, AnyFn any_lane) {
    const int nv = T.nv;
    const double eps = 2.220446049250313e-16;
    for (int r = 0; r < ml; ++r) {
        double s = ws[T.off_e + r0 + r];
        for (int c = 0; c < nv; ++c) s = dfma(-ws[T.off_J + (r0 + r) * nv + c], ws[T.off_dq + c], s);
        ws[T.off_de + r] = s;
        for (int c = 0; c < nv; ++c) {
            double a = 0.0;
            for (int k = 0; k < nv; ++k) a = dfma(ws[T.off_J + (r0 + r) * nv + k], ws[T.off_P + k * nv + c], a);
            ws[T.off_Jb + r * nv + c] = a;
        }
    }
    jacobi_rows(ws, T.off_Jb, T.off_de, ml, nv, any_lane);  // rows of [Jb | de]
    // sigma_i^2 = |row_i|^2, recomputed where it is used (the rows are short)
    double smax2 = 0.0;
    for (int i = 0; i < ml; ++i) {
        double s2 = 0.0;
        for (int c = 0; c < nv; ++c) { const double x = ws[T.off_Jb + i * nv + c]; s2 = dfma(x, x, s2); }
        smax2 = dmax(smax2, s2);
    }
    const double kk = eps * static_cast<double>(ml < nv ? ml : nv);
    const double thr2 = kk * kk * smax2;
    for (int i = 0; i < ml; ++i) {
        double s2 = 0.0;
        for (int c = 0; c < nv; ++c) { const double x = ws[T.off_Jb + i * nv + c]; s2 = dfma(x, x, s2); }
        const double f = ws[T.off_de + i] / (lam2 + s2);
        for (int c = 0; c < nv; ++c) ws[T.off_dq + c] = dfma(-ws[T.off_Jb + i * nv + c], f, ws[T.off_dq + c]);
        if (update_P) {
            const double inv = dsel(s2 > thr2, 1.0 / s2, 0.0);
            for (int a = 0; a < nv; ++a) {
                const double ra = ws[T.off_Jb + i * nv + a] * inv;
                for (int b = 0; b < nv; ++b) ws[T.off_P + a * nv + b] = dfma(-ra, ws[T.off_Jb + i * nv + b], ws[T.off_P + a * nv + b]);
            }
        }
    }
}

template <class AnyFn>
IKD_FN void generic_pik(const GenericTables &T, const PikParams &prm, const Ws &ws, const double *targets_lane, int64_t tstride,
                        int &iters_out, bool &success_out, AnyFn any_lane) {
    bool active = true, success = false;
    int iters = prm.max_iterations;
    const int nv = T.nv;
    int last_level = 0;
    for (int l = 0; l < T.nlevels; ++l)
        if (T.lvl_row0[l + 1] > T.lvl_row0[l]) last_level = l;
    for (int it = 0; it < prm.max_iterations; ++it) {
        const double e0sq = generic_evaluate(T, ws, targets_lane, tstride);   // pik.cpp:41
        for (int a = 0; a < nv; ++a) {                                        // pik.cpp:44-45
            for (int b = 0; b < nv; ++b) ws[T.off_P + a * nv + b] = (a == b) ? 1.0 : 0.0;
            ws[T.off_dq + a] = 0.0;
        }
        for (int l = 0; l < T.nlevels; ++l) {                                 // pik.cpp:47
            const int r0 = T.lvl_row0[l], ml = T.lvl_row0[l + 1] - r0;
            if (ml == 0) continue;
            // the projector left by the last level is only read by `P da` below
            pik_level(T, ws, r0, ml, prm.lam2[l], l != last_level || prm.has_da != 0, any_lane);
        }
        if (prm.has_da)                                                       // pik.cpp:65
            for (int c = 0; c < nv; ++c) {
                double s = ws[T.off_dq + c];
                for (int k = 0; k < nv; ++k) s = dfma(ws[T.off_P + c * nv + k], prm.da[k], s);
                ws[T.off_dq + c] = s;
            }
        const bool stop_now = active && (prm.stop_sq_tol >= 0.0) && (e0sq < prm.stop_sq_tol);  // pik.cpp:67-70
        if (stop_now) { success = true; iters = it; }
        active = active && !stop_now;
        generic_integrate_clip(T, ws, prm.step_length, active);               // pik.cpp:73-77
        if (!any_lane(active)) break;
    }
    iters_out = iters;
    success_out = success;
}

#ifdef IKD_STATIC_TABLES
// ---- ik::pik as a compiled lane program (rtc.cpp generic_static_source, kernel ikgpu_lane_pik) -----------------------------------
// The level loop of reference ik/ik/pik.cpp:47-61 with everything a register program needs known at compile time: TB carries the
// rows of every level (lvl_row0), so every loop below unrolls and the arrays are registers.
//   * the projector is kept FACTORED, P = I - V^T V, V an orthonormal basis (rows) of the row space of the levels done so far
//     (pik.cpp:58-61: P -= pinv(Jbar) Jbar -- pinv(Jbar) Jbar is the orthogonal projector onto rowspace(Jbar), and rowspace(Jbar) is
//     orthogonal to the earlier levels' because Jbar = J_l P): Jbar = J_l - (J_l V^T) V costs rows x |V| x nv instead of rows x nv x nv,
//     and V has the structural zeros of the Jacobians it came from;
//   * the damped pseudo-inverse of pik.cpp:5-21, sum_i sigma_i / (sigma_i^2 + lambda^2) v_i u_i^T, is Jbar^T (Jbar Jbar^T +
//     lambda^2 I)^-1 for lambda > 0: the m_l x m_l dual solve of the DLS program (unrolled Cholesky) -- no SVD, no sweep count that
//     depends on the data (lambda = 0 stays on the one-sided-Jacobi interpreter above);
//   * rank decisions as the complete orthogonal decomposition behind the reference's pinv makes them: pivoted Gram-Schmidt (largest
//     remaining row first, every projection applied twice), a pivot below epsilon x min(rows, cols) x the first one ends the basis;
//     the rows beyond the rank enter V as zero rows (selects, no branch).
// HAS_DA: the secondary step dq += P da (pik.cpp:65) -- then the last level's rows enter V too.
template <bool HAS_DA, class TB, class WS, class AnyFn>
IKD_FN void static_pik(const TB &T, const PikParams &prm, const WS &ws, const LaneRows &targets, int &iters_out, bool &success_out,
                       AnyFn any_lane) {
    constexpr int NV = TB::nv, NL = TB::nlevels, MM = TB::pik_max_rows;
    constexpr int VR = HAS_DA ? TB::M : TB::pik_basis_rows;   // rows V can hold: every level's but the last one's (all with da)
    bool active = true, success = false;
    int iters = prm.max_iterations;
    for (int it = 0; it < prm.max_iterations; ++it) {
        const double e0sq = generic_evaluate(T, ws, targets);                 // pik.cpp:41
        double dq[NV], V[VR > 0 ? VR : 1][NV];
        IKD_UNROLL
        for (int c = 0; c < NV; ++c) dq[c] = 0.0;                              // pik.cpp:44-45
        IKD_UNROLL
        for (int l = 0; l < NL; ++l) {                                         // pik.cpp:47
            const int r0 = TB::lvl_row0[l], ml = TB::lvl_row0[l + 1] - TB::lvl_row0[l];
            const int v0 = r0;                                                 // rows of V before this level (a dropped row is a zero row)
            const bool keep = HAS_DA || l != TB::pik_last_level;               // the projector left by the last level is only read by `P da`
            if (ml == 0) continue;
            double Jb[MM][NV], de[MM], A[MM * (MM + 1) / 2];
            IKD_UNROLL
            for (int r = 0; r < ml; ++r) {
                double s = ws[T.off_e + r0 + r];                               // de = e_l - J_l dq, pik.cpp:49
                IKD_UNROLL
                for (int c = 0; c < NV; ++c) s = dfma(-ws[T.off_J + (r0 + r) * NV + c], dq[c], s);
                de[r] = s;
                IKD_UNROLL
                for (int c = 0; c < NV; ++c) Jb[r][c] = ws[T.off_J + (r0 + r) * NV + c];
                IKD_UNROLL
                for (int k = 0; k < v0; ++k) {                                 // Jbar = J_l P, pik.cpp:51
                    double d = 0.0;
                    IKD_UNROLL
                    for (int c = 0; c < NV; ++c) d = dfma(ws[T.off_J + (r0 + r) * NV + c], V[k][c], d);
                    IKD_UNROLL
                    for (int c = 0; c < NV; ++c) Jb[r][c] = dfma(-d, V[k][c], Jb[r][c]);
                }
            }
            const double lam2 = prm.lam2[l < kMaxPikLevels ? l : kMaxPikLevels - 1];
            IKD_UNROLL
            for (int i = 0; i < ml; ++i)
                IKD_UNROLL
                for (int j = 0; j <= i; ++j) {
                    double s = (i == j) ? lam2 : 0.0;
                    IKD_UNROLL
                    for (int c = 0; c < NV; ++c) s = dfma(Jb[i][c], Jb[j][c], s);
                    A[tri(i, j)] = s;
                }
            IKD_UNROLL
            for (int k = 0; k < ml; ++k) {   // Cholesky in place (diagonal holds 1 / L_kk), as generic_dls
                double d = A[tri(k, k)];
                IKD_UNROLL
                for (int m = 0; m < k; ++m) { const double x = A[tri(k, m)]; d = dfma(-x, x, d); }
                const double inv = drsqrt(d);
                A[tri(k, k)] = inv;
                IKD_UNROLL
                for (int i = k + 1; i < ml; ++i) {
                    double s = A[tri(i, k)];
                    IKD_UNROLL
                    for (int m = 0; m < k; ++m) s = dfma(-A[tri(i, m)], A[tri(k, m)], s);
                    A[tri(i, k)] = s * inv;
                }
            }
            IKD_UNROLL
            for (int k = 0; k < ml; ++k) {
                double s = de[k];
                IKD_UNROLL
                for (int m = 0; m < k; ++m) s = dfma(-A[tri(k, m)], de[m], s);
                de[k] = s * A[tri(k, k)];
            }
            IKD_UNROLL
            for (int k = ml - 1; k >= 0; --k) {
                double s = de[k];
                IKD_UNROLL
                for (int m = ml - 1; m > k; --m) s = dfma(-A[tri(m, k)], de[m], s);
                de[k] = s * A[tri(k, k)];
            }
            IKD_UNROLL
            for (int c = 0; c < NV; ++c) {                                     // dq -= damp_pseudoinverse(Jbar, lambda_l) de, pik.cpp:54-55
                double s = dq[c];
                IKD_UNROLL
                for (int r = 0; r < ml; ++r) s = dfma(-Jb[r][c], de[r], s);
                dq[c] = s;
            }
            if (keep) {                                                        // P -= pinv(Jbar) Jbar, pik.cpp:58-61
                // Gram-Schmidt with PIVOTING -- the largest remaining row first, as the column-pivoted QR behind the reference's
                // complete orthogonal decomposition takes its columns -- and its rank rule: a pivot below epsilon x min(rows, cols)
                // x the first pivot ends the basis (a Cassie leg's Full task has rank 5: its sixth pivot is rounding noise, 1e-16
                // of the first, and WITHOUT pivoting up to 7e-16 when the fifth happens to be small -- the threshold is 1.3e-15).
                // No row moves: the pivot row is picked by selects (a per-lane row index would put Jb into scratch memory).
                const double thr = 2.220446049250313e-16 * static_cast<double>(ml < NV ? ml : NV);
                const double thr2 = thr * thr;
                bool used[MM], live = true;
                double maxpiv2 = 0.0;
                IKD_UNROLL
                for (int r = 0; r < ml; ++r) used[r] = false;
                IKD_UNROLL
                for (int k = 0; k < ml; ++k) {
                    double n2[MM], best = -1.0;
                    IKD_UNROLL
                    for (int r = 0; r < ml; ++r) {
                        double s = 0.0;
                        IKD_UNROLL
                        for (int c = 0; c < NV; ++c) s = dfma(Jb[r][c], Jb[r][c], s);
                        n2[r] = dsel(used[r], -1.0, s);
                        best = dmax(best, n2[r]);
                    }
                    if (k == 0) maxpiv2 = best;
                    live = live && best > thr2 * maxpiv2 && best > 0.0;
                    const double inv = dsel(live, drsqrt(dmax(best, 1e-300)), 0.0);
                    double v[NV];
                    IKD_UNROLL
                    for (int c = 0; c < NV; ++c) v[c] = 0.0;
                    bool found = false;
                    IKD_UNROLL
                    for (int r = 0; r < ml; ++r) {
                        const bool is_p = !found && !used[r] && n2[r] == best;
                        found = found || is_p;
                        used[r] = used[r] || is_p;
                        IKD_UNROLL
                        for (int c = 0; c < NV; ++c) v[c] = dsel(is_p, Jb[r][c] * inv, v[c]);
                    }
                    IKD_UNROLL
                    for (int c = 0; c < NV; ++c) V[v0 + k][c] = v[c];
                    IKD_UNROLL
                    for (int r = 0; r < ml; ++r) {
                        IKD_UNROLL
                        for (int pass = 0; pass < 2; ++pass) {
                            double d = 0.0;
                            IKD_UNROLL
                            for (int c = 0; c < NV; ++c) d = dfma(v[c], Jb[r][c], d);
                            d = dsel(used[r], 0.0, d);
                            IKD_UNROLL
                            for (int c = 0; c < NV; ++c) Jb[r][c] = dfma(-d, v[c], Jb[r][c]);
                        }
                    }
                }
            }
        }
        if (HAS_DA) {                                                          // dq += P da, pik.cpp:65
            double t[NV];
            IKD_UNROLL
            for (int c = 0; c < NV; ++c) t[c] = prm.da[c < kMaxPikDa ? c : kMaxPikDa - 1];
            IKD_UNROLL
            for (int k = 0; k < VR; ++k) {
                double d = 0.0;
                IKD_UNROLL
                for (int c = 0; c < NV; ++c) d = dfma(V[k][c], prm.da[c < kMaxPikDa ? c : kMaxPikDa - 1], d);
                IKD_UNROLL
                for (int c = 0; c < NV; ++c) t[c] = dfma(-d, V[k][c], t[c]);
            }
            IKD_UNROLL
            for (int c = 0; c < NV; ++c) dq[c] += t[c];
        }
        IKD_UNROLL
        for (int c = 0; c < NV; ++c) ws[T.off_dq + c] = dq[c];
        const bool stop_now = active && (prm.stop_sq_tol >= 0.0) && (e0sq < prm.stop_sq_tol);  // pik.cpp:67-70
        if (stop_now) { success = true; iters = it; }
        active = active && !stop_now;
        generic_integrate_clip(T, ws, prm.step_length, active);               // pik.cpp:73-77
        if (!any_lane(active)) break;
    }
    iters_out = iters;
    success_out = success;
}
#endif  // IKD_STATIC_TABLES

struct PikKernelArgs {
    GenericTables T;
    PikParams prm;
    int layout;
    int64_t B;
    const double *q0, *targets;
    double *q_out;
    uint8_t *success;
    int32_t *iters;
    double *ws;          // [T.ws_words_pik][ws_stride]
    int64_t ws_stride;
};

template <class AnyFn>
IKD_FN void pik_generic_body(const PikKernelArgs &a, int64_t gid, AnyFn any_lane) {
    const bool valid = gid < a.B;
    const int64_t b = valid ? gid : a.B - 1;
    const Ws ws{a.ws + gid, a.ws_stride};
    for (int i = 0; i < a.T.nq; ++i) ws[a.T.off_q + i] = a.q0[at(a.layout, a.B, a.T.nq, i, b)];
    const double *tl = a.layout == LAYOUT_SOA ? a.targets + b : a.targets + b * a.T.ntasks * 12;
    const int64_t ts = a.layout == LAYOUT_SOA ? a.B : 1;
    int iters;
    bool success;
    generic_pik(a.T, a.prm, ws, tl, ts, iters, success, any_lane);
    if (!valid) return;
    for (int i = 0; i < a.T.nq; ++i) a.q_out[at(a.layout, a.B, a.T.nq, i, b)] = ws[a.T.off_q + i];
    if (a.success) a.success[b] = success ? 1 : 0;
    if (a.iters) a.iters[b] = iters;
}

#ifdef IKD_STATIC_TABLES
// What one lane of the compiled ik::pik program does (kernel ikgpu_lane_pik / ikgpu_lane_pik_da; a.T and a.ws are unused: the
// tables are TB's constants, the workspace is the caller's local array).
template <bool HAS_DA, class TB, class WS, class AnyFn>
IKD_FN void pik_static_body(const PikKernelArgs &a, const TB &T, int64_t gid, const WS &ws, AnyFn any_lane) {
    const bool valid = gid < a.B;
    const int64_t b = valid ? gid : a.B - 1;
    IKD_UNROLL
    for (int i = 0; i < T.nq; ++i) ws[T.off_q + i] = a.q0[at(a.layout, a.B, T.nq, i, b)];
    const LaneRows tl{reinterpret_cast<const char *>(a.layout == LAYOUT_SOA ? a.targets + b : a.targets + b * T.ntasks * 12), 0u,
                      static_cast<int64_t>(a.layout == LAYOUT_SOA ? a.B * 8 : 8), false};
    int iters;
    bool success;
    static_pik<HAS_DA>(T, a.prm, ws, tl, iters, success, any_lane);
    if (!valid) return;
    IKD_UNROLL
    for (int i = 0; i < T.nq; ++i) a.q_out[at(a.layout, a.B, T.nq, i, b)] = ws[T.off_q + i];
    if (a.success) a.success[b] = success ? 1 : 0;
    if (a.iters) a.iters[b] = iters;
}
#endif

}  // namespace ikdev
