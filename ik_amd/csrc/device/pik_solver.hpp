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

}  // namespace ikdev
