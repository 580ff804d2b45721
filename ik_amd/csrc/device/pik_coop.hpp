// pik_coop.hpp -- ik::pik(), the reference's prioritised IK (reference ik/ik/pik.cpp:31-103), in the cooperative LDS-resident form
// of coop_solver.hpp: sixteen lanes per problem, the workspace (now with the projector P and the projected Jacobian of the
// level) on chip.  The per-lane program (pik_solver.hpp) streams ~17 GB of workspace through HBM per iteration at B = 65536
// (profiles/r01_pmc/pik_*.csv: 872 GB per 50-iteration launch); here HBM sees q0 and the targets on entry and q on exit.
//
// Per iteration: evaluate_problem_data (coop_evaluate), P = I, dq = 0, then for every priority level
//     de   = e_l - J_l dq                                   pik.cpp:49      one lane per row
//     Jbar = J_l P                                          pik.cpp:51      one lane per entry
//     dq  -= damp_pseudoinverse(Jbar, lambda_l) de          pik.cpp:54-55   = Jbar^T (Jbar Jbar^T + lambda^2 I)^-1 de: the m x m
//                                                                           SPD solve of coop_chol_solve (needs lambda > 0)
//     P   -= pinv(Jbar) Jbar                                pik.cpp:58-61   an orthonormal basis v_1 .. v_r of the row space of Jbar
//                                                                           by Gram-Schmidt with row pivoting on the largest
//                                                                           remaining norm, every projection applied twice;
//                                                                           the pivot norms are |R_kk| of Eigen's column-pivoted
//                                                                           QR of Jbar^T (what its COD starts from) and its rank
//                                                                           rule is applied to them: |R_kk| > eps min(m, n) max|R|
// and dq += P da (pik.cpp:65), the stop test on e[0], integrate + clip.  The projector left by the last level is only built
// when da != 0.  Where the rank decision is clear-cut this agrees with pik_solver.hpp (one-sided Jacobi) to rounding; where a
// level's projected Jacobian is exactly rank deficient the decision is made on rounding noise in either form (and in the
// reference: see pik_solver.hpp).
#pragma once
#include "coop_solver.hpp"
#include "pik_solver.hpp"

namespace ikdev {

struct PikCoopLayout {
    int P, Jb, de, nrm, words;  // offsets into the group's workspace beyond CoopLayout's; `words` = the whole PIK workspace
    int factored;               // 1: keep the projector as its orthonormal basis V (see coop_pik), never form the nv x nv P
};

// The level loop of ik::pik (pik.cpp:44-65) with the projector kept in factored form.  The reference updates
// P -= pinv(Jbar) Jbar level by level; pinv(Jbar) Jbar is the orthogonal projector onto the row space of Jbar = J_l P, which
// lies in the complement of every earlier level's row space -- so P = I - V^T V with V the orthonormal bases of all levels
// so far stacked row-wise (R rows, kept in the nv x nv region the dense P would occupy).  Then
//   Jbar = J_l P = J_l - (J_l V^T) V        (the first level: Jbar = J_l, nothing to project)
//   dq += P da   = da  - V^T (V da)
// and the nv x nv matrix is never formed: O(ml R nv) per level instead of O(ml nv^2 + rank nv^2).  The coefficients J_l V^T
// (ml x R) sit behind V's current rows; the host checks that they fit (PikCoopLayout::factored).
template <class AnyFn>
IKD_FN void coop_pik_levels_factored(const GenericTables &T, const CoopLayout &L, const PikCoopLayout &K, const PikParams &prm,
                                     const int g, double *ws, int last_level, AnyFn any_fn IKC_TICK_ARG) {
    (void)g;
    const int nv = T.nv;
    int R = 0;  // rows of V so far (per problem: the four groups of a workgroup may differ; no barrier depends on it)
    IKC_FOR(c, nv) ws[L.dq + c] = 0.0;
    IKC_SYNC();
    for (int l = 0; l < T.nlevels; ++l) {                              // pik.cpp:47
        const int r0 = T.lvl_row0[l], ml = T.lvl_row0[l + 1] - r0;
        if (ml == 0) continue;
        const bool update_P = l != last_level || prm.has_da != 0;
        const int cf = K.P + R * nv;                                   // ml x R coefficients, behind V
        IKC_FOR(r, ml) {                                               // de = e_l - J_l dq
            double s = ws[L.e + r0 + r];
#pragma unroll 8
            for (int c = 0; c < nv; ++c) s = dfma(-ws[L.J + (r0 + r) * nv + c], ws[L.dq + c], s);
            ws[K.de + r] = s;
        }
        IKC_FOR(idx, ml * R) {                                         // J_l V^T
            const int r = idx / R, k = idx % R;
            double a = 0.0;
#pragma unroll 8
            for (int c = 0; c < nv; ++c) a = dfma(ws[L.J + (r0 + r) * nv + c], ws[K.P + k * nv + c], a);
            ws[cf + idx] = a;
        }
        IKC_SYNC();
        IKC_FOR(idx, ml * nv) {                                        // Jbar = J_l - (J_l V^T) V
            const int r = idx / nv, c = idx % nv;
            double a = ws[L.J + (r0 + r) * nv + c];
            for (int k = 0; k < R; ++k) a = dfma(-ws[cf + r * R + k], ws[K.P + k * nv + c], a);
            ws[K.Jb + idx] = a;
        }
        IKC_SYNC();
        IKC_TICK(9);
        // damped step: (Jbar Jbar^T + lambda^2 I) x = de, dq -= Jbar^T x
        IKC_FOR(p, tri(ml, 0) + ml) {
            const int i = L.pair_i[p], j = L.pair_j[p];
            double s;
            if (i == ml) {
                s = ws[K.de + j];
            } else {
                s = (i == j) ? prm.lam2[l] : 0.0;
#pragma unroll 8
                for (int c = 0; c < nv; ++c) s = dfma(ws[K.Jb + i * nv + c], ws[K.Jb + j * nv + c], s);
                if (update_P) ws[cf + tri(i, j)] = (i == j) ? s - prm.lam2[l] : s;   // Jbar Jbar^T, for the Cholesky-QR basis below
            }
            ws[L.G + tri(i, j)] = s;
        }
        IKC_SYNC();
        IKC_TICK(10);
        coop_chol_solve(L, g, ws, L.G, L.dinv, L.x, ml);
        IKC_FOR(c, nv) {
            double s = ws[L.dq + c];
#pragma unroll 8
            for (int r = 0; r < ml; ++r) s = dfma(-ws[K.Jb + r * nv + c], ws[L.x + r], s);
            ws[L.dq + c] = s;
        }
        IKC_SYNC();
        IKC_TICK(11);
        if (!update_P) continue;
        int rank = ml;                                                 // v_1 .. v_rank in the first rows of Jb
        if (!coop_rowspace_basis_cholqr(L.pair_i, L.pair_j, g, ws, K.Jb, cf, cf + tri(ml, 0), ml, nv, any_fn)) rank = coop_rowspace_basis(g, ws, K.Jb, K.nrm, ml, nv);
        if (rank > nv - R) rank = nv - R;                              // (cannot happen in exact arithmetic)
        IKC_FOR(idx, rank * nv) ws[K.P + R * nv + idx] = ws[K.Jb + idx];   // append them to V
        R += rank;
        IKC_SYNC();
        IKC_TICK(12);
    }
    if (prm.has_da) {                                                  // pik.cpp:65: dq += P da = da - V^T (V da)
        const int cf = K.P + R * nv;
        IKC_FOR(k, R) {
            double a = 0.0;
            for (int c = 0; c < nv; ++c) a = dfma(ws[K.P + k * nv + c], prm.da[c], a);
            ws[cf + k] = a;
        }
        IKC_SYNC();
        IKC_FOR(c, nv) {
            double s = ws[L.dq + c] + prm.da[c];
            for (int k = 0; k < R; ++k) s = dfma(-ws[cf + k], ws[K.P + k * nv + c], s);
            ws[L.dq + c] = s;
        }
        IKC_SYNC();
    }
}

template <class AnyFn>
IKD_FN void coop_pik(const GenericTables &T, const CoopLayout &L, const PikCoopLayout &K, const PikParams &prm, const int g, double *ws,
                     int &iters_out, bool &success_out, AnyFn any_active) {
    (void)g;
    const int nv = T.nv;
    bool active = true, success = false;
    int iters = prm.max_iterations;
    int last_level = 0;
    for (int l = 0; l < T.nlevels; ++l)
        if (T.lvl_row0[l + 1] > T.lvl_row0[l]) last_level = l;
    IKC_TICK_INIT;
    for (int it = 0; it < prm.max_iterations; ++it) {
        const double e0sq = coop_evaluate(T, L, g, ws IKC_TICK_PASS);    // pik.cpp:41
        if (K.factored) {
            coop_pik_levels_factored(T, L, K, prm, g, ws, last_level, any_active IKC_TICK_PASS);
        } else {
        IKC_FOR(i, nv * nv) ws[K.P + i] = (i / nv == i % nv) ? 1.0 : 0.0;   // pik.cpp:44-45
        IKC_FOR(c, nv) ws[L.dq + c] = 0.0;
        IKC_SYNC();
        for (int l = 0; l < T.nlevels; ++l) {                              // pik.cpp:47
            const int r0 = T.lvl_row0[l], ml = T.lvl_row0[l + 1] - r0;
            if (ml == 0) continue;
            const bool update_P = l != last_level || prm.has_da != 0;
            IKC_FOR(r, ml) {
                double s = ws[L.e + r0 + r];
#pragma unroll 8
                for (int c = 0; c < nv; ++c) s = dfma(-ws[L.J + (r0 + r) * nv + c], ws[L.dq + c], s);
                ws[K.de + r] = s;
            }
            IKC_FOR(idx, ml * nv) {
                const int r = idx / nv, c = idx % nv;
                double a = 0.0;
#pragma unroll 8
                for (int k = 0; k < nv; ++k) a = dfma(ws[L.J + (r0 + r) * nv + k], ws[K.P + k * nv + c], a);
                ws[K.Jb + idx] = a;
            }
            IKC_SYNC();
            // damped step: (Jbar Jbar^T + lambda^2 I) x = de, dq -= Jbar^T x
            IKC_FOR(p, tri(ml, 0) + ml) {
                const int i = L.pair_i[p], j = L.pair_j[p];
                double s;
                if (i == ml) {
                    s = ws[K.de + j];
                } else {
                    s = (i == j) ? prm.lam2[l] : 0.0;
#pragma unroll 8
                    for (int c = 0; c < nv; ++c) s = dfma(ws[K.Jb + i * nv + c], ws[K.Jb + j * nv + c], s);
                }
                ws[L.G + tri(i, j)] = s;
            }
            IKC_SYNC();
            coop_chol_solve(L, g, ws, L.G, L.dinv, L.x, ml);
            IKC_FOR(c, nv) {
                double s = ws[L.dq + c];
#pragma unroll 8
                for (int r = 0; r < ml; ++r) s = dfma(-ws[K.Jb + r * nv + c], ws[L.x + r], s);
                ws[L.dq + c] = s;
            }
            IKC_SYNC();
            if (!update_P) continue;
            const int rank = coop_rowspace_basis(g, ws, K.Jb, K.nrm, ml, nv);   // v_1 .. v_rank in the first rows of Jb
            IKC_FOR(idx, nv * nv) {   // P -= sum_{k < rank} v_k v_k^T
                const int a = idx / nv, b = idx % nv;
                double s = ws[K.P + idx];
                for (int k = 0; k < rank; ++k) s = dfma(-ws[K.Jb + k * nv + a], ws[K.Jb + k * nv + b], s);
                ws[K.P + idx] = s;
            }
            IKC_SYNC();
        }
        if (prm.has_da) {                                                  // pik.cpp:65 (da lives in the kernel arguments)
            IKC_FOR(c, nv) {
                double s = ws[L.dq + c];
                for (int k = 0; k < nv; ++k) s = dfma(ws[K.P + c * nv + k], prm.da[k], s);
                ws[L.dq + c] = s;
            }
            IKC_SYNC();
        }
        }  // !K.factored
        const bool stop_now = active && (prm.stop_sq_tol >= 0.0) && (e0sq < prm.stop_sq_tol);   // pik.cpp:67-70
        if (stop_now) { success = true; iters = it; }
        active = active && !stop_now;
        coop_integrate(T, L, g, ws, prm.step_length, active);             // pik.cpp:73-77
        IKC_TICK(8);
        if (!any_active(active)) break;
    }
    iters_out = iters;
    success_out = success;
}

struct PikCoopKernelArgs {
    GenericTables T;
    CoopLayout L;
    PikCoopLayout K;
    PikParams prm;
    int layout;
    int64_t B;
    const double *q0, *targets;
    double *q_out;
    uint8_t *success;
    int32_t *iters;
};

// T, L: the tables the phases read (on the device: the copies staged in LDS, see kernels.hip; a.T / a.L otherwise).
template <class AnyFn>
IKD_FN void pik_coop_body(const PikCoopKernelArgs &a, const GenericTables &T, const CoopLayout &L, int64_t problem, const int g, double *ws,
                          AnyFn any_active) {
    (void)g;
    const bool valid = problem < a.B;
    const int64_t b = valid ? problem : a.B - 1;
    const int nq = T.nq, nslots = T.ntasks * 12;
    IKC_FOR(i, nq) ws[L.q + i] = a.q0[at(a.layout, a.B, nq, i, b)];
    IKC_FOR(i, L.ntg) {
        const int slot = L.tg_src[i];
        ws[L.tg + i] = a.layout == LAYOUT_SOA ? a.targets[static_cast<int64_t>(slot) * a.B + b] : a.targets[b * nslots + slot];
    }
    IKC_SYNC();
    int iters;
    bool success;
    coop_pik(T, L, a.K, a.prm, g, ws, iters, success, any_active);
    if (valid) {   // (no early return: the workgroup goes on to its next group of problems, see pik_coop_kernel)
        IKC_FOR(i, nq) a.q_out[at(a.layout, a.B, nq, i, b)] = ws[L.q + i];
        IKC_FOR(one, 1) {
            if (a.success) a.success[b] = success ? 1 : 0;
            if (a.iters) a.iters[b] = iters;
        }
    }
    IKC_SYNC();    // the workspace is free for the next problem
}

}  // namespace ikdev
