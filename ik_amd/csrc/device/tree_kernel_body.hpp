// tree_kernel_body.hpp -- what ONE lane of the free-flyer tree kernels does around the lane solver
// (device/tree_solver.hpp): load its problem from HBM, solve, store.  Shared by the __global__
// wrappers (kernels.hip) and by the CPU lane emulator under tests/.
#pragma once
#include <cstdint>
#include <type_traits>

#include "chain_kernel_body.hpp"
#include "tree_solver.hpp"

namespace ikdev {

template <int NA, int NB>
struct TreeKernelArgs {
    const TreeDesc<NA, NB> *desc;  // HBM copy of the table (uploaded at problem creation)
    TreeParams prm;
    int qidxA[NA > 0 ? NA : 1], qidxB[NB > 0 ? NB : 1];  // index of each chain joint in q
    int vidxA[NA > 0 ? NA : 1], vidxB[NB > 0 ? NB : 1];
    int tslot[3];      // row of `targets` holding the target of chain A / chain B / the base task
    int trow[3];       // first row of each task in the stacked error vector (stage kernel)
    int tdim[3], trow0[3];  // rows the task's kinematic type keeps: tdim of them starting at local row trow0
    int nq, nv, ntasks, layout;
    int64_t B;
    const double *q0;
    const double *targets;
    double *q_out;
    uint8_t *success;
    int32_t *iters;
    const double *lower, *upper;  // [nq]
    const uint8_t *q_in_chain;    // [nq] 1 where the kernel integrates the entry itself
    double *e_out, *J_out, *oMf_out;  // stage kernels
};

// B independent ik::dls() calls (reference ik/ik/dls.cpp:5-78) on a free-flyer model, lane `gid`.
template <int NA, int NB, class Park, class AnyFn>
IKD_FN void dls_tree_body(const TreeKernelArgs<NA, NB> &a, const TreeDesc<NA, NB> &d, int64_t gid, Park park,
                          AnyFn any_active) {
    const bool valid = gid < a.B;
    const int64_t b = valid ? gid : a.B - 1;
    double qb[7], qa[NA > 0 ? NA : 1], qbj[NB > 0 ? NB : 1];
#pragma unroll
    for (int k = 0; k < 7; ++k) qb[k] = a.q0[at(a.layout, a.B, a.nq, k, b)];  // free-flyer: idx_q = 0
#pragma unroll
    for (int j = 0; j < NA; ++j) qa[j] = a.q0[at(a.layout, a.B, a.nq, a.qidxA[j], b)];
#pragma unroll
    for (int j = 0; j < NB; ++j) qbj[j] = a.q0[at(a.layout, a.B, a.nq, a.qidxB[j], b)];
    const double *tl = a.layout == LAYOUT_SOA ? a.targets + b : a.targets + b * a.ntasks * 12;
    const int64_t ts = a.layout == LAYOUT_SOA ? a.B : 1;

    int iters;
    bool success;
    tree_dls<NA, NB>(d, a.prm, qb, qa, qbj, tl, ts, a.tslot, iters, success, park, any_active);

    if (!valid) return;
#pragma unroll
    for (int k = 0; k < 7; ++k) a.q_out[at(a.layout, a.B, a.nq, k, b)] = qb[k];
#pragma unroll
    for (int j = 0; j < NA; ++j) a.q_out[at(a.layout, a.B, a.nq, a.qidxA[j], b)] = qa[j];
#pragma unroll
    for (int j = 0; j < NB; ++j) a.q_out[at(a.layout, a.B, a.nq, a.qidxB[j], b)] = qbj[j];
    for (int i = 0; i < a.nq; ++i) {  // entries outside every task support: only ever clamped
        if (a.q_in_chain[i]) continue;
        const double v = a.q0[at(a.layout, a.B, a.nq, i, b)];
        const double c = dmin(a.upper[i], dmax(v, a.lower[i]));
        a.q_out[at(a.layout, a.B, a.nq, i, b)] = (iters > 0) ? c : v;
    }
    if (a.success) a.success[b] = success ? 1 : 0;
    if (a.iters) a.iters[b] = iters;
}

// Stage kernel: world placement of every task frame and the stacked weighted error / dense Jacobian
// (reference ik/ik/data.cpp:25-58).  Rows are emitted in task order with each task's kinematic type
// selecting its rows, exactly as the reference stacks them.
template <int NA, int NB>
IKD_FN void eval_tree_body(const TreeKernelArgs<NA, NB> &a, const TreeDesc<NA, NB> &d, int64_t b) {
    if (b >= a.B) return;
    double qb[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) qb[k] = a.q0[at(a.layout, a.B, a.nq, k, b)];
    double R1[9];
    quat_to_R(qb, R1);
    const double p1[3] = {qb[0], qb[1], qb[2]};
    const double *tl = a.layout == LAYOUT_SOA ? a.targets + b : a.targets + b * a.ntasks * 12;
    const int64_t ts = a.layout == LAYOUT_SOA ? a.B : 1;
    int Mtot = 0;
    for (int s = 0; s < 3; ++s) Mtot += a.tdim[s];

    auto emit = [&](int slot, const TaskTerms &t, const double (&JL)[3][3], const double (&JA)[3][6], const double (*col)[6],
                    const int *vidx, int nj) {
        for (int r = 0; r < a.tdim[slot]; ++r) {
            const int lr = a.trow0[slot] + r, row = a.trow[slot] + r;
            if (a.e_out) a.e_out[at(a.layout, a.B, Mtot, row, b)] = t.e[lr];
            if (a.J_out) {
                for (int c = 0; c < a.nv; ++c) a.J_out[at(a.layout, a.B, Mtot * a.nv, row * a.nv + c, b)] = 0.0;
                for (int c = 0; c < 3; ++c) {
                    a.J_out[at(a.layout, a.B, Mtot * a.nv, row * a.nv + c, b)] = lr < 3 ? JL[c][lr] : 0.0;
                    a.J_out[at(a.layout, a.B, Mtot * a.nv, row * a.nv + 3 + c, b)] = JA[c][lr];
                }
                for (int j = 0; j < nj; ++j) a.J_out[at(a.layout, a.B, Mtot * a.nv, row * a.nv + vidx[j], b)] = col[j][lr];
            }
        }
    };
    auto chain = [&](auto njc, int slot, const double (*pl)[12], const double *fr, const double *w6, const int *qidx,
                     const int *vidx) {
        constexpr int NJ = decltype(njc)::value;
        constexpr int NJs = NJ > 0 ? NJ : 1;
        double R[9], p[3], zax[NJs][3], org[NJs][3];
        for (int k = 0; k < 9; ++k) R[k] = R1[k];
        for (int k = 0; k < 3; ++k) p[k] = p1[k];
        for (int j = 0; j < NJ; ++j) {
            se3_compose_const(R, p, pl[j]);
            double s, c;
            dsincos(a.q0[at(a.layout, a.B, a.nq, qidx[j], b)], s, c);
            rot_z_right(R, s, c);
            zax[j][0] = R[2]; zax[j][1] = R[5]; zax[j][2] = R[8];
            org[j][0] = p[0]; org[j][1] = p[1]; org[j][2] = p[2];
        }
        se3_compose_const(R, p, fr);
        if (a.oMf_out) {
            for (int k = 0; k < 9; ++k) a.oMf_out[at(a.layout, a.B, a.ntasks * 12, a.tslot[slot] * 12 + k, b)] = R[k];
            for (int k = 0; k < 3; ++k) a.oMf_out[at(a.layout, a.B, a.ntasks * 12, a.tslot[slot] * 12 + 9 + k, b)] = p[k];
        }
        if (!a.e_out && !a.J_out) return;
        double oMt[12];
        for (int k = 0; k < 12; ++k) oMt[k] = tl[(a.tslot[slot] * 12 + k) * ts];
        TaskTerms t;
        task_terms(R, p, oMt, w6, t);
        double JL[3][3], JA[3][6], col[NJs][6];
        base_columns(t, R, p, R1, p1, JL, JA);
        for (int j = 0; j < NJ; ++j) {
            double wl[3], r[3], vl[3];
            rotT_vec(R, zax[j], wl);
            const double dj[3] = {org[j][0] - p[0], org[j][1] - p[1], org[j][2] - p[2]};
            rotT_vec(R, dj, r);
            cross(r, wl, vl);
            for (int i = 0; i < 3; ++i) {
                col[j][i] = dfma(t.At[3 * i], vl[0], dfma(t.At[3 * i + 1], vl[1], dfma(t.At[3 * i + 2], vl[2],
                            dfma(t.Bt[3 * i], wl[0], dfma(t.Bt[3 * i + 1], wl[1], t.Bt[3 * i + 2] * wl[2])))));
                col[j][3 + i] = dfma(t.Ab[3 * i], wl[0], dfma(t.Ab[3 * i + 1], wl[1], t.Ab[3 * i + 2] * wl[2]));
            }
        }
        emit(slot, t, JL, JA, col, vidx, NJ);
    };
    using NAc = std::integral_constant<int, NA>;
    using NBc = std::integral_constant<int, NB>;
    using N0c = std::integral_constant<int, 0>;
    if (NA > 0) chain(NAc{}, 0, d.plA, d.frA, d.wA, a.qidxA, a.vidxA);
    if (NB > 0) chain(NBc{}, 1, d.plB, d.frB, d.wB, a.qidxB, a.vidxB);
    if (a.prm.hasP) chain(N0c{}, 2, d.plA, d.frP, d.wP, a.qidxA, a.vidxA);
}

}  // namespace ikdev
