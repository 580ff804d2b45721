// tree_kernel_body.hpp -- what ONE lane of the free-flyer tree kernels does around the lane solver
// (device/tree_solver.hpp): load its problem from HBM, solve, store.  Shared by the __global__
// wrappers (kernels.hip) and by the CPU lane emulator under tests/.
#pragma once
#include <cstdint>

#include "chain_kernel_body.hpp"
#include "tree_solver.hpp"

namespace ikdev {

template <int NJ, int NCH>
struct TreeKernelArgs {
    const TreeDesc<NJ, NCH> *desc;  // HBM copy of the table (uploaded at problem creation)
    TreeParams prm;
    int qidx[NCH][NJ], vidx[NCH][NJ];  // index of each chain joint in q / in the tangent vector
    int tslot[3];                      // row of `targets` holding the target of chain 0 / chain 1 / the base task
    int trow[3];                       // first row of each task in the stacked error vector (stage kernel)
    int tdim[3], trow0[3];             // rows the task's kinematic type keeps: tdim of them from local row trow0
    int nq, nv, ntasks, layout;
    int64_t B;
    const double *q0;
    const double *targets;
    double *q_out;
    uint8_t *success;
    int32_t *iters;
    const double *lower, *upper;  // [nq]
    const uint8_t *q_in_chain;    // [nq] 1 where the kernel integrates the entry itself
    double *e_out, *J_out, *oMf_out;  // stage kernel
    // second phase of a two-phase stop-rule solve (as ChainKernelArgs): the refill kernel walks the list of problems the lock-step first
    // phase left unfinished, continuing each from its iterate in q_out at the count in iters[].  Null: the whole batch.
    const int32_t *worklist;
    const unsigned long long *count;
    int leave_active, leave_after;   // ... and of its first phase (chain_kernel_body.hpp KeepGoing, append_unfinished)
    int32_t *append_list;
    unsigned long long *append_count;
};

// B independent ik::dls() calls (reference ik/ik/dls.cpp:5-78) on a free-flyer model, lane `gid`.
// post_lane / post_stride: where a posture build keeps the joints outside the chains between iterations -- an LDS column of
// the lane (row k at post_lane[k * post_stride]) when the kernel has LDS to spare, else (nullptr) the lane's column of q_out.
// group0: the first problem of this lane's workgroup (wave-uniform; what LaneRows addresses from), or -1: per-lane addressing.
template <int NJ, int NCH, int SPEC = -1, class Desc, class Park, class AnyFn>
IKD_FN void dls_tree_body(const TreeKernelArgs<NJ, NCH> &a, const Desc &d, int64_t gid, Park park,
                          AnyFn any_active, double *post_lane = nullptr, int64_t post_stride = 0, int64_t group0 = -1) {
    const bool valid = gid < a.B;
    const int64_t b = valid ? gid : a.B - 1;
    double qb[7], qj0[NJ], qj1[NJ];  // chain 1's angles stay unused (zero) when NCH == 1
    // general builds: a fixed-base model has no base entries in q (do not read them: nq may be smaller than 7), its base pose is
    // the world
    const bool fixed_base = (spec_is_general(SPEC) && (SPEC < 0 || !spec_has_constraint(SPEC))) ? a.prm.fixed_base != 0 : false;  // (no constraints on fixed-base models)
#pragma unroll
    for (int k = 0; k < 7; ++k) qb[k] = fixed_base ? (k == 6 ? 1.0 : 0.0) : a.q0[at(a.layout, a.B, a.nq, k, b)];  // free-flyer: idx_q = 0
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        qj0[j] = a.q0[at(a.layout, a.B, a.nq, a.qidx[0][j], b)];
        qj1[j] = NCH > 1 ? a.q0[at(a.layout, a.B, a.nq, a.qidx[NCH - 1][j], b)] : 0.0;
    }
    const int64_t g0 = group0 >= 0 ? group0 : b;
    const int64_t tcol = a.layout == LAYOUT_SOA ? 1 : static_cast<int64_t>(a.ntasks) * 12;   // doubles between consecutive problems
    const LaneRows tl{reinterpret_cast<const char *>(a.targets + g0 * tcol), static_cast<uint32_t>((b - g0) * tcol * 8),
                      static_cast<int64_t>(a.layout == LAYOUT_SOA ? a.B * 8 : 8), group0 >= 0};

    int iters;
    bool success;
    // joints outside the chains that carry a posture row live in this lane's column of q_out while the loop runs
    constexpr bool kPost = spec_has_posture(SPEC);
    // (LDS layout of a posture build with room to spare: post_n rows of q, post_n rows of outside targets, NJ rows of chain targets)
    double *t_out = post_lane ? post_lane + a.prm.post_n * post_stride : nullptr;
    double *t_chain = post_lane ? t_out + a.prm.post_n * post_stride : nullptr;
    const PostureState ps = post_lane ? PostureState{post_lane, post_stride, true, a.lower, a.upper, valid, t_out, t_chain}
                                      : PostureState{a.layout == LAYOUT_SOA ? a.q_out + b : a.q_out + b * a.nq,
                                                     a.layout == LAYOUT_SOA ? a.B : 1, false, a.lower, a.upper, valid, nullptr, nullptr};
    if (kPost && a.prm.post_on && valid) {
        for (int k = 0; k < a.prm.post_n; ++k)
            ps.q_lane[(ps.by_row ? k : a.prm.post_q[k]) * ps.stride] = a.q0[at(a.layout, a.B, a.nq, a.prm.post_q[k], b)];
    }
    if (kPost && a.prm.post_on && post_lane) {  // every lane (a tail lane shadows the last problem): its own LDS column
        for (int k = 0; k < a.prm.post_n; ++k) t_out[k * post_stride] = tl(a.prm.post_slot[k] * 12 + 9);
        for (int j = 0; j < NJ; ++j)
            t_chain[j * post_stride] = a.prm.postc_slot[0][j] >= 0 ? tl(a.prm.postc_slot[0][j] * 12 + 9) : 0.0;
    }
    tree_dls<NJ, NCH, SPEC>(d, a.prm, qb, qj0, qj1, tl, a.tslot, ps, iters, success, park, any_active);
    if (kPost && a.prm.post_on && valid && ps.by_row) {
        for (int k = 0; k < a.prm.post_n; ++k) a.q_out[at(a.layout, a.B, a.nq, a.prm.post_q[k], b)] = ps.q_lane[k * ps.stride];
    }

    if (a.append_count) append_unfinished(a.append_list, a.append_count, valid && !success && iters < a.prm.max_iterations, b);   // (wave-uniform test)
    if (!valid) return;
#pragma unroll
    for (int k = 0; k < 7; ++k)
        if (!fixed_base) a.q_out[at(a.layout, a.B, a.nq, k, b)] = qb[k];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        a.q_out[at(a.layout, a.B, a.nq, a.qidx[0][j], b)] = qj0[j];
        if (NCH > 1) a.q_out[at(a.layout, a.B, a.nq, a.qidx[NCH - 1][j], b)] = qj1[j];
    }
    for (int i = 0; i < a.nq; ++i) {  // entries outside every task support: only ever clamped
        if (a.q_in_chain[i]) continue;
        const double v = a.q0[at(a.layout, a.B, a.nq, i, b)];
        const double c = dmin(a.upper[i], dmax(v, a.lower[i]));
        a.q_out[at(a.layout, a.B, a.nq, i, b)] = (iters > 0) ? c : v;
    }
    if (a.success) a.success[b] = success ? 1 : 0;
    if (a.iters) a.iters[b] = iters;
}

#if IKD_HIP_LANG
// ---- lane refill for the tree kernels: chain_kernel_body.hpp's chain_refill_loop with the tree kernel's lane state ------------------
// (the floating base's 7 numbers + the chains' joint angles + the lane's target pointer).  tree_dls calls step() at the end of every
// iteration; builds without per-lane state outside q only (no posture rows, no ik::pik level).  queue / chunk: as the chain kernels'.
template <int NJ, int NCH>
struct TreeRefill {
    static constexpr bool on = true;
    const TreeKernelArgs<NJ, NCH> *a;
    unsigned long long *queue;
    int chunk, batch;   // problems pulled from the head at a time; idle lanes a refill event waits for
    bool fixed_base;
    int64_t b, first_round, pool_lo, pool_hi;
    bool exhausted;
    bool start;   // this lane holds a problem of the first (static) round
    bool took;    // (set by step) this lane has just taken a new problem: its iteration count restarts
    int64_t nwork;   // work items: the batch's problems, or the entries of a->worklist (`b` is always a PROBLEM index)
    __device__ __forceinline__ int64_t problem(int64_t w) const { return a->worklist ? static_cast<int64_t>(a->worklist[w]) : w; }
    __device__ __forceinline__ int it0() const { return a->worklist ? a->iters[b] : 0; }   // (of the lane's CURRENT problem)

    __device__ __forceinline__ void load(int64_t bb, double (&qb)[7], double (&qj0)[NJ], double (&qj1)[NJ]) const {
        const TreeKernelArgs<NJ, NCH> &A = *a;
        const double *src = A.worklist ? A.q_out : A.q0;   // (second phase: the first phase's iterate)
#pragma unroll
        for (int k = 0; k < 7; ++k) qb[k] = fixed_base ? (k == 6 ? 1.0 : 0.0) : src[at(A.layout, A.B, A.nq, k, bb)];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            qj0[j] = src[at(A.layout, A.B, A.nq, A.qidx[0][j], bb)];
            qj1[j] = NCH > 1 ? src[at(A.layout, A.B, A.nq, A.qidx[NCH - 1][j], bb)] : 0.0;
        }
    }
    __device__ __forceinline__ LaneRows target_rows(int64_t bb) const {
        return LaneRows{reinterpret_cast<const char *>(a->layout == LAYOUT_SOA ? a->targets + bb : a->targets + bb * a->ntasks * 12), 0u,
                        static_cast<int64_t>(a->layout == LAYOUT_SOA ? a->B * 8 : 8), false};
    }

    // done lanes store (q, success, iters) and take the next problem; returns the wave-uniform "some lane still holds a problem"
    template <class ReloadFn>
    __device__ __forceinline__ bool step(bool done, bool stopped, int iters, double (&qb)[7], double (&qj0)[NJ], double (&qj1)[NJ], LaneRows &tl,
                                         bool &active, ReloadFn reload_targets) {
        const TreeKernelArgs<NJ, NCH> &A = *a;
        const int lane = static_cast<int>(threadIdx.x) & 63;
        if (done) {   // the result leaves at once; the lane then idles until the wave's next refill event (batched: chain_kernel_body.hpp)
#pragma unroll
            for (int k = 0; k < 7; ++k)
                if (!fixed_base) A.q_out[at(A.layout, A.B, A.nq, k, b)] = qb[k];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                A.q_out[at(A.layout, A.B, A.nq, A.qidx[0][j], b)] = qj0[j];
                if (NCH > 1) A.q_out[at(A.layout, A.B, A.nq, A.qidx[NCH - 1][j], b)] = qj1[j];
            }
            if (A.success) A.success[b] = stopped ? 1 : 0;
            A.iters[b] = iters;                                       // never null here: the pass-through kernel reads it
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
                load(b, qb, qj0, qj1);
                tl = target_rows(b);
                reload_targets(tl);
            }
        }
        return __any(active) != 0;
    }
};

// B independent ik::dls() calls under lane refill (persistent waves; `wave` counts 64-lane waves across the grid).
template <int NJ, int NCH, int SPEC, class Desc, class Park>
__device__ __forceinline__ void dls_tree_refill_body(const TreeKernelArgs<NJ, NCH> &a, const Desc &d, int64_t wave, int64_t nwaves, Park park,
                                                     unsigned long long *queue, int chunk) {
    const int lane = static_cast<int>(threadIdx.x) & 63;
    const bool fixed_base = spec_is_general(SPEC) ? a.prm.fixed_base != 0 : false;
    const int64_t nwork = a.worklist ? static_cast<int64_t>(*a.count) : a.B;
    // waves without a share of the first round leave at once and are not counted below (chain_kernel_body.hpp chain_refill_loop)
    const int64_t working = (nwork + 63) / 64 < nwaves ? (nwork + 63) / 64 : nwaves;
    if (wave >= working) return;
    {
        TreeRefill<NJ, NCH> rf{&a, queue, chunk & 0xffff, chunk >> 16, fixed_base, 0, nwaves * 64, 0, 0, nwaves * 64 >= nwork, wave * 64 + lane < nwork, false, nwork};
        rf.b = rf.problem(rf.start ? wave * 64 + lane : 0);   // a tail lane of the first round shadows a valid problem
        double qb[7], qj0[NJ], qj1[NJ];
        rf.load(rf.b, qb, qj0, qj1);
        const LaneRows tl = rf.target_rows(rf.b);
        const PostureState ps{nullptr, 0, false, a.lower, a.upper, false, nullptr, nullptr};
        int iters;
        bool success;
        tree_dls<NJ, NCH, SPEC>(d, a.prm, qb, qj0, qj1, tl, a.tslot, ps, iters, success, park, [](bool act) { return __any(act) != 0; }, rf);
    }
    if (lane == 0) {   // the last wave out resets the slot
        __threadfence();
        if (atomicAdd(queue + 1, 1ull) == static_cast<unsigned long long>(working) - 1ull) {
            queue[0] = 0ull;
            queue[1] = 0ull;
            queue[2] = 0ull;   // (the two-phase worklist's length)
            __threadfence();
        }
    }
}
#endif  // IKD_HIP_LANG

// Stage kernel: world placement of every task frame and the stacked weighted error / dense Jacobian
// (reference ik/ik/data.cpp:25-58).  Rows are emitted in task order with each task's kinematic type
// selecting its rows, exactly as the reference stacks them.  slot 0 / 1: chains, slot 2: base task.
template <int NJ, int NCH>
IKD_FN void eval_tree_body(const TreeKernelArgs<NJ, NCH> &a, const TreeDesc<NJ, NCH> &d, int64_t b) {
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

    for (int slot = 0; slot < 3; ++slot) {
        const bool is_chain = slot < NCH;
        if (!is_chain && !(slot == 2 && a.prm.hasP)) continue;
        const int nj = is_chain ? NJ : 0;
        const int c = is_chain ? slot : 0;
        const ChainTable<NJ> &ct = d.chain[c];
        double R[9], p[3], zax[NJ][3], org[NJ][3];
        for (int k = 0; k < 9; ++k) R[k] = R1[k];
        for (int k = 0; k < 3; ++k) p[k] = p1[k];
        for (int j = 0; j < nj; ++j) {
            se3_compose_const(R, p, ct.pl[j], (a.prm.idmask[c] >> j) & 1);
            double s, cs;
            dsincos(a.q0[at(a.layout, a.B, a.nq, a.qidx[c][j], b)], s, cs);
            rot_z_right(R, s, cs);
            zax[j][0] = R[2]; zax[j][1] = R[5]; zax[j][2] = R[8];
            org[j][0] = p[0]; org[j][1] = p[1]; org[j][2] = p[2];
        }
        se3_compose_const(R, p, is_chain ? ct.fr : d.frP, is_chain ? ((a.prm.idmask[c] >> NJ) & 1) : (a.prm.idmaskP & 1));
        if (a.oMf_out) {
            for (int k = 0; k < 9; ++k) a.oMf_out[at(a.layout, a.B, a.ntasks * 12, a.tslot[slot] * 12 + k, b)] = R[k];
            for (int k = 0; k < 3; ++k) a.oMf_out[at(a.layout, a.B, a.ntasks * 12, a.tslot[slot] * 12 + 9 + k, b)] = p[k];
        }
        if (!a.e_out && !a.J_out) continue;
        double oMt[12];
        for (int k = 0; k < 12; ++k) oMt[k] = tl[(a.tslot[slot] * 12 + k) * ts];
        TaskTerms t;
        task_terms(R, p, oMt, is_chain ? ct.w : d.wP, (is_chain ? a.prm.unit[c] : a.prm.unitP) != 0, t);
        double JL[3][3], JA[3][6], col[NJ][6];
        base_columns(t, p, R1, p1, JL, JA);
        for (int j = 0; j < nj; ++j) {
            const double dj[3] = {org[j][0] - p[0], org[j][1] - p[1], org[j][2] - p[2]};
            double vw[3];
            cross(dj, zax[j], vw);
            for (int i = 0; i < 3; ++i) {
                col[j][i] = dfma(t.At[3 * i], vw[0], dfma(t.At[3 * i + 1], vw[1], dfma(t.At[3 * i + 2], vw[2],
                            dfma(t.Bt[3 * i], zax[j][0], dfma(t.Bt[3 * i + 1], zax[j][1], t.Bt[3 * i + 2] * zax[j][2])))));
                col[j][3 + i] = dfma(t.Ab[3 * i], zax[j][0], dfma(t.Ab[3 * i + 1], zax[j][1], t.Ab[3 * i + 2] * zax[j][2]));
            }
        }
        for (int r = 0; r < a.tdim[slot]; ++r) {
            const int lr = a.trow0[slot] + r, row = a.trow[slot] + r;
            if (a.e_out) a.e_out[at(a.layout, a.B, Mtot, row, b)] = t.e[lr];
            if (a.J_out) {
                for (int cc = 0; cc < a.nv; ++cc) a.J_out[at(a.layout, a.B, Mtot * a.nv, row * a.nv + cc, b)] = 0.0;
                for (int cc = 0; cc < 3; ++cc) {
                    a.J_out[at(a.layout, a.B, Mtot * a.nv, row * a.nv + cc, b)] = lr < 3 ? -JL[cc][lr] : 0.0;
                    a.J_out[at(a.layout, a.B, Mtot * a.nv, row * a.nv + 3 + cc, b)] = -JA[cc][lr];
                }
                for (int j = 0; j < nj; ++j) a.J_out[at(a.layout, a.B, Mtot * a.nv, row * a.nv + a.vidx[c][j], b)] = -col[j][lr];
            }
        }
    }
}

}  // namespace ikdev
