// chain_kernel_body.hpp -- what ONE lane of the chain kernels does around the lane solver:
// load its problem from HBM, solve, store.  Shared by the __global__ wrappers (kernels.hip) and
// by the CPU lane emulator under tests/ (which runs it in a plain loop over b).
#pragma once
#if !defined(__HIPCC_RTC__)
#include <cstdint>
#endif

#include "chain_solver.hpp"

namespace ikdev {

enum : int { LAYOUT_SOA = 0, LAYOUT_AOS = 1 };  // == ikgpu_layout

template <int NJ>
struct ChainKernelArgs {
    const ChainDesc<NJ> *desc;  // HBM copy of the chain table (uploaded at problem creation)
    LoopParams prm;
    double ref_pl[12];          // world placement of the task's (world-fixed) reference frame
    int qidx[NJ];               // index of each chain joint in q
    int vidx[NJ];               // ... and in the tangent vector
    int nq, nv, layout;
    int64_t B;
    const double *q0;
    const double *targets;
    double *q_out;
    uint8_t *success;
    int32_t *iters;
    const double *lower, *upper;  // [nq]
    const uint8_t *q_in_chain;    // [nq]
    double *e_out, *J_out, *oMf_out;  // stage kernels
    // second phase of a two-phase stop-rule solve (kernels.hip two_phase_*): the refill kernel walks a LIST of problems -- those the
    // lock-step first phase left unfinished -- continuing each from its iterate in q_out at the iteration count the first phase left
    // in iters[].  Null: the whole batch.
    const int32_t *worklist;
    const unsigned long long *count;   // (device) length of the list
    // ... and of its first phase: a wave of the lock-step kernel LEAVES its loop once no more than `leave_active` of its lanes are still
    // iterating (and at least `leave_after` iterations are done; KeepGoing below) and appends the problems it leaves unfinished
    // (append_unfinished below).  leave_active == 0: an ordinary lock-step solve.
    int leave_active, leave_after;
    int32_t *append_list;
    unsigned long long *append_count;
};

// First phase of a two-phase solve: the lanes of a wave whose problem is still open after the lock-step iterations append its index to
// the list -- one atomic per wave.  Called by EVERY lane of the wave (tail lanes with open = false).
IKD_FN void append_unfinished(int32_t *list, unsigned long long *count, bool open, int64_t b) {
#if IKD_ON_DEVICE
    const unsigned long long mask = __ballot(open);
    if (mask == 0ull) return;
    const int lane = static_cast<int>(threadIdx.x) & 63;
    unsigned long long base = 0;
    if (lane == __ffsll(static_cast<long long>(mask)) - 1) base = atomicAdd(count, static_cast<unsigned long long>(__popcll(mask)));
    const int src = __ffsll(static_cast<long long>(mask)) - 1;
    const unsigned lo = __shfl(static_cast<unsigned>(base), src), hi = __shfl(static_cast<unsigned>(base >> 32), src);
    base = (static_cast<unsigned long long>(hi) << 32) | lo;
    if (open) list[base + __popcll(mask & ((1ull << lane) - 1ull))] = static_cast<int32_t>(b);
#else
    if (open) list[(*count)++] = static_cast<int32_t>(b);
#endif
}

IKD_FN int64_t at(int layout, int64_t B, int ncomp, int c, int64_t b) {
    return layout == LAYOUT_SOA ? static_cast<int64_t>(c) * B + b : b * ncomp + c;
}

// oMt = oMr * target (reference ik/ik/frame.hpp:48); oMr is constant for a world-fixed reference.
template <int NJ>
IKD_FN void load_target(const ChainKernelArgs<NJ> &a, int64_t b, double (&oMt)[12]) {
    double t[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) t[k] = a.targets[at(a.layout, a.B, 12, k, b)];
    const double *r = a.ref_pl;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
            oMt[3 * i + j] = dfma(r[3 * i], t[j], dfma(r[3 * i + 1], t[3 + j], r[3 * i + 2] * t[6 + j]));
        oMt[9 + i] = dfma(r[3 * i], t[9], dfma(r[3 * i + 1], t[10], dfma(r[3 * i + 2], t[11], r[9 + i])));
    }
}

// B independent ik::dls() calls (reference ik/ik/dls.cpp:5-78), lane `gid`.
template <int NJ, int KT, int SMASK = -1, class Desc, class AnyFn>
IKD_FN void dls_chain_body(const ChainKernelArgs<NJ> &a, const Desc &d, int64_t gid, AnyFn any_active) {
    const bool valid = gid < a.B;
    const int64_t b = valid ? gid : a.B - 1;  // tail lanes shadow the last problem and store nothing

    double q[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) q[j] = a.q0[at(a.layout, a.B, a.nq, a.qidx[j], b)];
    double oMt[12];
    load_target(a, b, oMt);

    int iters;
    bool success;
    chain_dls<NJ, KT, SMASK>(d, a.prm, q, oMt, iters, success, any_active);

    // (wave-uniform test; a problem that ran out of iterations is finished, not open)
    if (a.append_count) append_unfinished(a.append_list, a.append_count, valid && !success && iters < a.prm.max_iterations, b);
    if (!valid) return;
#pragma unroll
    for (int j = 0; j < NJ; ++j) a.q_out[at(a.layout, a.B, a.nq, a.qidx[j], b)] = q[j];
    // Entries outside the task support: dq = 0 there, so the loop only ever clamps them
    // (reference ik/ik/dls.cpp:71 clips the whole q after each step; no step is taken when iters == 0).
    for (int i = 0; i < a.nq; ++i) {
        if (a.q_in_chain[i]) continue;
        const double v = a.q0[at(a.layout, a.B, a.nq, i, b)];
        const double c = dmin(a.upper[i], dmax(v, a.lower[i]));
        a.q_out[at(a.layout, a.B, a.nq, i, b)] = (iters > 0) ? c : v;
    }
    if (a.success) a.success[b] = success ? 1 : 0;
    if (a.iters) a.iters[b] = iters;
}

#if IKD_HIP_LANG
// ---- lane refill: the stop-rule mode without lock-step divergence ----------------------------------------------------------------
// With the reference's default visitor (ik/ik/visitor.hpp:15-21) a problem stops after a handful of iterations or never
// (max_iterations, ik/ik/dls.cpp:76-77); in the lock-step kernels a wave runs until its LAST lane stops, so one lane that never
// converges keeps 63 finished ones idle.  Here a wave is persistent: a lane whose visitor fired (or whose iteration count reached
// max_iterations) stores its result and takes the next unsolved problem from the launch's queue head; the wave leaves when the
// queue is exhausted and every lane is idle.  A lane's state is (q, target, it): the iteration is the same code as the lock-step
// loop's, so results are bit-identical whatever the batch composition.  Worth it when the batch is larger than the machine
// (B > resident lanes); at B = resident lanes every problem has its own lane anyway and the launch lasts as long as its slowest problem.
//
// queue[0]: head -- problems handed out (in chunks) beyond the first (static) round, queue[1]: waves that have left; the last wave out
// zeroes both, so the next launch on the same stream finds the slot clean (host: QueuePool in kernels.hpp).
// iterate(q, oMt, have) runs ONE iteration of this lane's problem in place and returns "the visitor stopped it before the step".
template <int NJ, class IterFn>
__device__ __forceinline__ void chain_refill_loop(const ChainKernelArgs<NJ> &a, unsigned long long *queue, int chunk_and_batch, IterFn iterate) {
    const int lane = static_cast<int>(threadIdx.x) & 63;                 // one wave per workgroup
    const int64_t first_round = static_cast<int64_t>(gridDim.x) * 64;
    // work items: the batch's problems, or (second phase) the entries of the worklist; `b` is always a PROBLEM index
    const bool listed = a.worklist != nullptr;
    const int64_t nwork = listed ? static_cast<int64_t>(*a.count) : a.B;
    const double *qsrc = listed ? a.q_out : a.q0;
    auto problem = [&](int64_t w) { return listed ? static_cast<int64_t>(a.worklist[w]) : w; };
    const int64_t w0 = static_cast<int64_t>(blockIdx.x) * 64 + lane;
    bool have = w0 < nwork;
    double q[NJ], oMt[12];
    int64_t b = 0;
    int it = 0;
    const int max_it = a.prm.max_iterations;                             // >= 1: the host sends max_iterations == 0 to the lock-step kernel
    // Waves without a share of the first round never touch the queue: they leave at once, and the slot's bookkeeping below counts the
    // others only.  (An EMPTY list -- every problem stopped in the first phase -- costs the launch and one load per wave: the slot is
    // still zero.  A thousand arrivals on one address took ~15 us, a quarter of a near-target solve.)
    const int64_t working = (nwork + 63) / 64 < static_cast<int64_t>(gridDim.x) ? (nwork + 63) / 64 : static_cast<int64_t>(gridDim.x);
    if (static_cast<int64_t>(blockIdx.x) >= working) return;
    b = problem(have ? w0 : 0);                                          // idle lanes shadow a valid problem
#pragma unroll
    for (int j = 0; j < NJ; ++j) q[j] = qsrc[at(a.layout, a.B, a.nq, a.qidx[j], b)];
    load_target(a, b, oMt);
    if (listed) it = a.iters[b];                                         // (the iterations the first phase took over this problem)
    // The wave's own reserve [pool_lo, pool_hi) of unsolved problems (wave-uniform): finished lanes are served from it, and only when
    // it runs dry does the wave pull `chunk` (>= 64) more from the launch's head.  One atomic per finished LANE-GROUP on one address
    // saturated the memory-side atomic unit (measured: ~30-70 M same-address atomics/s device-wide, every refill waiting ~30 us).
    int64_t pool_lo = 0, pool_hi = 0;
    bool exhausted = first_round >= nwork;                               // the head has passed the end of the batch
    // Refill events are BATCHED: a finished lane stores its result at once and then waits until `batch` lanes of the wave are idle
    // (or no lane is active any more) -- one ballot, one reserve update and one round trip of loads per event instead of per
    // iteration.  With targets near the start every lane is done within two or three iterations, an event per iteration cost
    // as much as the iterations themselves (round 3: refill 1.5-2x slower than lock-step there); with far targets the wave's
    // stragglers set the pace either way.  chunk_and_batch: chunk | batch << 16 (kernels.hip refill_chunk).
    const int batch = chunk_and_batch >> 16, chunk = chunk_and_batch & 0xffff;
    while (__any(have)) {
        const bool stop_now = iterate(q, oMt, have) && have;
        ++it;
        const bool done = have && (stop_now || it >= max_it);
        if (done) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) a.q_out[at(a.layout, a.B, a.nq, a.qidx[j], b)] = q[j];
            if (a.success) a.success[b] = stop_now ? 1 : 0;
            a.iters[b] = stop_now ? it - 1 : max_it;                     // never null here: the pass-through kernel reads it
            have = false;
        }
        const unsigned long long mask = __ballot(!have);                 // idle lanes: finished (or never had a problem)
        const int need = __popcll(mask);
        const bool supply = pool_hi > pool_lo || !exhausted;             // (wave-uniform)
        if (supply && (need >= batch || need == 64)) {
            const int rank = __popcll(mask & ((1ull << lane) - 1ull));
            const int64_t avail = pool_hi - pool_lo;
            int64_t nb = pool_lo + rank;                                 // ranks below `avail` are served from the reserve
            bool got = rank < avail;
            if (avail < need && !exhausted) {                            // wave-uniform: pull the next chunk, serve the other ranks from it
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
            if (!have && got) {
                have = true;
                b = problem(nb);
#pragma unroll
                for (int j = 0; j < NJ; ++j) q[j] = qsrc[at(a.layout, a.B, a.nq, a.qidx[j], b)];
                load_target(a, b, oMt);
                it = listed ? a.iters[b] : 0;
            }
        }
    }
    // the last wave out resets the slot (every wave's atomics on queue[0] are ordered before its own arrival on queue[1])
    if (lane == 0) {
        __threadfence();
        if (atomicAdd(queue + 1, 1ull) == static_cast<unsigned long long>(working) - 1ull) {
            queue[0] = 0ull;
            queue[1] = 0ull;
            queue[2] = 0ull;   // (the two-phase worklist's length)
            __threadfence();
        }
    }
}

// The general chain program under lane refill (device/chain_solver.hpp chain_dls, one iteration at a time).
template <int NJ, int KT, int SMASK, class Desc>
__device__ __forceinline__ void dls_chain_refill_body(const ChainKernelArgs<NJ> &a, const Desc &d_in, unsigned long long *queue, int chunk) {
    const Desc *dp = &d_in;
    chain_refill_loop<NJ>(a, queue, chunk, [&](double (&q)[NJ], const double (&oMt)[12], bool) {
        asm volatile("" ::: "memory");
        if constexpr (!std::is_same<Desc, ChainDesc<NJ>>::value) IKD_LAUNDER(dp);   // see chain_dls
        return chain_iteration<NJ, KT, SMASK>(*dp, a.prm, q, oMt, true);
    });
}

#endif  // IKD_HIP_LANG

// evaluate_problem_data + stacking (reference ik/ik/data.cpp:25-58, ik/ik/dls.cpp:18-24):
// e [M x B], dense J [M x nv x B] (zero outside the support, as the reference's zero-initialised
// frame Jacobian, ik/ik/frame.hpp:110).
template <int NJ, int KT>
IKD_FN void eval_chain_body(const ChainKernelArgs<NJ> &a, const ChainDesc<NJ> &d, int64_t b) {
    constexpr int M = TaskDim<KT>::value;
    if (b >= a.B) return;
    double q[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) q[j] = a.q0[at(a.layout, a.B, a.nq, a.qidx[j], b)];
    double oMt[12];
    load_target(a, b, oMt);
    double e[M], col[NJ][M], Rf[9], pf[3];
    chain_evaluate<NJ, KT>(d, q, oMt, a.prm.idmask, a.prm.unit_weights != 0, e, col, Rf, pf);
#pragma unroll
    for (int r = 0; r < M; ++r) a.e_out[at(a.layout, a.B, M, r, b)] = e[r];
    if (a.J_out) {
        for (int r = 0; r < M; ++r)
            for (int c = 0; c < a.nv; ++c) a.J_out[at(a.layout, a.B, M * a.nv, r * a.nv + c, b)] = 0.0;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < M; ++r) a.J_out[at(a.layout, a.B, M * a.nv, r * a.nv + a.vidx[j], b)] = -col[j][r];
    }
}

// World placement of the task frame (reference ik/ik/data.cpp:28-29, one entry of data.oMf).
template <int NJ>
IKD_FN void fk_chain_body(const ChainKernelArgs<NJ> &a, const ChainDesc<NJ> &d, int64_t b) {
    if (b >= a.B) return;
    double R[9], p[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = d.pl[0][k];
#pragma unroll
    for (int k = 0; k < 3; ++k) p[k] = d.pl[0][9 + k];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        if (j > 0) se3_compose_const(R, p, d.pl[j], (a.prm.idmask >> j) & 1);
        double s, c;
        dsincos(a.q0[at(a.layout, a.B, a.nq, a.qidx[j], b)], s, c);
        rot_z_right(R, s, c);
    }
    se3_compose_const(R, p, d.frame_pl, (a.prm.idmask >> NJ) & 1);
#pragma unroll
    for (int k = 0; k < 9; ++k) a.oMf_out[at(a.layout, a.B, 12, k, b)] = R[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) a.oMf_out[at(a.layout, a.B, 12, 9 + k, b)] = p[k];
}

}  // namespace ikdev
