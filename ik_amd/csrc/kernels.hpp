// kernels.hpp -- host-visible launchers of the gfx950 kernels (defined in kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "problem.hpp"

namespace ikgpu {

// Work-queue heads of the persistent kernels (the cooperative generic kernels' groups of problems, the chain kernels' lane refill).
// A slot is two 64-bit words {head, workgroups that have left}: the LAST workgroup out of a launch zeroes both, so launches that
// follow each other on ONE stream share one slot with no host-side reset between them.  Launches on different streams may overlap,
// so every stream owns a slot; a launch recorded during stream capture gets a slot no live launch ever uses (the graph bakes the
// pointer in and may be replayed next to anything on any stream; a graph never overlaps itself).  Nobody else can hold a launch's
// slot: there is no ring to wrap around.
struct QueuePool {
    static constexpr int kChunkSlots = 256;     // the first chunk is allocated (and zeroed) when the problem is created
    std::mutex mu;
    std::vector<unsigned long long *> chunks;   // device allocations of kChunkSlots slots each
    int used_in_last = kChunkSlots;
    std::unordered_map<hipStream_t, unsigned long long *> by_stream;
    hipError_t grow();                           // one more zeroed chunk (not during stream capture: hipMalloc / hipMemset are not capturable)
    // the slot of a launch on `stream`, or nullptr with *err set.  A slot is FOUR words: {head, workgroups that have left, length of
    // the two-phase worklist, spare}; the last workgroup out of a refill launch zeroes the first three.
    unsigned long long *slot_for(hipStream_t stream, hipError_t *err);
    // the stream's worklist of a two-phase solve (kernels.hpp stop_rule_mode): n 32-bit problem indices, grown on demand and kept (two
    // solves on one stream are ordered; solves on different streams have their own).  Not during stream capture (hipMalloc).
    std::unordered_map<hipStream_t, std::pair<int32_t *, size_t>> lists;
    int32_t *worklist_for(hipStream_t stream, size_t n, hipError_t *err);
    void release();
};

struct DeviceTables {  // per-problem constant arrays resident in HBM
    double *lower = nullptr, *upper = nullptr;  // [nq]
    uint8_t *q_in_chain = nullptr;              // [nq]
    double *chain_desc = nullptr;               // ikdev::ChainDesc<NJ> / TreeDesc as a flat array of doubles
    int32_t *g_ints = nullptr;                  // generic kernel: packed int tables
    double *g_dbls = nullptr;                   // generic kernel: packed double tables
    mutable struct QueuePool queues;            // work-queue heads of the persistent kernels (below)
};

// The derived-visitor members of ikgpu_dls_params (include/ikgpu.h) into the loop parameters the generic lane program reads.
template <class LP>
inline void fill_visitor(LP &lp, const ikgpu_dls_params &prm) {
    lp.dq_sq_tol = prm.dq_sq_tol;
    lp.nlt = prm.num_level_tols;
    for (int l = 0; l < 8; ++l) lp.lvl_tol[l] = l < prm.num_level_tols ? prm.level_sq_tol[l] : 0.0;
}
inline bool visitor_extended(const ikgpu_dls_params &prm) { return prm.dq_sq_tol > 0.0 || prm.num_level_tols > 0; }

struct BatchIO {
    int64_t B;
    const double *q0;       // [nq x B]
    const double *targets;  // [ntasks x 12 x B]
    double *q_out;          // [nq x B]
    uint8_t *success;       // [B] or null
    int32_t *iters;         // [B] or null
    int layout;             // ikgpu_layout
};

// Returns hipSuccess or the launch error. Throws std::runtime_error for an un-instantiated shape.
hipError_t launch_dls_chain(const ProblemHost &ph, const DeviceTables &dt, const BatchIO &io,
                            const ikgpu_dls_params &prm, hipStream_t stream);
hipError_t launch_eval_chain(const ProblemHost &ph, const DeviceTables &dt, int64_t B, const double *q, const double *targets,
                             double *e_out, double *J_out, int layout, hipStream_t stream);
hipError_t launch_fk_chain(const ProblemHost &ph, const DeviceTables &dt, int64_t B, const double *q, double *oMf_out, int layout,
                           hipStream_t stream);
bool chain_shape_built(int nj, int type);
// The structure-specialised builds of the chain kernel (kernels_hot.hip, device/chain_hot.hpp): one Full task with unit
// weights on a chain whose placement-structure code has an instantiation.  launch_dls_chain takes that route when it exists.
bool chain_hot_built(const ProblemHost &ph);
// Which build a Chain problem launches (ProblemHost::chain_build), decided once at problem creation: 1 when chain_hot_built, else 2
// when the structure-specialised kernel could be compiled for this chain's structure code at run time (rtc.cpp; needs libhiprtc
// and, the first time, a few seconds), else 0 = the general build.  IKGPU_CHAIN_HOT=0 forces the general build, IKGPU_RTC=0 keeps
// run-time compilation off.  compile: false = plan only (ikgpu_problem_plan: names what WOULD run without compiling anything).
int select_chain_build(const ProblemHost &ph, bool compile);
// ",hot>" / ",hot-rtc>" / ",general>" appended to "dls_chain<NJ=..,type"
std::string chain_kernel_name(const ProblemHost &ph);
// Lane refill (device/chain_kernel_body.hpp chain_refill_loop): persistent one-wave workgroups whose lanes take the next unsolved
// problem as soon as their visitor stops them.  Taken for the stop-rule mode (stop_sq_tol >= 0, max_iterations >= 1) when the batch
// is larger than the lanes the device keeps resident; IKGPU_REFILL=0 never, =1 whenever the mode allows.  `resident_waves`: what
// hipOccupancyMaxActiveBlocksPerMultiprocessor reports for the refill kernel x the device's CUs (IKGPU_REFILL_WAVES_PER_CU overrides).
bool refill_wanted(const ikgpu_dls_params &prm, int64_t B, int64_t resident_waves);
// How a stop-rule solve of B problems runs (chain and tree kernels).  The kernel cannot know the targets' distribution in advance:
// with targets near the start every lane stops within 2-3 iterations and lock-step is 1.5-2x faster than persistent refill waves
// (whose loads are exposed); with far targets a few per cent of the problems run to max_iterations and lock-step is 2-4x slower.
// TWO PHASES get both: (1) the lock-step kernel over the whole batch, each WAVE leaving its loop once no more than a quarter of its
// lanes are still iterating (and at least K iterations are done) and appending the problems it leaves open to a worklist (one atomic per
// wave, no host synchronisation); (2) the refill kernel over the worklist, continuing each problem from its iterate in q_out at the
// iteration count the first phase left in iters[].  The switch point follows the batch: near targets never reach it (the list stays
// empty, the second launch finds nothing to do), far targets reach it when the bulk has converged and the stragglers fit the
// resident lanes.  Results are the lock-step kernel's bits (a problem's state is its q).
//   kStopLockStep: B fits the resident lanes, max_iterations < 1, never-stop visitor, or IKGPU_REFILL=0
//   kStopRefill:   IKGPU_REFILL=1, or (B large and) max_iterations <= K, or the stream is being captured (the worklist is a
//                  stream-ordered allocation)
//   kStopTwoPhase: B larger than the resident lanes (default), or IKGPU_REFILL=2
enum { kStopLockStep = 0, kStopRefill = 1, kStopTwoPhase = 2 };
int stop_rule_mode(const ikgpu_dls_params &prm, int64_t B, int64_t resident_waves, hipStream_t stream, bool tree);
// K: iterations before a wave of the first phase may leave (IKGPU_TWO_PHASE_ITERS overrides), and N: it does so with this many lanes or
// fewer still iterating (IKGPU_TWO_PHASE_ACTIVE, 1..63).  Defaults by kernel family, from tools/two_phase_probe.py on both target
// distributions at 262144 and 2^20 problems (profiles/r04_two_phase_probe_*.txt): the chain kernels K = 8, N = 32 (by then a tenth of a
// uniform batch is open and the list fits the resident lanes twice over; leaving earlier doubles the second phase), the tree kernels
// K = 4, N = 48 (their first phase runs one wave per SIMD in many rounds: an idle lane there is a lane lost, so they hand over early);
// the static lane programs take the tree kernels' (profiles/r04_refill_timing_static.txt).
int two_phase_iterations(bool tree);
int two_phase_active(bool tree);
struct TwoPhase {
    int32_t *worklist = nullptr;            // [B] (QueuePool::worklist_for)
    unsigned long long *count = nullptr;    // word 2 of the launch's queue slot: zero before the solve, zeroed again by the refill kernel's last wave
    uint8_t *success = nullptr;             // the caller's arrays, or stream-ordered temporaries when it passed none
    int32_t *iters = nullptr;
    void *tmp[2] = {nullptr, nullptr};
};
hipError_t two_phase_begin(QueuePool &queues, const BatchIO &io, hipStream_t stream, unsigned long long *queue, TwoPhase *tp);
hipError_t two_phase_end(TwoPhase *tp, hipStream_t stream);                                    // free the temporaries (stream-ordered)
// The two launches of a two-phase solve around a kernel-argument block `a` with the members {prm.max_iterations, success, iters,
// append_list, append_count, worklist, count, leave_active, leave_after} (ChainKernelArgs / TreeKernelArgs): lockstep() launches the problem's lock-step
// kernel with `a` -- it appends the problems it leaves unfinished to the list (one atomic per wave, in its epilogue) -- refill(queue)
// its refill kernel over that list.  The first phase writes every entry of q_out (also those outside the task supports, clipped once
// a step was taken); the second rewrites the rows a solve moves and the flags of the listed problems only: no pass-through launch,
// no compaction launch, no memset (the list's length lives in the queue slot, which the refill kernel's last wave zeroes).
template <class Args, class LockStepFn, class RefillFn>
hipError_t run_two_phase(QueuePool &queues, const BatchIO &io, hipStream_t stream, Args &a, bool tree, LockStepFn lockstep, RefillFn refill) {
    hipError_t qe = hipSuccess;
    unsigned long long *queue = queues.slot_for(stream, &qe);
    if (!queue) return qe;
    TwoPhase tp;
    hipError_t e = two_phase_begin(queues, io, stream, queue, &tp);
    if (e == hipSuccess) {
        a.success = tp.success; a.iters = tp.iters;
        a.worklist = nullptr; a.count = nullptr;
        a.leave_active = two_phase_active(tree); a.leave_after = two_phase_iterations(tree);
        a.append_list = tp.worklist; a.append_count = tp.count;
        lockstep();
        e = hipGetLastError();
        if (e == hipSuccess) {
            a.leave_active = 0; a.leave_after = 0;
            a.append_list = nullptr; a.append_count = nullptr;
            a.worklist = tp.worklist; a.count = tp.count;
            refill(queue);
            e = hipGetLastError();
        }
    }
    const hipError_t f = two_phase_end(&tp, stream);
    return e != hipSuccess ? e : f;
}
// the lock-step builds of the tree kernel a refill launch can mirror (kernels.hip run_dls_tree / kernels_tree_refill.hip)
enum { kTreeBuildGeneral = 0, kTreeBuildHot = 1, kTreeBuildMask = 2, kTreeBuildFold = 3 };
int64_t refill_grid(const void *kernel, int64_t B);
int64_t refill_resident(int64_t occupancy_waves, int64_t B);   // persistent waves of a refill launch given what the device can hold
int refill_chunk(int64_t B, int64_t grid);   // problems a wave reserves per pull from the head (IKGPU_REFILL_CHUNK overrides) | refill_batch() << 16
int refill_batch();                          // idle lanes a refill event waits for (IKGPU_REFILL_BATCH overrides)
// After a refill launch: the entries of q outside the chain (q0 clipped when iters > 0, else q0 -- reference ik/ik/dls.cpp:61-71),
// one thread per problem; `iters` is the launch's own iteration-count array (never null).
hipError_t launch_chain_pass_through(const ProblemHost &ph, const DeviceTables &dt, const BatchIO &io, const int32_t *iters, hipStream_t stream);
// Run-time compiled hot build (rtc.cpp): dls_chain_hot_kernel<NJ, code...> for THIS chain's structure code through hipRTC.
// available(compile = false): would it be attempted (libhiprtc loads, not disabled); (compile = true): compiled and loaded, or cached.
bool rtc_chain_hot_available(const ProblemHost &ph, bool compile);
// The generic lane program specialised for one problem at run time (rtc.cpp: tables as compile-time constants, workspace in
// registers).  available(compile = true) compiles / fetches it and returns the key rtc_launch_generic_static takes.
bool rtc_generic_static_available(const ProblemHost &gen, bool compile, uint64_t *key_out);
hipError_t rtc_launch_generic_static(const ProblemHost &gen, uint64_t key, const BatchIO &io, const ikgpu_dls_params &prm, hipStream_t stream,
                                     QueuePool *queues = nullptr);   // queues: where the refill program's work-queue slots come from (none: lock-step only)
int rtc_static_solve_rows(const ProblemHost &ph);   // rows of the linear system the static program solves (PostureTask rows eliminated when there are many)
bool rtc_generic_static_precompile_refill(const ProblemHost &gen);   // the refill program, ahead of the first large stop-rule batch
// ik::pik on a compiled lane program (device/pik_solver.hpp static_pik): available(compile = true) compiles / fetches the program for
// calls with (with_da) or without a secondary step; launch needs every lambda > 0 (the caller checks).
bool rtc_pik_static_available(const ProblemHost &gen, bool with_da, bool compile, uint64_t *key_out);
hipError_t rtc_launch_pik_static(const ProblemHost &gen, uint64_t key, const BatchIO &io, const ikgpu_pik_params &prm, hipStream_t stream);
std::string rtc_last_log();   // compiler log (or cache note) of the calling process's last run-time compilation attempt
hipError_t rtc_launch_chain_hot(const ProblemHost &ph, const DeviceTables &dt, const BatchIO &io, const ikgpu_dls_params &prm, hipStream_t stream);
hipError_t launch_targets_from_pose7(int64_t B, int ntasks, const double *pose7, double *targets12, int layout, hipStream_t stream);
int64_t persistent_grid(const void *kernel, int block, size_t lds, int64_t nblocks);
bool raise_lds_limit(const void *kernel, size_t lds);
hipError_t launch_dls_chain_hot(const ProblemHost &ph, const DeviceTables &dt, const BatchIO &io, const ikgpu_dls_params &prm,
                                hipStream_t stream);

}  // namespace ikgpu

namespace ikgpu {
// Free-flyer tree kernels (shape F).  dt.chain_desc holds ikdev::TreeDesc<NA, NB>.
// pik_lambda1 != nullptr: ik::pik with two priority levels in the shape tree_takes_two_level_pik() accepts -- prm.damping is
// lambda[0], *pik_lambda1 is lambda[1] (device/tree_solver.hpp PikRow).
hipError_t launch_dls_tree(const ProblemHost &ph, const DeviceTables &dt, const BatchIO &io,
                           const ikgpu_dls_params &prm, hipStream_t stream, const double *pik_lambda1 = nullptr);
// Level 0 = every frame task, among them a Full task on the base link with six non-zero weights; level 1 = the AlignAxisTask row.
bool tree_takes_two_level_pik(const ProblemHost &ph);
// e_out / J_out / oMf_out: any may be null.
hipError_t launch_eval_tree(const ProblemHost &ph, const DeviceTables &dt, int64_t B, const double *q,
                            const double *targets, double *e_out, double *J_out, double *oMf_out, int layout,
                            hipStream_t stream);
bool tree_shape_built(int nj, int nch);
// Generic fallback kernel.  The per-lane workspace (ws_words doubles x roundup(B, 64) lanes) is allocated and
// freed in stream order (hipMallocAsync / hipFreeAsync), so the call stays asynchronous and re-entrant.
// True when launch_dls_generic runs the cooperative LDS-resident program (device/coop_solver.hpp) for this problem: no
// constraints, no centre-of-mass task, four workspaces fit 64 KB of LDS, and IKGPU_GENERIC_KERNEL is not "lane".
bool generic_runs_cooperative(const ProblemHost &ph);
// force_lane: the per-lane memory-resident form whatever the problem's size (the derived-visitor family is implemented there and in
// the run-time specialised program, not in the cooperative kernel)
hipError_t launch_dls_generic(const ProblemHost &ph, const DeviceTables &dt, const BatchIO &io,
                              const ikgpu_dls_params &prm, hipStream_t stream, bool force_lane = false);
// ik::pik (reference ik/ik/pik.cpp:31-103): always the generic lane program; `gen` must be a Generic-kind analysis and
// dt.g_ints / dt.g_dbls its uploaded tables.
hipError_t launch_pik_generic(const ProblemHost &gen, const DeviceTables &dt, const BatchIO &io, const ikgpu_pik_params &prm,
                              hipStream_t stream);
hipError_t launch_eval_generic(const ProblemHost &ph, const DeviceTables &dt, int64_t B, const double *q,
                               const double *targets, double *e_out, double *J_out, double *oMf_out, int layout,
                               hipStream_t stream);
}  // namespace ikgpu
