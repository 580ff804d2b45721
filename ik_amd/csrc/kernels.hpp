// kernels.hpp -- host-visible launchers of the gfx950 kernels (defined in kernels.hip).
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "problem.hpp"

namespace ikgpu {

struct DeviceTables {  // per-problem constant arrays resident in HBM
    double *lower = nullptr, *upper = nullptr;  // [nq]
    uint8_t *q_in_chain = nullptr;              // [nq]
    double *chain_desc = nullptr;               // ikdev::ChainDesc<NJ> / TreeDesc as a flat array of doubles
    int32_t *g_ints = nullptr;                  // generic kernel: packed int tables
    double *g_dbls = nullptr;                   // generic kernel: packed double tables
    // cooperative kernels: work-queue heads of the persistent workgroups, one slot per launch in flight (a ring: the launch zeroes
    // its slot on its stream, so launches of one problem on several streams -- or replays of a captured graph -- do not share one)
    static constexpr int kQueueSlots = 64;
    unsigned long long *queue = nullptr;        // [kQueueSlots]
    mutable std::atomic<unsigned> queue_next{0};
};

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
hipError_t launch_dls_generic(const ProblemHost &ph, const DeviceTables &dt, const BatchIO &io,
                              const ikgpu_dls_params &prm, hipStream_t stream);
// ik::pik (reference ik/ik/pik.cpp:31-103): always the generic lane program; `gen` must be a Generic-kind analysis and
// dt.g_ints / dt.g_dbls its uploaded tables.
hipError_t launch_pik_generic(const ProblemHost &gen, const DeviceTables &dt, const BatchIO &io, const ikgpu_pik_params &prm,
                              hipStream_t stream);
hipError_t launch_eval_generic(const ProblemHost &ph, const DeviceTables &dt, int64_t B, const double *q,
                               const double *targets, double *e_out, double *J_out, double *oMf_out, int layout,
                               hipStream_t stream);
}  // namespace ikgpu
