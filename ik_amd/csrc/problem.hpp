// problem.hpp -- host-side flattening of (model, task table) into the constant tables the
// gfx950 kernels read.  Stands in for what InverseKinematicsProblem + dls_data hold between
// calls (reference ik/ik/problem.hpp:17-22,183-189; ik/ik/dls.hpp:36-52; ik/ik/data.cpp:8-23).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "ikgpu.h"
#include "model.hpp"

namespace ikgpu {

constexpr int kMaxChain = 8;

enum class KernelKind { Chain };

// Host copy of one serial chain (support of one task), axis-folded: every joint rotates about
// its local z.  Filled by analyse_chain().
struct ChainHost {
    int nj = 0;
    int qidx[kMaxChain] = {};
    int vidx[kMaxChain] = {};
    double pl[kMaxChain][12] = {};
    double frame_pl[12] = {};
    double lo[kMaxChain] = {}, hi[kMaxChain] = {};
};

struct ProblemHost {
    KernelKind kind = KernelKind::Chain;
    std::string kernel_name;
    int nq = 0, nv = 0, ntasks = 0, rows = 0;
    std::vector<ikgpu_task> tasks;
    // chain kernels (single task)
    ChainHost chain;
    double ref_pl[12] = {};  // world placement of the (fixed) reference frame
    std::vector<uint8_t> q_in_chain;  // [nq] 1 if the entry is integrated by the kernel
    std::vector<double> lower, upper;
};

// Throws std::runtime_error (unsupported shapes say so explicitly).
ProblemHost analyse_problem(const Model &m, const ikgpu_task *tasks, int ntasks);

// ikdev::ChainDesc<nj> as the flat array of doubles the kernels stage into LDS.
std::vector<double> chain_desc_table(const ProblemHost &ph);
// The per-problem scalars of ikdev::ChainKernelArgs<nj>.
void fill_chain_args(const ProblemHost &ph, double *ref_pl12, int *qidx, int *vidx, int *nq, int *nv, int *priority);

}  // namespace ikgpu
