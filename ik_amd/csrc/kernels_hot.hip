// kernels_hot.hip -- the structure-specialised builds of the headline chain kernel (device/chain_hot.hpp): one Full FrameTask
// with unit weights on a fixed-base serial chain, i.e. BASELINE.json's configs 1, 2, 4 (Cassie leg) and 5 (UR arm).
//
// This translation unit is compiled with -fno-signed-zeros -fno-honor-nans -fno-honor-infinities (see the Makefile): the lane
// program multiplies by LITERAL 0.0 / +-1.0 where a placement entry is structurally zero / one, and those flags let the
// compiler fold such products (x * 0.0 -> 0.0 needs "no NaN, no signed zero") -- no reassociation, no reciprocal or
// approximate maths, contraction as in kernels.hip: on finite data the results are the bits of the full products.
//
// The instantiation list is generated: tools/print_struct_codes.cpp prints the placement-structure code (ikgpu::chain_structure)
// of the fixture models' chains.  A chain whose code is not in the list runs on the general chain kernel (kernels.hip).
#include "kernels.hpp"

#include <cstdlib>
#include <cstring>

#include "device/chain_hot.hpp"

namespace ikgpu {
namespace {

#ifndef IKGPU_HOT_PIN_LO
// the placement values are parked in vector registers, the joint limits stay in scalar registers (A/B on one box, B = 65536:
// everything in VGPRs 0.1440 ms -- 22 v_accvgpr_read per iteration --, limits in SGPRs 0.1417 ms)
#define IKGPU_HOT_PIN_LO 0
#define IKGPU_HOT_PIN_HI (S::offset(NJ + 1))
#endif
constexpr int kBlock = 64;  // one wave64 per workgroup: 1024 workgroups at B = 65536 cover 256 CUs x 4 SIMDs

using ikdev::ChainKernelArgs;
using ikdev::ChainStruct;
using ikdev::HotTable;

template <int NJ, uint64_t C0, uint64_t C1, uint64_t C2, bool NEVERSTOP>
__global__ __launch_bounds__(kBlock) void dls_chain_hot_kernel(const ChainKernelArgs<NJ> a, const HotTable t) {
    typedef ChainStruct<C0, C1, C2> S;
    const int64_t gid = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    // The compact table (<= 36 doubles for the shapes built here) arrives in the kernel-argument segment and is parked in
    // vector registers for the whole loop: in scalar registers it competes with the ~40 polynomial constants for the 100
    // SGPRs (31 v_readlane + 39 s_mov of spill code per iteration in the round-1 kernel); a lone wave has 512 VGPRs to itself.
    HotTable tv;
    constexpr int kUsed = S::offset(NJ + 1) + 2 * NJ;
    static_assert(kUsed <= ikdev::kHotTableMax, "compact table too long");
#pragma unroll
    for (int k = 0; k < ikdev::kHotTableMax; ++k) {
        tv.v[k] = k < kUsed ? t.v[k] : 0.0;
        if (k >= IKGPU_HOT_PIN_LO && k < kUsed && k < IKGPU_HOT_PIN_HI) IKD_PIN(tv.v[k]);
    }
    ikdev::hot_chain_body<NJ, S, NEVERSTOP>(a, tv, gid, [](bool act) { return __any(act) != 0; });
}

// X(NJ, code0, code1, code2)
#define IKGPU_HOT_SHAPES(X)                                                                                                   \
    X(7, 0x04f0208cce8c7664ull, 0x395959cacad65656ull, 0x000001cacace5656ull) /* Cassie leg: Left / RightFootFront, 22 values */ \
    X(6, 0x695959272b925656ull, 0x47655a33aaca549cull, 0x0000000000121256ull) /* UR5 / UR10 tool0, 14 values */

}  // namespace

bool chain_hot_built(const ProblemHost &ph) {
    if (ph.kind != KernelKind::Chain || ph.ntasks != 1 || ph.tasks[0].type != IKGPU_FULL || !task_has_unit_weights(ph.tasks[0])) return false;
    const char *env = std::getenv("IKGPU_CHAIN_HOT");
    if (env && std::strcmp(env, "0") == 0) return false;   // A/B switch: the general chain kernel
    const ChainStructure &s = ph.chain_struct;
    if (!s.fits) return false;
#define X(N, K0, K1, K2) \
    if (ph.chain.nj == N && s.code[0] == K0 && s.code[1] == K1 && s.code[2] == K2) return true;
    IKGPU_HOT_SHAPES(X)
#undef X
    return false;
}

hipError_t launch_dls_chain_hot(const ProblemHost &ph, const DeviceTables &dt, const BatchIO &io, const ikgpu_dls_params &prm,
                                hipStream_t stream) {
    const ChainStructure &s = ph.chain_struct;
    const std::vector<double> &tab = ph.chain_hot;
    HotTable t{};
    if (tab.size() > static_cast<size_t>(ikdev::kHotTableMax)) return hipErrorInvalidValue;
    std::memcpy(t.v, tab.data(), tab.size() * sizeof(double));
    const dim3 grid(static_cast<unsigned>((io.B + kBlock - 1) / kBlock));
#define X(N, K0, K1, K2)                                                                                                 \
    if (ph.chain.nj == N && s.code[0] == K0 && s.code[1] == K1 && s.code[2] == K2) {                                        \
        ChainKernelArgs<N> a{};                                                                                          \
        fill_chain_args(ph, a.ref_pl, a.qidx, a.vidx, &a.nq, &a.nv, &a.prm.priority, &a.prm.idmask, &a.prm.unit_weights); \
        a.lower = dt.lower; a.upper = dt.upper; a.q_in_chain = dt.q_in_chain;                                            \
        a.prm.max_iterations = prm.max_iterations;                                                                       \
        a.prm.lam2 = prm.damping * prm.damping;                                                                          \
        a.prm.step_length = prm.step_length;                                                                             \
        a.prm.stop_sq_tol = prm.stop_sq_tol;                                                                             \
        a.layout = io.layout; a.B = io.B; a.q0 = io.q0; a.targets = io.targets;                                          \
        a.q_out = io.q_out; a.success = io.success; a.iters = io.iters;                                                  \
        if (prm.stop_sq_tol < 0.0) hipLaunchKernelGGL((dls_chain_hot_kernel<N, K0, K1, K2, true>), grid, dim3(kBlock), 0, stream, a, t); \
        else hipLaunchKernelGGL((dls_chain_hot_kernel<N, K0, K1, K2, false>), grid, dim3(kBlock), 0, stream, a, t);          \
        return hipGetLastError();                                                                                        \
    }
    IKGPU_HOT_SHAPES(X)
#undef X
    return hipErrorInvalidValue;
}

}  // namespace ikgpu
