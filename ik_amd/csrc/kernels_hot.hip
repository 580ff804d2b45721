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

constexpr int kBlock = 64;  // one wave64 per workgroup: 1024 workgroups at B = 65536 cover 256 CUs x 4 SIMDs

using ikdev::ChainKernelArgs;
using ikdev::ChainStruct;
using ikdev::HotTable;

template <int NJ, uint64_t C0, uint64_t C1, uint64_t C2, bool NEVERSTOP>
__global__ __launch_bounds__(kBlock) void dls_chain_hot_kernel(const ChainKernelArgs<NJ> a, const HotTable t) {
    ikdev::hot_kernel_entry<NJ, ChainStruct<C0, C1, C2>, NEVERSTOP>(a, t);
}

// The same program with lane refill (device/chain_kernel_body.hpp chain_refill_loop): the stop-rule mode on batches larger than the machine.
// Two waves per SIMD asked for (<= 256 registers): a wave that refills waits for its gathered loads (~2 us, most iterations have a
// lane that finishes) and the other wave of the SIMD computes meanwhile.
#ifndef IKGPU_HOT_REFILL_WAVES
#define IKGPU_HOT_REFILL_WAVES 2   // (tools/build_variant.sh onewave -DIKGPU_HOT_REFILL_WAVES=1: the A/B of DESIGN.md section 3.1)
#endif
template <int NJ, uint64_t C0, uint64_t C1, uint64_t C2>
__global__ __launch_bounds__(kBlock, IKGPU_HOT_REFILL_WAVES) void dls_chain_hot_refill_kernel(const ChainKernelArgs<NJ> a, const HotTable t, unsigned long long *queue, int chunk) {
    ikdev::hot_refill_entry<NJ, ChainStruct<C0, C1, C2>>(a, t, queue, chunk);
}

// X(NJ, code0, code1, code2)
#define IKGPU_HOT_SHAPES(X)                                                                                                   \
    X(7, 0x04f0208cce8c7664ull, 0x395959cacad65656ull, 0x000001cacace5656ull) /* Cassie leg: Left / RightFootFront, 22 values */ \
    X(6, 0x695959272b925656ull, 0x47655a33aaca549cull, 0x0000000000121256ull) /* UR5 / UR10 tool0, 14 values */

}  // namespace

// The hot program's preconditions: one Full FrameTask with unit weights on a chain whose structure code fits three words.
static bool chain_hot_eligible(const ProblemHost &ph) {
    if (ph.kind != KernelKind::Chain || ph.ntasks != 1 || ph.tasks[0].type != IKGPU_FULL || !task_has_unit_weights(ph.tasks[0])) return false;
    const char *env = std::getenv("IKGPU_CHAIN_HOT");
    if (env && std::strcmp(env, "0") == 0) return false;   // A/B switch: the general chain kernel
    return ph.chain_struct.fits;
}

// A lane-refill launch of a hot build: queue slot, an iteration-count array when the caller passed none, the kernel, then the
// entries of q outside the chain (kernels.hip launch_chain_pass_through).
template <int NJ, class LaunchFn>
static hipError_t hot_refill_launch(const ProblemHost &ph, const DeviceTables &dt, const BatchIO &io, ChainKernelArgs<NJ> &a, hipStream_t stream, LaunchFn launch) {
    hipError_t e = hipSuccess;
    unsigned long long *queue = dt.queues.slot_for(stream, &e);
    if (!queue) return e;
    void *tmp = nullptr;
    if (!a.iters) {
        if ((e = hipMallocAsync(&tmp, sizeof(int32_t) * static_cast<size_t>(io.B), stream)) != hipSuccess) return e;
        a.iters = static_cast<int32_t *>(tmp);
    }
    launch(queue);
    e = hipGetLastError();
    if (e == hipSuccess) e = launch_chain_pass_through(ph, dt, io, a.iters, stream);
    if (tmp) {
        const hipError_t f = hipFreeAsync(tmp, stream);
        if (e == hipSuccess) e = f;
    }
    return e;
}

bool chain_hot_built(const ProblemHost &ph) {
    if (!chain_hot_eligible(ph)) return false;
    const ChainStructure &s = ph.chain_struct;
    if (!s.fits) return false;
#define X(N, K0, K1, K2) \
    if (ph.chain.nj == N && s.code[0] == K0 && s.code[1] == K1 && s.code[2] == K2) return true;
    IKGPU_HOT_SHAPES(X)
#undef X
    return false;
}

int select_chain_build(const ProblemHost &ph, bool compile) {
    if (chain_hot_built(ph)) return 1;
    if (chain_hot_eligible(ph) && rtc_chain_hot_available(ph, compile)) return 2;
    return 0;
}

std::string chain_kernel_name(const ProblemHost &ph) {
    std::string n = ph.kernel_name;
    const size_t cut = n.rfind(',');
    if (cut == std::string::npos) return n;
    return n.substr(0, cut) + (ph.chain_build == 1 ? ",hot>" : ph.chain_build == 2 ? ",hot-rtc>" : ",general>");
}

hipError_t launch_dls_chain_hot(const ProblemHost &ph, const DeviceTables &dt, const BatchIO &io, const ikgpu_dls_params &prm,
                                hipStream_t stream) {
    if (ph.chain_build == 2) return rtc_launch_chain_hot(ph, dt, io, prm, stream);
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
        else {                                                                                                           \
            const void *rk = reinterpret_cast<const void *>(dls_chain_hot_refill_kernel<N, K0, K1, K2>);                   \
            const int64_t rgrid = refill_grid(rk, io.B);                                                                 \
            const int mode = stop_rule_mode(prm, io.B, rgrid, stream, false);                                                  \
            if (mode == kStopRefill) return hot_refill_launch<N>(ph, dt, io, a, stream, [&](unsigned long long *queue) {  \
                hipLaunchKernelGGL((dls_chain_hot_refill_kernel<N, K0, K1, K2>), dim3(static_cast<unsigned>(rgrid)), dim3(kBlock), 0, stream, a, t, queue, refill_chunk(io.B, rgrid)); \
            });                                                                                                          \
            if (mode == kStopTwoPhase) return run_two_phase(dt.queues, io, stream, a, false, [&] {                                    \
                hipLaunchKernelGGL((dls_chain_hot_kernel<N, K0, K1, K2, false>), grid, dim3(kBlock), 0, stream, a, t);    \
            }, [&](unsigned long long *queue) {                                                                          \
                hipLaunchKernelGGL((dls_chain_hot_refill_kernel<N, K0, K1, K2>), dim3(static_cast<unsigned>(rgrid)), dim3(kBlock), 0, stream, a, t, queue, refill_chunk(io.B, rgrid)); \
            });                                                                                                          \
            hipLaunchKernelGGL((dls_chain_hot_kernel<N, K0, K1, K2, false>), grid, dim3(kBlock), 0, stream, a, t);          \
        }                                                                                                                \
        return hipGetLastError();                                                                                        \
    }
    IKGPU_HOT_SHAPES(X)
#undef X
    return hipErrorInvalidValue;
}

}  // namespace ikgpu
