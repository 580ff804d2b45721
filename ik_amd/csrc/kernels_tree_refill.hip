// kernels_tree_refill.hip -- the free-flyer tree kernels under lane refill (device/tree_kernel_body.hpp TreeRefill): the stop-rule
// mode of BASELINE.json's config 3 (Cassie full body) and of the general tree build on batches larger than the machine.  Persistent
// two-wave workgroups (the LDS park of chain 0's factor is per wave, as in kernels.hip); a lane whose visitor fired (reference
// ik/ik/visitor.hpp:15-21, ik/ik/dls.cpp:61-64) or whose iteration count reached max_iterations (dls.cpp:76-77) stores its result
// and takes the next unsolved problem; results are bit-identical to the lock-step kernels' (same lane program).
// A separate translation unit: the tree kernels are the slowest to compile.
#include "kernels.hpp"

#include <algorithm>
#include <cstdlib>

#include "device/tree_kernel_body.hpp"

namespace ikgpu {
namespace {

using ikdev::LegFactor;
using ikdev::TreeDesc;
using ikdev::TreeKernelArgs;

constexpr int kTreeWaves = 2;
constexpr int kTreeBlock = 64 * kTreeWaves;

template <int NJ>
struct LdsPark {   // (as kernels.hip)
    static constexpr int kL = NJ * (NJ + 1) / 2;
    static constexpr int kEntries = kL + NJ * 6 + NJ;
    double (*buf)[64];
    int lane;
    __device__ __forceinline__ void store(const LegFactor<NJ> &F) const {
#pragma unroll
        for (int e = 0; e < kL; ++e) buf[e][lane] = F.L[e];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#pragma unroll
            for (int c = 0; c < 6; ++c) buf[kL + j * 6 + c][lane] = F.W[j][c];
            buf[kL + NJ * 6 + j][lane] = F.u[j];
        }
    }
    __device__ __forceinline__ void load(LegFactor<NJ> &F) const {
#pragma unroll
        for (int e = 0; e < kL; ++e) F.L[e] = buf[e][lane];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#pragma unroll
            for (int c = 0; c < 6; ++c) F.W[j][c] = buf[kL + j * 6 + c][lane];
            F.u[j] = buf[kL + NJ * 6 + j][lane];
        }
    }
};

template <int NJ, int NCH, int SPEC>
__global__ __launch_bounds__(kTreeBlock) void dls_tree_refill_kernel(const TreeKernelArgs<NJ, NCH> a, unsigned long long *queue, int chunk) {
    __shared__ double lds_park[kTreeWaves][NCH > 1 ? LdsPark<NJ>::kEntries : 1][64];
    __shared__ double lds_desc[sizeof(TreeDesc<NJ, NCH>) / sizeof(double)];
    {
        constexpr int kWords = sizeof(TreeDesc<NJ, NCH>) / sizeof(double);
        const double *g = reinterpret_cast<const double *>(a.desc);
        for (int i = threadIdx.x; i < kWords; i += kTreeBlock) lds_desc[i] = g[i];
        __syncthreads();
    }
    const TreeDesc<NJ, NCH> &d = *reinterpret_cast<const TreeDesc<NJ, NCH> *>(lds_desc);
    LdsPark<NJ> park{lds_park[threadIdx.x / 64], static_cast<int>(threadIdx.x % 64)};
    const int64_t wave = static_cast<int64_t>(blockIdx.x) * kTreeWaves + threadIdx.x / 64;
    ikdev::dls_tree_refill_body<NJ, NCH, SPEC>(a, d, wave, static_cast<int64_t>(gridDim.x) * kTreeWaves, park, queue, chunk);
}

template <int NJ> struct HotMask { static constexpr int value = 0; };
template <> struct HotMask<7> { static constexpr int value = 0xf8; };

}  // namespace

// Lane-refill launch of a tree problem when the mode and the batch ask for it (kernels.hpp refill_wanted); returns false when this
// problem's build has no refill instantiation (posture rows, ik::pik levels, constraints, shapes other than Cassie's) -- the caller
// then launches the lock-step kernel.  `a` is the fully prepared argument block of the lock-step launch; `build` names the lock-step
// build the problem runs on (kernels.hip run_dls_tree), whose lane program the refill kernel must share to return the same bits:
// kTreeBuildHot (mask, unit weights, base task at a translation: all folded), kTreeBuildMask (the placement mask folded only),
// kTreeBuildFold (the mask next to the general extras: base-relative references, alignment row, fixed base), kTreeBuildGeneral.
// *err == hipErrorNotReady (returned true): the batch wants the TWO-PHASE solve -- the caller launches its lock-step kernel for the first
// iterations, compacts, and calls again with phase2_queue set (kernels.hpp run_two_phase).
template <int NJ, int NCH>
bool launch_tree_refill(const ProblemHost &ph, const DeviceTables &dt, const BatchIO &io, const ikgpu_dls_params &prm, hipStream_t stream,
                        ikdev::TreeKernelArgs<NJ, NCH> a, int build, hipError_t *err, unsigned long long *phase2_queue) {
    if constexpr (NJ != 7) {
        return false;
    } else {
        constexpr int kMask = HotMask<NJ>::value;
        constexpr int kHot = kMask | (1 << ikdev::kSpecUnit) | (1 << ikdev::kSpecUnitP) | (1 << ikdev::kSpecIdP);
        constexpr int kFold = kMask | (1 << ikdev::kSpecGen);
        if (ph.has_posture || ph.cons_on) return false;
        const void *kern = build == kTreeBuildHot    ? reinterpret_cast<const void *>(dls_tree_refill_kernel<NJ, NCH, kHot>)
                           : build == kTreeBuildMask ? reinterpret_cast<const void *>(dls_tree_refill_kernel<NJ, NCH, kMask>)
                           : build == kTreeBuildFold ? reinterpret_cast<const void *>(dls_tree_refill_kernel<NJ, NCH, kFold>)
                                                     : reinterpret_cast<const void *>(dls_tree_refill_kernel<NJ, NCH, 0>);
        // persistent workgroups of two waves: what the device holds, one wave per SIMD until every lane has >= 8 problems (kernels.hip refill_resident)
        const int64_t occ_waves = persistent_grid(kern, kTreeBlock, 0, INT64_MAX) * kTreeWaves;
        int64_t waves = refill_resident(occ_waves, io.B);
        waves = std::max<int64_t>(kTreeWaves, waves / kTreeWaves * kTreeWaves);
        const dim3 grid(static_cast<unsigned>(waves / kTreeWaves));
        const int chunk = refill_chunk(io.B, waves);
        auto launch = [&](unsigned long long *queue) {
            if (build == kTreeBuildHot) hipLaunchKernelGGL((dls_tree_refill_kernel<NJ, NCH, kHot>), grid, dim3(kTreeBlock), 0, stream, a, queue, chunk);
            else if (build == kTreeBuildMask) hipLaunchKernelGGL((dls_tree_refill_kernel<NJ, NCH, kMask>), grid, dim3(kTreeBlock), 0, stream, a, queue, chunk);
            else if (build == kTreeBuildFold) hipLaunchKernelGGL((dls_tree_refill_kernel<NJ, NCH, kFold>), grid, dim3(kTreeBlock), 0, stream, a, queue, chunk);
            else hipLaunchKernelGGL((dls_tree_refill_kernel<NJ, NCH, 0>), grid, dim3(kTreeBlock), 0, stream, a, queue, chunk);
            return hipGetLastError();
        };
        if (phase2_queue) {   // second phase of a two-phase solve (kernels.hpp run_two_phase): `a` carries the worklist; nothing else to do
            *err = launch(phase2_queue);
            return true;
        }
        const int mode = stop_rule_mode(prm, io.B, waves, stream, true);
        if (mode == kStopLockStep) return false;
        if (mode == kStopTwoPhase) { *err = hipErrorNotReady; return true; }   // (the caller runs the phases: it owns the lock-step launch)
        hipError_t e = hipSuccess;
        unsigned long long *queue = dt.queues.slot_for(stream, &e);
        if (!queue) { *err = e; return true; }
        void *tmp = nullptr;
        if (!a.iters) {
            if ((e = hipMallocAsync(&tmp, sizeof(int32_t) * static_cast<size_t>(io.B), stream)) != hipSuccess) { *err = e; return true; }
            a.iters = static_cast<int32_t *>(tmp);
        }
        e = launch(queue);
        if (e == hipSuccess) e = launch_chain_pass_through(ph, dt, io, a.iters, stream);
        if (tmp) {
            const hipError_t f = hipFreeAsync(tmp, stream);
            if (e == hipSuccess) e = f;
        }
        *err = e;
        return true;
    }
}

template bool launch_tree_refill<7, 2>(const ProblemHost &, const DeviceTables &, const BatchIO &, const ikgpu_dls_params &, hipStream_t, ikdev::TreeKernelArgs<7, 2>, int, hipError_t *, unsigned long long *);
template bool launch_tree_refill<7, 1>(const ProblemHost &, const DeviceTables &, const BatchIO &, const ikgpu_dls_params &, hipStream_t, ikdev::TreeKernelArgs<7, 1>, int, hipError_t *, unsigned long long *);
template bool launch_tree_refill<6, 2>(const ProblemHost &, const DeviceTables &, const BatchIO &, const ikgpu_dls_params &, hipStream_t, ikdev::TreeKernelArgs<6, 2>, int, hipError_t *, unsigned long long *);
template bool launch_tree_refill<6, 1>(const ProblemHost &, const DeviceTables &, const BatchIO &, const ikgpu_dls_params &, hipStream_t, ikdev::TreeKernelArgs<6, 1>, int, hipError_t *, unsigned long long *);

}  // namespace ikgpu
