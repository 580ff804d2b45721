// kernels.hip -- gfx950 kernels of the batched DLS path: one IK problem per wavefront lane,
// the whole iteration loop on-chip, HBM touched once on entry and once on exit; the shared
// kinematic table comes in by scalar loads (chain kernels) or is staged HBM -> LDS once per workgroup
// and read by broadcast (tree kernels, stage kernels).
//
// Reference path: ik::dls (ik/ik/dls.cpp:5-78) and everything it calls per iteration; the lane
// program is device/chain_solver.hpp, the per-lane load/solve/store is device/chain_kernel_body.hpp.
#include "kernels.hpp"

#include <algorithm>
#include <mutex>
#include <utility>
#include <vector>
#include <cstdlib>
#include <stdexcept>
#include <string>

#include "device/chain_kernel_body.hpp"
#include "device/pik_solver.hpp"
#include "device/tree_kernel_body.hpp"
#include "generic_tables.hpp"

namespace ikgpu {
namespace {

constexpr int kBlock = 64;  // one wave64 per workgroup: 1024 workgroups at B = 65536 cover 256 CUs x 4 SIMDs

using ikdev::ChainDesc;
using ikdev::ChainKernelArgs;

// Stage the chain table HBM -> LDS with coalesced loads; every lane then reads it by broadcast.
template <int NJ>
__device__ __forceinline__ const ChainDesc<NJ> &stage_desc(const ChainDesc<NJ> *src, double *lds) {
    constexpr int kWords = sizeof(ChainDesc<NJ>) / sizeof(double);
    const double *g = reinterpret_cast<const double *>(src);
    for (int i = threadIdx.x; i < kWords; i += kBlock) lds[i] = g[i];
    __syncthreads();
    return *reinterpret_cast<const ChainDesc<NJ> *>(lds);
}

// SMASK: compile-time identity-rotation placement mask (0: skip nothing), see device/chain_solver.hpp
template <int NJ, int KT, int SMASK>
__global__ __launch_bounds__(kBlock) void dls_chain_kernel(const ChainKernelArgs<NJ> a) {
    const int64_t gid = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
#ifndef IKGPU_CHAIN_TABLE_IN_LDS
    // The chain table stays in HBM and is read through the constant address space: wave-uniform scalar loads (scalar cache)
    // straight into SGPR operands of the FP64 instructions -- no LDS, no s_waitcnt on ds_read, and 223 instead of 268 VGPRs,
    // i.e. two waves per SIMD fit.  A/B on one box against the LDS-staged table (-DIKGPU_CHAIN_TABLE_IN_LDS), leg / UR5:
    //   B = 65536 (one wave per SIMD either way): 0.1750 / 0.1761 ms  vs  0.1753 / 0.1750 ms   -- equal
    //   B = 262144:                               0.615  / 0.625  ms  vs  0.674  / 0.676  ms   -- 9 % faster
    //   B = 1048576:                              2.13   / 2.13   ms  vs  2.39   / 2.36   ms   -- 11 % faster (4.9e8 solves/s)
    // The table competes with the polynomial constants for the 100 SGPRs (31 v_readlane + 39 s_mov per iteration of SGPR
    // spill code), which is why the single-wave case gains nothing from the missing LDS waits.
    typedef const IKD_CONST_AS ChainDesc<NJ> ConstDesc;
    ikdev::dls_chain_body<NJ, KT, SMASK>(a, *(ConstDesc *)a.desc, gid, ikdev::KeepGoing{a.leave_active, a.leave_after, 0});
#else
    __shared__ double lds_desc[sizeof(ChainDesc<NJ>) / sizeof(double)];
    const ChainDesc<NJ> &d = stage_desc<NJ>(a.desc, lds_desc);
    ikdev::dls_chain_body<NJ, KT, SMASK>(a, d, gid, ikdev::KeepGoing{a.leave_active, a.leave_after, 0});
#endif
}

// The same lane program with lane refill (device/chain_kernel_body.hpp): persistent one-wave workgroups, stop-rule mode only.
template <int NJ, int KT, int SMASK>
__global__ __launch_bounds__(kBlock) void dls_chain_refill_kernel(const ChainKernelArgs<NJ> a, unsigned long long *queue, int chunk) {
    typedef const IKD_CONST_AS ChainDesc<NJ> ConstDesc;
    ikdev::dls_chain_refill_body<NJ, KT, SMASK>(a, *(ConstDesc *)a.desc, queue, chunk);
}

struct PassThroughArgs {
    const double *q0, *lower, *upper;
    const uint8_t *q_in_chain;
    const int32_t *iters;
    double *q_out;
    int64_t B;
    int nq, layout;
};
__global__ __launch_bounds__(256) void chain_pass_through_kernel(const PassThroughArgs a) {
    const int64_t b = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (b >= a.B) return;
    const bool stepped = a.iters[b] > 0;
    for (int i = 0; i < a.nq; ++i) {
        if (a.q_in_chain[i]) continue;
        const double v = a.q0[ikdev::at(a.layout, a.B, a.nq, i, b)];
        const double c = ikdev::dmin(a.upper[i], ikdev::dmax(v, a.lower[i]));
        a.q_out[ikdev::at(a.layout, a.B, a.nq, i, b)] = stepped ? c : v;
    }
}

// Placement masks with a dedicated instantiation: the Cassie leg chains (knee, shin, tarsus, foot and the foot frame
// are pure translations: 0xf8) and the UR5 arm (0x05).  Any other model runs the SMASK = 0 build.
template <int NJ> struct HotMask { static constexpr int value = 0; };
template <> struct HotMask<7> { static constexpr int value = 0xf8; };
template <> struct HotMask<6> { static constexpr int value = 0x05; };

template <int NJ, int KT>
__global__ __launch_bounds__(kBlock) void eval_chain_kernel(const ChainKernelArgs<NJ> a) {
    __shared__ double lds_desc[sizeof(ChainDesc<NJ>) / sizeof(double)];
    const ChainDesc<NJ> &d = stage_desc<NJ>(a.desc, lds_desc);
    ikdev::eval_chain_body<NJ, KT>(a, d, static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x);
}

template <int NJ>
__global__ __launch_bounds__(kBlock) void fk_chain_kernel(const ChainKernelArgs<NJ> a) {
    __shared__ double lds_desc[sizeof(ChainDesc<NJ>) / sizeof(double)];
    const ChainDesc<NJ> &d = stage_desc<NJ>(a.desc, lds_desc);
    ikdev::fk_chain_body<NJ>(a, d, static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x);
}

template <int NJ>
ChainKernelArgs<NJ> make_args(const ProblemHost &ph, const DeviceTables &dt) {
    ChainKernelArgs<NJ> a{};
    fill_chain_args(ph, a.ref_pl, a.qidx, a.vidx, &a.nq, &a.nv, &a.prm.priority, &a.prm.idmask, &a.prm.unit_weights);
    a.desc = reinterpret_cast<const ChainDesc<NJ> *>(dt.chain_desc);
    a.lower = dt.lower;
    a.upper = dt.upper;
    a.q_in_chain = dt.q_in_chain;
    return a;
}

inline dim3 grid_for(int64_t B) { return dim3(static_cast<unsigned>((B + kBlock - 1) / kBlock)); }

// A lane-refill launch: the launch's queue slot, an iteration-count array when the caller passed none (the pass-through kernel
// reads it), the refill kernel, then the entries of q outside the chain.
template <int NJ, class LaunchFn>
hipError_t run_refill(const ProblemHost &ph, const DeviceTables &dt, const BatchIO &io, ChainKernelArgs<NJ> &a, hipStream_t stream, LaunchFn launch) {
    hipError_t e = hipSuccess;
    unsigned long long *queue = dt.queues.slot_for(stream, &e);
    if (!queue) return e;
    int32_t *iters = io.iters;
    void *tmp = nullptr;
    if (!iters) {
        if ((e = hipMallocAsync(&tmp, sizeof(int32_t) * static_cast<size_t>(io.B), stream)) != hipSuccess) return e;
        iters = static_cast<int32_t *>(tmp);
    }
    launch(queue, iters);
    e = hipGetLastError();
    if (e == hipSuccess) e = launch_chain_pass_through(ph, dt, io, iters, stream);
    if (tmp) {
        const hipError_t f = hipFreeAsync(tmp, stream);
        if (e == hipSuccess) e = f;
    }
    return e;
}

template <int NJ, int KT>
hipError_t run_dls(const ProblemHost &ph, const DeviceTables &dt, const BatchIO &io, const ikgpu_dls_params &prm,
                   hipStream_t stream) {
    ChainKernelArgs<NJ> a = make_args<NJ>(ph, dt);
    a.prm.max_iterations = prm.max_iterations;
    a.prm.lam2 = prm.damping * prm.damping;
    a.prm.step_length = prm.step_length;
    a.prm.stop_sq_tol = prm.stop_sq_tol;
    a.layout = io.layout;
    a.B = io.B;
    a.q0 = io.q0;
    a.targets = io.targets;
    a.q_out = io.q_out;
    a.success = io.success;
    a.iters = io.iters;
    // two builds per shape: the hot one (known placement mask + unit weights, all folded at compile time) and the
    // general one (nothing skipped, weights always applied); neither has a branch inside the iteration
    // ... and, for the shapes with a known mask, one in between: the mask folded, the weights applied (a weighted task on a
    // Cassie leg or a UR arm keeps the skipped placement products)
    constexpr int kMask = HotMask<NJ>::value;
    constexpr int kHot = kMask | (1 << ikdev::kSpecUnit);
    // stop-rule mode on a batch larger than the machine: lane refill (same lane program, persistent waves), then the entries outside the chain
#define IKGPU_CHAIN_LAUNCH(SM)                                                                                                  \
    do {                                                                                                                        \
        const void *rk = reinterpret_cast<const void *>(dls_chain_refill_kernel<NJ, KT, SM>);                                    \
        const int64_t rgrid = refill_grid(rk, io.B);                                                                            \
        const int mode = stop_rule_mode(prm, io.B, rgrid, stream, false);                                                             \
        if (mode == kStopRefill) return run_refill<NJ>(ph, dt, io, a, stream, [&](unsigned long long *queue, int32_t *it) {      \
            a.iters = it;                                                                                                       \
            hipLaunchKernelGGL((dls_chain_refill_kernel<NJ, KT, SM>), dim3(static_cast<unsigned>(rgrid)), dim3(kBlock), 0, stream, a, queue, refill_chunk(io.B, rgrid)); \
        });                                                                                                                     \
        if (mode == kStopTwoPhase) return run_two_phase(dt.queues, io, stream, a, false, [&] {                                               \
            hipLaunchKernelGGL((dls_chain_kernel<NJ, KT, SM>), grid_for(io.B), dim3(kBlock), 0, stream, a);                      \
        }, [&](unsigned long long *queue) {                                                                                     \
            hipLaunchKernelGGL((dls_chain_refill_kernel<NJ, KT, SM>), dim3(static_cast<unsigned>(rgrid)), dim3(kBlock), 0, stream, a, queue, refill_chunk(io.B, rgrid)); \
        });                                                                                                                     \
        hipLaunchKernelGGL((dls_chain_kernel<NJ, KT, SM>), grid_for(io.B), dim3(kBlock), 0, stream, a);                          \
        return hipGetLastError();                                                                                               \
    } while (0)
    if constexpr (kMask != 0) {
        if ((a.prm.idmask & kMask) == kMask) {
            if (a.prm.unit_weights) IKGPU_CHAIN_LAUNCH(kHot);
            else IKGPU_CHAIN_LAUNCH(kMask);
        }
    }
    IKGPU_CHAIN_LAUNCH(0);
#undef IKGPU_CHAIN_LAUNCH
}

template <int NJ, int KT>
hipError_t run_eval(const ProblemHost &ph, const DeviceTables &dt, int64_t B, const double *q, const double *targets,
                    double *e_out, double *J_out, int layout, hipStream_t stream) {
    ChainKernelArgs<NJ> a = make_args<NJ>(ph, dt);
    a.layout = layout;
    a.B = B;
    a.q0 = q;
    a.targets = targets;
    a.e_out = e_out;
    a.J_out = J_out;
    hipLaunchKernelGGL((eval_chain_kernel<NJ, KT>), grid_for(B), dim3(kBlock), 0, stream, a);
    return hipGetLastError();
}

template <int NJ>
hipError_t run_fk(const ProblemHost &ph, const DeviceTables &dt, int64_t B, const double *q, double *oMf_out,
                  int layout, hipStream_t stream) {
    ChainKernelArgs<NJ> a = make_args<NJ>(ph, dt);
    a.layout = layout;
    a.B = B;
    a.q0 = q;
    a.oMf_out = oMf_out;
    hipLaunchKernelGGL((fk_chain_kernel<NJ>), grid_for(B), dim3(kBlock), 0, stream, a);
    return hipGetLastError();
}

[[noreturn]] void not_built(int nj, int type) {
    throw std::runtime_error("no kernel instantiated for chain length " + std::to_string(nj) + ", kinematic type " +
                             std::to_string(type));
}

}  // namespace

#define IKGPU_FOR_NJ(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8)

namespace {
// (x y z qx qy qz qw) -> rotation row-major (9) + translation (3), one thread per (problem, task)
__global__ __launch_bounds__(256) void targets_from_pose7_kernel(int64_t B, int ntasks, const double *pose7, double *out, int layout) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= B * ntasks) return;
    const int64_t b = i % B;
    const int t = static_cast<int>(i / B);
    double in[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) in[k] = pose7[ikdev::at(layout, B, ntasks * 7, t * 7 + k, b)];
    double R[9];
    ikdev::quat_to_R(in, R);   // reads entries 3..6 (qx qy qz qw): the free-flyer's layout
#pragma unroll
    for (int k = 0; k < 9; ++k) out[ikdev::at(layout, B, ntasks * 12, t * 12 + k, b)] = R[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) out[ikdev::at(layout, B, ntasks * 12, t * 12 + 9 + k, b)] = in[k];
}
}  // namespace

hipError_t launch_targets_from_pose7(int64_t B, int ntasks, const double *pose7, double *targets12, int layout, hipStream_t stream) {
    const int64_t n = B * ntasks;
    hipLaunchKernelGGL(targets_from_pose7_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, stream, B, ntasks, pose7, targets12, layout);
    return hipGetLastError();
}

bool refill_wanted(const ikgpu_dls_params &prm, int64_t B, int64_t resident_waves) {
    if (!(prm.stop_sq_tol >= 0.0) || prm.max_iterations < 1) return false;   // the never-stop visitor: every lane takes the same number of steps
    const char *env = std::getenv("IKGPU_REFILL");
    if (env && env[0] == '0') return false;
    if (env && (env[0] == '1' || env[0] == '2')) return true;
    return B > resident_waves * kBlock;   // otherwise every problem has its own lane from the start: nothing to refill
}

int two_phase_iterations(bool tree) {
    if (const char *env = std::getenv("IKGPU_TWO_PHASE_ITERS")) {
        const long v = std::strtol(env, nullptr, 10);
        if (v >= 1 && v <= 64) return static_cast<int>(v);
    }
    return tree ? 4 : 8;
}

int two_phase_active(bool tree) {
    if (const char *env = std::getenv("IKGPU_TWO_PHASE_ACTIVE")) {
        const long v = std::strtol(env, nullptr, 10);
        if (v >= 1 && v <= 63) return static_cast<int>(v);
    }
    return tree ? 48 : 32;
}

int stop_rule_mode(const ikgpu_dls_params &prm, int64_t B, int64_t resident_waves, hipStream_t stream, bool tree) {
    if (!(prm.stop_sq_tol >= 0.0) || prm.max_iterations < 1) return kStopLockStep;
    const char *env = std::getenv("IKGPU_REFILL");
    if (env && env[0] == '0') return kStopLockStep;
    if (env && env[0] == '1') return kStopRefill;
    const bool forced = env && env[0] == '2';
    if (!forced && B <= resident_waves * kBlock) return kStopLockStep;
    if (prm.max_iterations <= two_phase_iterations(tree)) return forced ? kStopLockStep : kStopRefill;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (stream && hipStreamIsCapturing(stream, &cap) == hipSuccess && cap == hipStreamCaptureStatusActive) return kStopRefill;
    return kStopTwoPhase;
}

hipError_t two_phase_begin(QueuePool &queues, const BatchIO &io, hipStream_t stream, unsigned long long *queue, TwoPhase *tp) {
    if (io.B > 0x7fffffff) return hipErrorInvalidValue;   // (the worklist holds 32-bit problem indices)
    hipError_t e = hipSuccess;
    tp->worklist = queues.worklist_for(stream, static_cast<size_t>(io.B), &e);
    if (!tp->worklist) return e;
    tp->count = queue + 2;
    tp->success = io.success;
    tp->iters = io.iters;
    if (!tp->success) {
        if ((e = hipMallocAsync(&tp->tmp[0], static_cast<size_t>(io.B), stream)) != hipSuccess) return e;
        tp->success = static_cast<uint8_t *>(tp->tmp[0]);
    }
    if (!tp->iters) {
        if ((e = hipMallocAsync(&tp->tmp[1], sizeof(int32_t) * static_cast<size_t>(io.B), stream)) != hipSuccess) return e;
        tp->iters = static_cast<int32_t *>(tp->tmp[1]);
    }
    return hipSuccess;
}

hipError_t two_phase_end(TwoPhase *tp, hipStream_t stream) {
    hipError_t e = hipSuccess;
    for (void *&p : tp->tmp)
        if (p) { const hipError_t f = hipFreeAsync(p, stream); if (e == hipSuccess) e = f; p = nullptr; }
    return e;
}

int64_t refill_resident(int64_t occupancy_waves, int64_t B) {
    const int64_t waves = (B + kBlock - 1) / kBlock;
    if (const char *env = std::getenv("IKGPU_REFILL_WAVES_PER_CU")) {
        const long w = std::strtol(env, nullptr, 10);
        if (w > 0) return std::min<int64_t>(waves, w * 256);
    }
    // One wave per SIMD until every lane has >= 8 problems to go through: the launch ends with problems that run to max_iterations,
    // and a wave that shares its SIMD takes twice as long over them (measured, B = 262144: 0.45 ms with 4 waves per CU, 0.58 with 8;
    // B = 1048576: 0.99 against 0.84).
    const int64_t want = std::max<int64_t>(1024, waves / 8);
    return std::min<int64_t>(waves, std::min<int64_t>(want, occupancy_waves));
}

int64_t refill_grid(const void *kernel, int64_t B) {
    return refill_resident(persistent_grid(kernel, kBlock, 0, INT64_MAX), B);
}

// Problems a wave pulls from the head at a time (a multiple of 64, >= 64): about a quarter of an even share of what is left after the
// static first round, so that the head sees ~4 atomics per wave and the last chunks are small against a wave's whole share.
int refill_chunk(int64_t B, int64_t grid) {
    if (const char *env = std::getenv("IKGPU_REFILL_CHUNK")) {
        const long c = std::strtol(env, nullptr, 10);
        if (c >= 64) return static_cast<int>(std::min<long>(c / 64 * 64, 4096)) | (refill_batch() << 16);
    }
    const int64_t left = B - grid * kBlock;
    int64_t c = left > 0 ? left / (grid * 4) : 0;
    c = (c + 63) / 64 * 64;
    return static_cast<int>(std::min<int64_t>(std::max<int64_t>(c, 64), 4096)) | (refill_batch() << 16);
}

// Idle lanes a wave waits for before a refill event (1..64; device/chain_kernel_body.hpp chain_refill_loop): a quarter of the wave.
// IKGPU_REFILL_BATCH overrides (1 = round 3's behaviour: an event whenever a lane finishes).
int refill_batch() {
    if (const char *env = std::getenv("IKGPU_REFILL_BATCH")) {
        const long v = std::strtol(env, nullptr, 10);
        if (v >= 1 && v <= 64) return static_cast<int>(v);
    }
    return 16;
}

hipError_t launch_chain_pass_through(const ProblemHost &ph, const DeviceTables &dt, const BatchIO &io, const int32_t *iters, hipStream_t stream) {
    bool any_outside = false;
    for (uint8_t in : ph.q_in_chain) any_outside = any_outside || !in;
    if (!any_outside) return hipSuccess;
    const PassThroughArgs p{io.q0, dt.lower, dt.upper, dt.q_in_chain, iters, io.q_out, io.B, ph.nq, io.layout};
    hipLaunchKernelGGL(chain_pass_through_kernel, dim3(static_cast<unsigned>((io.B + 255) / 256)), dim3(256), 0, stream, p);
    return hipGetLastError();
}

bool chain_shape_built(int nj, int type) {
    return nj >= 1 && nj <= kMaxChain && (type == IKGPU_FULL || type == IKGPU_POSITION || type == IKGPU_ORIENTATION);
}

hipError_t launch_dls_chain(const ProblemHost &ph, const DeviceTables &dt, const BatchIO &io,
                            const ikgpu_dls_params &prm, hipStream_t stream) {
    if (ph.chain_build != 0) return launch_dls_chain_hot(ph, dt, io, prm, stream);   // structure-specialised build (kernels_hot.hip / rtc.cpp)
    const int nj = ph.chain.nj, type = ph.tasks[0].type;
#define X(N)                                                                                              \
    if (nj == N) {                                                                                        \
        if (type == IKGPU_FULL) return run_dls<N, ikdev::KT_FULL>(ph, dt, io, prm, stream);               \
        if (type == IKGPU_POSITION) return run_dls<N, ikdev::KT_POSITION>(ph, dt, io, prm, stream);       \
        if (type == IKGPU_ORIENTATION) return run_dls<N, ikdev::KT_ORIENTATION>(ph, dt, io, prm, stream); \
    }
    IKGPU_FOR_NJ(X)
#undef X
    not_built(nj, type);
}

hipError_t launch_eval_chain(const ProblemHost &ph, const DeviceTables &dt, int64_t B, const double *q,
                             const double *targets, double *e_out, double *J_out, int layout, hipStream_t stream) {
    const int nj = ph.chain.nj, type = ph.tasks[0].type;
#define X(N)                                                                                                             \
    if (nj == N) {                                                                                                       \
        if (type == IKGPU_FULL) return run_eval<N, ikdev::KT_FULL>(ph, dt, B, q, targets, e_out, J_out, layout, stream); \
        if (type == IKGPU_POSITION)                                                                                      \
            return run_eval<N, ikdev::KT_POSITION>(ph, dt, B, q, targets, e_out, J_out, layout, stream);                 \
        if (type == IKGPU_ORIENTATION)                                                                                   \
            return run_eval<N, ikdev::KT_ORIENTATION>(ph, dt, B, q, targets, e_out, J_out, layout, stream);              \
    }
    IKGPU_FOR_NJ(X)
#undef X
    not_built(nj, type);
}

hipError_t launch_fk_chain(const ProblemHost &ph, const DeviceTables &dt, int64_t B, const double *q, double *oMf_out,
                           int layout, hipStream_t stream) {
    const int nj = ph.chain.nj;
#define X(N) \
    if (nj == N) return run_fk<N>(ph, dt, B, q, oMf_out, layout, stream);
    IKGPU_FOR_NJ(X)
#undef X
    not_built(nj, 0);
}


// ------------------------------------------------------------------------------------------------
// Free-flyer tree kernels (shape F: Cassie full body, two foot chains + the pelvis task)
// ------------------------------------------------------------------------------------------------
namespace {

using ikdev::LegFactor;
using ikdev::TreeDesc;
using ikdev::TreeKernelArgs;

// While chain 1 is evaluated, chain 0's factor (L packed, W, u: 77 doubles for NJ = 7) is parked in LDS as
// [entry][lane]: consecutive lanes hit consecutive 8-byte words, so every ds_write_b64 / ds_read_b64 is
// conflict-free.  77 x 512 B = 38.5 KB per wave: two waves share a workgroup (and one copy of the constant
// table) so that four waves -- one per SIMD -- fit the CU's 160 KB of LDS.
constexpr int kTreeWaves = 2;
constexpr int kTreeBlock = 64 * kTreeWaves;

template <int NJ>
struct LdsPark {
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

template <int NJ, int NCH>
__device__ __forceinline__ const TreeDesc<NJ, NCH> &stage_tree_desc(const TreeDesc<NJ, NCH> *src, double *lds, int nthreads) {
    constexpr int kWords = sizeof(TreeDesc<NJ, NCH>) / sizeof(double);
    const double *g = reinterpret_cast<const double *>(src);
    for (int i = threadIdx.x; i < kWords; i += nthreads) lds[i] = g[i];
    __syncthreads();
    return *reinterpret_cast<const TreeDesc<NJ, NCH> *>(lds);
}

template <int NJ, int NCH, int SPEC>
__global__ __launch_bounds__(kTreeBlock) void dls_tree_kernel(const TreeKernelArgs<NJ, NCH> a) {
    __shared__ double lds_park[kTreeWaves][NCH > 1 ? LdsPark<NJ>::kEntries : 1][64];
#ifdef IKGPU_TREE_TABLE_SCALAR
    typedef const IKD_CONST_AS TreeDesc<NJ, NCH> ConstDesc;   // scalar loads from HBM, as the chain kernels do
    ConstDesc &d = *(ConstDesc *)a.desc;
#else
    __shared__ double lds_desc[sizeof(TreeDesc<NJ, NCH>) / sizeof(double)];
    const TreeDesc<NJ, NCH> &d = stage_tree_desc<NJ, NCH>(a.desc, lds_desc, kTreeBlock);
#endif
    const int64_t gid = static_cast<int64_t>(blockIdx.x) * kTreeBlock + threadIdx.x;
    LdsPark<NJ> park{lds_park[threadIdx.x / 64], static_cast<int>(threadIdx.x % 64)};
    // posture build with one chain (no factor to park, LDS to spare): the joints outside the chain that carry a posture row
    // live in dynamic LDS [row][lane] between iterations; with two chains the LDS is full and they stay in the q_out column
    extern __shared__ double lds_post[];
    double *post_lane = (SPEC > 0 && ikdev::spec_has_posture(SPEC) && NCH == 1) ? lds_post + threadIdx.x : nullptr;
    int64_t post_stride = kTreeBlock;
    // posture rows next to the constraint: chain 1 carries the constrained frame and no task, so no factor is ever parked -- the park
    // buffer's rows hold the outside joints instead (2 * post_n + NJ <= kEntries rows, checked by the launcher), [row][lane] per wave
    if constexpr (SPEC > 0 && NCH > 1 && (SPEC & ikdev::kSpecPostCons) == ikdev::kSpecPostCons) {
        if (a.prm.cons_on && 2 * a.prm.post_n + NJ <= LdsPark<NJ>::kEntries) {
            post_lane = &lds_park[threadIdx.x / 64][0][threadIdx.x % 64];
            post_stride = 64;
        }
    }
    // target rows as SGPR row pointer + lane offset (LaneRows) in the builds that spilled hoisted per-lane row addresses to scratch
    // memory without it: two chains with posture rows next to the constraint, and the general builds without a folded
    // placement mask.  Elsewhere the hoisted addresses fit the register file and are free, the scalar row arithmetic is not (same-box
    // A/B: posture build 0.652 -> 0.663 ms, pik 0.576 -> 0.591, pinned foot 0.707 -> 0.714; posture + pinned foot 1.17 -> 0.87-0.94)
    constexpr bool kRows = SPEC >= 0 && ((NCH > 1 && (SPEC & ikdev::kSpecPostCons) == ikdev::kSpecPostCons) ||
                                         (ikdev::spec_is_general(SPEC) && (SPEC & (1 << ikdev::kSpecGen)) == 0));
    ikdev::dls_tree_body<NJ, NCH, SPEC>(a, d, gid, park, ikdev::KeepGoing{a.leave_active, a.leave_after, 0}, post_lane, post_stride,
                                        kRows ? static_cast<int64_t>(blockIdx.x) * kTreeBlock : int64_t{-1});
}

template <int NJ, int NCH>
__global__ __launch_bounds__(kBlock) void eval_tree_kernel(const TreeKernelArgs<NJ, NCH> a) {
    __shared__ double lds_desc[sizeof(TreeDesc<NJ, NCH>) / sizeof(double)];
    const TreeDesc<NJ, NCH> &d = stage_tree_desc<NJ, NCH>(a.desc, lds_desc, kBlock);
    ikdev::eval_tree_body<NJ, NCH>(a, d, static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x);
}

template <int NJ, int NCH>
TreeKernelArgs<NJ, NCH> make_tree_args(const ProblemHost &ph, const DeviceTables &dt) {
    TreeKernelArgs<NJ, NCH> a{};
    const TreeArgsHost h = tree_args(ph);
    for (int c = 0; c < NCH; ++c)
        for (int j = 0; j < NJ; ++j) { a.qidx[c][j] = h.qidx[c][j]; a.vidx[c][j] = h.vidx[c][j]; }
    for (int s = 0; s < 3; ++s) { a.tslot[s] = h.tslot[s]; a.trow[s] = h.trow[s]; a.tdim[s] = h.tdim[s]; a.trow0[s] = h.trow0[s]; }
    a.prm.prio[0] = h.prio[0]; a.prm.prio[1] = h.prio[1]; a.prm.prioP = h.prio[2];
    a.prm.hasP = h.hasP;
    a.prm.idmask[0] = h.idmask[0]; a.prm.idmask[1] = h.idmask[1]; a.prm.idmaskP = h.idmaskP;
    a.prm.unit[0] = h.unit[0]; a.prm.unit[1] = h.unit[1]; a.prm.unitP = h.unit[2];
    a.prm.ref_base[0] = h.ref_base[0]; a.prm.ref_base[1] = h.ref_base[1];
    a.prm.align_chain = h.align_chain; a.prm.align_axis = h.align_axis; a.prm.align_slot = h.align_slot;
    a.prm.align_prio = h.align_prio; a.prm.align_w = h.align_w;
    static_assert(ikdev::kMaxPostOut == kMaxPostureOut && kMaxChain == 8, "TreeParams posture arrays");
    a.prm.fixed_base = h.fixed_base;
    a.prm.cons_on = h.cons_on; a.prm.cons_type = h.cons_type;
    a.prm.post_on = h.post_on; a.prm.post_prio = h.post_prio; a.prm.post_n = h.post_n;
    for (int k = 0; k < h.post_n; ++k) {
        a.prm.post_q[k] = h.post_q[k]; a.prm.post_slot[k] = h.post_slot[k]; a.prm.post_w[k] = h.post_w[k]; a.prm.post_m[k] = h.post_m[k];
    }
    for (int c = 0; c < 2; ++c)
        for (int j = 0; j < 8; ++j) { a.prm.postc_slot[c][j] = h.postc_slot[c][j]; a.prm.postc_w[c][j] = h.postc_w[c][j]; a.prm.postc_m[c][j] = h.postc_m[c][j]; }
    a.desc = reinterpret_cast<const TreeDesc<NJ, NCH> *>(dt.chain_desc);
    a.nq = ph.nq; a.nv = ph.nv; a.ntasks = ph.ntasks;
    a.lower = dt.lower; a.upper = dt.upper; a.q_in_chain = dt.q_in_chain;
    return a;
}

}  // namespace
// kernels_tree_refill.hip
template <int NJ, int NCH>
bool launch_tree_refill(const ProblemHost &ph, const DeviceTables &dt, const BatchIO &io, const ikgpu_dls_params &prm, hipStream_t stream,
                        ikdev::TreeKernelArgs<NJ, NCH> a, int build, hipError_t *err, unsigned long long *phase2_queue);
namespace {

template <int NJ>
bool tree_hot_structure_matches(const ProblemHost &ph) {
    if constexpr (std::is_same<typename ikdev::TreeHotStruct<NJ>::type, void>::value) return true;   // no structure folded for this NJ
    else {
        for (const ChainHost *c : {&ph.chain, &ph.chainB}) {
            if (c->nj == 0) continue;
            const ChainStructure s = chain_structure(*c);
            if (!s.fits || s.code[0] != ikdev::kTreeHotCode7[0] || s.code[1] != ikdev::kTreeHotCode7[1] || s.code[2] != ikdev::kTreeHotCode7[2]) return false;
        }
        return true;
    }
}

template <int NJ, int NCH>
hipError_t run_dls_tree(const ProblemHost &ph, const DeviceTables &dt, const BatchIO &io, const ikgpu_dls_params &prm,
                        hipStream_t stream, const double *pik_lambda1) {
    TreeKernelArgs<NJ, NCH> a = make_tree_args<NJ, NCH>(ph, dt);
    a.prm.max_iterations = prm.max_iterations;
    a.prm.lam2 = prm.damping * prm.damping;
    a.prm.step_length = prm.step_length;
    a.prm.stop_sq_tol = prm.stop_sq_tol;
    if (pik_lambda1) { a.prm.pik_on = 1; a.prm.pik_lam2_1 = *pik_lambda1 * *pik_lambda1; }
    a.layout = io.layout; a.B = io.B; a.q0 = io.q0; a.targets = io.targets;
    a.q_out = io.q_out; a.success = io.success; a.iters = io.iters;
    // hot build: both chains carry the shape's known placement mask, every task is Full with unit weights and the base
    // task sits at a pure translation from the base joint -- all folded at compile time; otherwise the general build
    constexpr int kMask = HotMask<NJ>::value;
    constexpr int kHot = kMask | (1 << ikdev::kSpecUnit) | (1 << ikdev::kSpecUnitP) | (1 << ikdev::kSpecIdP);
    // (and, where the hot builds fold the chains' placement STRUCTURE -- ikdev::TreeHotStruct, NJ = 7 -- both chains carry that code)
    const bool hot = !ph.tree_extras() && kMask != 0 && (a.prm.idmask[0] & kMask) == kMask && (NCH == 1 || (a.prm.idmask[1] & kMask) == kMask) &&
                     a.prm.unit[0] && (NCH == 1 || a.prm.unit[1]) && (!a.prm.hasP || (a.prm.unitP && (a.prm.idmaskP & 1))) &&
                     tree_hot_structure_matches<NJ>(ph);
    const dim3 grid(static_cast<unsigned>((io.B + kTreeBlock - 1) / kTreeBlock));
    // in between (as for the chain kernels): the placement mask folded, weights / task types / base-frame placement general
    // (A/B on one box, full body: pelvis task weighted 1.07 -> 0.91 ms, feet as Position tasks 0.98 -> 0.83 ms)
    const bool structure = tree_hot_structure_matches<NJ>(ph);
    const bool mask_only = !hot && !ph.tree_extras() && structure && kMask != 0 && (a.prm.idmask[0] & kMask) == kMask && (NCH == 1 || (a.prm.idmask[1] & kMask) == kMask);
    // the general builds (extras: base reference, alignment row, fixed base, posture rows, the constraint, level 1 of ik::pik) exist
    // twice where the shape's mask is known: with it folded (bit kSpecGen next to the mask) and without
    const bool fold = structure && kMask != 0 && (a.prm.idmask[0] & kMask) == kMask && (NCH == 1 || (a.prm.idmask[1] & kMask) == kMask);
    const size_t post_lds = NCH == 1 ? sizeof(double) * kTreeBlock * static_cast<size_t>(2 * std::max(1, a.prm.post_n) + NJ) : 0;
#define IKGPU_TREE_GENERAL(FLAGS, LDS)                                                                                              \
    do {                                                                                                                            \
        if (fold) hipLaunchKernelGGL((dls_tree_kernel<NJ, NCH, (kMask != 0 ? ((FLAGS) | kMask | (1 << ikdev::kSpecGen)) : (FLAGS))>), grid, \
                                     dim3(kTreeBlock), LDS, stream, a);                                                             \
        else hipLaunchKernelGGL((dls_tree_kernel<NJ, NCH, (FLAGS)>), grid, dim3(kTreeBlock), LDS, stream, a);                       \
    } while (0)
    auto launch_lockstep = [&]() {
    if (hot && !(a.prm.stop_sq_tol >= 0.0) && !std::getenv("IKGPU_TREE_NEVER_OFF"))   // the never-stop visitor: its own instantiation, as the hot chain kernel's
        hipLaunchKernelGGL((dls_tree_kernel<NJ, NCH, (kHot | (1 << ikdev::kSpecNever))>), grid, dim3(kTreeBlock), 0, stream, a);
    else if (hot) hipLaunchKernelGGL((dls_tree_kernel<NJ, NCH, kHot>), grid, dim3(kTreeBlock), 0, stream, a);
    else if (mask_only) hipLaunchKernelGGL((dls_tree_kernel<NJ, NCH, (kMask != 0 ? kMask : kHot)>), grid, dim3(kTreeBlock), 0, stream, a);
    else if (pik_lambda1)   // two-level ik::pik (tree_takes_two_level_pik): the general build + the level-1 row's projection
        IKGPU_TREE_GENERAL((1 << ikdev::kSpecPik), 0);
    else if (ph.cons_on) {   // one FrameConstraint on the second chain: the constraint build (general + the projection), with or
                             // without the posture code
        if constexpr (NCH == 2) {
            if (ph.has_posture) IKGPU_TREE_GENERAL(ikdev::kSpecPostCons, 0);
            else IKGPU_TREE_GENERAL((1 << ikdev::kSpecCons), 0);
        }
    } else if (ph.has_posture)
        IKGPU_TREE_GENERAL((1 << ikdev::kSpecPost), post_lds);
    else IKGPU_TREE_GENERAL(0, 0);
    };
    if constexpr (NCH != 2) {
        if (ph.cons_on && !pik_lambda1 && !hot && !mask_only) return hipErrorInvalidValue;
    }
    // stop-rule mode on a batch larger than the machine: lane refill (kernels_tree_refill.hip) for the builds that have it -- every
    // build without per-lane state outside q (no posture rows, no constraint, no ik::pik level) -- directly, or as the second phase
    // after the lock-step kernel's first iterations (kernels.hpp stop_rule_mode)
    if (!pik_lambda1 && !ph.cons_on && !ph.has_posture && prm.stop_sq_tol >= 0.0 && prm.max_iterations >= 1) {
        hipError_t re = hipSuccess;
        const int build = hot ? kTreeBuildHot : mask_only ? kTreeBuildMask : fold ? kTreeBuildFold : kTreeBuildGeneral;
        if (launch_tree_refill<NJ, NCH>(ph, dt, io, prm, stream, a, build, &re, nullptr)) {
            if (re != hipErrorNotReady) return re;
            return run_two_phase(dt.queues, io, stream, a, true, launch_lockstep, [&](unsigned long long *queue) {
                hipError_t pe = hipSuccess;
                (void)launch_tree_refill<NJ, NCH>(ph, dt, io, prm, stream, a, build, &pe, queue);
            });
        }
    }
    launch_lockstep();
#undef IKGPU_TREE_GENERAL
    return hipGetLastError();
}

template <int NJ, int NCH>
hipError_t run_eval_tree(const ProblemHost &ph, const DeviceTables &dt, int64_t B, const double *q, const double *targets,
                         double *e_out, double *J_out, double *oMf_out, int layout, hipStream_t stream) {
    TreeKernelArgs<NJ, NCH> a = make_tree_args<NJ, NCH>(ph, dt);
    a.layout = layout; a.B = B; a.q0 = q; a.targets = targets;
    a.e_out = e_out; a.J_out = J_out; a.oMf_out = oMf_out;
    hipLaunchKernelGGL((eval_tree_kernel<NJ, NCH>), grid_for(B), dim3(kBlock), 0, stream, a);
    return hipGetLastError();
}

}  // namespace

// (chain length, number of chains) pairs with a compiled kernel
#define IKGPU_FOR_TREE(X) X(7, 2) X(7, 1) X(6, 2) X(6, 1)

bool tree_shape_built(int nj, int nch) {
#define X(N, C) if (nj == N && nch == C) return true;
    IKGPU_FOR_TREE(X)
#undef X
    return false;
}

bool tree_takes_two_level_pik(const ProblemHost &ph) {
    if (ph.kind != KernelKind::Tree || ph.align_task < 0 || ph.base_task < 0 || ph.has_posture || ph.fixed_base || ph.cons_on) return false;
    const ikgpu_task &base = ph.tasks[ph.base_task];
    if (base.type != IKGPU_FULL) return false;
    for (int i = 0; i < 6; ++i)
        if (base.weight[i] == 0.0) return false;          // the base task's 6 x 6 block must span the base directions
    for (int i = 0; i < ph.ntasks; ++i)                    // level 0: everything but the alignment row; level 1: that row alone
        if (ph.tasks[i].priority != (i == ph.align_task ? 1 : 0)) return false;
    return true;
}

hipError_t launch_dls_tree(const ProblemHost &ph, const DeviceTables &dt, const BatchIO &io, const ikgpu_dls_params &prm,
                           hipStream_t stream, const double *pik_lambda1) {
    const int nj = ph.chain.nj, nch = ph.chainB.nj > 0 ? 2 : 1;
#define X(N, C) if (nj == N && nch == C) return run_dls_tree<N, C>(ph, dt, io, prm, stream, pik_lambda1);
    IKGPU_FOR_TREE(X)
#undef X
    not_built(nj, nch);
}

hipError_t launch_eval_tree(const ProblemHost &ph, const DeviceTables &dt, int64_t B, const double *q, const double *targets,
                            double *e_out, double *J_out, double *oMf_out, int layout, hipStream_t stream) {
    const int nj = ph.chain.nj, nch = ph.chainB.nj > 0 ? 2 : 1;
#define X(N, C) if (nj == N && nch == C) return run_eval_tree<N, C>(ph, dt, B, q, targets, e_out, J_out, oMf_out, layout, stream);
    IKGPU_FOR_TREE(X)
#undef X
    not_built(nj, nch);
}


// ------------------------------------------------------------------------------------------------
// Generic fallback kernel: any tree, any task list; workspace in HBM as [word][lane]
// ------------------------------------------------------------------------------------------------
namespace {

__global__ __launch_bounds__(kBlock) void dls_generic_kernel(const ikdev::GenericKernelArgs a) {
    const int64_t gid = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    ikdev::dls_generic_body_ws(a, ikdev::const_tables(a.T), gid, ikdev::Ws{a.ws + gid, a.ws_stride}, [](bool act) { return __any(act) != 0; });
}

// The same per-lane program with its workspace in LDS, [word][lane]: one 64-lane workgroup, ws_words x 512 bytes.
__global__ __launch_bounds__(64) void dls_generic_lds_kernel(const ikdev::GenericKernelArgs a) {
    extern __shared__ double lane_lds[];
    ikdev::dls_generic_body_ws(a, ikdev::const_tables(a.T), static_cast<int64_t>(blockIdx.x) * 64 + threadIdx.x, ikdev::Ws{lane_lds + threadIdx.x, 64},
                               [](bool act) { return __any(act) != 0; });
}

__global__ __launch_bounds__(kBlock) void eval_generic_kernel(const ikdev::GenericKernelArgs a) {
    ikdev::eval_generic_body(a, static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x);
}

hipError_t with_workspace(const ProblemHost &ph, const DeviceTables &dt, ikdev::GenericKernelArgs &a, int64_t B, hipStream_t stream,
                          bool solve) {
    a.T = bind_generic_tables(ph, dt.g_ints, dt.g_dbls);
    a.B = B;
    a.ws_stride = (B + kBlock - 1) / kBlock * kBlock;
    const size_t bytes = sizeof(double) * static_cast<size_t>(ph.generic.ws_words) * static_cast<size_t>(a.ws_stride);
    void *ws = nullptr;
    hipError_t e = hipMallocAsync(&ws, bytes, stream);
    if (e != hipSuccess) return e;
    a.ws = static_cast<double *>(ws);
    if (solve) hipLaunchKernelGGL(dls_generic_kernel, grid_for(B), dim3(kBlock), 0, stream, a);
    else hipLaunchKernelGGL(eval_generic_kernel, grid_for(B), dim3(kBlock), 0, stream, a);
    e = hipGetLastError();
    const hipError_t f = hipFreeAsync(ws, stream);
    return e != hipSuccess ? e : f;
}

__global__ __launch_bounds__(kBlock) void pik_generic_kernel(const ikdev::PikKernelArgs a) {
    ikdev::pik_generic_body(a, static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x, [](bool act) { return __any(act) != 0; });
}

}  // namespace

namespace {

// Cooperative form: 16 lanes per problem, 4 problems per 64-lane workgroup, the workspace in (dynamic) LDS.
struct CoopStaging {  // the problem's packed tables in HBM, copied to LDS by every workgroup before the first phase
    const int32_t *ints;
    const double *dbls;
    int n_ints, n_dbls;
};

template <class P>
__device__ __forceinline__ void rebase(const P *&p, const P *from, const P *to) { p = to + (p - from); }

// LDS of a cooperative workgroup: [double tables | int tables | one workspace per problem].  Copies the tables in, points
// the table pointers of T and L at the copies and returns the first workspace: a table read inside the phases is then an LDS
// read, not an L2 round trip (73 global loads per iteration otherwise, each a dependent ~1 us stall with so few waves per CU).
__device__ __forceinline__ double *coop_stage(double *lds, ikdev::GenericTables &T, ikdev::CoopLayout &L, const CoopStaging &s) {
    double *ld = lds;
    int32_t *li = reinterpret_cast<int32_t *>(lds + s.n_dbls);
    for (int i = threadIdx.x; i < s.n_dbls; i += blockDim.x) ld[i] = s.dbls[i];
    for (int i = threadIdx.x; i < s.n_ints; i += blockDim.x) li[i] = s.ints[i];
    __syncthreads();
    rebase(T.jtype, s.ints, li); rebase(T.parent, s.ints, li); rebase(T.idx_q, s.ints, li); rebase(T.idx_v, s.ints, li);
    rebase(T.t_type, s.ints, li); rebase(T.t_fjoint, s.ints, li); rebase(T.t_rjoint, s.ints, li); rebase(T.t_row, s.ints, li);
    rebase(T.t_dim, s.ints, li); rebase(T.t_prio, s.ints, li); rebase(T.lvl_row0, s.ints, li);
    rebase(T.placement, s.dbls, ld); rebase(T.axis, s.dbls, ld); rebase(T.lower, s.dbls, ld); rebase(T.upper, s.dbls, ld);
    rebase(T.t_fpl, s.dbls, ld); rebase(T.t_rpl, s.dbls, ld); rebase(T.t_w, s.dbls, ld);
    rebase(L.support, s.ints, li); rebase(L.pair_i, s.ints, li); rebase(L.pair_j, s.ints, li); rebase(L.order, s.ints, li);
    rebase(L.lvl_start, s.ints, li); rebase(L.chain_start, s.ints, li); rebase(L.tb_index, s.ints, li); rebase(L.btask, s.ints, li); rebase(L.frow, s.ints, li); rebase(L.pstart, s.ints, li); rebase(L.ptask, s.ints, li); rebase(L.tg_off, s.ints, li); rebase(L.tg_src, s.ints, li); rebase(L.jrow, s.ints, li); rebase(L.col_joint, s.ints, li);
    rebase(T.j_mass, s.dbls, ld); rebase(T.j_lever, s.dbls, ld); rebase(T.j_submass, s.dbls, ld);
    rebase(T.c_type, s.ints, li); rebase(T.c_fjoint, s.ints, li); rebase(T.c_rjoint, s.ints, li); rebase(T.c_row, s.ints, li);
    rebase(T.c_dim, s.ints, li); rebase(T.c_fpl, s.dbls, ld); rebase(T.c_rpl, s.dbls, ld);
    rebase(L.csupp_f, s.ints, li); rebase(L.csupp_r, s.ints, li); rebase(L.cpair_i, s.ints, li); rebase(L.cpair_j, s.ints, li);
    return lds + s.n_dbls + (s.n_ints + 1) / 2;
}

// (two waves per SIMD asked for: the kernel sits at 261 registers otherwise -- one wave per SIMD, four workgroups per CU where
// the LDS has room for five: 22 instead of 18.6 ms on the demo task set)
// The next group of problems for this (one-wave) workgroup, from the launch's queue head.  Dynamic, not blockIdx + k gridDim: the
// LDS lets five workgroups live on a CU's four SIMDs, so two of them share a SIMD and run at about half the pace of the other
// three -- with equal shares the launch waited for those two.  Every wave leaves once the head has passed the last group.
__device__ __forceinline__ int64_t next_block(unsigned long long *queue) {
    unsigned long long v = 0;
    if (threadIdx.x == 0) v = atomicAdd(queue, 1ull);
    const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v)), hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v >> 32));
    return static_cast<int64_t>((static_cast<unsigned long long>(hi) << 32) | lo);
}

// A workgroup that has taken its last (failed) pull from the head signs out on queue[1]; the last one out zeroes the slot, so the
// next launch on this stream finds it clean (QueuePool, kernels.hpp) -- no host-side reset, no other launch can hold this slot.
__device__ __forceinline__ void leave_queue(unsigned long long *queue) {
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(queue + 1, 1ull) == static_cast<unsigned long long>(gridDim.x) - 1ull) {
            queue[0] = 0ull;
            queue[1] = 0ull;
            __threadfence();
        }
    }
}

#ifndef IKGPU_COOP_WAVES
#define IKGPU_COOP_WAVES 2   // (A/B knob: 1 = no register cap, four workgroups per CU)
#endif
__global__ __launch_bounds__(kBlock, IKGPU_COOP_WAVES) void dls_coop_kernel(ikdev::CoopKernelArgs a, const CoopStaging s, unsigned long long *queue) {
    extern __shared__ double coop_lds[];
    double *ws0 = coop_stage(coop_lds, a.T, a.L, s);
    const int grp = threadIdx.x / ikdev::kCoopGroup, g = threadIdx.x % ikdev::kCoopGroup;
    const int per_block = blockDim.x / ikdev::kCoopGroup;
    // persistent workgroups: the tables are staged once per workgroup (7 KB for the demo task set) and serve every group of
    // problems the workgroup takes -- staged per four problems they were 2/3 of the launch's HBM traffic (137 MB against 43 MB of
    // algorithmic bytes at B = 65536)
    const int64_t nblocks = (a.B + per_block - 1) / per_block;
    for (int64_t blk = next_block(queue); blk < nblocks; blk = next_block(queue))
        ikdev::dls_coop_body<false>(a, blk * per_block + grp, g, ws0 + grp * a.L.words, [](bool act) { return __any(act) != 0; });
    leave_queue(queue);
}

// The same for 16 <= M <= 31 (rows left after the posture elimination): two matrix rows per lane in the register Cholesky, no
// register cap -- workspaces of that size leave at most four workgroups on a CU, one per SIMD.
__global__ __launch_bounds__(kBlock) void dls_coop_big_kernel(ikdev::CoopKernelArgs a, const CoopStaging s, unsigned long long *queue) {
    extern __shared__ double coop_lds[];
    double *ws0 = coop_stage(coop_lds, a.T, a.L, s);
    const int grp = threadIdx.x / ikdev::kCoopGroup, g = threadIdx.x % ikdev::kCoopGroup;
    const int per_block = blockDim.x / ikdev::kCoopGroup;
    const int64_t nblocks = (a.B + per_block - 1) / per_block;
    for (int64_t blk = next_block(queue); blk < nblocks; blk = next_block(queue))
        ikdev::dls_coop_body<true>(a, blk * per_block + grp, g, ws0 + grp * a.L.words, [](bool act) { return __any(act) != 0; });
    leave_queue(queue);
}

__global__ __launch_bounds__(kBlock) void pik_coop_kernel(const ikdev::PikCoopKernelArgs a, const CoopStaging s, unsigned long long *queue) {
    extern __shared__ double coop_lds[];
    // the arguments stay in the kernel-argument segment (they hold the 1 KB `da` array: a mutable copy would live in
    // scratch memory, 1.3 KB per lane); only the two small table structs are copied, to be pointed at LDS
    ikdev::GenericTables T = a.T;
    ikdev::CoopLayout L = a.L;
    double *ws0 = coop_stage(coop_lds, T, L, s);
    const int grp = threadIdx.x / ikdev::kCoopGroup, g = threadIdx.x % ikdev::kCoopGroup;
    const int per_block = blockDim.x / ikdev::kCoopGroup;
    const int64_t nblocks = (a.B + per_block - 1) / per_block;
    for (int64_t blk = next_block(queue); blk < nblocks; blk = next_block(queue))   // persistent workgroups, as dls_coop_kernel
        ikdev::pik_coop_body(a, T, L, blk * per_block + grp, g, ws0 + grp * a.K.words, [](bool act) { return __any(act) != 0; });
    leave_queue(queue);
}

}  // namespace

// A launch asking for more than the default 64 KB of dynamic LDS has to raise the kernel's limit first (once per kernel and device).
bool raise_lds_limit(const void *kernel, size_t lds) {
    if (lds <= 64 * 1024) return true;
    static std::mutex mu;
    static std::vector<std::pair<const void *, int>> done;   // (kernel, device) pairs already raised
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    for (const auto &d : done)
        if (d.first == kernel && d.second == dev) return true;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
    done.emplace_back(kernel, dev);
    return true;
}

hipError_t QueuePool::grow() {
    std::lock_guard<std::mutex> lock(mu);
    void *p = nullptr;
    const size_t bytes = sizeof(unsigned long long) * 4 * kChunkSlots;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return e;
    if ((e = hipMemset(p, 0, bytes)) != hipSuccess) { (void)hipFree(p); return e; }
    chunks.push_back(static_cast<unsigned long long *>(p));
    used_in_last = 0;
    return hipSuccess;
}

unsigned long long *QueuePool::slot_for(hipStream_t stream, hipError_t *err) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (stream && hipStreamIsCapturing(stream, &cap) != hipSuccess) cap = hipStreamCaptureStatusNone;
    const bool capturing = cap == hipStreamCaptureStatusActive;
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!capturing) {
            auto it = by_stream.find(stream);
            if (it != by_stream.end()) return it->second;
        }
        if (used_in_last < kChunkSlots && !chunks.empty()) {
            unsigned long long *slot = chunks.back() + 4 * used_in_last++;
            if (!capturing) by_stream.emplace(stream, slot);
            return slot;
        }
    }
    if (capturing) {   // allocation and memset are not capturable: a problem takes kChunkSlots captured launches (+ one chunk per grow() outside a capture)
        *err = hipErrorStreamCaptureUnsupported;
        return nullptr;
    }
    if ((*err = grow()) != hipSuccess) return nullptr;
    return slot_for(stream, err);
}

int32_t *QueuePool::worklist_for(hipStream_t stream, size_t n, hipError_t *err) {
    std::lock_guard<std::mutex> lock(mu);
    std::pair<int32_t *, size_t> &l = lists[stream];
    if (l.second < n) {
        // (the stream's previous solve may still read the old list: wait for it before the buffer goes)
        if (l.first) { (void)hipStreamSynchronize(stream); (void)hipFree(l.first); l = {nullptr, 0}; }
        void *p = nullptr;
        const size_t cap = std::max<size_t>(n, 65536);
        if ((*err = hipMalloc(&p, cap * sizeof(int32_t))) != hipSuccess) return nullptr;
        l = {static_cast<int32_t *>(p), cap};
    }
    return l.first;
}

void QueuePool::release() {
    std::lock_guard<std::mutex> lock(mu);
    for (auto &kv : lists) (void)hipFree(kv.second.first);
    lists.clear();
    for (unsigned long long *c : chunks) (void)hipFree(c);
    chunks.clear();
    by_stream.clear();
    used_in_last = kChunkSlots;
}

// Grid of a persistent launch: as many workgroups as the device keeps resident (LDS-bound: five per CU for the demo task set),
// never more than there are groups of problems.  The occupancy query costs ~10 us: cached per (kernel, block, lds, device).  Call
// raise_lds_limit first: the query fails above the default 64 KB of dynamic LDS otherwise.
int64_t persistent_grid(const void *kernel, int block, size_t lds, int64_t nblocks) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    struct Key { const void *k; int block; size_t lds; int dev; int64_t resident; };
    static std::mutex mu;
    static std::vector<Key> cache;
    int64_t resident = 0;
    {
        std::lock_guard<std::mutex> lock(mu);
        for (const Key &c : cache)
            if (c.k == kernel && c.block == block && c.lds == lds && c.dev == dev) { resident = c.resident; break; }
    }
    if (resident == 0) {
        int per_cu = 0, cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, lds) != hipSuccess || per_cu < 1) per_cu = 1;
        resident = static_cast<int64_t>(per_cu) * cus;
        std::lock_guard<std::mutex> lock(mu);
        cache.push_back(Key{kernel, block, lds, dev, resident});
    }
    return nblocks < resident ? nblocks : resident;
}

#ifdef IKGPU_COOP_PROFILE
// Debug builds only: per-phase cycle counters of workgroup 0 of the cooperative kernel, accumulated over its iterations.
extern "C" int ikgpu_debug_coop_profile(long long *out16, int reset) {
    long long zero[16] = {};
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(ikdev::g_coop_prof), sizeof(zero)) != hipSuccess) return 1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(ikdev::g_coop_prof), zero, sizeof(zero)) != hipSuccess) return 1;
    return 0;
}
#endif

// ik::pik (reference ik/ik/pik.cpp:31-103) on the generic lane program; `gen` is the problem's generic analysis.
bool pik_runs_cooperative(const ProblemHost &gen, const ikgpu_pik_params &prm) {
    if (!gen.generic.coop_pik_ok) return false;
    for (int l = 0; l < gen.generic.nlevels; ++l)  // levels beyond the tasks' own are empty (capi.cpp: check_pik_params)
        if (!(prm.lambda[l] > 0.0)) return false;  // the cooperative form solves (Jbar Jbar^T + lambda^2 I) by Cholesky
    const char *force = std::getenv("IKGPU_GENERIC_KERNEL");
    return !(force && std::string(force) == "lane");
}

hipError_t launch_pik_generic(const ProblemHost &gen, const DeviceTables &dt, const BatchIO &io, const ikgpu_pik_params &prm,
                              hipStream_t stream) {
    ikdev::PikParams pp{};
    pp.max_iterations = prm.max_iterations;
    pp.step_length = prm.step_length;
    pp.stop_sq_tol = prm.stop_sq_tol;
    for (int l = 0; l < ikdev::kMaxPikLevels; ++l) pp.lam2[l] = l < prm.num_levels ? prm.lambda[l] * prm.lambda[l] : 1.0;
    pp.has_da = 0;
    if (prm.da)
        for (int k = 0; k < gen.nv; ++k) {
            pp.da[k] = prm.da[k];
            if (prm.da[k] != 0.0) pp.has_da = 1;
        }
    if (pik_runs_cooperative(gen, prm)) {
        ikdev::PikCoopKernelArgs c{};
        c.T = bind_generic_tables(gen, dt.g_ints, dt.g_dbls);
        c.L = bind_coop_layout(gen, dt.g_ints, /*for_pik=*/true);   // (the elimination of posture rows is the DLS solver's)
        c.K = bind_pik_coop_layout(gen);
        c.prm = pp;
        c.layout = io.layout; c.B = io.B; c.q0 = io.q0; c.targets = io.targets;
        c.q_out = io.q_out; c.success = io.success; c.iters = io.iters;
        const CoopStaging s{dt.g_ints, dt.g_dbls, static_cast<int>(gen.generic.ints.size()), static_cast<int>(gen.generic.dbls.size())};
        const int per_block = ikdev::kCoopPerBlock;
        const size_t lds = sizeof(double) * (static_cast<size_t>(per_block) * static_cast<size_t>(c.K.words) +
                                             static_cast<size_t>(s.n_dbls) + static_cast<size_t>((s.n_ints + 1) / 2));
        if (!raise_lds_limit(reinterpret_cast<const void *>(pik_coop_kernel), lds)) return hipGetLastError();
        const int64_t blocks = persistent_grid(reinterpret_cast<const void *>(pik_coop_kernel), per_block * ikdev::kCoopGroup, lds, (io.B + per_block - 1) / per_block);
        hipError_t qe = hipSuccess;
        unsigned long long *queue = dt.queues.slot_for(stream, &qe);
        if (!queue) return qe;
        hipLaunchKernelGGL(pik_coop_kernel, dim3(static_cast<unsigned>(blocks)), dim3(per_block * ikdev::kCoopGroup), lds, stream, c, s, queue);
        return hipGetLastError();
    }
    ikdev::PikKernelArgs a{};
    a.T = bind_generic_tables(gen, dt.g_ints, dt.g_dbls);
    a.prm = pp;
    a.layout = io.layout; a.B = io.B; a.q0 = io.q0; a.targets = io.targets;
    a.q_out = io.q_out; a.success = io.success; a.iters = io.iters;
    a.ws_stride = (io.B + kBlock - 1) / kBlock * kBlock;
    const size_t bytes = sizeof(double) * static_cast<size_t>(gen.generic.ws_words_pik) * static_cast<size_t>(a.ws_stride);
    void *ws = nullptr;
    hipError_t e = hipMallocAsync(&ws, bytes, stream);
    if (e != hipSuccess) return e;
    a.ws = static_cast<double *>(ws);
    hipLaunchKernelGGL(pik_generic_kernel, grid_for(io.B), dim3(kBlock), 0, stream, a);
    e = hipGetLastError();
    const hipError_t f = hipFreeAsync(ws, stream);
    return e != hipSuccess ? e : f;
}


// The per-lane program with its workspace column in LDS: when 64 columns fit the CU's 160 KB
bool generic_runs_in_lds(const ProblemHost &ph) {
    const char *force = std::getenv("IKGPU_GENERIC_KERNEL");
    if (!(force && std::string(force) == "lds")) return false;
    return static_cast<size_t>(ph.generic.ws_words) * 64 * sizeof(double) <= 160 * 1024;
}

bool generic_runs_cooperative(const ProblemHost &ph) {
    if (generic_runs_in_lds(ph)) return false;
    if (!ph.generic.coop_ok) return false;
    // beyond 31 rows the solve falls back to the LDS Cholesky, one barrier per pivot: measured SLOWER than the per-lane program there
    // (M = 32: 490 against 367 ms per launch; M = 31 with the two-row register form: 151 against 341)
    if ((ph.generic.coop_post_elim ? ph.generic.coop_Mf : ph.rows) > 31 && !std::getenv("IKGPU_GENERIC_COOP_ANY_SIZE")) return false;
    const char *force = std::getenv("IKGPU_GENERIC_KERNEL");  // "lane": keep the memory-resident per-lane program (tests, profiling)
    return !(force && std::string(force) == "lane");
}

hipError_t launch_dls_generic(const ProblemHost &ph, const DeviceTables &dt, const BatchIO &io, const ikgpu_dls_params &prm,
                              hipStream_t stream, bool force_lane) {
    if (!force_lane && generic_runs_cooperative(ph)) {
        ikdev::CoopKernelArgs c{};
        c.T = bind_generic_tables(ph, dt.g_ints, dt.g_dbls);
        c.L = bind_coop_layout(ph, dt.g_ints);
        c.prm.max_iterations = prm.max_iterations;
        c.prm.lam2 = prm.damping * prm.damping;
        c.prm.step_length = prm.step_length;
        c.prm.stop_sq_tol = prm.stop_sq_tol;
        c.layout = io.layout; c.B = io.B; c.q0 = io.q0; c.targets = io.targets;
        c.q_out = io.q_out; c.success = io.success; c.iters = io.iters;
        const CoopStaging s{dt.g_ints, dt.g_dbls, static_cast<int>(ph.generic.ints.size()), static_cast<int>(ph.generic.dbls.size())};
        const int per_block = ikdev::kCoopPerBlock;  // 4 x 16 lanes = one wave; 2 and 1 problems per workgroup measured slower
        const size_t lds = sizeof(double) * (static_cast<size_t>(per_block) * static_cast<size_t>(c.L.words) +
                                             static_cast<size_t>(s.n_dbls) + static_cast<size_t>((s.n_ints + 1) / 2));
        hipError_t qe = hipSuccess;
        unsigned long long *queue = dt.queues.slot_for(stream, &qe);
        if (!queue) return qe;
        const int rows = c.L.post_elim ? c.L.Mf : c.T.M;   // the size of the system the solver factorises
        if (rows >= 16 && rows <= 31) {
            if (!raise_lds_limit(reinterpret_cast<const void *>(dls_coop_big_kernel), lds)) return hipGetLastError();
            const int64_t blocks_big = persistent_grid(reinterpret_cast<const void *>(dls_coop_big_kernel), per_block * ikdev::kCoopGroup, lds, (io.B + per_block - 1) / per_block);
            hipLaunchKernelGGL(dls_coop_big_kernel, dim3(static_cast<unsigned>(blocks_big)), dim3(per_block * ikdev::kCoopGroup), lds, stream, c, s, queue);
            return hipGetLastError();
        }
        if (!raise_lds_limit(reinterpret_cast<const void *>(dls_coop_kernel), lds)) return hipGetLastError();
        const int64_t blocks = persistent_grid(reinterpret_cast<const void *>(dls_coop_kernel), per_block * ikdev::kCoopGroup, lds, (io.B + per_block - 1) / per_block);
        hipLaunchKernelGGL(dls_coop_kernel, dim3(static_cast<unsigned>(blocks)), dim3(per_block * ikdev::kCoopGroup), lds, stream, c, s, queue);
        return hipGetLastError();
    }
    ikdev::GenericKernelArgs a{};
    a.prm.max_iterations = prm.max_iterations;
    a.prm.lam2 = prm.damping * prm.damping;
    a.prm.step_length = prm.step_length;
    a.prm.stop_sq_tol = prm.stop_sq_tol;
    fill_visitor(a.prm, prm);
    a.layout = io.layout; a.q0 = io.q0; a.targets = io.targets;
    a.q_out = io.q_out; a.success = io.success; a.iters = io.iters;
    if (!force_lane && generic_runs_in_lds(ph)) {
        a.T = bind_generic_tables(ph, dt.g_ints, dt.g_dbls);
        a.B = io.B;
        const size_t lds = sizeof(double) * 64 * static_cast<size_t>(ph.generic.ws_words);
        if (!raise_lds_limit(reinterpret_cast<const void *>(dls_generic_lds_kernel), lds)) return hipGetLastError();
        hipLaunchKernelGGL(dls_generic_lds_kernel, dim3(static_cast<unsigned>((io.B + 63) / 64)), dim3(64), lds, stream, a);
        return hipGetLastError();
    }
    return with_workspace(ph, dt, a, io.B, stream, true);
}

hipError_t launch_eval_generic(const ProblemHost &ph, const DeviceTables &dt, int64_t B, const double *q, const double *targets,
                               double *e_out, double *J_out, double *oMf_out, int layout, hipStream_t stream) {
    ikdev::GenericKernelArgs a{};
    a.layout = layout; a.q0 = q; a.targets = targets;
    a.e_out = e_out; a.J_out = J_out; a.oMf_out = oMf_out;
    return with_workspace(ph, dt, a, B, stream, false);
}

}  // namespace ikgpu
