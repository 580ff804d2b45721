// capi.cpp -- implementation of include/ikgpu.h.  No exception crosses the boundary; every
// solve runs the gfx950 kernels (kernels.hip) or fails with a message -- there is no CPU path.
#include <hip/hip_runtime.h>

#include <time.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "ikgpu.h"
#include "kernels.hpp"
#include "model.hpp"
#include "problem.hpp"

// Staging area of the host-pointer entry points for small batches (host_solve below): grows on demand, guarded by a mutex a
// caller only ever try-locks.
struct Staging {
    std::mutex mu;
    void *dev = nullptr, *host = nullptr;
    size_t cap = 0;
};
constexpr size_t kStageLimit = size_t(1) << 20;  // batches whose buffers total at most 1 MiB take the staged path

// Per-phase wall clock of the pipelined host entry (host_solve below), switched on by IKGPU_HOST_TRACE=<file>: one line per call --
// total and the time spent waiting for the pipe's mutex, in set-up, enqueueing copies and launches, and in each of the final waits.
// (Round 3 saw 30-75 ms stalls about once per hundred calls: this is how the phase that carries them is found;
// tools/host_entry_tails.py reads the file.)
struct HostTrace {
    FILE *f = nullptr;
    bool on = false;
    HostTrace() {
        if (const char *path = std::getenv("IKGPU_HOST_TRACE")) { f = std::fopen(path, "a"); on = f != nullptr; }
    }
    ~HostTrace() { if (f) std::fclose(f); }
    static double now() {
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return 1e3 * static_cast<double>(ts.tv_sec) + 1e-6 * static_cast<double>(ts.tv_nsec);
    }
};

// Larger batches through the host-pointer entry points: a pipeline the problem keeps -- one device arena (grow-only), three
// streams, events -- so that a call allocates nothing and never synchronises the device: chunks of problems flow
// H2D (chunk k + 1)  ||  solve (chunk k)  ||  D2H (chunk k - 1).
struct HostPipe {
    std::mutex mu;
    void *dev = nullptr;
    size_t cap = 0;
    static constexpr int kRunStreams = 8;
    hipStream_t in = nullptr, out = nullptr;
    hipStream_t run[kRunStreams] = {};   // chunk k solves on run[k % 8]
    std::vector<hipEvent_t> ev_in, ev_run;
};

struct ikgpu_problem {
    mutable Staging stage;
    mutable HostPipe pipe;
    ikgpu::ProblemHost host;
    ikgpu::ProblemHost gen;  // the same problem analysed for the generic lane program (what ik::pik runs on)
    ikgpu::DeviceTables dev;
    int device = 0;
    int nframes = 0;
    // A Tree-kind problem with few rows runs ik::dls on the lane program specialised for it at run time instead (gen.generic_build == 2):
    // the dense M x M system of a small task set is cheaper than the tree kernel's arrow elimination (the reference demo's own task
    // set, M = 10: 0.39 against 0.51 ms per 65536 problems); `host` stays the tree analysis (stage kernels, two-level ik::pik)
    bool dls_on_static_gen = false;
    // A derived visitor (ikgpu_dls_params::dq_sq_tol / level_sq_tol) runs on the generic lane program whatever kernel ik::dls itself
    // uses.  For Chain / Tree problems that program's static build is compiled at the FIRST such solve (most callers never use a
    // derived visitor, and creation should not pay for it); until it exists -- or where it cannot -- the per-lane interpreter runs.
    mutable std::once_flag visitor_static_once;
    mutable uint64_t visitor_static_key = 0;
    mutable bool visitor_static = false;
    // ik::pik beyond one level / the tree kernel's two: a compiled lane program (rtc.cpp rtc_pik_static_available), one per
    // (problem, with / without the secondary step da), compiled at the FIRST ik::pik call that wants it (or by
    // ikgpu_problem_precompile) -- most problems are only ever handed to ik::dls
    mutable std::once_flag pik_static_once[2];
    mutable uint64_t pik_static_key[2] = {0, 0};
    mutable bool pik_static[2] = {false, false};
    mutable std::string pik_static_name;
    std::string dls_name;    // what ikgpu_problem_kernel reports
    std::string pik_name;    // name of the generic PIK kernel instance
    std::string pik_tree_name;  // ... and of the tree kernel running a two-level ik::pik (when the problem has that shape)
};

namespace {

thread_local std::string g_last_error;

bool shape_built(const ikgpu::ProblemHost &ph) {
    if (ph.kind == ikgpu::KernelKind::Generic) return true;
    return ph.kind == ikgpu::KernelKind::Chain ? ikgpu::chain_shape_built(ph.chain.nj, ph.tasks[0].type)
                                               : ikgpu::tree_shape_built(ph.chain.nj, ph.chainB.nj > 0 ? 2 : 1);
}

// Analysis + the "is this specialisation compiled" check; a specialised shape without an instantiation
// falls back to the generic kernel.  Throws std::runtime_error on invalid input.
ikgpu::ProblemHost analyse(const ikgpu::Model &m, const ikgpu_task *tasks, int32_t ntasks, const ikgpu_task *cons = nullptr,
                           int32_t ncons = 0, bool compile_rtc = false) {
    // IKGPU_DLS_KERNEL=generic skips the register-resident specialisations (the parity tests compare them with the generic kernel)
    const char *force = std::getenv("IKGPU_DLS_KERNEL");
    ikgpu::ProblemHost ph = ikgpu::analyse_problem(m, tasks, ntasks, force && std::strcmp(force, "generic") == 0, cons, ncons);
    if (!shape_built(ph)) ph = ikgpu::analyse_problem(m, tasks, ntasks, /*force_generic=*/true, cons, ncons);
    if (ph.kind == ikgpu::KernelKind::Chain) {   // which build of the chain kernel: decided here, once, and part of the name
        ph.chain_build = ikgpu::select_chain_build(ph, compile_rtc);
        ph.kernel_name = ikgpu::chain_kernel_name(ph);
    }
    if (ph.kind == ikgpu::KernelKind::Generic && ikgpu::rtc_generic_static_available(ph, compile_rtc, &ph.generic_key)) {
        ph.generic_build = 2;   // the lane program specialised for this problem (the name says so: "...,static>")
        ph.kernel_name = ph.kernel_name.substr(0, ph.kernel_name.size() - 1) + ",static>";
    }
    return ph;
}

int fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

// Tree-kind problems the static generic program is tried for first: few rows in the system it solves (IKGPU_TREE_STATIC_ROWS, default
// 12; 0 = never; PostureTask rows do not count when the program eliminates them) and no constraint -- measured, B = 65536, 50
// iterations: the demo's task set (M = 10) 0.40 ms static against 0.51 ms on the tree kernel, with the posture regulariser on all 16
// joints (M = 26, a 10 x 10 system) 0.53 against 0.65,
// but with the right foot pinned 0.78 against 0.71 (the tree kernel projects on the constrained chain's 13 columns only; the static
// program orthogonalises three dense rows of 22).  IKGPU_TREE_STATIC_CONSTRAINED=1 routes those too (tests).
bool tree_prefers_static(const ikgpu::ProblemHost &ph) {
    if (ph.kind != ikgpu::KernelKind::Tree) return false;
    long rows = 12;
    if (const char *env = std::getenv("IKGPU_TREE_STATIC_ROWS")) rows = std::strtol(env, nullptr, 10);
    if (ph.cons_on && !std::getenv("IKGPU_TREE_STATIC_CONSTRAINED")) return false;
    return ikgpu::rtc_static_solve_rows(ph) <= rows;
}

std::string static_name(const ikgpu::ProblemHost &gen) {
    const std::string &n = gen.kernel_name;
    return n.size() > 8 && n.compare(n.size() - 8, 8, ",static>") == 0 ? n : n.substr(0, n.size() - 1) + ",static>";
}

}  // namespace

// (shard.cpp reports through the same thread-local message)
int ikgpu_set_last_error(int code, const std::string &msg) { return fail(code, msg); }

namespace {

int hip_fail(hipError_t e, const char *what) {
    return fail(IKGPU_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}

template <class F>
int guarded(F &&f) {
    try {
        return f();
    } catch (const std::bad_alloc &) {
        return fail(IKGPU_ERR_INVALID, "out of host memory");
    } catch (const std::exception &e) {
        return fail(IKGPU_ERR_INVALID, e.what());
    } catch (...) {
        return fail(IKGPU_ERR_INVALID, "unknown C++ exception");
    }
}

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

int check_params(const ikgpu_dls_params *p) {
    if (!p) return fail(IKGPU_ERR_INVALID, "params is null");
    if (p->max_iterations < 0) return fail(IKGPU_ERR_INVALID, "max_iterations must be >= 0");
    if (!(p->damping > 0.0))
        return fail(IKGPU_ERR_INVALID, "damping must be > 0: the device solves JJ^T + damping^2 I by Cholesky (SPD)");
    if (p->num_level_tols < 0 || p->num_level_tols > IKGPU_MAX_VISITOR_LEVELS)
        return fail(IKGPU_ERR_INVALID, "num_level_tols must be in 0.." + std::to_string(IKGPU_MAX_VISITOR_LEVELS));
    return IKGPU_OK;
}

int check_pik_params(const ikgpu_problem *p, const ikgpu_pik_params *prm) {
    if (!prm) return fail(IKGPU_ERR_INVALID, "params is null");
    if (!p) return fail(IKGPU_ERR_INVALID, "null problem");
    if (prm->max_iterations < 0) return fail(IKGPU_ERR_INVALID, "max_iterations must be >= 0");
    const int levels = p->gen.generic.nlevels;
    if (levels > IKGPU_MAX_PIK_LEVELS)
        return fail(IKGPU_ERR_UNSUPPORTED, "the problem has " + std::to_string(levels) + " priority levels, ik::pik on the device takes at most " +
                                               std::to_string(IKGPU_MAX_PIK_LEVELS));
    // a problem may declare more levels than its tasks use (the demo does, reference ik_ros/src/cassie.cpp:43): the
    // reference's loop over 0..max_priority_level (ik/ik/pik.cpp:47) is a no-op on a level without rows
    if (prm->num_levels < levels || prm->num_levels > IKGPU_MAX_PIK_LEVELS)
        return fail(IKGPU_ERR_INVALID, "num_levels is " + std::to_string(prm->num_levels) + " but the problem's tasks use " +
                                           std::to_string(levels) + " priority levels (at most " + std::to_string(IKGPU_MAX_PIK_LEVELS) + ")");
    for (int l = 0; l < levels; ++l)
        if (!(prm->lambda[l] >= 0.0)) return fail(IKGPU_ERR_INVALID, "lambda[" + std::to_string(l) + "] must be >= 0");
    if (prm->da && p->gen.nv > IKGPU_MAX_PIK_DA)
        return fail(IKGPU_ERR_UNSUPPORTED, "da is carried by value for nv <= " + std::to_string(IKGPU_MAX_PIK_DA));
    return IKGPU_OK;
}

// ik::pik with ONE priority level and no secondary velocity is the DLS iteration: P = I, so the level's step is
// dq = -damp_pinv(J, lambda) e = -J^T (J J^T + lambda^2 I)^-1 e (reference ik/ik/pik.cpp:5-21,47-61 against ik/ik/dls.cpp:39-53),
// and the stop test, integration and clamp are the same statements in the same order (pik.cpp:67-77, dls.cpp:61-71).
// ik::pik does not read the problem's constraints, so a problem that has any stays on the PIK kernel.
// IKGPU_PIK_KERNEL=generic keeps every ik::pik call on the PIK kernel (the parity tests compare the two).
bool pik_is_one_dls_level(const ikgpu_problem *p, const ikgpu_pik_params *prm) {
    const char *force = std::getenv("IKGPU_PIK_KERNEL");
    if (force && (std::strcmp(force, "generic") == 0 || std::strcmp(force, "static") == 0)) return false;
    return p->gen.generic.nlevels == 1 && !prm->da && prm->lambda[0] > 0.0 && p->host.constraints.empty();
}

// ik::pik with TWO levels in the shape the tree kernel takes (kernels.hpp tree_takes_two_level_pik: level 0 = the frame tasks with a
// Full task on the base link, level 1 = the AlignAxisTask row -- the reference demo's task set split over two levels): level 0 is
// the tree kernel's arrow solve with damping lambda[0], level 1 a rank-one correction on the chain's joints (device/tree_solver.hpp
// PikRow).  No secondary velocity, lambda > 0 on both levels.
bool pik_is_two_levels_on_the_tree(const ikgpu_problem *p, const ikgpu_pik_params *prm) {
    const char *force = std::getenv("IKGPU_PIK_KERNEL");   // generic: the interpreter forms; static: the compiled lane program where there is one
    if (force && (std::strcmp(force, "generic") == 0 || std::strcmp(force, "static") == 0)) return false;
    return p->gen.generic.nlevels == 2 && prm->num_levels == 2 && !prm->da && prm->lambda[0] > 0.0 && prm->lambda[1] > 0.0 &&
           p->host.constraints.empty() && ikgpu::tree_takes_two_level_pik(p->host);
}

// ik::pik on its compiled lane program: every level's lambda > 0 (the program factors Jbar Jbar^T + lambda^2 I), the program exists
// (compiled here on first use).  IKGPU_PIK_KERNEL=generic / IKGPU_PIK_STATIC=0 keep the interpreter forms.
bool pik_runs_static(const ikgpu_problem *p, const ikgpu_pik_params *prm) {
    const char *force = std::getenv("IKGPU_PIK_KERNEL");
    if (force && std::strcmp(force, "generic") == 0) return false;
    for (int l = 0; l < prm->num_levels; ++l)
        if (!(prm->lambda[l] > 0.0)) return false;
    bool has_da = false;
    if (prm->da)
        for (int k = 0; k < p->gen.nv; ++k) has_da = has_da || prm->da[k] != 0.0;
    const int v = has_da ? 1 : 0;
    if (!ikgpu::rtc_pik_static_available(p->gen, has_da, /*compile=*/false, nullptr)) return false;
    std::call_once(p->pik_static_once[v], [&] {
        p->pik_static[v] = ikgpu::rtc_pik_static_available(p->gen, has_da, /*compile=*/true, &p->pik_static_key[v]);
    });
    return p->pik_static[v];
}

// Host-pointer form of a batched solve: copy in, run `launch` on device buffers, synchronise, copy out.
template <class Launch>
int host_solve(const ikgpu_problem *p, int64_t B, const double *q0, const double *targets, double *q_out, uint8_t *success,
               int32_t *iters, int layout_in, Launch &&launch) {
    const bool pose7 = (layout_in & IKGPU_TARGETS_POSE7) != 0;   // targets arrive as 7 doubles per task and are expanded on the device
    const int layout = layout_in & ~IKGPU_TARGETS_POSE7;
    return guarded([&] {
        DeviceGuard g(p->device);
        if (!g.ok) return fail(IKGPU_ERR_DEVICE, "hipSetDevice failed");
        const size_t nb_q = sizeof(double) * p->host.nq * B, nb_t = sizeof(double) * 12 * p->host.ntasks * B;
        const size_t nb_t7 = pose7 ? sizeof(double) * 7 * p->host.ntasks * B : 0;
        // Small batches -- the reference's own call pattern is ONE problem per call, 50 times a second (ik_ros/src/cassie.cpp:112)
        // -- go through a staging area the problem keeps: one pinned host buffer and one device buffer laid out
        // [q0 | targets | q_out | iters | success], so a call is two copies and a launch instead of five allocations, five
        // copies, a device synchronise and five frees.  A second thread calling on the same problem meanwhile takes the path below.
        const size_t off_t = nb_q, off_q = off_t + nb_t, off_i = off_q + nb_q, off_s = off_i + sizeof(int32_t) * B;
        const size_t off_7 = (off_s + B + 7) / 8 * 8;                  // (pose7 targets land behind everything else)
        const size_t total = off_7 + nb_t7;
        if (total <= kStageLimit) {
            std::unique_lock<std::mutex> lock(p->stage.mu, std::try_to_lock);
            if (lock.owns_lock()) {
                Staging &st = p->stage;
                if (st.cap < total) {
                    if (st.dev) (void)hipFree(st.dev);
                    if (st.host) (void)hipHostFree(st.host);
                    st.dev = st.host = nullptr;
                    st.cap = 0;
                    const size_t cap = std::max<size_t>(total, 4096);
                    if (hipMalloc(&st.dev, cap) == hipSuccess && hipHostMalloc(&st.host, cap, hipHostMallocDefault) == hipSuccess) st.cap = cap;
                }
                if (st.cap >= total) {
                    char *h = static_cast<char *>(st.host), *d = static_cast<char *>(st.dev);
                    std::memcpy(h, q0, nb_q);
                    hipError_t e = hipSuccess;
                    if (pose7) {
                        std::memcpy(h + off_7, targets, nb_t7);
                        e = hipMemcpy(d, h, nb_q, hipMemcpyHostToDevice);
                        if (e == hipSuccess) e = hipMemcpy(d + off_7, h + off_7, nb_t7, hipMemcpyHostToDevice);
                        if (e == hipSuccess) e = ikgpu::launch_targets_from_pose7(B, p->host.ntasks, reinterpret_cast<const double *>(d + off_7),
                                                                                  reinterpret_cast<double *>(d + off_t), layout, nullptr);
                    } else {
                        std::memcpy(h + off_t, targets, nb_t);
                        e = hipMemcpy(d, h, off_q, hipMemcpyHostToDevice);
                    }
                    if (e != hipSuccess) return hip_fail(e, "host-pointer solve (staged copy in)");
                    const int rc = launch(B, reinterpret_cast<double *>(d), reinterpret_cast<double *>(d + off_t), reinterpret_cast<double *>(d + off_q),
                                          reinterpret_cast<uint8_t *>(d + off_s), reinterpret_cast<int32_t *>(d + off_i), nullptr);
                    if (rc != IKGPU_OK) return rc;
                    e = hipMemcpy(h + off_q, d + off_q, total - off_q, hipMemcpyDeviceToHost);   // waits for the launch on the null stream
                    if (e != hipSuccess) return hip_fail(e, "host-pointer solve (staged copy out)");
                    std::memcpy(q_out, h + off_q, nb_q);
                    if (iters) std::memcpy(iters, h + off_i, sizeof(int32_t) * B);
                    if (success) std::memcpy(success, h + off_s, B);
                    return static_cast<int>(IKGPU_OK);
                }
            }
        }
        // The pipelined path.  Chunk k lives compactly on the device ([rows][b_k], component-major, or [b_k][rows]); for the
        // component-major layout one 2-D copy per array gathers / scatters the chunk's columns of the caller's [rows][B] arrays.
        // Pinned caller buffers make every copy asynchronous; pageable ones still work (the runtime stages them).
        static HostTrace trace;
        const double t_enter = trace.on ? HostTrace::now() : 0.0;
        std::lock_guard<std::mutex> plock(p->pipe.mu);
        const double t_locked = trace.on ? HostTrace::now() : 0.0;
        HostPipe &pp = p->pipe;
        // Chunks that fill the device: one problem per lane means a launch lasts as long as ONE wave whatever its size, and kernels of
        // different streams were measured NOT to overlap here (B = 65536 in 8 chunks on 8 streams: 0.93 ms; 2 chunks: 0.55 ms; 1 chunk,
        // i.e. no overlap at all: 0.61 ms; tools/host_entry_timing.py) -- so two halves up to 131072 problems, 65536 per chunk above
        // (B = 262144: 1.49 ms against 2.21 unpipelined).
        int64_t chunk = B <= 131072 ? ((B + 1) / 2 + 63) / 64 * 64 : 65536;
        if (const char *env = std::getenv("IKGPU_HOST_CHUNK")) { const long c = std::strtol(env, nullptr, 10); if (c >= 64) chunk = c; }
        const int64_t nchunks = (B + chunk - 1) / chunk;
        const size_t nq = static_cast<size_t>(p->host.nq), nt = static_cast<size_t>(12 * p->host.ntasks);
        const size_t nt7 = pose7 ? static_cast<size_t>(7 * p->host.ntasks) : 0;
        const size_t per_problem = 8 * nq + 8 * nt + 8 * nq + 4 + 1 + 8 * nt7;
        const size_t need = per_problem * static_cast<size_t>(B) + 64 * static_cast<size_t>(nchunks) * 6;   // (every array 64-byte aligned)
        hipError_t e = hipSuccess;
        auto step = [&](hipError_t r) { if (e == hipSuccess) e = r; };
        if (!pp.in) {
            step(hipStreamCreateWithFlags(&pp.in, hipStreamNonBlocking));
            step(hipStreamCreateWithFlags(&pp.out, hipStreamNonBlocking));
            for (hipStream_t &r : pp.run) step(hipStreamCreateWithFlags(&r, hipStreamNonBlocking));
        }
        if (e == hipSuccess && pp.cap < need) {
            if (pp.dev) { (void)hipFree(pp.dev); pp.dev = nullptr; pp.cap = 0; }   // (every call leaves its streams idle)
            step(hipMalloc(&pp.dev, need));
            if (e == hipSuccess) pp.cap = need;
        }
        while (e == hipSuccess && static_cast<int64_t>(pp.ev_in.size()) < nchunks) {
            hipEvent_t a = nullptr, b = nullptr;
            step(hipEventCreateWithFlags(&a, hipEventDisableTiming));
            step(hipEventCreateWithFlags(&b, hipEventDisableTiming));
            if (e == hipSuccess) { pp.ev_in.push_back(a); pp.ev_run.push_back(b); }
        }
        if (e != hipSuccess) return hip_fail(e, "host-pointer solve (pipeline set-up)");
        char *cur = static_cast<char *>(pp.dev);
        auto take = [&](size_t bytes) { char *r = cur; cur += (bytes + 63) / 64 * 64; return r; };
        const bool soa = layout == IKGPU_SOA;
        // rows x [b0, b0 + bk) of a host array with B columns  <->  a compact rows x bk device array (SoA), or bk x rows contiguous (AoS)
        auto copy = [&](void *dev, const void *host_c, void *host, size_t rows, size_t elem, int64_t b0, int64_t bk, bool to_device, hipStream_t st) {
            if (soa && rows > 1) {
                const size_t w = static_cast<size_t>(bk) * elem, hp = static_cast<size_t>(B) * elem;
                return to_device ? hipMemcpy2DAsync(dev, w, static_cast<const char *>(host_c) + static_cast<size_t>(b0) * elem, hp, w, rows, hipMemcpyHostToDevice, st)
                                 : hipMemcpy2DAsync(static_cast<char *>(host) + static_cast<size_t>(b0) * elem, hp, dev, w, w, rows, hipMemcpyDeviceToHost, st);
            }
            const size_t off = static_cast<size_t>(b0) * rows * elem, bytes = static_cast<size_t>(bk) * rows * elem;
            return to_device ? hipMemcpyAsync(dev, static_cast<const char *>(host_c) + off, bytes, hipMemcpyHostToDevice, st)
                             : hipMemcpyAsync(static_cast<char *>(host) + off, dev, bytes, hipMemcpyDeviceToHost, st);
        };
        int rc = IKGPU_OK;
        const double t_setup = trace.on ? HostTrace::now() : 0.0;
        for (int64_t k = 0; k < nchunks && rc == IKGPU_OK && e == hipSuccess; ++k) {
            const int64_t b0 = k * chunk, bk = std::min<int64_t>(chunk, B - b0);
            double *d_q0 = reinterpret_cast<double *>(take(8 * nq * bk)), *d_t = reinterpret_cast<double *>(take(8 * nt * bk));
            double *d_q = reinterpret_cast<double *>(take(8 * nq * bk));
            int32_t *d_i = reinterpret_cast<int32_t *>(take(4 * bk));
            uint8_t *d_s = reinterpret_cast<uint8_t *>(take(bk));
            double *d_t7 = pose7 ? reinterpret_cast<double *>(take(8 * nt7 * bk)) : nullptr;
            step(copy(d_q0, q0, nullptr, nq, 8, b0, bk, true, pp.in));
            if (pose7) step(copy(d_t7, targets, nullptr, nt7, 8, b0, bk, true, pp.in));
            else step(copy(d_t, targets, nullptr, nt, 8, b0, bk, true, pp.in));
            step(hipEventRecord(pp.ev_in[k], pp.in));
            const hipStream_t run = pp.run[k % HostPipe::kRunStreams];
            step(hipStreamWaitEvent(run, pp.ev_in[k], 0));
            if (pose7) step(ikgpu::launch_targets_from_pose7(bk, p->host.ntasks, d_t7, d_t, layout, run));
            if (e != hipSuccess) break;
            rc = launch(bk, d_q0, d_t, d_q, d_s, d_i, run);
            if (rc != IKGPU_OK) break;
            step(hipEventRecord(pp.ev_run[k], run));
            step(hipStreamWaitEvent(pp.out, pp.ev_run[k], 0));
            step(copy(d_q, nullptr, q_out, nq, 8, b0, bk, false, pp.out));
            if (iters) step(copy(d_i, nullptr, iters, 1, 4, b0, bk, false, pp.out));
            if (success) step(copy(d_s, nullptr, success, 1, 1, b0, bk, false, pp.out));
        }
        // The arena is reused by the next call: everything in flight has to land first.  When every chunk was enqueued, the copy-out
        // stream is the LAST link of every chain (in -> run[k] -> out, by events), so one wait on it covers all three; after a failure
        // half way through every stream that was touched is waited for.  (Round 3 waited for all ten streams, used or not.)
        const double t_enqueued = trace.on ? HostTrace::now() : 0.0;
        hipError_t w = hipSuccess;
        if (rc != IKGPU_OK || e != hipSuccess) {
            w = hipStreamSynchronize(pp.in);
            for (int64_t k = 0; k < std::min<int64_t>(nchunks, HostPipe::kRunStreams); ++k) { const hipError_t x = hipStreamSynchronize(pp.run[k]); if (w == hipSuccess) w = x; }
        }
        { const hipError_t x = hipStreamSynchronize(pp.out); if (w == hipSuccess) w = x; }
        if (trace.on) {
            const double t_done = HostTrace::now();
            std::fprintf(trace.f, "B %lld chunks %lld total_ms %.4f lock %.4f setup %.4f enqueue %.4f wait %.4f\n", static_cast<long long>(B),
                         static_cast<long long>(nchunks), t_done - t_enter, t_locked - t_enter, t_setup - t_locked, t_enqueued - t_setup, t_done - t_enqueued);
            std::fflush(trace.f);
        }
        if (rc != IKGPU_OK) return rc;
        step(w);
        if (e != hipSuccess) return hip_fail(e, "host-pointer solve (pipeline)");
        return static_cast<int>(IKGPU_OK);
    });
}

}  // namespace

extern "C" {

int ikgpu_abi_version(void) { return IKGPU_ABI_VERSION; }

const char *ikgpu_last_error(void) { return g_last_error.c_str(); }

void ikgpu_dls_params_default(ikgpu_dls_params *p) {
    if (!p) return;
    p->max_iterations = 100;  // reference ik/ik/common.hpp:61
    p->damping = 1e-2;        // reference ik/ik/dls.hpp:25
    p->step_length = 1.0;     // reference ik/ik/common.hpp:65
    p->stop_sq_tol = 1e-4;    // reference ik/ik/visitor.hpp:19
    p->dq_sq_tol = 0.0;       // the derived-visitor family: off (the reference's own visitor)
    p->num_level_tols = 0;
    for (double &t : p->level_sq_tol) t = 0.0;
}

int ikgpu_model_from_urdf(const char *xml, size_t len, int root_joint, ikgpu_model **out) {
    if (!xml || !out) return fail(IKGPU_ERR_INVALID, "null argument");
    if (root_joint != IKGPU_ROOT_FIXED && root_joint != IKGPU_ROOT_FREEFLYER)
        return fail(IKGPU_ERR_INVALID, "root_joint must be IKGPU_ROOT_FIXED or IKGPU_ROOT_FREEFLYER");
    *out = nullptr;
    try {
        auto *h = new ikgpu_model{ikgpu::Model::from_urdf(xml, len, root_joint == IKGPU_ROOT_FREEFLYER)};
        h->m.finalize();  // name views must point into the final object
        *out = h;
        return IKGPU_OK;
    } catch (const std::exception &e) {
        return fail(IKGPU_ERR_PARSE, e.what());
    } catch (...) {
        return fail(IKGPU_ERR_PARSE, "unknown C++ exception");
    }
}

int ikgpu_model_create(const ikgpu_flat_model *flat, ikgpu_model **out) {
    if (!flat || !out) return fail(IKGPU_ERR_INVALID, "null argument");
    *out = nullptr;
    return guarded([&] {
        auto *h = new ikgpu_model{ikgpu::Model::from_flat(*flat)};
        h->m.finalize();
        *out = h;
        return static_cast<int>(IKGPU_OK);
    });
}

void ikgpu_model_destroy(ikgpu_model *m) { delete m; }

int ikgpu_model_get_flat(const ikgpu_model *h, ikgpu_flat_model *out) {
    if (!h || !out) return fail(IKGPU_ERR_INVALID, "null argument");
    const ikgpu::Model &m = h->m;
    out->njoints = m.njoints();
    out->nq = m.nq;
    out->nv = m.nv;
    out->nframes = m.nframes();
    out->joint_type = m.joint_type.data();
    out->joint_parent = m.joint_parent.data();
    out->joint_idx_q = m.joint_idx_q.data();
    out->joint_idx_v = m.joint_idx_v.data();
    out->joint_placement = m.joint_placement.empty() ? nullptr : m.joint_placement[0].data();
    out->joint_axis = m.joint_axis.empty() ? nullptr : m.joint_axis[0].data();
    out->joint_mass = m.joint_mass.data();
    out->joint_com = m.joint_com.empty() ? nullptr : m.joint_com[0].data();
    out->lower = m.lower.data();
    out->upper = m.upper.data();
    out->frame_parent = m.frame_parent.data();
    out->frame_placement = m.frame_placement.empty() ? nullptr : m.frame_placement[0].data();
    out->joint_names = m.joint_name_ptrs.data();
    out->frame_names = m.frame_name_ptrs.data();
    return IKGPU_OK;
}

int32_t ikgpu_model_frame_id(const ikgpu_model *h, const char *name) {
    if (!h || !name) return -1;
    return h->m.frame_id(name);
}

int32_t ikgpu_model_joint_id(const ikgpu_model *h, const char *name) {
    if (!h || !name) return -1;
    return h->m.joint_id(name);
}

int ikgpu_problem_create(const ikgpu_model *h, const ikgpu_task *tasks, int32_t ntasks, int32_t device,
                         ikgpu_problem **out) {
    return ikgpu_problem_create_constrained(h, tasks, ntasks, nullptr, 0, device, out);
}

int ikgpu_problem_create_constrained(const ikgpu_model *h, const ikgpu_task *tasks, int32_t ntasks, const ikgpu_task *constraints,
                                     int32_t nconstraints, int32_t device, ikgpu_problem **out) {
    if (!h || !tasks || !out || (nconstraints > 0 && !constraints)) return fail(IKGPU_ERR_INVALID, "null argument");
    if (nconstraints < 0) return fail(IKGPU_ERR_INVALID, "negative constraint count");
    *out = nullptr;
    ikgpu::ProblemHost ph, gen;
    try {
        ph = analyse(h->m, tasks, ntasks, constraints, nconstraints, /*compile_rtc=*/true);
        gen = ph.kind == ikgpu::KernelKind::Generic
                  ? ph
                  : ikgpu::analyse_problem(h->m, tasks, ntasks, /*force_generic=*/true, constraints, nconstraints);
    } catch (const std::exception &e) {
        return fail(IKGPU_ERR_INVALID, e.what());
    }

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) return fail(IKGPU_ERR_DEVICE, "no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail(IKGPU_ERR_INVALID, "device ordinal out of range");
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return hip_fail(e, "hipGetDeviceProperties");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(IKGPU_ERR_DEVICE, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");

    return guarded([&] {
        DeviceGuard g(device);
        if (!g.ok) return fail(IKGPU_ERR_DEVICE, "hipSetDevice failed");
        auto *p = new ikgpu_problem;
        p->host = std::move(ph);
        p->gen = std::move(gen);
        p->dls_name = p->host.kernel_name;
        if (tree_prefers_static(p->host) && ikgpu::rtc_generic_static_available(p->gen, /*compile=*/true, &p->gen.generic_key)) {
            p->gen.generic_build = 2;
            p->dls_on_static_gen = true;
            p->dls_name = static_name(p->gen);
        }
        if (p->host.kind == ikgpu::KernelKind::Generic) {   // (analyse() compiled it: gen is a copy of host)
            p->gen.generic_build = p->host.generic_build;
            p->gen.generic_key = p->host.generic_key;
        }
        {
            std::string gname = p->gen.kernel_name;   // ("...,static>": the DLS program's build, not ik::pik's)
            const size_t st = gname.find(",static>");
            if (st != std::string::npos) gname = gname.substr(0, st) + ">";
            p->pik_name = "pik_generic" + gname.substr(std::min(gname.find('<'), gname.size()));
        }
        p->pik_tree_name = p->host.kernel_name.substr(0, p->host.kernel_name.size() - (p->host.kernel_name.empty() ? 0 : 1)) + ",pik_levels=2>";
        p->device = device;
        p->nframes = h->m.nframes();
        const size_t nq = static_cast<size_t>(p->host.nq);
        hipError_t err = hipSuccess;
        auto up = [&](auto **dst, const void *src, size_t bytes) {
            if (err != hipSuccess) return;
            err = hipMalloc(reinterpret_cast<void **>(dst), bytes ? bytes : 8);
            if (err == hipSuccess && bytes) err = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
        };
        up(&p->dev.lower, p->host.lower.data(), nq * sizeof(double));
        up(&p->dev.upper, p->host.upper.data(), nq * sizeof(double));
        up(&p->dev.q_in_chain, p->host.q_in_chain.data(), nq);
        up(&p->dev.g_ints, p->gen.generic.ints.data(), p->gen.generic.ints.size() * sizeof(int32_t));
        up(&p->dev.g_dbls, p->gen.generic.dbls.data(), p->gen.generic.dbls.size() * sizeof(double));
        if (err == hipSuccess) err = p->dev.queues.grow();
        if (p->host.kind != ikgpu::KernelKind::Generic) {
            const std::vector<double> desc = p->host.kind == ikgpu::KernelKind::Chain ? ikgpu::chain_desc_table(p->host)
                                                                                     : ikgpu::tree_desc_table(p->host);
            up(&p->dev.chain_desc, desc.data(), desc.size() * sizeof(double));
        }
        if (err != hipSuccess) {
            ikgpu_problem_destroy(p);
            return hip_fail(err, "uploading problem tables");
        }
        *out = p;
        return static_cast<int>(IKGPU_OK);
    });
}

int ikgpu_problem_plan(const ikgpu_model *h, const ikgpu_task *tasks, int32_t ntasks, char *out, size_t cap) {
    return ikgpu_problem_plan_constrained(h, tasks, ntasks, nullptr, 0, out, cap);
}

int ikgpu_problem_plan_constrained(const ikgpu_model *h, const ikgpu_task *tasks, int32_t ntasks, const ikgpu_task *constraints,
                                   int32_t nconstraints, char *out, size_t cap) {
    if (!h || !tasks || (nconstraints > 0 && !constraints)) return fail(IKGPU_ERR_INVALID, "null argument");
    if (nconstraints < 0) return fail(IKGPU_ERR_INVALID, "negative constraint count");
    try {
        const ikgpu::ProblemHost ph = analyse(h->m, tasks, ntasks, constraints, nconstraints);
        std::string name = ph.kernel_name;
        if (tree_prefers_static(ph)) {
            const ikgpu::ProblemHost gen = ikgpu::analyse_problem(h->m, tasks, ntasks, /*force_generic=*/true, constraints, nconstraints);
            if (ikgpu::rtc_generic_static_available(gen, /*compile=*/false, nullptr)) name = static_name(gen);
        }
        if (out && cap) {
            std::strncpy(out, name.c_str(), cap - 1);
            out[cap - 1] = '\0';
        }
        return IKGPU_OK;
    } catch (const std::exception &e) {
        return fail(IKGPU_ERR_INVALID, e.what());
    }
}

int ikgpu_problem_precompile(const ikgpu_model *h, const ikgpu_task *tasks, int32_t ntasks, const ikgpu_task *constraints,
                             int32_t nconstraints, char *out, size_t cap) {
    if (!h || !tasks || (nconstraints > 0 && !constraints)) return fail(IKGPU_ERR_INVALID, "null argument");
    if (nconstraints < 0) return fail(IKGPU_ERR_INVALID, "negative constraint count");
    try {
        const ikgpu::ProblemHost planned = analyse(h->m, tasks, ntasks, constraints, nconstraints, /*compile_rtc=*/false);
        const ikgpu::ProblemHost ph = analyse(h->m, tasks, ntasks, constraints, nconstraints, /*compile_rtc=*/true);
        std::string want = planned.kernel_name, got = ph.kernel_name;
        if (tree_prefers_static(ph)) {
            ikgpu::ProblemHost gen = ikgpu::analyse_problem(h->m, tasks, ntasks, /*force_generic=*/true, constraints, nconstraints);
            if (ikgpu::rtc_generic_static_available(gen, /*compile=*/false, nullptr)) {
                want = static_name(gen);
                if (ikgpu::rtc_generic_static_available(gen, /*compile=*/true, &gen.generic_key)) {
                    got = want;
                    (void)ikgpu::rtc_generic_static_precompile_refill(gen);
                }
            }
        } else if (ph.generic_build == 2) {
            (void)ikgpu::rtc_generic_static_precompile_refill(ph);
        }
        {   // a problem with several priority levels may be handed to ik::pik: its compiled lane program (without the secondary step)
            const ikgpu::ProblemHost gen = ph.kind == ikgpu::KernelKind::Generic
                                               ? ph : ikgpu::analyse_problem(h->m, tasks, ntasks, /*force_generic=*/true, constraints, nconstraints);
            // (one level needs the program only for a secondary step, da != 0: compiled at the first such call)
            if (gen.generic.nlevels >= 2 && ikgpu::rtc_pik_static_available(gen, false, /*compile=*/false, nullptr))
                (void)ikgpu::rtc_pik_static_available(gen, false, /*compile=*/true, nullptr);
        }
        if (out && cap) {
            std::strncpy(out, got.c_str(), cap - 1);
            out[cap - 1] = '\0';
        }
        if (want != got)   // planned a run-time compiled build, got the pre-built one
            return fail(IKGPU_ERR_UNSUPPORTED, "run-time compilation failed, the problem runs on " + got + ": " + ikgpu::rtc_last_log());
        return IKGPU_OK;
    } catch (const std::exception &e) {
        return fail(IKGPU_ERR_INVALID, e.what());
    }
}

void ikgpu_problem_destroy(ikgpu_problem *p) {
    if (!p) return;
    DeviceGuard g(p->device);
    (void)hipFree(p->dev.lower);
    (void)hipFree(p->dev.upper);
    (void)hipFree(p->dev.q_in_chain);
    (void)hipFree(p->dev.chain_desc);
    (void)hipFree(p->dev.g_ints);
    (void)hipFree(p->dev.g_dbls);
    p->dev.queues.release();
    if (p->pipe.dev) (void)hipFree(p->pipe.dev);
    for (hipEvent_t ev : p->pipe.ev_in) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : p->pipe.ev_run) (void)hipEventDestroy(ev);
    if (p->pipe.in) { (void)hipStreamDestroy(p->pipe.in); (void)hipStreamDestroy(p->pipe.out); }
    for (hipStream_t r : p->pipe.run) if (r) (void)hipStreamDestroy(r);
    if (p->stage.dev) (void)hipFree(p->stage.dev);
    if (p->stage.host) (void)hipHostFree(p->stage.host);
    delete p;
}

int32_t ikgpu_problem_rows(const ikgpu_problem *p) { return p ? p->host.rows : -1; }

const char *ikgpu_problem_kernel(const ikgpu_problem *p) { return p ? p->dls_name.c_str() : ""; }

int ikgpu_problem_support(const ikgpu_problem *p, uint8_t *support) {
    if (!p || !support) return fail(IKGPU_ERR_INVALID, "ikgpu_problem_support: null argument");
    for (int i = 0; i < p->host.nq; ++i) support[i] = p->host.q_in_chain[static_cast<size_t>(i)] ? 1 : 0;
    return IKGPU_OK;
}

int ikgpu_dls_solve_batch(const ikgpu_problem *p, int64_t B, const double *q0, const double *targets,
                          const ikgpu_dls_params *params, double *q_out, uint8_t *success, int32_t *iters, int layout,
                          void *stream) {
    if (!p) return fail(IKGPU_ERR_INVALID, "null problem");
    if (B < 0) return fail(IKGPU_ERR_INVALID, "negative batch size");
    if (layout != IKGPU_SOA && layout != IKGPU_AOS) return fail(IKGPU_ERR_INVALID, "unknown layout");
    if (int rc = check_params(params)) return rc;
    if (B == 0) return IKGPU_OK;  // an empty batch is a no-op (its pointers may be null)
    if (!q0 || !targets || !q_out) return fail(IKGPU_ERR_INVALID, "null argument");
    if (B > (int64_t(1) << 31) * 32) return fail(IKGPU_ERR_INVALID, "batch too large for one launch");
    return guarded([&] {
        DeviceGuard g(p->device);
        if (!g.ok) return fail(IKGPU_ERR_DEVICE, "hipSetDevice failed");
        ikgpu::BatchIO io{B, q0, targets, q_out, success, iters, layout};
        const hipStream_t st = static_cast<hipStream_t>(stream);
        if (ikgpu::visitor_extended(*params)) {
            // a derived visitor (step tolerance / per-level tolerances): the generic lane program implements the family -- the one
            // specialised for this problem when there is one, else its memory-resident per-lane form
            if (p->gen.generic_build != 2)
                std::call_once(p->visitor_static_once, [&] {
                    p->visitor_static = ikgpu::rtc_generic_static_available(p->gen, /*compile=*/true, &p->visitor_static_key);
                });
            const bool on_static = p->gen.generic_build == 2 || p->visitor_static;
            const uint64_t key = p->gen.generic_build == 2 ? p->gen.generic_key : p->visitor_static_key;
            const hipError_t ev = on_static ? ikgpu::rtc_launch_generic_static(p->gen, key, io, *params, st, &p->dev.queues)
                                            : ikgpu::launch_dls_generic(p->gen, p->dev, io, *params, st, /*force_lane=*/true);
            if (ev != hipSuccess) return hip_fail(ev, "launching the generic DLS kernel (derived visitor)");
            return static_cast<int>(IKGPU_OK);
        }
        hipError_t e = p->dls_on_static_gen                      ? ikgpu::rtc_launch_generic_static(p->gen, p->gen.generic_key, io, *params, st, &p->dev.queues)
                       : p->host.kind == ikgpu::KernelKind::Chain  ? ikgpu::launch_dls_chain(p->host, p->dev, io, *params, st)
                       : p->host.kind == ikgpu::KernelKind::Tree ? ikgpu::launch_dls_tree(p->host, p->dev, io, *params, st)
                       : p->host.generic_build == 2              ? ikgpu::rtc_launch_generic_static(p->host, p->host.generic_key, io, *params, st, &p->dev.queues)
                                                                 : ikgpu::launch_dls_generic(p->host, p->dev, io, *params, st);
        if (e != hipSuccess) return hip_fail(e, "launching the DLS kernel");
        return static_cast<int>(IKGPU_OK);
    });
}

int ikgpu_dls_solve_batch_host(const ikgpu_problem *p, int64_t B, const double *q0, const double *targets,
                               const ikgpu_dls_params *params, double *q_out, uint8_t *success, int32_t *iters,
                               int layout) {
    if (!p) return fail(IKGPU_ERR_INVALID, "null problem");
    if (B < 0) return fail(IKGPU_ERR_INVALID, "negative batch size");
    if (int rc = check_params(params)) return rc;
    if (B == 0) return IKGPU_OK;  // an empty batch is a no-op (its pointers may be null)
    if (!q0 || !targets || !q_out) return fail(IKGPU_ERR_INVALID, "null argument");
    const int lay = layout & ~IKGPU_TARGETS_POSE7;
    if (lay != IKGPU_SOA && lay != IKGPU_AOS) return fail(IKGPU_ERR_INVALID, "unknown layout");
    return host_solve(p, B, q0, targets, q_out, success, iters, layout, [&](int64_t Bk, const double *d_q0, const double *d_t, double *d_q, uint8_t *d_s, int32_t *d_i, hipStream_t st) {
        return ikgpu_dls_solve_batch(p, Bk, d_q0, d_t, params, d_q, d_s, d_i, lay, st);
    });
}

void ikgpu_pik_params_default(ikgpu_pik_params *p, int32_t num_levels) {
    if (!p) return;
    p->max_iterations = 100;  // reference ik/ik/pik.hpp:12
    p->step_length = 1.0;     // reference ik/ik/pik.hpp:14
    p->stop_sq_tol = 1e-4;    // reference ik/ik/visitor.hpp:19
    p->num_levels = num_levels;
    for (double &l : p->lambda) l = 1.0;  // reference ik/ik/pik.hpp:24
    p->da = nullptr;                      // reference ik/ik/pik.hpp:26
}

const char *ikgpu_pik_kernel(const ikgpu_problem *p, const ikgpu_pik_params *params) {
    if (!p || !params) return "";
    if (pik_is_one_dls_level(p, params)) return p->dls_name.c_str();
    if (pik_is_two_levels_on_the_tree(p, params)) return p->pik_tree_name.c_str();
    if (pik_runs_static(p, params)) {
        if (p->pik_static_name.empty()) p->pik_static_name = p->pik_name.substr(0, p->pik_name.size() - 1) + ",static>";
        return p->pik_static_name.c_str();
    }
    return p->pik_name.c_str();
}

int ikgpu_pik_solve_batch(const ikgpu_problem *p, int64_t B, const double *q0, const double *targets,
                          const ikgpu_pik_params *params, double *q_out, uint8_t *success, int32_t *iters, int layout,
                          void *stream) {
    if (!p) return fail(IKGPU_ERR_INVALID, "null problem");
    if (B < 0) return fail(IKGPU_ERR_INVALID, "negative batch size");
    if (layout != IKGPU_SOA && layout != IKGPU_AOS) return fail(IKGPU_ERR_INVALID, "unknown layout");
    if (int rc = check_pik_params(p, params)) return rc;
    if (B == 0) return IKGPU_OK;
    if (!q0 || !targets || !q_out) return fail(IKGPU_ERR_INVALID, "null argument");
    if (B > (int64_t(1) << 31) * 32) return fail(IKGPU_ERR_INVALID, "batch too large for one launch");
    if (pik_is_one_dls_level(p, params)) {
        ikgpu_dls_params d{};   // (the derived-visitor members stay off)
        d.max_iterations = params->max_iterations; d.damping = params->lambda[0]; d.step_length = params->step_length; d.stop_sq_tol = params->stop_sq_tol;
        return ikgpu_dls_solve_batch(p, B, q0, targets, &d, q_out, success, iters, layout, stream);
    }
    if (pik_is_two_levels_on_the_tree(p, params)) {
        return guarded([&] {
            DeviceGuard g(p->device);
            if (!g.ok) return fail(IKGPU_ERR_DEVICE, "hipSetDevice failed");
            ikgpu::BatchIO io{B, q0, targets, q_out, success, iters, layout};
            ikgpu_dls_params d{};
            d.max_iterations = params->max_iterations; d.damping = params->lambda[0]; d.step_length = params->step_length; d.stop_sq_tol = params->stop_sq_tol;
            const hipError_t e = ikgpu::launch_dls_tree(p->host, p->dev, io, d, static_cast<hipStream_t>(stream), &params->lambda[1]);
            if (e != hipSuccess) return hip_fail(e, "launching the tree kernel (two-level ik::pik)");
            return static_cast<int>(IKGPU_OK);
        });
    }
    const bool on_static = pik_runs_static(p, params);
    return guarded([&] {
        DeviceGuard g(p->device);
        if (!g.ok) return fail(IKGPU_ERR_DEVICE, "hipSetDevice failed");
        ikgpu::BatchIO io{B, q0, targets, q_out, success, iters, layout};
        bool has_da = false;
        if (params->da)
            for (int k = 0; k < p->gen.nv; ++k) has_da = has_da || params->da[k] != 0.0;
        const hipError_t e = on_static ? ikgpu::rtc_launch_pik_static(p->gen, p->pik_static_key[has_da ? 1 : 0], io, *params, static_cast<hipStream_t>(stream))
                                       : ikgpu::launch_pik_generic(p->gen, p->dev, io, *params, static_cast<hipStream_t>(stream));
        if (e != hipSuccess) return hip_fail(e, "launching the PIK kernel");
        return static_cast<int>(IKGPU_OK);
    });
}

int ikgpu_pik_solve_batch_host(const ikgpu_problem *p, int64_t B, const double *q0, const double *targets,
                               const ikgpu_pik_params *params, double *q_out, uint8_t *success, int32_t *iters,
                               int layout) {
    if (!p) return fail(IKGPU_ERR_INVALID, "null problem");
    if (B < 0) return fail(IKGPU_ERR_INVALID, "negative batch size");
    if (int rc = check_pik_params(p, params)) return rc;
    if (B == 0) return IKGPU_OK;
    if (!q0 || !targets || !q_out) return fail(IKGPU_ERR_INVALID, "null argument");
    const int lay = layout & ~IKGPU_TARGETS_POSE7;
    if (lay != IKGPU_SOA && lay != IKGPU_AOS) return fail(IKGPU_ERR_INVALID, "unknown layout");
    return host_solve(p, B, q0, targets, q_out, success, iters, layout, [&](int64_t Bk, const double *d_q0, const double *d_t, double *d_q, uint8_t *d_s, int32_t *d_i, hipStream_t st) {
        return ikgpu_pik_solve_batch(p, Bk, d_q0, d_t, params, d_q, d_s, d_i, lay, st);
    });
}

int ikgpu_targets_from_pose7(int64_t B, int32_t ntasks, const double *pose7, double *targets12, int layout, void *stream) {
    if (!pose7 || !targets12) return fail(IKGPU_ERR_INVALID, "null argument");
    if (B < 0 || ntasks < 0) return fail(IKGPU_ERR_INVALID, "negative size");
    if (layout != IKGPU_SOA && layout != IKGPU_AOS) return fail(IKGPU_ERR_INVALID, "unknown layout");
    if (B == 0 || ntasks == 0) return IKGPU_OK;
    const hipError_t e = ikgpu::launch_targets_from_pose7(B, ntasks, pose7, targets12, layout, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail(e, "launching the target expansion kernel");
    return IKGPU_OK;
}

int ikgpu_evaluate_batch(const ikgpu_problem *p, int64_t B, const double *q, const double *targets, double *e_out,
                         double *J_out, int layout, void *stream) {
    if (!p || !q || !targets || !e_out) return fail(IKGPU_ERR_INVALID, "null argument");
    if (B < 0) return fail(IKGPU_ERR_INVALID, "negative batch size");
    if (layout != IKGPU_SOA && layout != IKGPU_AOS) return fail(IKGPU_ERR_INVALID, "unknown layout");
    if (B == 0) return IKGPU_OK;
    return guarded([&] {
        DeviceGuard g(p->device);
        if (!g.ok) return fail(IKGPU_ERR_DEVICE, "hipSetDevice failed");
        const hipStream_t st = static_cast<hipStream_t>(stream);
        hipError_t e = p->host.kind == ikgpu::KernelKind::Chain  ? ikgpu::launch_eval_chain(p->host, p->dev, B, q, targets, e_out, J_out, layout, st)
                       // (a tree problem with the demo's extras -- base-relative reference, alignment row -- has its stages evaluated
                       // by the generic program: the tree stage kernel does not know them)
                       : p->host.kind == ikgpu::KernelKind::Tree && !p->host.tree_extras()
                           ? ikgpu::launch_eval_tree(p->host, p->dev, B, q, targets, e_out, J_out, nullptr, layout, st)
                           : ikgpu::launch_eval_generic(p->gen, p->dev, B, q, targets, e_out, J_out, nullptr, layout, st);
        if (e != hipSuccess) return hip_fail(e, "launching the evaluate kernel");
        return static_cast<int>(IKGPU_OK);
    });
}

int ikgpu_task_frames_fk_batch(const ikgpu_problem *p, int64_t B, const double *q, double *oMf_out, int layout,
                               void *stream) {
    if (!p || !q || !oMf_out) return fail(IKGPU_ERR_INVALID, "null argument");
    if (B < 0) return fail(IKGPU_ERR_INVALID, "negative batch size");
    if (layout != IKGPU_SOA && layout != IKGPU_AOS) return fail(IKGPU_ERR_INVALID, "unknown layout");
    if (B == 0) return IKGPU_OK;
    return guarded([&] {
        DeviceGuard g(p->device);
        if (!g.ok) return fail(IKGPU_ERR_DEVICE, "hipSetDevice failed");
        const hipStream_t st = static_cast<hipStream_t>(stream);
        hipError_t e = p->host.kind == ikgpu::KernelKind::Chain  ? ikgpu::launch_fk_chain(p->host, p->dev, B, q, oMf_out, layout, st)
                       : p->host.kind == ikgpu::KernelKind::Tree && !p->host.tree_extras()
                           ? ikgpu::launch_eval_tree(p->host, p->dev, B, q, q, nullptr, nullptr, oMf_out, layout, st)
                           : ikgpu::launch_eval_generic(p->gen, p->dev, B, q, q, nullptr, nullptr, oMf_out, layout, st);
        if (e != hipSuccess) return hip_fail(e, "launching the FK kernel");
        return static_cast<int>(IKGPU_OK);
    });
}

}  // extern "C"
