// fast_cpu.cpp -- BENCH BASELINE ONLY (bench.py's cpu_baseline leg): the *optimised* CPU variant of the ik::dls path that
// SURVEY.md 8(d) asks to be timed next to the faithful port (oracle/ik_oracle.c), "so the speed-up is not flattered".
//
// It is the device's own lane program (ik_amd/csrc/device/chain_kernel_body.hpp: support-sparse chain FK, shared log3 for
// log6 / Jlog6, 21-entry Gram, unrolled Cholesky, no allocation, no whole-tree pass) compiled for the host with
// g++ -O3 -march=x86-64-v3 and run one problem after another on every hardware thread.  Same algorithm as the reference's
// loop (ik/ik/dls.cpp:5-78) with the restructuring a careful CPU implementation would also make.  Chain problems only
// (the headline shape S and the UR arms).  libikgpu.so neither contains nor calls this file; nothing in ik_amd/ loads it.
#include <algorithm>
#include <cstring>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "device/chain_kernel_body.hpp"
#include "ikgpu.h"
#include "model.hpp"
#include "problem.hpp"

namespace {
thread_local std::string g_err;

template <int NJ>
void solve_range(const ikgpu::ProblemHost &ph, const ikgpu_dls_params &prm, int64_t B, const double *q0, const double *targets,
                 double *q_out, uint8_t *success, int32_t *iters, int layout, int64_t lo, int64_t hi) {
    ikdev::ChainKernelArgs<NJ> a{};
    ikdev::ChainDesc<NJ> d{};
    const std::vector<double> t = ikgpu::chain_desc_table(ph);
    if (t.size() * sizeof(double) != sizeof d) throw std::runtime_error("chain desc table size mismatch");
    std::memcpy(&d, t.data(), sizeof d);
    ikgpu::fill_chain_args(ph, a.ref_pl, a.qidx, a.vidx, &a.nq, &a.nv, &a.prm.priority, &a.prm.idmask, &a.prm.unit_weights);
    a.lower = ph.lower.data(); a.upper = ph.upper.data(); a.q_in_chain = ph.q_in_chain.data();
    a.layout = layout; a.B = B; a.q0 = q0; a.targets = targets; a.q_out = q_out; a.success = success; a.iters = iters;
    a.prm.max_iterations = prm.max_iterations;
    a.prm.lam2 = prm.damping * prm.damping;
    a.prm.step_length = prm.step_length;
    a.prm.stop_sq_tol = prm.stop_sq_tol;
    for (int64_t b = lo; b < hi; ++b) ikdev::dls_chain_body<NJ, ikdev::KT_FULL, 0>(a, d, b, [](bool act) { return act; });
}
}  // namespace

extern "C" {

const char *fastcpu_last_error(void) { return g_err.c_str(); }

// Host pointers, layouts as include/ikgpu.h.  One Full FrameTask whose support is a serial chain on a fixed-base model.
int fastcpu_dls_chain(const char *urdf, size_t len, const ikgpu_task *task, int64_t B, const double *q0, const double *targets,
                      const ikgpu_dls_params *prm, double *q_out, uint8_t *success, int32_t *iters, int layout, int nthreads) {
    try {
        const ikgpu::Model m = ikgpu::Model::from_urdf(urdf, len, false);
        const ikgpu::ProblemHost ph = ikgpu::analyse_problem(m, task, 1, false);
        if (ph.kind != ikgpu::KernelKind::Chain || task->type != IKGPU_FULL) throw std::runtime_error("fast CPU variant: chain problems with one Full task only");
        nthreads = std::max(1, nthreads);
        std::vector<std::thread> pool;
        std::vector<std::string> errs(static_cast<size_t>(nthreads));
        for (int t = 0; t < nthreads; ++t) {
            const int64_t lo = B * t / nthreads, hi = B * (t + 1) / nthreads;
            pool.emplace_back([&, t, lo, hi] {
                try {
                    switch (ph.chain.nj) {
#define X(N) case N: solve_range<N>(ph, *prm, B, q0, targets, q_out, success, iters, layout, lo, hi); break;
                        X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8)
#undef X
                        default: throw std::runtime_error("chain length not instantiated");
                    }
                } catch (const std::exception &e) { errs[static_cast<size_t>(t)] = e.what(); }
            });
        }
        for (auto &th : pool) th.join();
        for (const auto &e : errs) if (!e.empty()) throw std::runtime_error(e);
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return 1;
    }
}

}  // extern "C"
