// lane_emu.cpp -- TEST HARNESS ONLY.  Runs the exact per-lane program of the gfx950 chain
// kernels (ik_amd/csrc/device/chain_kernel_body.hpp) on the CPU, one "lane" after another, so
// that the lane program and the host-side problem analysis can be checked against the oracle in
// the GPU-less build container.  It is compiled by tests/ with g++ into its own shared object;
// libikgpu.so neither contains nor calls it (the product has no CPU path).
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "device/chain_kernel_body.hpp"
#include "ikgpu.h"
#include "model.hpp"
#include "problem.hpp"

namespace {

thread_local std::string g_err;

template <int NJ>
struct Ctx {
    ikdev::ChainKernelArgs<NJ> a{};
    ikdev::ChainDesc<NJ> d{};
};

template <int NJ>
Ctx<NJ> make_ctx(const ikgpu::ProblemHost &ph) {
    Ctx<NJ> c;
    const std::vector<double> t = ikgpu::chain_desc_table(ph);
    if (t.size() * sizeof(double) != sizeof(ikdev::ChainDesc<NJ>)) throw std::runtime_error("desc table size mismatch");
    std::memcpy(&c.d, t.data(), sizeof c.d);
    ikgpu::fill_chain_args(ph, c.a.ref_pl, c.a.qidx, c.a.vidx, &c.a.nq, &c.a.nv, &c.a.prm.priority);
    c.a.lower = ph.lower.data();
    c.a.upper = ph.upper.data();
    c.a.q_in_chain = ph.q_in_chain.data();
    return c;
}

template <int NJ, int KT>
void run(const ikgpu::ProblemHost &ph, int mode, int64_t B, const double *q0, const double *targets,
         const ikgpu_dls_params *prm, double *q_out, uint8_t *success, int32_t *iters, double *e_out, double *J_out,
         double *oMf_out, int layout) {
    Ctx<NJ> c = make_ctx<NJ>(ph);
    c.a.layout = layout;
    c.a.B = B;
    c.a.q0 = q0;
    c.a.targets = targets;
    c.a.q_out = q_out;
    c.a.success = success;
    c.a.iters = iters;
    c.a.e_out = e_out;
    c.a.J_out = J_out;
    c.a.oMf_out = oMf_out;
    if (prm) {
        c.a.prm.max_iterations = prm->max_iterations;
        c.a.prm.lam2 = prm->damping * prm->damping;
        c.a.prm.step_length = prm->step_length;
        c.a.prm.stop_sq_tol = prm->stop_sq_tol;
    }
    for (int64_t b = 0; b < B; ++b) {
        if (mode == 0) ikdev::dls_chain_body<NJ, KT>(c.a, c.d, b, [](bool act) { return act; });
        else if (mode == 1) ikdev::eval_chain_body<NJ, KT>(c.a, c.d, b);
        else ikdev::fk_chain_body<NJ>(c.a, c.d, b);
    }
}

}  // namespace

extern "C" {

const char *lane_emu_last_error(void) { return g_err.c_str(); }

// mode 0: dls, 1: evaluate, 2: task-frame FK.  Host pointers, same layouts as include/ikgpu.h.
int lane_emu_run(const char *urdf, size_t len, int root_joint, const ikgpu_task *task, int mode, int64_t B,
                 const double *q0, const double *targets, const ikgpu_dls_params *prm, double *q_out,
                 uint8_t *success, int32_t *iters, double *e_out, double *J_out, double *oMf_out, int layout) {
    try {
        ikgpu::Model m = ikgpu::Model::from_urdf(urdf, len, root_joint != 0);
        ikgpu::ProblemHost ph = ikgpu::analyse_problem(m, task, 1);
        const int nj = ph.chain.nj, kt = task->type;
#define X(N)                                                                                                               \
    if (nj == N) {                                                                                                         \
        if (kt == 2) run<N, 2>(ph, mode, B, q0, targets, prm, q_out, success, iters, e_out, J_out, oMf_out, layout);       \
        else if (kt == 0) run<N, 0>(ph, mode, B, q0, targets, prm, q_out, success, iters, e_out, J_out, oMf_out, layout);  \
        else run<N, 1>(ph, mode, B, q0, targets, prm, q_out, success, iters, e_out, J_out, oMf_out, layout);               \
        return 0;                                                                                                          \
    }
        X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8)
#undef X
        g_err = "chain length not instantiated";
        return 1;
    } catch (const std::exception &e) {
        g_err = e.what();
        return 1;
    }
}

}  // extern "C"
