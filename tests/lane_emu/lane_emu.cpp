// lane_emu.cpp -- TEST HARNESS ONLY.  Runs the exact per-lane programs of the gfx950 kernels
// (ik_amd/csrc/device/*_kernel_body.hpp) on the CPU, one "lane" after another, so that the lane
// programs and the host-side problem analysis can be checked against the oracle in the GPU-less
// build container.  It is compiled by tests/ with g++ into its own shared object; libikgpu.so
// neither contains nor calls it (the product has no CPU path).
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "device/chain_kernel_body.hpp"
#include "device/chain_hot.hpp"
#include "device/tree_kernel_body.hpp"
#include "device/pik_solver.hpp"
#include "generic_tables.hpp"
#include "ikgpu.h"
#include "model.hpp"
#include "problem.hpp"

namespace {

thread_local std::string g_err;

struct IO {
    int mode;  // 0: dls, 1: evaluate (e, J), 2: task-frame FK
    int64_t B;
    const double *q0, *targets;
    const ikgpu_dls_params *prm;
    double *q_out;
    uint8_t *success;
    int32_t *iters;
    double *e_out, *J_out, *oMf_out;
    int layout;
};

template <int NJ, int KT>
void run_chain(const ikgpu::ProblemHost &ph, const IO &io) {
    ikdev::ChainKernelArgs<NJ> a{};
    ikdev::ChainDesc<NJ> d{};
    const std::vector<double> t = ikgpu::chain_desc_table(ph);
    if (t.size() * sizeof(double) != sizeof d) throw std::runtime_error("chain desc table size mismatch");
    std::memcpy(&d, t.data(), sizeof d);
    ikgpu::fill_chain_args(ph, a.ref_pl, a.qidx, a.vidx, &a.nq, &a.nv, &a.prm.priority, &a.prm.idmask, &a.prm.unit_weights);
    a.lower = ph.lower.data(); a.upper = ph.upper.data(); a.q_in_chain = ph.q_in_chain.data();
    a.layout = io.layout; a.B = io.B; a.q0 = io.q0; a.targets = io.targets;
    a.q_out = io.q_out; a.success = io.success; a.iters = io.iters;
    a.e_out = io.e_out; a.J_out = io.J_out; a.oMf_out = io.oMf_out;
    if (io.prm) {
        a.prm.max_iterations = io.prm->max_iterations;
        a.prm.lam2 = io.prm->damping * io.prm->damping;
        a.prm.step_length = io.prm->step_length;
        a.prm.stop_sq_tol = io.prm->stop_sq_tol;
    }
    // LANE_EMU_TRIG set: run the device's general build (SMASK = 0: compile-time "skip nothing", sin / cos by dsincos_fast);
    // unset: the runtime-parameter build (SMASK = -1), dsincos throughout
    const char *tr = std::getenv("LANE_EMU_TRIG");
    for (int64_t b = 0; b < io.B; ++b) {
        if (io.mode == 0 && tr) ikdev::dls_chain_body<NJ, KT, 0>(a, d, b, [](bool act) { return act; });
        else if (io.mode == 0) ikdev::dls_chain_body<NJ, KT>(a, d, b, [](bool act) { return act; });
        else if (io.mode == 1) ikdev::eval_chain_body<NJ, KT>(a, d, b);
        else ikdev::fk_chain_body<NJ>(a, d, b);
    }
}

// The structure-specialised chain program (device/chain_hot.hpp) with the structure codes kernels_hot.hip instantiates.
template <int NJ, uint64_t C0, uint64_t C1, uint64_t C2>
void run_chain_hot(const ikgpu::ProblemHost &ph, const IO &io) {
    ikdev::ChainKernelArgs<NJ> a{};
    ikgpu::fill_chain_args(ph, a.ref_pl, a.qidx, a.vidx, &a.nq, &a.nv, &a.prm.priority, &a.prm.idmask, &a.prm.unit_weights);
    a.lower = ph.lower.data(); a.upper = ph.upper.data(); a.q_in_chain = ph.q_in_chain.data();
    a.layout = io.layout; a.B = io.B; a.q0 = io.q0; a.targets = io.targets;
    a.q_out = io.q_out; a.success = io.success; a.iters = io.iters;
    a.prm.max_iterations = io.prm->max_iterations;
    a.prm.lam2 = io.prm->damping * io.prm->damping;
    a.prm.step_length = io.prm->step_length;
    a.prm.stop_sq_tol = io.prm->stop_sq_tol;
    ikdev::HotTable t{};
    const std::vector<double> tab = ikgpu::chain_hot_table(ph.chain);
    if (tab.size() > static_cast<size_t>(ikdev::kHotTableMax)) throw std::runtime_error("compact table too long");
    std::memcpy(t.v, tab.data(), tab.size() * sizeof(double));
    typedef ikdev::ChainStruct<C0, C1, C2> S;
    for (int64_t b = 0; b < io.B; ++b) {
        if (io.prm->stop_sq_tol < 0.0) ikdev::hot_chain_body<NJ, S, true>(a, t, b, [](bool act) { return act; });
        else ikdev::hot_chain_body<NJ, S, false>(a, t, b, [](bool act) { return act; });
    }
}

// true when the problem was run by the hot program (LANE_EMU_HOT set, a Full task with unit weights, a known structure code)
bool try_chain_hot(const ikgpu::ProblemHost &ph, const IO &io) {
    if (!std::getenv("LANE_EMU_HOT") || io.mode != 0 || ph.tasks[0].type != IKGPU_FULL || !ikgpu::task_has_unit_weights(ph.tasks[0])) return false;
    const ikgpu::ChainStructure s = ikgpu::chain_structure(ph.chain);
    if (!s.fits) return false;
#define X(N, K0, K1, K2)                                                                      \
    if (ph.chain.nj == N && s.code[0] == K0 && s.code[1] == K1 && s.code[2] == K2) {         \
        run_chain_hot<N, K0, K1, K2>(ph, io);                                                 \
        return true;                                                                          \
    }
    X(7, 0x04f0208cce8c7664ull, 0x395959cacad65656ull, 0x000001cacace5656ull)
    X(6, 0x695959272b925656ull, 0x47655a33aaca549cull, 0x0000000000121256ull)
#undef X
    return false;
}

template <int NJ>
struct HostPark {  // the device parks chain 0's factor in LDS; the host keeps a copy
    mutable ikdev::LegFactor<NJ> saved;
    void store(const ikdev::LegFactor<NJ> &F) const { saved = F; }
    void load(ikdev::LegFactor<NJ> &F) const {
        for (int e = 0; e < NJ * (NJ + 1) / 2; ++e) F.L[e] = saved.L[e];
        for (int j = 0; j < NJ; ++j) {
            for (int c = 0; c < 6; ++c) F.W[j][c] = saved.W[j][c];
            F.u[j] = saved.u[j];
        }
    }
};

template <int NJ, int NCH>
void run_tree(const ikgpu::ProblemHost &ph, const IO &io) {
    ikdev::TreeKernelArgs<NJ, NCH> a{};
    ikdev::TreeDesc<NJ, NCH> d{};
    const std::vector<double> t = ikgpu::tree_desc_table(ph);
    if (t.size() * sizeof(double) != sizeof d) throw std::runtime_error("tree desc table size mismatch");
    std::memcpy(&d, t.data(), sizeof d);
    const ikgpu::TreeArgsHost h = ikgpu::tree_args(ph);
    for (int c = 0; c < NCH; ++c)
        for (int j = 0; j < NJ; ++j) { a.qidx[c][j] = h.qidx[c][j]; a.vidx[c][j] = h.vidx[c][j]; }
    for (int s = 0; s < 3; ++s) { a.tslot[s] = h.tslot[s]; a.trow[s] = h.trow[s]; a.tdim[s] = h.tdim[s]; a.trow0[s] = h.trow0[s]; }
    a.prm.prio[0] = h.prio[0]; a.prm.prio[1] = h.prio[1]; a.prm.prioP = h.prio[2]; a.prm.hasP = h.hasP;
    a.prm.idmask[0] = h.idmask[0]; a.prm.idmask[1] = h.idmask[1]; a.prm.idmaskP = h.idmaskP;
    a.prm.unit[0] = h.unit[0]; a.prm.unit[1] = h.unit[1]; a.prm.unitP = h.unit[2];
    a.prm.ref_base[0] = h.ref_base[0]; a.prm.ref_base[1] = h.ref_base[1];
    a.prm.align_chain = h.align_chain; a.prm.align_axis = h.align_axis; a.prm.align_slot = h.align_slot;
    a.prm.align_prio = h.align_prio; a.prm.align_w = h.align_w;
    a.prm.fixed_base = h.fixed_base;
    a.prm.cons_on = h.cons_on; a.prm.cons_type = h.cons_type;
    a.prm.post_on = h.post_on; a.prm.post_prio = h.post_prio; a.prm.post_n = h.post_n;
    for (int k = 0; k < h.post_n; ++k) {
        a.prm.post_q[k] = h.post_q[k]; a.prm.post_slot[k] = h.post_slot[k]; a.prm.post_w[k] = h.post_w[k]; a.prm.post_m[k] = h.post_m[k];
    }
    for (int c = 0; c < 2; ++c)
        for (int j = 0; j < 8; ++j) { a.prm.postc_slot[c][j] = h.postc_slot[c][j]; a.prm.postc_w[c][j] = h.postc_w[c][j]; a.prm.postc_m[c][j] = h.postc_m[c][j]; }
    a.nq = ph.nq; a.nv = ph.nv; a.ntasks = ph.ntasks;
    a.lower = ph.lower.data(); a.upper = ph.upper.data(); a.q_in_chain = ph.q_in_chain.data();
    a.layout = io.layout; a.B = io.B; a.q0 = io.q0; a.targets = io.targets;
    a.q_out = io.q_out; a.success = io.success; a.iters = io.iters;
    if (io.mode == 1) { a.e_out = io.e_out; a.J_out = io.J_out; }
    if (io.mode == 2) { a.oMf_out = io.oMf_out; a.targets = io.q0; }
    if (io.prm) {
        a.prm.max_iterations = io.prm->max_iterations;
        a.prm.lam2 = io.prm->damping * io.prm->damping;
        a.prm.step_length = io.prm->step_length;
        a.prm.stop_sq_tol = io.prm->stop_sq_tol;
    }
    // LANE_EMU_TREE_PIK_LAMBDA1=<lambda_1>: ik::pik with two levels on the tree program (level 1 = the AlignAxisTask row, damping
    // factor lambda_1; lambda_0 is the DLS damping), see device/tree_solver.hpp PikRow
    if (const char *l1 = std::getenv("LANE_EMU_TREE_PIK_LAMBDA1")) { a.prm.pik_on = 1; a.prm.pik_lam2_1 = std::atof(l1) * std::atof(l1); }
    const char *tr = std::getenv("LANE_EMU_TRIG");  // set: the device's general build (SPEC = 0), which takes sin / cos by dsincos_fast
    for (int64_t b = 0; b < io.B; ++b) {
        if (io.mode == 0 && tr && ph.has_posture)
            ikdev::dls_tree_body<NJ, NCH, (1 << ikdev::kSpecPost)>(a, d, b, HostPark<NJ>{}, [](bool act) { return act; });
        else if (io.mode == 0 && tr) ikdev::dls_tree_body<NJ, NCH, 0>(a, d, b, HostPark<NJ>{}, [](bool act) { return act; });
        else if (io.mode == 0) ikdev::dls_tree_body<NJ, NCH>(a, d, b, HostPark<NJ>{}, [](bool act) { return act; });
        else ikdev::eval_tree_body<NJ, NCH>(a, d, b);
    }
}

void run_generic(const ikgpu::ProblemHost &ph, const IO &io) {
    ikdev::GenericKernelArgs a{};
    a.T = ikgpu::bind_generic_tables(ph, ph.generic.ints.data(), ph.generic.dbls.data());
    a.layout = io.layout; a.B = io.B; a.q0 = io.q0; a.targets = io.targets;
    a.q_out = io.q_out; a.success = io.success; a.iters = io.iters;
    if (io.mode == 1) { a.e_out = io.e_out; a.J_out = io.J_out; }
    if (io.mode == 2) { a.oMf_out = io.oMf_out; a.targets = io.q0; }
    if (io.prm) {
        a.prm.max_iterations = io.prm->max_iterations;
        a.prm.lam2 = io.prm->damping * io.prm->damping;
        a.prm.step_length = io.prm->step_length;
        a.prm.stop_sq_tol = io.prm->stop_sq_tol;
    }
    a.ws_stride = (io.B + 63) / 64 * 64;
    std::vector<double> ws(static_cast<size_t>(ph.generic.ws_words) * a.ws_stride, 0.0);
    a.ws = ws.data();
    for (int64_t b = 0; b < io.B; ++b) {
        if (io.mode == 0) ikdev::dls_generic_body(a, b, [](bool act) { return act; });
        else ikdev::eval_generic_body(a, b);
    }
}

}  // namespace

extern "C" {

const char *lane_emu_last_error(void) { return g_err.c_str(); }

// The device's sin / cos routines (device/lane_math.hpp): D = 0: dsincos, D = 1: dsincos_fast, D = 2 / 3: dsincos_bounded<D>, D = 4: dsincos_hot.
void lane_emu_sincos(int D, int64_t n, const double *x, double *s, double *c) {
    for (int64_t i = 0; i < n; ++i) {
        if (D == 1) ikdev::dsincos_fast(x[i], s[i], c[i]);
        else if (D == 4) ikdev::dsincos_hot(x[i], s[i], c[i]);
        else if (D == 2) ikdev::dsincos_bounded<2>(x[i], s[i], c[i]);
        else if (D == 3) ikdev::dsincos_bounded<3>(x[i], s[i], c[i]);
        else ikdev::dsincos(x[i], s[i], c[i]);
    }
}

// The two front ends of log6(fMt) + Jlog6(tMf) (device/lane_math.hpp): which = 0: log6_and_jlog6_inv (general builds), 1:
// log6_and_jlog6_hot (the headline loop and the tree kernels).  Re [n][9] row-major, pe [n][3]; out [n][24] = e (6), A (9), Bm (9).
void lane_emu_log6(int which, int64_t n, const double *Re, const double *pe, double *out) {
    for (int64_t i = 0; i < n; ++i) {
        double R[9], p[3];
        for (int k = 0; k < 9; ++k) R[k] = Re[9 * i + k];
        for (int k = 0; k < 3; ++k) p[k] = pe[3 * i + k];
        ikdev::LogAndJlog o;
        if (which == 1) ikdev::log6_and_jlog6_hot<true>(R, p, o);
        else ikdev::log6_and_jlog6_inv(R, p, o);
        for (int k = 0; k < 6; ++k) out[24 * i + k] = o.e[k];
        for (int k = 0; k < 9; ++k) { out[24 * i + 6 + k] = o.A[k]; out[24 * i + 15 + k] = o.Bm[k]; }
    }
}

// Host pointers, same layouts as include/ikgpu.h.  tasks must be in stacking order.
int lane_emu_run(const char *urdf, size_t len, int root_joint, const ikgpu_task *tasks, int ntasks, int mode, int64_t B,
                 const double *q0, const double *targets, const ikgpu_dls_params *prm, double *q_out, uint8_t *success,
                 int32_t *iters, double *e_out, double *J_out, double *oMf_out, int layout) {
    try {
        // root_joint bit 1 forces the generic kernel (to test it on shapes the specialisations also take)
        ikgpu::Model m = ikgpu::Model::from_urdf(urdf, len, (root_joint & 1) != 0);
        ikgpu::ProblemHost ph = ikgpu::analyse_problem(m, tasks, ntasks, (root_joint & 2) != 0);
        // stages of a tree problem with the demo's extras are evaluated by the generic program (as capi.cpp does)
        if (ph.kind == ikgpu::KernelKind::Tree && ph.tree_extras() && mode != 0) ph = ikgpu::analyse_problem(m, tasks, ntasks, true);
        const IO io{mode, B, q0, targets, prm, q_out, success, iters, e_out, J_out, oMf_out, layout};
        if (ph.kind == ikgpu::KernelKind::Generic) {
            run_generic(ph, io);
            return 0;
        }
        if (ph.kind == ikgpu::KernelKind::Chain) {
            if (try_chain_hot(ph, io)) return 0;
            const int nj = ph.chain.nj, kt = tasks[0].type;
#define X(N)                                       \
    if (nj == N) {                                 \
        if (kt == 2) run_chain<N, 2>(ph, io);      \
        else if (kt == 0) run_chain<N, 0>(ph, io); \
        else run_chain<N, 1>(ph, io);              \
        return 0;                                  \
    }
            X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8)
#undef X
        } else {
            const int nj = ph.chain.nj, nch = ph.chainB.nj > 0 ? 2 : 1;
            if (nj == 7 && nch == 2) { run_tree<7, 2>(ph, io); return 0; }
            if (nj == 7 && nch == 1) { run_tree<7, 1>(ph, io); return 0; }
            if (nj == 6 && nch == 2) { run_tree<6, 2>(ph, io); return 0; }
            if (nj == 6 && nch == 1) { run_tree<6, 1>(ph, io); return 0; }
            // a tree shape without a compiled kernel runs on the generic program (as capi.cpp: shape_built())
            run_generic(ikgpu::analyse_problem(m, tasks, ntasks, true), io);
            return 0;
        }
        g_err = "shape not instantiated in the lane emulator: " + ph.kernel_name;
        return 1;
    } catch (const std::exception &e) {
        g_err = e.what();
        return 1;
    }
}

// ik::dls through the cooperative program (device/coop_solver.hpp): one host "lane" takes every item of each phase.
// Returns 2 when the problem has no cooperative form (constraints, centre-of-mass task, workspace too large for LDS).
int lane_emu_dls_coop_constrained(const char *urdf, size_t len, int root_joint, const ikgpu_task *tasks, int ntasks, const ikgpu_task *cons,
                                  int ncons, int64_t B, const double *q0, const double *targets, const ikgpu_dls_params *prm, double *q_out,
                                  uint8_t *success, int32_t *iters, int layout);

int lane_emu_dls_coop(const char *urdf, size_t len, int root_joint, const ikgpu_task *tasks, int ntasks, int64_t B, const double *q0,
                      const double *targets, const ikgpu_dls_params *prm, double *q_out, uint8_t *success, int32_t *iters, int layout) {
    return lane_emu_dls_coop_constrained(urdf, len, root_joint, tasks, ntasks, nullptr, 0, B, q0, targets, prm, q_out, success, iters, layout);
}

// ... with ik::FrameConstraint entries (the step projected into the null space of their Jacobian)
int lane_emu_dls_coop_constrained(const char *urdf, size_t len, int root_joint, const ikgpu_task *tasks, int ntasks, const ikgpu_task *cons,
                                  int ncons, int64_t B, const double *q0, const double *targets, const ikgpu_dls_params *prm, double *q_out,
                                  uint8_t *success, int32_t *iters, int layout) {
    try {
        ikgpu::Model m = ikgpu::Model::from_urdf(urdf, len, (root_joint & 1) != 0);
        const ikgpu::ProblemHost ph = ikgpu::analyse_problem(m, tasks, ntasks, /*force_generic=*/true, cons, ncons);
        if (!ph.generic.coop_ok) { g_err = "no cooperative form for this problem"; return 2; }
        ikdev::CoopKernelArgs a{};
        a.T = ikgpu::bind_generic_tables(ph, ph.generic.ints.data(), ph.generic.dbls.data());
        a.L = ikgpu::bind_coop_layout(ph, ph.generic.ints.data());
        a.prm.max_iterations = prm->max_iterations;
        a.prm.lam2 = prm->damping * prm->damping;
        a.prm.step_length = prm->step_length;
        a.prm.stop_sq_tol = prm->stop_sq_tol;
        a.layout = layout; a.B = B; a.q0 = q0; a.targets = targets;
        a.q_out = q_out; a.success = success; a.iters = iters;
        std::vector<double> ws(static_cast<size_t>(a.L.words), 0.0);
        for (int64_t b = 0; b < B; ++b) ikdev::dls_coop_body(a, b, 0, ws.data(), [](bool act) { return act; });
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return 1;
    }
}

// ik::pik through the cooperative program (device/pik_coop.hpp).  Returns 2 when the problem has no cooperative form.
int lane_emu_pik_coop(const char *urdf, size_t len, int root_joint, const ikgpu_task *tasks, int ntasks, int64_t B, const double *q0,
                      const double *targets, const ikgpu_pik_params *prm, double *q_out, uint8_t *success, int32_t *iters, int layout) {
    try {
        ikgpu::Model m = ikgpu::Model::from_urdf(urdf, len, (root_joint & 1) != 0);
        const ikgpu::ProblemHost ph = ikgpu::analyse_problem(m, tasks, ntasks, /*force_generic=*/true);
        if (prm->num_levels != ph.generic.nlevels) { g_err = "num_levels does not match the task table"; return 1; }
        if (!ph.generic.coop_pik_ok) { g_err = "no cooperative form for this problem"; return 2; }
        ikdev::PikCoopKernelArgs a{};
        a.T = ikgpu::bind_generic_tables(ph, ph.generic.ints.data(), ph.generic.dbls.data());
        a.L = ikgpu::bind_coop_layout(ph, ph.generic.ints.data(), /*for_pik=*/true);
        a.K = ikgpu::bind_pik_coop_layout(ph);
        a.prm.max_iterations = prm->max_iterations;
        a.prm.step_length = prm->step_length;
        a.prm.stop_sq_tol = prm->stop_sq_tol;
        for (int l = 0; l < ikdev::kMaxPikLevels; ++l) a.prm.lam2[l] = l < prm->num_levels ? prm->lambda[l] * prm->lambda[l] : 1.0;
        if (prm->da)
            for (int k = 0; k < ph.nv; ++k) {
                a.prm.da[k] = prm->da[k];
                if (prm->da[k] != 0.0) a.prm.has_da = 1;
            }
        a.layout = layout; a.B = B; a.q0 = q0; a.targets = targets;
        a.q_out = q_out; a.success = success; a.iters = iters;
        std::vector<double> ws(static_cast<size_t>(a.K.words), 0.0);
        for (int64_t b = 0; b < B; ++b) ikdev::pik_coop_body(a, a.T, a.L, b, 0, ws.data(), [](bool act) { return act; });
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return 1;
    }
}

// ik::dls with ik::FrameConstraint entries (reference ik/ik/dls.cpp:26-34,43-53) through the generic lane program.
int lane_emu_dls_constrained(const char *urdf, size_t len, int root_joint, const ikgpu_task *tasks, int ntasks, const ikgpu_task *cons,
                             int ncons, int64_t B, const double *q0, const double *targets, const ikgpu_dls_params *prm, double *q_out,
                             uint8_t *success, int32_t *iters, int layout) {
    try {
        ikgpu::Model m = ikgpu::Model::from_urdf(urdf, len, (root_joint & 1) != 0);
        // root_joint bit 2: let the analysis pick the tree kernel's constraint build when the problem has that shape
        const ikgpu::ProblemHost ph = ikgpu::analyse_problem(m, tasks, ntasks, /*force_generic=*/(root_joint & 4) == 0, cons, ncons);
        const IO io{0, B, q0, targets, prm, q_out, success, iters, nullptr, nullptr, nullptr, layout};
        if (ph.kind == ikgpu::KernelKind::Tree) {
            const int nj = ph.chain.nj, nch = ph.chainB.nj > 0 ? 2 : 1;
            if (nj == 7 && nch == 2) { run_tree<7, 2>(ph, io); return 0; }
            if (nj == 6 && nch == 2) { run_tree<6, 2>(ph, io); return 0; }
            g_err = "constraint tree shape not instantiated in the lane emulator: " + ph.kernel_name;
            return 1;
        }
        if ((root_joint & 4) != 0) { g_err = "the problem did not map onto the tree kernel: " + ph.kernel_name; return 1; }
        run_generic(ph, io);
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return 1;
    }
}

// ik::pik (reference ik/ik/pik.cpp:31-103) through the device lane program (device/pik_solver.hpp), one lane at a time.
int lane_emu_pik(const char *urdf, size_t len, int root_joint, const ikgpu_task *tasks, int ntasks, int64_t B, const double *q0,
                 const double *targets, const ikgpu_pik_params *prm, double *q_out, uint8_t *success, int32_t *iters, int layout) {
    try {
        ikgpu::Model m = ikgpu::Model::from_urdf(urdf, len, (root_joint & 1) != 0);
        const ikgpu::ProblemHost ph = ikgpu::analyse_problem(m, tasks, ntasks, /*force_generic=*/true);
        if (prm->num_levels != ph.generic.nlevels) { g_err = "num_levels does not match the task table"; return 1; }
        ikdev::PikKernelArgs a{};
        a.T = ikgpu::bind_generic_tables(ph, ph.generic.ints.data(), ph.generic.dbls.data());
        a.prm.max_iterations = prm->max_iterations;
        a.prm.step_length = prm->step_length;
        a.prm.stop_sq_tol = prm->stop_sq_tol;
        for (int l = 0; l < ikdev::kMaxPikLevels; ++l) a.prm.lam2[l] = l < prm->num_levels ? prm->lambda[l] * prm->lambda[l] : 1.0;
        if (prm->da)
            for (int k = 0; k < ph.nv; ++k) {
                a.prm.da[k] = prm->da[k];
                if (prm->da[k] != 0.0) a.prm.has_da = 1;
            }
        a.layout = layout; a.B = B; a.q0 = q0; a.targets = targets;
        a.q_out = q_out; a.success = success; a.iters = iters;
        a.ws_stride = (B + 63) / 64 * 64;
        std::vector<double> ws(static_cast<size_t>(ph.generic.ws_words_pik) * a.ws_stride, 0.0);
        a.ws = ws.data();
        for (int64_t b = 0; b < B; ++b) ikdev::pik_generic_body(a, b, [](bool act) { return act; });
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return 1;
    }
}

}  // extern "C"
