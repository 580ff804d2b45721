// problem.cpp -- analyse the support of each task and build the kernel tables.
#include "problem.hpp"

#include <cmath>
#include <cstring>
#include <stdexcept>

namespace ikgpu {
namespace {

// Rotation P with P e_z = a (a unit).  Axis-aligned cases are exact permutations.
void axis_frame(const std::array<double, 3> &a, double *P) {
    auto set = [&](std::initializer_list<double> v) { int i = 0; for (double x : v) P[i++] = x; };
    if (a == std::array<double, 3>{0, 0, 1}) { set({1, 0, 0, 0, 1, 0, 0, 0, 1}); return; }
    if (a == std::array<double, 3>{1, 0, 0}) { set({0, 0, 1, 0, 1, 0, -1, 0, 0}); return; }  // Ry(+90deg)
    if (a == std::array<double, 3>{0, 1, 0}) { set({1, 0, 0, 0, 0, 1, 0, -1, 0}); return; }  // Rx(-90deg)
    // general: b1 orthogonal to a, b2 = a x b1, columns [b1 b2 a]
    int k = 0;
    if (std::fabs(a[1]) < std::fabs(a[k])) k = 1;
    if (std::fabs(a[2]) < std::fabs(a[k])) k = 2;
    double t[3] = {0, 0, 0};
    t[k] = 1.0;
    double b1[3] = {t[1] * a[2] - t[2] * a[1], t[2] * a[0] - t[0] * a[2], t[0] * a[1] - t[1] * a[0]};
    const double n = std::sqrt(b1[0] * b1[0] + b1[1] * b1[1] + b1[2] * b1[2]);
    for (double &x : b1) x /= n;
    const double b2[3] = {a[1] * b1[2] - a[2] * b1[1], a[2] * b1[0] - a[0] * b1[2], a[0] * b1[1] - a[1] * b1[0]};
    for (int i = 0; i < 3; ++i) { P[3 * i] = b1[i]; P[3 * i + 1] = b2[i]; P[3 * i + 2] = a[i]; }
}

SE3 rot_only(const double *P, bool transpose) {
    SE3 s = se3_identity();
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) s[3 * i + j] = transpose ? P[3 * j + i] : P[3 * i + j];
    return s;
}

int task_dim(const ikgpu_task &t) { return t.type == IKGPU_FULL ? 6 : 3; }

}  // namespace

ProblemHost analyse_problem(const Model &m, const ikgpu_task *tasks, int ntasks) {
    if (ntasks < 1) throw std::runtime_error("a problem needs at least one task");
    ProblemHost ph;
    ph.nq = m.nq;
    ph.nv = m.nv;
    ph.ntasks = ntasks;
    ph.lower = m.lower;
    ph.upper = m.upper;
    ph.q_in_chain.assign(m.nq, 0);
    for (int i = 0; i < ntasks; ++i) {
        const ikgpu_task &t = tasks[i];
        if (t.frame < 0 || t.frame >= m.nframes()) throw std::runtime_error("task " + std::to_string(i) + ": frame id out of range");
        if (t.reference < 0 || t.reference >= m.nframes()) throw std::runtime_error("task " + std::to_string(i) + ": reference frame id out of range");
        if (t.type != IKGPU_POSITION && t.type != IKGPU_ORIENTATION && t.type != IKGPU_FULL)
            throw std::runtime_error("task " + std::to_string(i) + ": unknown kinematic type");
        if (t.priority < 0) throw std::runtime_error("task " + std::to_string(i) + ": negative priority");
        ph.tasks.push_back(t);
        ph.rows += task_dim(t);
    }

    if (ntasks != 1)
        throw std::runtime_error("unsupported on the device yet: more than one task (multi-task kernels are not built in this round)");

    const ikgpu_task &t = tasks[0];
    if (m.frame_parent[t.reference] != 0)
        throw std::runtime_error("unsupported on the device yet: reference frame '" + m.frame_names[t.reference] +
                                 "' moves with the configuration (only world-fixed reference frames)");
    std::memcpy(ph.ref_pl, m.frame_placement[t.reference].data(), sizeof(double) * 12);

    // support chain, root first
    std::vector<int> chain;
    for (int j = m.frame_parent[t.frame]; j > 0; j = m.joint_parent[j]) chain.insert(chain.begin(), j);
    if (chain.empty()) throw std::runtime_error("task frame '" + m.frame_names[t.frame] + "' is fixed in the world: nothing to solve");
    if (static_cast<int>(chain.size()) > kMaxChain)
        throw std::runtime_error("unsupported on the device yet: support chain longer than " + std::to_string(kMaxChain) + " joints");
    for (int j : chain)
        if (m.joint_type[j] != IKGPU_JOINT_REVOLUTE)
            throw std::runtime_error("unsupported on the device yet: joint '" + m.joint_names[j] + "' in the task support is not revolute");

    ChainHost &c = ph.chain;
    c.nj = static_cast<int>(chain.size());
    double Pprev[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int k = 0; k < c.nj; ++k) {
        const int j = chain[k];
        double P[9];
        axis_frame(m.joint_axis[j], P);
        SE3 pl = se3_mul(se3_mul(rot_only(Pprev, true), m.joint_placement[j]), rot_only(P, false));
        std::memcpy(c.pl[k], pl.data(), sizeof(double) * 12);
        c.qidx[k] = m.joint_idx_q[j];
        c.vidx[k] = m.joint_idx_v[j];
        c.lo[k] = m.lower[m.joint_idx_q[j]];
        c.hi[k] = m.upper[m.joint_idx_q[j]];
        ph.q_in_chain[m.joint_idx_q[j]] = 1;
        std::memcpy(Pprev, P, sizeof P);
    }
    SE3 fpl = se3_mul(rot_only(Pprev, true), m.frame_placement[t.frame]);
    std::memcpy(c.frame_pl, fpl.data(), sizeof(double) * 12);

    ph.kind = KernelKind::Chain;
    static const char *kt[] = {"position", "orientation", "full"};
    ph.kernel_name = "dls_chain<NJ=" + std::to_string(c.nj) + "," + kt[t.type] + ">";
    return ph;
}

std::vector<double> chain_desc_table(const ProblemHost &ph) {
    const ChainHost &c = ph.chain;
    std::vector<double> t;
    for (int j = 0; j < c.nj; ++j) t.insert(t.end(), c.pl[j], c.pl[j] + 12);
    t.insert(t.end(), c.frame_pl, c.frame_pl + 12);
    t.insert(t.end(), c.lo, c.lo + c.nj);
    t.insert(t.end(), c.hi, c.hi + c.nj);
    t.insert(t.end(), ph.tasks[0].weight, ph.tasks[0].weight + 6);
    return t;
}

void fill_chain_args(const ProblemHost &ph, double *ref_pl12, int *qidx, int *vidx, int *nq, int *nv, int *priority) {
    const ChainHost &c = ph.chain;
    for (int j = 0; j < c.nj; ++j) {
        qidx[j] = c.qidx[j];
        vidx[j] = c.vidx[j];
    }
    std::memcpy(ref_pl12, ph.ref_pl, sizeof(double) * 12);
    *nq = ph.nq;
    *nv = ph.nv;
    *priority = ph.tasks[0].priority;
}

}  // namespace ikgpu
