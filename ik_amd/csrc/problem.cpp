// problem.cpp -- analyse the support of each task and build the kernel tables.
#include "problem.hpp"

#include <algorithm>
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

int task_dim(const ikgpu_task &t) {  // align / posture rows: 1; frame position / orientation and centre of mass: 3
    return t.type == IKGPU_FULL ? 6 : (t.type >= IKGPU_ALIGN_AXIS_X && t.type <= IKGPU_POSTURE_ROW ? 1 : 3);
}

bool is_identity(const SE3 &s) { return s == se3_identity(); }

// Axis-folded table of the revolute joints `joints` (root first) ending in frame `frame`.
// pl[0] is relative to the parent joint of joints[0] (the universe, or the free-flyer base).
ChainHost build_chain(const Model &m, const std::vector<int> &joints, int frame, int task_index,
                      std::vector<uint8_t> &q_in_chain) {
    if (static_cast<int>(joints.size()) > kMaxChain)
        throw std::runtime_error("unsupported on the device yet: support chain longer than " + std::to_string(kMaxChain) + " joints");
    ChainHost c;
    c.nj = static_cast<int>(joints.size());
    c.task = task_index;
    double Pprev[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int k = 0; k < c.nj; ++k) {
        const int j = joints[k];
        if (m.joint_type[j] != IKGPU_JOINT_REVOLUTE)
            throw std::runtime_error("unsupported on the device yet: joint '" + m.joint_names[j] + "' in the task support is not revolute");
        if (q_in_chain[m.joint_idx_q[j]])
            throw std::runtime_error("unsupported on the device yet: joint '" + m.joint_names[j] + "' is in the support of two tasks");
        double P[9];
        axis_frame(m.joint_axis[j], P);
        SE3 pl = se3_mul(se3_mul(rot_only(Pprev, true), m.joint_placement[j]), rot_only(P, false));
        std::memcpy(c.pl[k], pl.data(), sizeof(double) * 12);
        c.qidx[k] = m.joint_idx_q[j];
        c.vidx[k] = m.joint_idx_v[j];
        c.lo[k] = m.lower[m.joint_idx_q[j]];
        c.hi[k] = m.upper[m.joint_idx_q[j]];
        q_in_chain[m.joint_idx_q[j]] = 1;
        std::memcpy(Pprev, P, sizeof P);
    }
    SE3 fpl = se3_mul(rot_only(Pprev, true), m.frame_placement[frame]);
    std::memcpy(c.frame_pl, fpl.data(), sizeof(double) * 12);
    return c;
}

// Six-row weights of a task: rows its kinematic type drops get weight zero.
void weights6(const ikgpu_task &t, double *w6) {
    for (int i = 0; i < 6; ++i) w6[i] = 0.0;
    if (t.type == IKGPU_FULL) for (int i = 0; i < 6; ++i) w6[i] = t.weight[i];
    if (t.type == IKGPU_POSITION) for (int i = 0; i < 3; ++i) w6[i] = t.weight[i];
    if (t.type == IKGPU_ORIENTATION) for (int i = 0; i < 3; ++i) w6[3 + i] = t.weight[i];
}

}  // namespace

namespace {

struct Unsupported : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// Chain / Tree specialisations; throws Unsupported when the problem does not have that shape.
void specialise(ProblemHost &ph, const Model &m) {
    static const char *kt[] = {"position", "orientation", "full"};
    const int ntasks = ph.ntasks;
    const bool free_flyer = m.njoints() > 1 && m.joint_type[1] == IKGPU_JOINT_FREEFLYER;
    for (const ikgpu_task &t : ph.tasks) {
        if (t.type == IKGPU_CENTRE_OF_MASS) throw Unsupported("CentreOfMassTask rows run on the generic kernel");
    }
    // a "continuous" joint anywhere in the model: its (cos, sin) pair is renormalised by every integrate, also where dq = 0
    for (int j = 1; j < m.njoints(); ++j)
        if (m.joint_type[j] == IKGPU_JOINT_REVOLUTE_UNBOUNDED) throw Unsupported("a model with continuous joints runs on the generic kernel");
    std::vector<uint8_t> in_chain(m.nq, 0);

    auto chain_of = [&](const std::vector<int> &joints, int frame, int task_index) {
        if (static_cast<int>(joints.size()) > kMaxChain) throw Unsupported("support chain longer than the specialised kernels take");
        for (int j : joints) {
            if (m.joint_type[j] != IKGPU_JOINT_REVOLUTE) throw Unsupported("non-revolute joint in a task support");
            if (in_chain[m.joint_idx_q[j]]) throw Unsupported("a joint is in the support of two tasks");
        }
        return build_chain(m, joints, frame, task_index, in_chain);
    };

    if (!ph.constraints.empty() && !free_flyer) throw Unsupported("a constraint on a fixed-base model runs on the generic kernel");
    if (!free_flyer && ntasks == 1 && ph.tasks[0].type <= IKGPU_FULL) {
        const ikgpu_task &t = ph.tasks[0];
        if (m.frame_parent[t.reference] != 0) throw Unsupported("reference frame moves with the configuration");
        std::memcpy(ph.ref_pl, m.frame_placement[t.reference].data(), sizeof(double) * 12);
        std::vector<int> joints;
        for (int j = m.frame_parent[t.frame]; j > 0; j = m.joint_parent[j]) joints.insert(joints.begin(), j);
        if (joints.empty()) throw Unsupported("task frame fixed in the world");
        ph.chain = chain_of(joints, t.frame, 0);
        ph.chain_struct = chain_structure(ph.chain);
        ph.chain_hot = chain_hot_table(ph.chain);
        ph.kind = KernelKind::Chain;
        ph.kernel_name = "dls_chain<NJ=" + std::to_string(ph.chain.nj) + "," + kt[t.type] + ",general>";   // capi.cpp renames it when a hot build is taken
        ph.q_in_chain = in_chain;
        return;
    }
    // Tree kind.  Free-flyer base: chains hang off joint 1.  Fixed base (several tasks, or alignment / posture rows next to one
    // chain task): chains hang off the universe, and the kernel drops the base block it solves (TreeParams::fixed_base).
    const int root = free_flyer ? 1 : 0;
    ph.fixed_base = !free_flyer;
    if (free_flyer)
        for (int k = 0; k < 7; ++k) in_chain[k] = 1;
    int nchains = 0;
    int chain_frame[2] = {-1, -1};
    for (int i = 0; i < ntasks; ++i) {
        const ikgpu_task &t = ph.tasks[i];
        if (t.type > IKGPU_FULL) continue;  // alignment and posture rows are matched to the chains below
        const bool ref_world = m.frame_parent[t.reference] == 0 && is_identity(m.frame_placement[t.reference]);
        const bool ref_on_base = free_flyer && m.frame_parent[t.reference] == 1;
        if (!ref_world && !ref_on_base) throw Unsupported("reference frame neither the universe nor on the floating base");
        if (m.frame_parent[t.frame] == 0) throw Unsupported("task frame fixed in the world");
        std::vector<int> joints;
        for (int j = m.frame_parent[t.frame]; j > root; j = m.joint_parent[j]) joints.insert(joints.begin(), j);
        if (joints.empty()) {
            if (ph.base_task >= 0) throw Unsupported("two tasks on the floating base link");
            if (!ref_world) throw Unsupported("base-link task with a reference frame other than the universe");
            ph.base_task = i;
            std::memcpy(ph.base_frame_pl, m.frame_placement[t.frame].data(), sizeof(double) * 12);
        } else {
            if (nchains == 2) throw Unsupported("more than two chain tasks on a free-flyer model");
            (nchains == 0 ? ph.chain : ph.chainB) = chain_of(joints, t.frame, i);
            if (ref_on_base) {
                ph.ref_base[nchains] = 1;
                std::memcpy(ph.chain_ref_pl[nchains], m.frame_placement[t.reference].data(), sizeof(double) * 12);
            }
            chain_frame[nchains] = t.frame;
            ++nchains;
        }
    }
    for (int i = 0; i < ntasks; ++i) {  // one AlignAxisTask row, on the frame of a chain task, direction given in the world
        const ikgpu_task &t = ph.tasks[i];
        if (t.type <= IKGPU_FULL || t.type == IKGPU_POSTURE_ROW) continue;
        if (ph.align_task >= 0) throw Unsupported("more than one AlignAxisTask");
        if (m.frame_parent[t.reference] != 0 || !is_identity(m.frame_placement[t.reference]))
            throw Unsupported("AlignAxisTask with a reference frame other than the universe");
        const int c = t.frame == chain_frame[0] ? 0 : (t.frame == chain_frame[1] ? 1 : -1);
        if (c < 0) throw Unsupported("AlignAxisTask on a frame that carries no frame task");
        ph.align_task = i;
        ph.align_chain = c;
    }
    if (nchains == 0) throw Unsupported("free-flyer problem without a chain task");
    if (!ph.constraints.empty()) {
        // ONE FrameConstraint, reference frame = the universe, on a frame that ends a chain off the floating base which no task
        // touches (the pinned stance foot of a humanoid): that chain becomes chain B with no task
        const ikgpu_task &c = ph.constraints[0];
        if (ph.constraints.size() != 1) throw Unsupported("more than one constraint");
        if (nchains != 1) throw Unsupported("a constraint next to two chain tasks");
        if (m.frame_parent[c.reference] != 0 || !is_identity(m.frame_placement[c.reference]))
            throw Unsupported("constraint with a reference frame other than the universe");
        std::vector<int> joints;
        for (int j = m.frame_parent[c.frame]; j > root; j = m.joint_parent[j]) joints.insert(joints.begin(), j);
        if (joints.empty()) throw Unsupported("constraint on the floating base link");
        ph.chainB = chain_of(joints, c.frame, -1);   // throws when it shares a joint with the task chain
        if (ph.chainB.nj != ph.chain.nj) throw Unsupported("the constrained chain and the task chain differ in length");
        ph.cons_on = true;
        ph.cons_type = c.type;
    }
    if (nchains == 2 && ph.chain.nj != ph.chainB.nj) throw Unsupported("the two chains differ in length");
    // PostureTask rows (frame = tangent column, reference = index in q, weight[0] = weight, weight[1] = mask entry)
    for (int c = 0; c < 2; ++c)
        for (int j = 0; j < kMaxChain; ++j) ph.posture_chain_task[c][j] = -1;
    std::vector<uint8_t> has_row(m.nq, 0);
    for (int i = 0; i < ntasks; ++i) {
        const ikgpu_task &t = ph.tasks[i];
        if (t.type != IKGPU_POSTURE_ROW) continue;
        if (ph.has_posture && t.priority != ph.posture_prio) throw Unsupported("posture rows on different priority levels");
        ph.has_posture = true;
        ph.posture_prio = t.priority;
        const int v = t.frame, qi = t.reference;
        const int off_q = free_flyer ? 7 : 0, off_v = free_flyer ? 6 : 0;
        if (qi < off_q || qi >= m.nq || v < off_v || v >= m.nv || qi - off_q != v - off_v) throw Unsupported("posture row on the floating base");
        if (has_row[qi]) throw Unsupported("two posture rows on one joint");
        has_row[qi] = 1;
        int where = -1, at = -1;
        for (int c = 0; c < (ph.cons_on ? 2 : nchains) && where < 0; ++c) {   // (the constrained chain is chain B)
            const ChainHost &ch = c == 0 ? ph.chain : ph.chainB;
            for (int j = 0; j < ch.nj; ++j)
                if (ch.vidx[j] == v) { where = c; at = j; break; }
        }
        if (where >= 0) {
            ph.posture_chain_task[where][at] = i;
            ph.posture_chain_w[where][at] = t.weight[0];
            ph.posture_chain_mask[where][at] = t.weight[1];
        } else {
            if (static_cast<int>(ph.posture_out.size()) == kMaxPostureOut) throw Unsupported("more posture rows outside the chains than the tree kernel takes");
            ph.posture_out.push_back({i, qi, t.weight[0], t.weight[1]});
            in_chain[qi] = 1;  // the kernel steps and clamps this entry itself
        }
    }
    ph.kind = KernelKind::Tree;
    ph.kernel_name = "dls_tree<NJ=" + std::to_string(ph.chain.nj) + ",chains=" + std::to_string(nchains) +
                     (ph.base_task >= 0 ? ",base_task" : "") + (ph.ref_base[0] || ph.ref_base[1] ? ",base_reference" : "") +
                     (ph.align_task >= 0 ? ",align_axis" : "") + (ph.has_posture ? ",posture" : "") + (ph.fixed_base ? ",fixed_base" : "") +
                     (ph.cons_on ? ",constraint_rows=" + std::to_string(ph.crows) : std::string()) + ">";
    ph.q_in_chain = in_chain;
}

// Tables of the cooperative DLS program (device/coop_solver.hpp): which tangent columns each task row set touches, the
// (i, j) pairs of the lower triangle of the augmented Gram matrix, the joints ordered by tree depth for the forward
// kinematics, and the LDS layout.
constexpr size_t kCoopLdsCap = 160 * 1024;

void build_coop(ProblemHost &ph, const Model &m) {
    GenericHost &g = ph.generic;
    const int nj = m.njoints(), nt = ph.ntasks, nv = m.nv, M = ph.rows;
    auto put_i = [&](const std::vector<int32_t> &v) { int o = static_cast<int>(g.ints.size()); g.ints.insert(g.ints.end(), v.begin(), v.end()); return o; };
    std::vector<int32_t> support(static_cast<size_t>(nt) * nv, 0);
    for (int t = 0; t < nt; ++t) {
        const ikgpu_task &k = ph.tasks[t];
        if (k.type == IKGPU_POSTURE_ROW) { support[static_cast<size_t>(t) * nv + k.frame] = 1; continue; }
        if (k.type == IKGPU_CENTRE_OF_MASS) continue;
        for (int j = m.frame_parent[k.frame]; j > 0; j = m.joint_parent[j]) {
            const int n = m.joint_type[j] == IKGPU_JOINT_FREEFLYER ? 6 : 1;
            for (int c = m.joint_idx_v[j]; c < m.joint_idx_v[j] + n; ++c) support[static_cast<size_t>(t) * nv + c] = 1;
        }
    }
    g.o_csupport = put_i(support);
    std::vector<int32_t> tbi(nt, -1);  // slot of the task's 36-double block; posture rows need none
    int nblocks = 0;
    for (int t = 0; t < nt; ++t)
        if (ph.tasks[t].type != IKGPU_POSTURE_ROW) tbi[t] = nblocks++;
    g.o_ctbindex = put_i(tbi);
    std::vector<int32_t> btask;
    for (int t = 0; t < nt; ++t)
        if (tbi[t] >= 0) btask.push_back(t);
    g.o_cbtask = put_i(btask);
    g.coop_nblocks = nblocks;
    // FrameConstraint rows: which tangent columns move the constrained frame, which its reference frame
    const int ncons = static_cast<int>(ph.constraints.size());
    std::vector<int32_t> csf(static_cast<size_t>(ncons) * nv, 0), csr(static_cast<size_t>(ncons) * nv, 0);
    for (int k = 0; k < ncons; ++k)
        for (int side = 0; side < 2; ++side) {
            std::vector<int32_t> &mask = side == 0 ? csf : csr;
            const int frame = side == 0 ? ph.constraints[k].frame : ph.constraints[k].reference;
            for (int j = m.frame_parent[frame]; j > 0; j = m.joint_parent[j]) {
                const int n = m.joint_type[j] == IKGPU_JOINT_FREEFLYER ? 6 : 1;
                for (int c = m.joint_idx_v[j]; c < m.joint_idx_v[j] + n; ++c) mask[static_cast<size_t>(k) * nv + c] = 1;
            }
        }
    g.o_ccsf = put_i(csf);
    g.o_ccsr = put_i(csr);
    // PostureTask rows are eliminated from the linear system (device/coop_solver.hpp, coop_dls): the remaining rows, and the posture
    // rows by tangent column
    std::vector<int32_t> frow, pstart(nv + 1, 0), ptask;
    for (int t = 0; t < nt; ++t)
        if (ph.tasks[t].type != IKGPU_POSTURE_ROW)
            for (int r = 0; r < task_dim(ph.tasks[t]); ++r) frow.push_back(g.ints[g.o_trow + t] + r);
    for (int c = 0; c < nv; ++c) {
        pstart[c] = static_cast<int32_t>(ptask.size());
        for (int t = 0; t < nt; ++t)
            if (ph.tasks[t].type == IKGPU_POSTURE_ROW && ph.tasks[t].frame == c) ptask.push_back(t);
    }
    pstart[nv] = static_cast<int32_t>(ptask.size());
    g.coop_post_elim = ptask.empty() ? 0 : 1;
    g.coop_Mf = static_cast<int>(frow.size());
    std::vector<int32_t> jrow(nt, -1), tgoff(nt, 0), tgsrc;
    {
        int k = 0;
        for (int t = 0; t < nt; ++t) {
            if (ph.tasks[t].type == IKGPU_POSTURE_ROW) {
                tgoff[t] = static_cast<int32_t>(tgsrc.size()) - 9;
                tgsrc.push_back(t * 12 + 9);
            } else {
                jrow[t] = k; k += task_dim(ph.tasks[t]);
                tgoff[t] = static_cast<int32_t>(tgsrc.size());
                for (int w = 0; w < 12; ++w) tgsrc.push_back(t * 12 + w);
            }
        }
    }
    g.coop_ntg = static_cast<int>(tgsrc.size());
    g.o_cjrow = put_i(jrow);
    g.o_ctgoff = put_i(tgoff);
    g.o_ctgsrc = put_i(tgsrc);
    g.o_cfrow = put_i(frow);
    g.o_cpstart = put_i(pstart);
    g.o_cptask = put_i(ptask);
    std::vector<int32_t> colj(nv, 0);
    for (int j = 1; j < nj; ++j)
        for (int c = m.joint_idx_v[j]; c < m.joint_idx_v[j] + (m.joint_type[j] == IKGPU_JOINT_FREEFLYER ? 6 : 1); ++c) colj[c] = j;
    g.o_ccoljoint = put_i(colj);
    std::vector<int32_t> pi, pj;   // lower triangle of the (M + 1) x (M + 1) augmented matrix; row M carries the right-hand side
    for (int i = 0; i <= M; ++i)
        for (int j = 0; j <= i && j < M; ++j) { pi.push_back(i); pj.push_back(j); }
    g.coop_npairs = static_cast<int>(pi.size());
    g.o_cpair_i = put_i(pi);
    g.o_cpair_j = put_i(pj);
    std::vector<int32_t> ci, cj;   // lower triangle of Jc Jc^T (Cholesky-QR basis of the constraint Jacobian)
    for (int i = 0; i < ph.crows; ++i)
        for (int j = 0; j <= i; ++j) { ci.push_back(i); cj.push_back(j); }
    g.o_ccpair_i = put_i(ci);
    g.o_ccpair_j = put_i(cj);
    // forward kinematics by CHAIN: a chain is a run of joints each the only child of the one before; one lane walks a chain with
    // the running world placement in registers (the same products, in the same order, as the sequential pass of the per-lane
    // program).  A joint with several children ends its chain, every child starts one; the chains of one level hang off joints of
    // the levels before and are independent.  (By tree DEPTH -- one barrier and one LDS round trip per joint of the longest path,
    // eight for a Cassie leg -- the phase was 23 % of the kernel.)
    std::vector<int> nchild(nj, 0), chain_of(nj, -1);
    for (int j = 1; j < nj; ++j) nchild[m.joint_parent[j]]++;
    std::vector<std::vector<int32_t>> chains;
    std::vector<int> chain_level;
    for (int j = 1; j < nj; ++j) {   // (a joint's parent has a smaller index: model.cpp builds the tree in that order)
        const int p = m.joint_parent[j];
        if (p > 0 && nchild[p] == 1) { chain_of[j] = chain_of[p]; chains[chain_of[j]].push_back(j); continue; }
        chain_of[j] = static_cast<int>(chains.size());
        chains.push_back({j});
        chain_level.push_back(p > 0 ? chain_level[chain_of[p]] + 1 : 0);
    }
    int nlevels_fk = 0;
    for (int l : chain_level) nlevels_fk = std::max(nlevels_fk, l + 1);
    std::vector<int32_t> order, chain_start, lvl_start;
    for (int l = 0; l < nlevels_fk; ++l) {
        lvl_start.push_back(static_cast<int32_t>(chain_start.size()));
        for (size_t c = 0; c < chains.size(); ++c)
            if (chain_level[c] == l) { chain_start.push_back(static_cast<int32_t>(order.size())); order.insert(order.end(), chains[c].begin(), chains[c].end()); }
    }
    lvl_start.push_back(static_cast<int32_t>(chain_start.size()));
    chain_start.push_back(static_cast<int32_t>(order.size()));
    g.coop_rounds = nlevels_fk;
    g.o_cup = put_i(order);
    g.o_cchain = put_i(chain_start);
    g.o_clvl = put_i(lvl_start);
    // LDS layout of one problem.  Two pairs of arrays never live at the same time and share their space: the local joint
    // transforms (read by the forward kinematics only) with the task Jacobian (written after it), and the per-task blocks
    // (read by the Jacobian columns only) with the Gram matrix, its pivots and the solution (written after them).
    int o = 0;
    g.c_q = o; o += m.nq;
    g.c_tg = o + 9; o += g.coop_ntg + 9;   // (a leading posture row's block starts nine words before its one word)
    int mmax = 0;  // rows of the largest priority level: the prioritised solver parks that level's projected Jacobian where the
    {              // joint placements lived (dead once the task Jacobian is built) and two short vectors over the joint Jacobian
        std::vector<int> lvl(static_cast<size_t>(g.nlevels), 0);
        for (int t = 0; t < nt; ++t) lvl[ph.tasks[t].priority] += task_dim(ph.tasks[t]);
        for (int x : lvl) mmax = std::max(mmax, x);
    }
    g.c_A1 = o; o += std::max(12 * nj, mmax * nv);
    g.c_Jw = o; o += std::max(6 * nv, 2 * mmax);
    const int Mc = ph.crows;
    g.c_e = o; g.c_cnrm = o; o += std::max(M, Mc);   // (the constraint projection runs after the error vector is dead)
    g.c_dq = o; o += nv;
    g.c_Dd = o; o += g.coop_post_elim ? nv : 0;
    g.c_sf = o; o += g.has_com ? 3 * nj : 0;
    g.c_A0 = o; g.c_J = o; g.c_Jc = o; o += std::max(std::max(12 * nj, M * nv), Mc * nv);   // (and after the task Jacobian is)
    g.c_tb = o; g.c_G = o; g.c_cb = o;
    g.c_dinv = g.c_G + (M + 1) * (M + 2) / 2;
    g.c_x = g.c_dinv + M;
    o += std::max(std::max(36 * nblocks, (M + 1) * (M + 2) / 2 + 2 * M), 36 * ncons);
    g.coop_words = o + (o % 2 == 0 ? 1 : 0);  // odd stride between the groups of a block
    if (g.coop_post_elim) {   // the DLS solver's own layout when the posture rows are eliminated: Mf rows
        const int Mf = g.coop_Mf;
        int d = g.c_tg + g.coop_ntg;              // q and the targets as above
        g.d_A1 = d; d += 12 * nj;
        g.d_Jw = d; d += 6 * nv;
        g.d_e = d; d += std::max(M, Mc);
        g.d_dq = d; d += nv;
        g.d_Dd = d; d += nv;
        g.d_sf = d; d += g.has_com ? 3 * nj : 0;
        g.d_J = d; d += std::max(std::max(12 * nj, Mf * nv), Mc * nv);
        g.d_G = d; d += std::max(std::max(36 * nblocks, (Mf + 1) * (Mf + 2) / 2 + 2 * Mf), 36 * ncons);
        g.d_words = d + (d % 2 == 0 ? 1 : 0);
    }
    g.c_P = g.coop_words;  // ik::pik: the projector follows the DLS workspace
    g.coop_words_pik = g.coop_words + nv * nv;
    g.coop_words_pik += (g.coop_words_pik % 2 == 0 ? 1 : 0);
    g.coop_mmax = mmax;
    // one 64-lane block holds the packed tables and four workspaces, anything up to the CU's 160 KB of LDS (the launch raises the
    // kernel's dynamic-LDS limit past the 64 KB default).  Even one block per CU beats the memory-resident per-lane program by far:
    // M = 28 (two feet + pelvis + sixteen posture rows) 265 ms per launch there.
    const size_t lds_pik = 8 * (4 * static_cast<size_t>(g.coop_words_pik) + g.dbls.size() + (g.ints.size() + 1) / 2);
    g.coop_pik_ok = lds_pik <= kCoopLdsCap ? 1 : 0;
    const size_t lds_bytes = 8 * (4 * static_cast<size_t>(g.coop_post_elim ? g.d_words : g.coop_words) + g.dbls.size() + (g.ints.size() + 1) / 2);
    g.coop_ok = lds_bytes <= kCoopLdsCap ? 1 : 0;
}

void build_generic(ProblemHost &ph, const Model &m) {
    GenericHost &g = ph.generic;
    g = GenericHost();
    const int nj = m.njoints(), nt = ph.ntasks;
    g.njoints = nj;
    auto put_i = [&](const std::vector<int32_t> &v) { int o = static_cast<int>(g.ints.size()); g.ints.insert(g.ints.end(), v.begin(), v.end()); return o; };
    g.o_jtype = put_i(m.joint_type);
    g.o_parent = put_i(m.joint_parent);
    g.o_idx_q = put_i(m.joint_idx_q);
    g.o_idx_v = put_i(m.joint_idx_v);
    std::vector<int32_t> ttype, tfj, trj, trow, tdim, tprio;
    for (int i = 0; i < nt; ++i) {
        const ikgpu_task &t = ph.tasks[i];
        ttype.push_back(t.type);
        const bool posture = t.type == IKGPU_POSTURE_ROW;  // frame / reference are then tangent / configuration indices
        if (t.type == IKGPU_CENTRE_OF_MASS) {               // no frame of its own: the universe stands in
            g.has_com = 1;
            tfj.push_back(0);
            trj.push_back(m.frame_parent[t.reference]);
            trow.push_back(ph.task_row[i]);
            tdim.push_back(task_dim(t));
            tprio.push_back(t.priority);
            continue;
        }
        tfj.push_back(posture ? t.frame : m.frame_parent[t.frame]);
        trj.push_back(posture ? t.reference : m.frame_parent[t.reference]);
        trow.push_back(ph.task_row[i]);
        tdim.push_back(task_dim(t));
        tprio.push_back(t.priority);
    }
    g.o_ttype = put_i(ttype); g.o_tfjoint = put_i(tfj); g.o_trjoint = put_i(trj);
    g.o_trow = put_i(trow); g.o_tdim = put_i(tdim); g.o_tprio = put_i(tprio);
    // prioritised IK: first row of each priority level (tasks are listed in non-decreasing priority)
    g.nlevels = nt > 0 ? ph.tasks[nt - 1].priority + 1 : 1;
    std::vector<int32_t> lvl(g.nlevels + 1, 0);
    for (int i = 0; i < nt; ++i) lvl[ph.tasks[i].priority + 1] += task_dim(ph.tasks[i]);
    int mmax = 0;
    for (int l = 0; l < g.nlevels; ++l) {
        mmax = std::max(mmax, lvl[l + 1]);
        lvl[l + 1] += lvl[l];
    }
    g.o_lvlrow0 = put_i(lvl);
    std::vector<int32_t> ctype, cfj, crj, crow, cdim;
    int crows = 0;
    for (const ikgpu_task &c : ph.constraints) {
        ctype.push_back(c.type);
        cfj.push_back(m.frame_parent[c.frame]);
        crj.push_back(m.frame_parent[c.reference]);
        crow.push_back(crows);
        cdim.push_back(task_dim(c));
        crows += task_dim(c);
    }
    g.o_ctype = put_i(ctype); g.o_cfjoint = put_i(cfj); g.o_crjoint = put_i(crj); g.o_crow = put_i(crow); g.o_cdim = put_i(cdim);
    auto put_d = [&](const double *p, size_t n) { int o = static_cast<int>(g.dbls.size()); g.dbls.insert(g.dbls.end(), p, p + n); return o; };
    g.o_placement = static_cast<int>(g.dbls.size());
    for (int j = 0; j < nj; ++j) put_d(m.joint_placement[j].data(), 12);
    g.o_axis = static_cast<int>(g.dbls.size());
    for (int j = 0; j < nj; ++j) put_d(m.joint_axis[j].data(), 3);
    g.o_lower = put_d(m.lower.data(), m.lower.size());
    g.o_upper = put_d(m.upper.data(), m.upper.size());
    g.o_tfpl = static_cast<int>(g.dbls.size());
    const SE3 ident = se3_identity();
    for (int i = 0; i < nt; ++i)
        put_d(ph.tasks[i].type == IKGPU_POSTURE_ROW || ph.tasks[i].type == IKGPU_CENTRE_OF_MASS ? ident.data()
                                                                                              : m.frame_placement[ph.tasks[i].frame].data(), 12);
    g.o_trpl = static_cast<int>(g.dbls.size());
    for (int i = 0; i < nt; ++i)
        put_d(ph.tasks[i].type == IKGPU_POSTURE_ROW ? ident.data() : m.frame_placement[ph.tasks[i].reference].data(), 12);
    g.o_tw = static_cast<int>(g.dbls.size());
    for (int i = 0; i < nt; ++i) put_d(ph.tasks[i].weight, 6);
    {   // pinocchio::centerOfMass constants: mass and lever per joint, mass of every subtree, 1 / total mass
        std::vector<double> sub(m.joint_mass);
        if (!sub.empty()) sub[0] = 0.0;
        for (int j = nj - 1; j > 0; --j) sub[m.joint_parent[j]] += sub[j];
        g.o_jmass = put_d(m.joint_mass.data(), m.joint_mass.size());
        g.o_jlever = static_cast<int>(g.dbls.size());
        for (int j = 0; j < nj; ++j) put_d(m.joint_com[j].data(), 3);
        g.o_jsubmass = put_d(sub.data(), sub.size());
        g.inv_total_mass = sub.empty() || !(sub[0] > 0.0) ? 0.0 : 1.0 / sub[0];
    }
    g.o_cfpl = static_cast<int>(g.dbls.size());
    for (const ikgpu_task &c : ph.constraints) put_d(m.frame_placement[c.frame].data(), 12);
    g.o_crpl = static_cast<int>(g.dbls.size());
    for (const ikgpu_task &c : ph.constraints) put_d(m.frame_placement[c.reference].data(), 12);
    const int M = ph.rows, nv = m.nv;
    int o = 0;
    g.off_q = o; o += m.nq;
    g.off_oMi = o; o += 12 * nj;
    g.off_Jw = o; o += 6 * nv;
    g.off_e = o; o += M;
    g.off_J = o; o += M * nv;
    g.off_G = o; o += M * (M + 1) / 2;
    g.off_y = o; o += M;
    g.off_dq = o; o += nv;
    g.off_Jc = o; o += crows * nv;
    g.off_sf = o; o += g.has_com ? 3 * nj : 0;
    g.ws_words = o;
    build_coop(ph, m);
    g.off_P = o; o += nv * nv;
    g.off_Jb = o; o += mmax * nv;
    g.off_de = o; o += mmax;
    g.ws_words_pik = o;
    ph.kind = KernelKind::Generic;
    ph.kernel_name = "dls_generic<M=" + std::to_string(M) + ",nv=" + std::to_string(nv) + ",joints=" + std::to_string(nj - 1) +
                     (crows > 0 ? ",constraint_rows=" + std::to_string(crows) : std::string()) + ">";
    ph.q_in_chain.assign(m.nq, 1);
}

}  // namespace

ProblemHost analyse_problem(const Model &m, const ikgpu_task *tasks, int ntasks, bool force_generic,
                            const ikgpu_task *constraints, int nconstraints) {
    if (ntasks < 1) throw std::runtime_error("a problem needs at least one task");
    if (nconstraints < 0 || (nconstraints > 0 && !constraints)) throw std::runtime_error("invalid constraint list");
    ProblemHost ph;
    for (int i = 0; i < nconstraints; ++i) {
        const ikgpu_task &c = constraints[i];
        if (c.frame < 0 || c.frame >= m.nframes()) throw std::runtime_error("constraint " + std::to_string(i) + ": frame id out of range");
        if (c.reference < 0 || c.reference >= m.nframes())
            throw std::runtime_error("constraint " + std::to_string(i) + ": reference frame id out of range");
        if (c.type < IKGPU_POSITION || c.type > IKGPU_FULL)
            throw std::runtime_error("constraint " + std::to_string(i) + ": a frame constraint is of type Position, Orientation or Full");
        ph.constraints.push_back(c);
        ph.crows += task_dim(c);
    }
    if (nconstraints > 1) force_generic = true;   // one constraint may fit the tree kernel (specialise() decides), more run on the generic kernel
    ph.nq = m.nq;
    ph.nv = m.nv;
    ph.ntasks = ntasks;
    ph.lower = m.lower;
    ph.upper = m.upper;
    ph.q_in_chain.assign(m.nq, 0);
    int last_prio = 0;
    for (int i = 0; i < ntasks; ++i) {
        const ikgpu_task &t = tasks[i];
        if (t.type == IKGPU_CENTRE_OF_MASS) {
            if (t.reference < 0 || t.reference >= m.nframes()) throw std::runtime_error("task " + std::to_string(i) + ": reference frame id out of range");
            if (!(m.total_mass() > 0.0))
                throw std::runtime_error("task " + std::to_string(i) + ": a centre-of-mass task needs joint masses (the model has none)");
        } else if (t.type == IKGPU_POSTURE_ROW) {
            if (t.frame < 0 || t.frame >= m.nv) throw std::runtime_error("task " + std::to_string(i) + ": posture row tangent index out of range");
            if (t.reference < 0 || t.reference >= m.nq) throw std::runtime_error("task " + std::to_string(i) + ": posture row configuration index out of range");
        } else {
            if (t.frame < 0 || t.frame >= m.nframes()) throw std::runtime_error("task " + std::to_string(i) + ": frame id out of range");
            if (t.reference < 0 || t.reference >= m.nframes()) throw std::runtime_error("task " + std::to_string(i) + ": reference frame id out of range");
        }
        if (t.type < IKGPU_POSITION || t.type > IKGPU_CENTRE_OF_MASS) throw std::runtime_error("task " + std::to_string(i) + ": unknown kinematic type");
        if (t.priority < 0) throw std::runtime_error("task " + std::to_string(i) + ": negative priority");
        if (t.priority < last_prio) throw std::runtime_error("tasks must be listed in stacking order (non-decreasing priority)");
        last_prio = t.priority;
        ph.tasks.push_back(t);
        ph.task_row.push_back(ph.rows);
        ph.rows += task_dim(t);
    }
    if (!force_generic) {
        try {
            ProblemHost sp = ph;
            specialise(sp, m);
            return sp;
        } catch (const Unsupported &) {
        }
    }
    build_generic(ph, m);
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

std::vector<double> tree_desc_table(const ProblemHost &ph) {
    // ikdev::TreeDesc<NJ, NCH>: ChainTable chain[NCH] {pl, fr, lo, hi, w}, frP, wP
    std::vector<double> t;
    double w6[6];
    auto put_chain = [&](const ChainHost &c) {
        for (int j = 0; j < c.nj; ++j) t.insert(t.end(), c.pl[j], c.pl[j] + 12);
        t.insert(t.end(), c.frame_pl, c.frame_pl + 12);
        t.insert(t.end(), c.lo, c.lo + c.nj);
        t.insert(t.end(), c.hi, c.hi + c.nj);
        for (double &x : w6) x = 0.0;
        if (c.task >= 0) weights6(ph.tasks[c.task], w6);   // (the constrained chain of a constraint build carries no task)
        t.insert(t.end(), w6, w6 + 6);
    };
    put_chain(ph.chain);
    if (ph.chainB.nj > 0) put_chain(ph.chainB);
    t.insert(t.end(), ph.base_frame_pl, ph.base_frame_pl + 12);
    for (double &x : w6) x = 0.0;
    if (ph.base_task >= 0) weights6(ph.tasks[ph.base_task], w6);
    t.insert(t.end(), w6, w6 + 6);
    t.insert(t.end(), ph.chain_ref_pl[0], ph.chain_ref_pl[0] + 12);
    if (ph.chainB.nj > 0) t.insert(t.end(), ph.chain_ref_pl[1], ph.chain_ref_pl[1] + 12);
    return t;
}

namespace {
bool rot_is_identity(const double *pl) {
    static const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int k = 0; k < 9; ++k)
        if (pl[k] != I[k]) return false;
    return true;
}
}  // namespace

int chain_identity_mask(const ChainHost &c) {
    int m = 0;
    for (int j = 0; j < c.nj; ++j)
        if (rot_is_identity(c.pl[j])) m |= 1 << j;
    if (rot_is_identity(c.frame_pl)) m |= 1 << c.nj;
    return m;
}

ChainStructure chain_structure(const ChainHost &c) {
    ChainStructure out;
    if (c.nj < 1 || c.nj > 7) return out;
    out.fits = true;
    for (int i = 0; i <= c.nj; ++i) {
        const double *pl = i < c.nj ? c.pl[i] : c.frame_pl;
        uint64_t code = 0;
        for (int e = 0; e < 9; ++e) {
            const uint64_t cls = pl[e] == 0.0 ? 1u : (pl[e] == 1.0 ? 2u : (pl[e] == -1.0 ? 3u : 0u));
            if (cls == 0) ++out.values;
            code |= cls << (2 * e);
        }
        for (int k = 0; k < 3; ++k)
            if (pl[9 + k] != 0.0) { code |= uint64_t{1} << (18 + k); ++out.values; }
        out.code[i / 3] |= code << (21 * (i % 3));
    }
    return out;
}

std::vector<double> chain_hot_table(const ChainHost &c) {
    std::vector<double> t;
    for (int i = 0; i <= c.nj; ++i) {
        const double *pl = i < c.nj ? c.pl[i] : c.frame_pl;
        for (int e = 0; e < 9; ++e)
            if (!(pl[e] == 0.0 || pl[e] == 1.0 || pl[e] == -1.0)) t.push_back(pl[e]);
        for (int k = 0; k < 3; ++k)
            if (pl[9 + k] != 0.0) t.push_back(pl[9 + k]);
    }
    t.insert(t.end(), c.lo, c.lo + c.nj);
    t.insert(t.end(), c.hi, c.hi + c.nj);
    return t;
}

bool task_has_unit_weights(const ikgpu_task &t) {
    for (int i = 0; i < task_dim(t); ++i)
        if (t.weight[i] != 1.0) return false;
    return true;
}

void fill_chain_args(const ProblemHost &ph, double *ref_pl12, int *qidx, int *vidx, int *nq, int *nv, int *priority,
                     int *idmask, int *unit_weights) {
    const ChainHost &c = ph.chain;
    for (int j = 0; j < c.nj; ++j) {
        qidx[j] = c.qidx[j];
        vidx[j] = c.vidx[j];
    }
    std::memcpy(ref_pl12, ph.ref_pl, sizeof(double) * 12);
    *nq = ph.nq;
    *nv = ph.nv;
    *priority = ph.tasks[0].priority;
    *idmask = chain_identity_mask(c);
    *unit_weights = task_has_unit_weights(ph.tasks[0]) ? 1 : 0;
}

TreeArgsHost tree_args(const ProblemHost &ph) {
    TreeArgsHost a{};
    for (int j = 0; j < kMaxChain; ++j) {
        a.qidx[0][j] = ph.chain.qidx[j]; a.vidx[0][j] = ph.chain.vidx[j];
        a.qidx[1][j] = ph.chainB.qidx[j]; a.vidx[1][j] = ph.chainB.vidx[j];
    }
    const int slot_task[3] = {ph.chain.task, ph.chainB.nj > 0 ? ph.chainB.task : -1, ph.base_task};
    for (int s = 0; s < 3; ++s) {
        const int ti = slot_task[s];
        a.tslot[s] = ti >= 0 ? ti : 0;
        a.trow[s] = ti >= 0 ? ph.task_row[ti] : 0;
        a.tdim[s] = ti >= 0 ? task_dim(ph.tasks[ti]) : 0;
        a.trow0[s] = (ti >= 0 && ph.tasks[ti].type == IKGPU_ORIENTATION) ? 3 : 0;
        a.prio[s] = ti >= 0 ? ph.tasks[ti].priority : 1;
    }
    a.hasP = ph.base_task >= 0 ? 1 : 0;
    a.nch = ph.chainB.nj > 0 ? 2 : 1;
    a.idmask[0] = chain_identity_mask(ph.chain);
    a.idmask[1] = ph.chainB.nj > 0 ? chain_identity_mask(ph.chainB) : 0;
    a.idmaskP = rot_is_identity(ph.base_frame_pl) ? 1 : 0;
    for (int s = 0; s < 3; ++s) {
        // only a Full task with all-ones weights takes the unit path: a Position / Orientation task is a Full
        // task with zero-weight rows in the tree kernels
        const int ti = slot_task[s];
        a.unit[s] = (ti >= 0 && ph.tasks[ti].type == IKGPU_FULL && task_has_unit_weights(ph.tasks[ti])) ? 1 : 0;
    }
    a.ref_base[0] = ph.ref_base[0]; a.ref_base[1] = ph.ref_base[1];
    a.align_chain = ph.align_task >= 0 ? ph.align_chain : -1;
    if (ph.align_task >= 0) {
        const ikgpu_task &t = ph.tasks[ph.align_task];
        a.align_axis = t.type - IKGPU_ALIGN_AXIS_X;
        a.align_slot = ph.align_task;
        a.align_prio = t.priority;
        a.align_w = t.weight[0];
    }
    a.fixed_base = ph.fixed_base ? 1 : 0;
    a.cons_on = ph.cons_on ? 1 : 0;
    a.cons_type = ph.cons_type;
    a.post_on = ph.has_posture ? 1 : 0;
    a.post_prio = ph.posture_prio;
    a.post_n = static_cast<int>(ph.posture_out.size());
    for (int k = 0; k < a.post_n; ++k) {
        a.post_q[k] = ph.posture_out[k].qi;
        a.post_slot[k] = ph.posture_out[k].task;
        a.post_w[k] = ph.posture_out[k].w;
        a.post_m[k] = ph.posture_out[k].mask;
    }
    for (int c = 0; c < 2; ++c)
        for (int j = 0; j < kMaxChain; ++j) {
            a.postc_slot[c][j] = ph.has_posture ? ph.posture_chain_task[c][j] : -1;
            a.postc_w[c][j] = ph.posture_chain_w[c][j];
            a.postc_m[c][j] = ph.posture_chain_mask[c][j];
        }
    return a;
}

}  // namespace ikgpu
