// problem.hpp -- host-side flattening of (model, task table) into the constant tables the
// gfx950 kernels read.  Stands in for what InverseKinematicsProblem + dls_data hold between
// calls (reference ik/ik/problem.hpp:17-22,183-189; ik/ik/dls.hpp:36-52; ik/ik/data.cpp:8-23).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "ikgpu.h"
#include "model.hpp"

namespace ikgpu {

constexpr int kMaxChain = 8;
constexpr int kMaxPostureOut = 32;  // == ikdev::kMaxPostOut: posture rows on joints outside the chains the tree kernel takes

// Chain:   fixed base, ONE FrameTask whose support is a serial chain of revolute joints (shapes S, U).
// Tree:    free-flyer base, up to two chain FrameTasks + at most one on the base link (shape F).
// Generic: everything else the path can express (any tree, any task list incl. AlignAxisTask rows, moving
//          reference frames, prismatic joints): the memory-resident fallback kernel.
enum class KernelKind { Chain, Tree, Generic };

// Packed tables of the generic kernel: one int buffer and one double buffer, with the offsets of each table.
struct GenericHost {
    std::vector<int32_t> ints;
    std::vector<double> dbls;
    int o_jtype = 0, o_parent = 0, o_idx_q = 0, o_idx_v = 0;                       // into ints
    int o_ttype = 0, o_tfjoint = 0, o_trjoint = 0, o_trow = 0, o_tdim = 0, o_tprio = 0;
    int o_placement = 0, o_axis = 0, o_lower = 0, o_upper = 0, o_tfpl = 0, o_trpl = 0, o_tw = 0;  // into dbls
    int njoints = 0;
    int off_q = 0, off_oMi = 0, off_Jw = 0, off_e = 0, off_J = 0, off_G = 0, off_y = 0, off_dq = 0, ws_words = 0;
    int nlevels = 1, o_lvlrow0 = 0;                               // prioritised IK: level -> first row (into ints)
    int off_P = 0, off_Jb = 0, off_de = 0, ws_words_pik = 0;      // and its extra workspace
    int o_ctype = 0, o_cfjoint = 0, o_crjoint = 0, o_crow = 0, o_cdim = 0;  // FrameConstraint rows (into ints)
    int o_ccpair_i = 0, o_ccpair_j = 0;  // cooperative form: (i, j) of the lower triangle of the Mc x Mc matrix Jc Jc^T, by rows
    int o_cfpl = 0, o_crpl = 0;                                              // (into dbls)
    int off_Jc = 0;
    // cooperative form of the DLS program (device/coop_solver.hpp): 16 lanes per problem, workspace in LDS
    int coop_ok = 0, coop_rounds = 0, coop_words = 0, coop_npairs = 0, coop_nblocks = 0, o_cbtask = 0;
    int coop_post_elim = 0, coop_Mf = 0, c_Dd = 0, o_cfrow = 0, o_cpstart = 0, o_cptask = 0;  // posture rows eliminated (coop_dls)
    // ... whose DLS workspace is laid out for the Mf remaining rows (ik::pik keeps the full one above): offsets d_*, and per task the
    // first row of its block in that compact Jacobian (o_cjrow; -1: posture row)
    int d_words = 0, d_A1 = 0, d_Jw = 0, d_e = 0, d_dq = 0, d_Dd = 0, d_sf = 0, d_J = 0, d_G = 0, o_cjrow = 0;
    // targets in LDS: twelve words per task with a pose / direction, ONE per posture row (its target value); o_ctgoff: [ntasks] offset
    // of the task's block (a posture row's word sits at offset + 9, where the pose's first translation entry would), o_ctgsrc: the
    // target slot (task * 12 + k) each LDS word is loaded from
    int coop_ntg = 0, o_ctgoff = 0, o_ctgsrc = 0;
    int coop_pik_ok = 0, coop_words_pik = 0, coop_mmax = 0, c_P = 0;  // ik::pik in the same form (device/pik_coop.hpp)
    int o_csupport = 0, o_cpair_i = 0, o_cpair_j = 0, o_cup = 0, o_cchain = 0, o_clvl = 0, o_ctbindex = 0, o_ccoljoint = 0, o_ccsf = 0, o_ccsr = 0;  // into ints: [ntasks][nv], [npairs] x 2, joints by depth [njoints - 1] + level starts [rounds + 1]
    int c_q = 0, c_tg = 0, c_A0 = 0, c_A1 = 0, c_Jw = 0, c_tb = 0, c_e = 0, c_J = 0, c_G = 0, c_dinv = 0, c_x = 0, c_dq = 0, c_sf = 0, c_cb = 0, c_Jc = 0, c_cnrm = 0;
    int has_com = 0, o_jmass = 0, o_jlever = 0, o_jsubmass = 0, off_sf = 0;  // centre-of-mass task (into dbls / workspace)
    double inv_total_mass = 0.0;
};

// Host copy of one serial chain (support of one task below its base), axis-folded: every joint
// rotates about its local z.
struct ChainHost {
    int nj = 0;
    int task = -1;  // index of the task in the (priority-ordered) task list
    int qidx[kMaxChain] = {};
    int vidx[kMaxChain] = {};
    double pl[kMaxChain][12] = {};
    double frame_pl[12] = {};
    double lo[kMaxChain] = {}, hi[kMaxChain] = {};
};

// Structure of the chain's constant placements for the structure-specialised chain kernel (device/chain_hot.hpp): 21 bits per
// placement i = 0 .. nj (nj: the frame placement) -- nine 2-bit classes of the rotation entries, row-major (0 general, 1 exactly
// zero, 2 exactly +1, 3 exactly -1) and three bits "translation component is non-zero" -- three placements per 64-bit word.
// Chains longer than 7 joints do not fit the three words and get code[*] = 0 ("everything general") with `fits` false.
struct ChainStructure {
    uint64_t code[3] = {0, 0, 0};
    bool fits = false;
    int values = 0;   // non-structural entries: the length of the compact table before lo / hi
};
ChainStructure chain_structure(const ChainHost &c);
// The compact table of that kernel: the non-structural placement values in placement order (general rotation entries
// row-major, then the non-zero translation components), then lo[nj], hi[nj].
std::vector<double> chain_hot_table(const ChainHost &c);

struct ProblemHost {
    KernelKind kind = KernelKind::Chain;
    std::string kernel_name;
    int nq = 0, nv = 0, ntasks = 0, rows = 0;
    std::vector<ikgpu_task> tasks;    // in stacking order (priority, then insertion)
    std::vector<int> task_row;        // first row of each task in the stacked system
    ChainHost chain;                  // Chain kind: the chain; Tree kind: chain A
    ChainStructure chain_struct;      // Chain kind: the placement-structure code of `chain` and
    std::vector<double> chain_hot;    //   its compact table (device/chain_hot.hpp), computed once at analysis
    // Chain kind: which build of the chain kernel this problem launches, decided ONCE when the problem is created (capi.cpp) and
    // part of kernel_name: 0 "general" (device/chain_solver.hpp), 1 "hot" (structure-specialised, compiled into the library:
    // kernels_hot.hip), 2 "hot-rtc" (the same kernel template instantiated for this chain's structure code at run time: rtc.cpp)
    int chain_build = 0;
    // Generic kind: 0 = the cooperative / per-lane memory-resident forms (kernels.hip), 2 = "static": the per-lane program compiled
    // for this problem at run time with its tables as constants (rtc.cpp); generic_key names the code object
    int generic_build = 0;
    uint64_t generic_key = 0;
    ChainHost chainB;                 // Tree kind: chain B (nj = 0 when absent)
    int base_task = -1;               // Tree kind: index of the task on the base link, or -1
    double base_frame_pl[12] = {};    // base joint frame -> frame of the base task
    double ref_pl[12] = {};           // Chain kind: world placement of the (fixed) reference frame
    // Tree kind, what the reference's demo adds (ik_ros/src/cassie.cpp:45-81): a chain task whose reference frame rides on
    // the floating base, and one AlignAxisTask row on a chain task's frame
    int ref_base[2] = {0, 0};
    double chain_ref_pl[2][12] = {{1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0}, {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0}};
    int align_task = -1, align_chain = -1;
    // ... and PostureTask rows (one ABI record per joint, all on one priority level): on chain joints they join the chain's
    // normal equations, on other joints they are 1x1 systems the tree kernel steps by itself (device/tree_solver.hpp)
    struct PostureRow { int task, qi; double w, mask; };   // task index == target slot
    bool has_posture = false;
    int posture_prio = 0;
    std::vector<PostureRow> posture_out;            // joints outside the chains
    int posture_chain_task[2][kMaxChain];           // chain joints: task index, or -1 (set by specialise())
    double posture_chain_w[2][kMaxChain] = {}, posture_chain_mask[2][kMaxChain] = {};
    // ... or a FIXED base: the same kernel with the base block solved and dropped (dq_base = 0, base pose = the world), for
    // fixed-base problems with two disjoint chain tasks or with alignment / posture rows next to one chain task
    bool fixed_base = false;
    // ... or ONE FrameConstraint with the universe as reference frame on a frame that ends a second chain carrying no task (the
    // pinned stance foot): chainB is that chain (task = -1), the tree kernel projects the step onto the constraint's null space
    bool cons_on = false;
    int cons_type = 0;
    bool tree_extras() const { return ref_base[0] || ref_base[1] || align_task >= 0 || has_posture || fixed_base || cons_on; }
    GenericHost generic;              // Generic kind
    std::vector<ikgpu_task> constraints;  // ik::FrameConstraint list (frame, reference, type); forces the Generic kind
    int crows = 0;
    std::vector<uint8_t> q_in_chain;  // [nq] 1 if the entry is integrated by the kernel
    std::vector<double> lower, upper;
};

// Picks the kernel kind from the shape of the problem; force_generic skips the specialisations (used when a
// specialised shape has no compiled instantiation).  Throws std::runtime_error on invalid input.
// constraints: ik::FrameConstraint entries (reference ik/ik/frame.hpp:325-449) as ikgpu_task records of which frame, reference
// and type are read; any constraint sends the problem to the generic kernel.
ProblemHost analyse_problem(const Model &m, const ikgpu_task *tasks, int ntasks, bool force_generic = false,
                            const ikgpu_task *constraints = nullptr, int nconstraints = 0);

// ikdev::ChainDesc<nj> / ikdev::TreeDesc<na, nb> as the flat array of doubles the kernels stage into LDS.
std::vector<double> chain_desc_table(const ProblemHost &ph);
std::vector<double> tree_desc_table(const ProblemHost &ph);
// The per-problem scalars of ikdev::ChainKernelArgs<nj>.
void fill_chain_args(const ProblemHost &ph, double *ref_pl12, int *qidx, int *vidx, int *nq, int *nv, int *priority,
                     int *idmask, int *unit_weights);
// Bit j set when placement j of the chain has an exactly-identity rotation; bit nj for the frame placement.
int chain_identity_mask(const ChainHost &c);
bool task_has_unit_weights(const ikgpu_task &t);
// The per-problem scalars of ikdev::TreeKernelArgs<na, nb>.
struct TreeArgsHost {
    int qidx[2][kMaxChain], vidx[2][kMaxChain];
    int tslot[3], trow[3], tdim[3], trow0[3];
    int prio[3];
    int hasP, nch;
    int idmask[2], idmaskP, unit[3];
    int ref_base[2];                                        // chain c's target is given in a frame on the floating base
    int align_chain, align_axis, align_slot, align_prio;    // an AlignAxisTask row on a chain's task frame (-1: none)
    double align_w;
    // PostureTask rows (device/tree_solver.hpp: TreeParams::post_*)
    int fixed_base;
    int cons_on, cons_type;
    int post_on, post_prio, post_n;
    int post_q[kMaxPostureOut], post_slot[kMaxPostureOut];
    double post_w[kMaxPostureOut], post_m[kMaxPostureOut];
    int postc_slot[2][kMaxChain];
    double postc_w[2][kMaxChain], postc_m[2][kMaxChain];
};
TreeArgsHost tree_args(const ProblemHost &ph);

}  // namespace ikgpu
