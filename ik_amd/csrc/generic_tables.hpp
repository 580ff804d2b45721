// generic_tables.hpp -- binds the packed host tables of the generic kernel (problem.hpp: GenericHost) to the
// pointer struct its lane program reads (device/generic_solver.hpp: GenericTables), for base pointers that live
// either on the device (kernels.hip) or on the host (the lane emulator under tests/).
#pragma once
#include <algorithm>
#include <cstdlib>
#include <string>
#include "device/coop_solver.hpp"
#include "device/pik_coop.hpp"
#include "device/generic_solver.hpp"
#include "problem.hpp"

namespace ikgpu {

inline ikdev::GenericTables bind_generic_tables(const ProblemHost &ph, const int32_t *ibase, const double *dbase) {
    const GenericHost &g = ph.generic;
    ikdev::GenericTables T{};
    T.njoints = g.njoints; T.nq = ph.nq; T.nv = ph.nv; T.ntasks = ph.ntasks; T.M = ph.rows;
    T.jtype = ibase + g.o_jtype; T.parent = ibase + g.o_parent; T.idx_q = ibase + g.o_idx_q; T.idx_v = ibase + g.o_idx_v;
    T.placement = dbase + g.o_placement; T.axis = dbase + g.o_axis; T.lower = dbase + g.o_lower; T.upper = dbase + g.o_upper;
    T.t_type = ibase + g.o_ttype; T.t_fjoint = ibase + g.o_tfjoint; T.t_rjoint = ibase + g.o_trjoint;
    T.t_row = ibase + g.o_trow; T.t_dim = ibase + g.o_tdim; T.t_prio = ibase + g.o_tprio;
    T.t_fpl = dbase + g.o_tfpl; T.t_rpl = dbase + g.o_trpl; T.t_w = dbase + g.o_tw;
    T.off_q = g.off_q; T.off_oMi = g.off_oMi; T.off_Jw = g.off_Jw; T.off_e = g.off_e; T.off_J = g.off_J;
    T.off_G = g.off_G; T.off_y = g.off_y; T.off_dq = g.off_dq; T.ws_words = g.ws_words;
    T.nlevels = g.nlevels; T.lvl_row0 = ibase + g.o_lvlrow0;
    T.ncons = static_cast<int>(ph.constraints.size()); T.Mc = ph.crows;
    T.c_type = ibase + g.o_ctype; T.c_fjoint = ibase + g.o_cfjoint; T.c_rjoint = ibase + g.o_crjoint;
    T.c_row = ibase + g.o_crow; T.c_dim = ibase + g.o_cdim;
    T.c_fpl = dbase + g.o_cfpl; T.c_rpl = dbase + g.o_crpl; T.off_Jc = g.off_Jc;
    T.has_com = g.has_com; T.j_mass = dbase + g.o_jmass; T.j_lever = dbase + g.o_jlever; T.j_submass = dbase + g.o_jsubmass;
    T.inv_total_mass = g.inv_total_mass; T.off_sf = g.off_sf;
    T.off_P = g.off_P; T.off_Jb = g.off_Jb; T.off_de = g.off_de; T.ws_words_pik = g.ws_words_pik;
    return T;
}

// The cooperative DLS program's layout and index tables (problem.cpp: build_coop).
// for_pik: ik::pik keeps every row in its system, the DLS solver eliminates the posture rows and lays its workspace out for the rest
inline ikdev::CoopLayout bind_coop_layout(const ProblemHost &ph, const int32_t *ibase, bool for_pik = false) {
    const GenericHost &g = ph.generic;
    ikdev::CoopLayout L{};
    L.q = g.c_q; L.tg = g.c_tg; L.A0 = g.c_A0; L.A1 = g.c_A1; L.Jw = g.c_Jw; L.tb = g.c_tb; L.e = g.c_e; L.J = g.c_J; L.G = g.c_G;
    L.dinv = g.c_dinv; L.x = g.c_x; L.dq = g.c_dq; L.sf = g.c_sf; L.cb = g.c_cb; L.Jc = g.c_Jc; L.cnrm = g.c_cnrm; L.words = g.coop_words;
    L.rounds = g.coop_rounds; L.npairs = g.coop_npairs;
    {   // Cholesky-QR basis of the constraint Jacobian: Jc Jc^T (packed lower triangle) + its Mc pivots go where G / dinv / x and
        // the per-constraint placements live (all dead by then), indexed through its own pair tables (cpair_i / cpair_j)
        const int Mc = ph.crows, M = ph.rows, region = g.coop_words - 1 - g.c_G;
        const char *force = std::getenv("IKGPU_PIK_PROJECTOR");   // "dense": the reference-shaped routines everywhere (tests)
        (void)M;
        L.cholqr_c = (Mc > 0 && Mc * (Mc + 1) / 2 + Mc <= region && !(force && std::string(force) == "dense")) ? 1 : 0;
    }
    L.support = ibase + g.o_csupport; L.pair_i = ibase + g.o_cpair_i; L.pair_j = ibase + g.o_cpair_j; L.order = ibase + g.o_cup; L.chain_start = ibase + g.o_cchain; L.lvl_start = ibase + g.o_clvl; L.tb_index = ibase + g.o_ctbindex; L.btask = ibase + g.o_cbtask; L.nblocks = g.coop_nblocks; L.col_joint = ibase + g.o_ccoljoint; L.csupp_f = ibase + g.o_ccsf; L.csupp_r = ibase + g.o_ccsr;
    L.cpair_i = ibase + g.o_ccpair_i; L.cpair_j = ibase + g.o_ccpair_j;
    L.post_elim = for_pik ? 0 : g.coop_post_elim; L.Mf = g.coop_Mf; L.Dd = g.c_Dd;
    L.frow = ibase + g.o_cfrow; L.pstart = ibase + g.o_cpstart; L.ptask = ibase + g.o_cptask;
    L.ntg = g.coop_ntg; L.tg_off = ibase + g.o_ctgoff; L.tg_src = ibase + g.o_ctgsrc;
    L.jrow = ibase + g.o_trow;
    if (L.post_elim) {
        const int Mf = g.coop_Mf;
        L.A1 = g.d_A1; L.Jw = g.d_Jw; L.e = g.d_e; L.cnrm = g.d_e; L.dq = g.d_dq; L.Dd = g.d_Dd; L.sf = g.d_sf;
        L.A0 = L.J = L.Jc = g.d_J; L.tb = L.G = L.cb = g.d_G;
        L.dinv = L.G + (Mf + 1) * (Mf + 2) / 2; L.x = L.dinv + Mf; L.words = g.d_words;
        L.jrow = ibase + g.o_cjrow;
        const int Mc = ph.crows, region = g.d_words - 1 - g.d_G;
        const char *force = std::getenv("IKGPU_PIK_PROJECTOR");
        L.cholqr_c = (Mc > 0 && Mc * (Mc + 1) / 2 + Mc <= region && !(force && std::string(force) == "dense")) ? 1 : 0;
    }
    return L;
}

// ik::pik in the cooperative form: the projector after the DLS workspace; the level's projected Jacobian where the joint
// placements lived, its right-hand side and pivot norms over the joint Jacobian (all dead after coop_evaluate).
inline ikdev::PikCoopLayout bind_pik_coop_layout(const ProblemHost &ph) {
    const GenericHost &g = ph.generic;
    ikdev::PikCoopLayout K{};
    K.P = g.c_P; K.Jb = g.c_A1; K.de = g.c_Jw; K.nrm = g.c_Jw + g.coop_mmax; K.words = g.coop_words_pik;
    // factored projector (device/pik_coop.hpp): the coefficients J_l V^T (ml x R) of every level, and V da (R) when the last
    // level's basis is kept for `da`, must fit behind V's R rows in the nv x nv region; IKGPU_PIK_PROJECTOR=dense keeps P
    const char *force = std::getenv("IKGPU_PIK_PROJECTOR");
    bool ok = !(force && std::string(force) == "dense");
    const int nv = ph.nv;
    int rows_before = 0;
    for (int l = 0; l < g.nlevels && ok; ++l) {
        const int ml = g.ints[g.o_lvlrow0 + l + 1] - g.ints[g.o_lvlrow0 + l];
        const int R = std::min(rows_before, nv);
        if (std::max(ml * R, ml * (ml + 1) / 2 + ml) > (nv - R) * nv) ok = false;   // coefficients, then the Gram copy + pivots
        rows_before += ml;
    }
    const int Rend = std::min(rows_before, nv);
    if (Rend > (nv - Rend) * nv) ok = false;   // V da
    K.factored = ok ? 1 : 0;
    return K;
}

}  // namespace ikgpu
