/*
 * ik_oracle.h -- CPU restatement of the reference IK hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED.  The reference (dazzmo/ik) cannot be built in this environment: its
 * arithmetic lives in Pinocchio and Eigen, which are un-vendored, un-pinned
 * (ik/ik/CMakeLists.txt:1-3 `find_package(... REQUIRED)`, no version, no lock file) and absent
 * here, and its own tests pin no numbers (all TEST bodies in ik/test/{ik,dls,task}.cpp are
 * commented out).  This oracle is therefore a faithful restatement of the reference loop with
 * Pinocchio/Eigen semantics restated from their published algorithms,
 * cross-checked against an independently written numpy twin (oracle/twin.py), analytic
 * known-answer tests and the committed golden vectors (tests/golden/).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * Nothing under ik_amd/ or include/ links, imports or calls it.
 */
#ifndef IK_ORACLE_H
#define IK_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* IKO_JOINT_REVOLUTE_UNBOUNDED: Pinocchio's joint for a URDF "continuous" joint: configuration (cos, sin), nq = 2, nv = 1 */
enum { IKO_JOINT_UNIVERSE = 0, IKO_JOINT_REVOLUTE = 1, IKO_JOINT_PRISMATIC = 2, IKO_JOINT_FREEFLYER = 3, IKO_JOINT_REVOLUTE_UNBOUNDED = 4 };
/* ik::KinematicType, ik/ik/frame.hpp:20 */
enum { IKO_POSITION = 0, IKO_ORIENTATION = 1, IKO_FULL = 2,
       /* ik::AlignAxisTask with AlignAxisType X / Y / Z (ik/ik/frame.hpp:202-319): one row; its target direction
        * is the translation part (doubles 9..11) of the task's 12-double target slot, the rotation part is ignored */
       IKO_ALIGN_X = 3, IKO_ALIGN_Y = 4, IKO_ALIGN_Z = 5,
       /* one row of ik::PostureTask (ik/ik/posture.hpp:17-85): frame = tangent column, reference = index in q,
        * weight[0] = task weight, weight[1] = mask entry, target value in double 9 of the slot */
       IKO_POSTURE_ROW = 6,
       /* ik::CentreOfMassTask (ik/ik/centre_of_mass.hpp:14-62): three rows; reference = reference frame, frame unused,
        * target point in doubles 9..11 of the slot */
       IKO_COM = 7 };

/* Flat kinematic model with Pinocchio's conventions (joint 0 = universe). SE(3) values are 12
 * doubles: rotation row-major (9) then translation (3). */
typedef struct {
    int njoints, nq, nv, nframes;
    const int *jtype, *parent, *idx_q, *idx_v; /* [njoints] */
    const double *placement;                   /* [njoints][12] parent-joint frame -> joint frame */
    const double *axis;                        /* [njoints][3]  */
    const double *lower, *upper;               /* [nq] */
    const int *frame_parent;                   /* [nframes] parent joint */
    const double *frame_placement;             /* [nframes][12] */
    /* model.inertias[j] as pinocchio::centerOfMass reads it: mass and lever (centre of mass in the joint frame) of the
     * bodies attached to joint j; needed by IKO_COM tasks only */
    const double *mass;                        /* [njoints] */
    const double *lever;                       /* [njoints][3] */
} iko_model;

/* ik::FrameTask (ik/ik/frame.hpp:78-200): frame ids, kinematic type, priority level and the
 * Task::weighting() vector (ik/ik/task.hpp:40). */
typedef struct {
    int frame, reference, type, priority;
    double weight[6];
} iko_task;

/* ik::dls_parameters (ik/ik/dls.hpp:24-28) + the stop rule of ik/ik/visitor.hpp:19 as a number:
 * stop when ||e[0]||^2 < stop_sq_tol; negative == a visitor whose should_stop() is always false. */
typedef struct {
    int max_iterations;
    double damping, step_length, stop_sq_tol;
} iko_params;

/* The rest of what inverse_kinematics_visitor::should_stop(ik, e, dq) is handed (ik/ik/visitor.hpp:15-21: every level's error and
 * the step): a closed family of derived visitors.  nlevels == 0: the error test is the reference's own (||e[0]||^2 < stop_sq_tol of
 * iko_params); nlevels > 0: it is ||e[l]||^2 < level_sq_tol[l] for EVERY l < nlevels.  dq_sq_tol >= 0: the solve ALSO stops when
 * ||dq||^2 < dq_sq_tol.  Process-wide setting read by the dls entry points (test infrastructure); NULL restores the reference's visitor. */
typedef struct {
    double dq_sq_tol;
    int nlevels;
    double level_sq_tol[8];
} iko_visitor;
void iko_set_visitor(const iko_visitor *v);

int iko_task_rows(const iko_task *tasks, int ntasks);

/* framesForwardKinematics: oMi[njoints][12], oMf[nframes][12] (either may be NULL). */
void iko_fk(const iko_model *m, const double *q, double *oMi, double *oMf);

/* evaluate_problem_data (ik/ik/data.cpp:25-58) + the stacking of ik/ik/dls.cpp:18-24:
 * et[M], Jt[M][nv] row-major, rows ordered by priority then insertion. targets: [ntasks][12]. */
void iko_evaluate(const iko_model *m, const iko_task *tasks, int ntasks, const double *targets,
                  const double *q, double *et, double *Jt);

/* ik::dls (ik/ik/dls.cpp:5-78) for one problem. Returns 0 on success (of the call, not of the
 * solve). iters = index of the iteration at which should_stop fired, or max_iterations.
 * trace (optional): per iteration [q(nq) | et(M) | dq(nv)] appended, for stage parity tests. */
int iko_dls(const iko_model *m, const iko_task *tasks, int ntasks, const double *targets,
            const double *q0, const iko_params *p, double *q_out, int *success, int *iters,
            double *trace);

/* The same with ik::FrameConstraint entries (ik/ik/frame.hpp:325-449): iko_task records of which frame, reference and type
 * are read.  dq = -N Jt^T (JJ)^-1 et with N = I - pinv(Jc) Jc (ik/ik/dls.cpp:26-34,43-53). */
int iko_dls_constrained(const iko_model *m, const iko_task *tasks, int ntasks, const iko_task *cons, int ncons,
                        const double *targets, const double *q0, const iko_params *p, double *q_out, int *success, int *iters,
                        double *trace);
int iko_dls_batch_constrained(const iko_model *m, const iko_task *tasks, int ntasks, const iko_task *cons, int ncons, long B,
                              const double *targets, const double *q0, const iko_params *p, double *q_out, unsigned char *success,
                              int *iters, int nthreads);
/* Stacked constraint Jacobian Jc [Mc x nv] at q (ik/ik/frame.hpp:413-449). */
void iko_constraint_jacobian(const iko_model *m, const iko_task *cons, int ncons, const double *q, double *Jc);

/* Batch of independent problems, array-of-structures: q0[B][nq], targets[B][ntasks][12],
 * q_out[B][nq]; nthreads worker threads split the batch in contiguous blocks. */
int iko_dls_batch(const iko_model *m, const iko_task *tasks, int ntasks, long B,
                  const double *targets, const double *q0, const iko_params *p, double *q_out,
                  unsigned char *success, int *iters, int nthreads);

/* ik::pik_parameters (ik/ik/pik.hpp:11-16; `damping` there is never read) + the stop rule as above + the two members
 * of ik::pik_data a caller sets (ik/ik/pik.hpp:41,44): lambda[level] (default 1.0 each, pik.hpp:24) and da (default
 * zero; NULL == zero).  nlevels must be at least the tasks' max priority + 1
 * (the problem's declared max_priority_level + 1; a level without tasks is a no-op, pik.cpp:47). */
typedef struct {
    int max_iterations;
    double step_length, stop_sq_tol;
    int nlevels;
    const double *lambda; /* [nlevels] */
    const double *da;     /* [nv] or NULL */
} iko_pik_params;

/* ik::pik (ik/ik/pik.cpp:31-103) for one problem; same outputs and trace format as iko_dls. Returns -1 when
 * nlevels is smaller than the task table needs. */
int iko_pik(const iko_model *m, const iko_task *tasks, int ntasks, const double *targets, const double *q0,
            const iko_pik_params *p, double *q_out, int *success, int *iters, double *trace);
int iko_pik_batch(const iko_model *m, const iko_task *tasks, int ntasks, long B, const double *targets, const double *q0,
                  const iko_pik_params *p, double *q_out, unsigned char *success, int *iters, int nthreads);
/* The two matrix functions of the PIK step, exposed for tests: damp_pseudoinverse (pik.cpp:5-22; res is n x m) and
 * pinv(A) * A as Eigen's completeOrthogonalDecomposition would give it (pik.cpp:59-61; Pr is n x n). A is m x n. */
void iko_damp_pseudoinverse(const double *A, int m, int n, double lambda, double *res);
void iko_rowspace_projector(const double *A, int m, int n, double *Pr);

/* Batch FK of selected frames: q[B][nq] -> out[B][nsel][12]. */
void iko_fk_batch(const iko_model *m, long B, const double *q, const int *frames, int nsel,
                  double *out);

/* Lie-group maps exposed for analytic tests. M is 12 doubles (R row-major, p). */
void iko_log6(const double *M, double *out6);
void iko_Jlog6(const double *M, double *out36);
void iko_exp6(const double *nu6, double *M);
void iko_integrate(const iko_model *m, const double *q, const double *v, double *out);

#ifdef __cplusplus
}
#endif
#endif
