/*
 * ikgpu.h -- C ABI of the MI355X batched damped-least-squares IK path.
 *
 * This is the drop-in boundary for ONE path of dazzmo/ik: the iterative loop ik::dls()
 * (reference ik/ik/dls.cpp:5-78) together with what it calls per iteration
 * (ik/ik/data.cpp:25-58, ik/ik/frame.hpp:37-62,152-182, ik/ik/common.hpp:53-56,
 * ik/ik/visitor.hpp:15-21).  The reference has no FFI of its own -- its boundary is the C++
 * free function
 *
 *     vector_t dls(InverseKinematicsProblem&, const vector_t& q0, dls_data&,
 *                  const inverse_kinematics_visitor&, const dls_parameters&);   // ik/ik/dls.hpp:111-114
 *
 * so each entry point below names the reference interface it stands in for.  The C++ mirror of
 * the reference classes (ik_amd/csrc/host/ik/ *.hpp) and the Python ctypes binding (ik_amd/capi.py)
 * are thin layers over exactly these symbols.  Plain pointers and sizes only; no C++ or torch
 * types cross this boundary; nothing here ever throws.  There is NO CPU fallback: every solve
 * entry point runs hand-written gfx950 kernels or returns an error.
 *
 * Conventions
 *   - SE(3) values are 12 doubles: rotation row-major (9) then translation (3).
 *   - Configuration vectors follow Pinocchio: revolute/prismatic joints one entry; a free-flyer
 *     root is (x y z qx qy qz qw) in q and (v_lin, omega) body-frame in the tangent space.
 *   - Batch layouts: IKGPU_SOA is component-major  [component][B]  (device-native, coalesced);
 *     IKGPU_AOS is problem-major [B][component] (what a column-major nq x B Eigen matrix is).
 *     targets: SOA [ntasks][12][B], AOS [B][ntasks][12].
 *   - All functions return IKGPU_OK (0) or an ikgpu_status error; ikgpu_last_error() gives the
 *     message of the calling thread's last failure.
 */
#ifndef IKGPU_H
#define IKGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IKGPU_ABI_VERSION 2   /* 2: ikgpu_dls_params grew the derived-visitor fields */

typedef enum {
    IKGPU_OK = 0,
    IKGPU_ERR_INVALID = 1,     /* bad argument (null pointer, size mismatch, unknown id) */
    IKGPU_ERR_PARSE = 2,       /* URDF text could not be parsed */
    IKGPU_ERR_UNSUPPORTED = 3, /* model / task set has no device kernel (stated in last_error) */
    IKGPU_ERR_DEVICE = 4       /* HIP runtime failure, or no gfx950 device */
} ikgpu_status;

/* IKGPU_JOINT_REVOLUTE_UNBOUNDED: what Pinocchio builds for a URDF "continuous" joint (JointModelRevoluteUnbounded*): nq = 2 -- the
 * configuration holds (cos, sin) of the angle --, nv = 1; its position limits are -1.01 / +1.01 on both entries, and integration rotates
 * the pair by the step and renormalises it to first order (scale (3 - |.|^2) / 2).  Models with such a joint run on the generic kernel. */
typedef enum { IKGPU_JOINT_UNIVERSE = 0, IKGPU_JOINT_REVOLUTE = 1, IKGPU_JOINT_PRISMATIC = 2, IKGPU_JOINT_FREEFLYER = 3,
               IKGPU_JOINT_REVOLUTE_UNBOUNDED = 4 } ikgpu_joint_type;

/* ik::KinematicType, reference ik/ik/frame.hpp:20 (same order). */
typedef enum {
    IKGPU_POSITION = 0, IKGPU_ORIENTATION = 1, IKGPU_FULL = 2,
    /* ik::AlignAxisTask with AlignAxisType::AxisX / AxisY / AxisZ (reference ik/ik/frame.hpp:202-319): one row,
     * e = 1 - axis . target/|target|.  Its target direction (frame.hpp:307) rides in the translation part
     * (doubles 9..11) of the task's 12-double target slot; the rotation part is ignored. */
    IKGPU_ALIGN_AXIS_X = 3, IKGPU_ALIGN_AXIS_Y = 4, IKGPU_ALIGN_AXIS_Z = 5,
    /* ONE row of ik::PostureTask (reference ik/ik/posture.hpp:17-85; a task over nj joints is nj consecutive rows):
     * e = (q[reference] - target) * mask, J = unit row at tangent column `frame` (the reference does not apply the mask
     * to J), both times the weight.  Here `frame` is the tangent index, `reference` the index in q, weight[0] the
     * Task::weighting() entry, weight[1] the mask entry; the target value rides in double 9 of the row's 12-double slot. */
    IKGPU_POSTURE_ROW = 6,
    /* ik::CentreOfMassTask (reference ik/ik/centre_of_mass.hpp:14-62; data.cpp:31-34): three rows,
     * e = oMr^-1 * com(q) - target, J = R(oMr)^T Jcom (pinocchio::jacobianCenterOfMass).  `frame` is not read, `reference`
     * is the reference frame; the target point rides in doubles 9..11 of the task's 12-double slot.  Needs joint masses. */
    IKGPU_CENTRE_OF_MASS = 7
} ikgpu_kinematic_type;

typedef enum { IKGPU_SOA = 0, IKGPU_AOS = 1 } ikgpu_layout;
/* OR-ed into the `layout` argument of the HOST-pointer solve entry points: the targets array holds 7 doubles per task -- translation
 * (x y z) then quaternion (qx qy qz qw), the order of a free-flyer's configuration -- instead of 12; it is expanded on the device
 * (ikgpu_targets_from_pose7), so a target costs 56 instead of 96 bytes over PCIe. */
#define IKGPU_TARGETS_POSE7 0x100

typedef enum { IKGPU_ROOT_FIXED = 0, IKGPU_ROOT_FREEFLYER = 1 } ikgpu_root_joint;

/* Flat kinematic model (what pinocchio::Model holds that the path reads: ik/ik/common.hpp:16).
 * Joint 0 is "universe".  As an input every pointer is caller-owned; as an output of
 * ikgpu_model_get_flat the pointers alias the model and live as long as it does. */
typedef struct {
    int32_t njoints, nq, nv, nframes;
    const int32_t *joint_type;      /* [njoints] ikgpu_joint_type */
    const int32_t *joint_parent;    /* [njoints] */
    const int32_t *joint_idx_q;     /* [njoints] */
    const int32_t *joint_idx_v;     /* [njoints] */
    const double *joint_placement;  /* [njoints][12] parent joint frame -> joint frame */
    const double *joint_axis;       /* [njoints][3]  unit axis (revolute / prismatic) */
    const double *lower, *upper;    /* [nq] position limits (model.lowerPositionLimit / upper) */
    const int32_t *frame_parent;    /* [nframes] parent joint */
    const double *frame_placement;  /* [nframes][12] */
    const char *const *joint_names; /* [njoints] */
    const char *const *frame_names; /* [nframes] */
    /* what pinocchio::centerOfMass reads of model.inertias[j]: mass and lever (centre of mass in the joint frame) of the
     * bodies attached to each joint.  As an input both may be NULL (no masses: a centre-of-mass task is then refused). */
    const double *joint_mass;       /* [njoints] */
    const double *joint_com;        /* [njoints][3] */
} ikgpu_flat_model;

/* One ik::FrameTask (reference ik/ik/frame.hpp:78-200) as the device sees it: frame and
 * reference-frame ids (model.getFrameId), kinematic type, priority level
 * (InverseKinematicsProblem::add_frame_task's third argument, ik/ik/problem.hpp:55-66) and the
 * Task::weighting() vector (ik/ik/task.hpp:40; first `dimension` entries used, default ones).
 * The task's `target` (frame.hpp:189) is per-problem data and is passed to the solve call. */
typedef struct {
    int32_t frame, reference, type, priority;
    double weight[6];
} ikgpu_task;

/* ik::dls_parameters (reference ik/ik/dls.hpp:24-28, ik/ik/common.hpp:59-66) plus the stop rule
 * of inverse_kinematics_visitor::should_stop (ik/ik/visitor.hpp:15-21) as a number: a problem
 * stops, *before* stepping, when ||e[0]||^2 < stop_sq_tol (1e-4 in the reference).  A negative
 * value means "a visitor that never stops": exactly max_iterations steps are taken.
 * max_time and random_restart are unused by the reference loop and have no counterpart. */
#define IKGPU_MAX_VISITOR_LEVELS 8
typedef struct {
    int32_t max_iterations; /* default 100 */
    double damping;         /* default 1e-2; damping^2 is added to the Gram diagonal (dls.cpp:41) */
    double step_length;     /* default 1.0 */
    double stop_sq_tol;     /* default 1e-4; < 0 never stops */
    /* The rest of what a class derived from inverse_kinematics_visitor can decide on: should_stop(ik, e, dq) is virtual and is
     * handed EVERY level's error and the step (ik/ik/visitor.hpp:15-21, called at ik/ik/dls.cpp:61).  A C++ visitor cannot run
     * inside a kernel; this closed family covers what such a visitor realistically tests.  Set by ikgpu_dls_params_default to the
     * reference's own visitor (both switched off; so does zero-initialising the three members).  A solve with either switched on runs on the problem's generic lane program. */
    double dq_sq_tol;       /* > 0: ALSO stop when ||dq||^2 < dq_sq_tol (the step about to be taken is negligible); <= 0 (default 0): off */
    int32_t num_level_tols; /* 0 (default): the error test is ||e[0]||^2 < stop_sq_tol.  n > 0: it is ||e[l]||^2 < level_sq_tol[l] for
                             * every priority level l < n (stop_sq_tol is then not read) */
    double level_sq_tol[IKGPU_MAX_VISITOR_LEVELS];
} ikgpu_dls_params;

typedef struct ikgpu_model ikgpu_model;     /* immutable host-side kinematic model */
typedef struct ikgpu_problem ikgpu_problem; /* model + task table flattened onto one device */

int ikgpu_abi_version(void);
const char *ikgpu_last_error(void);
void ikgpu_dls_params_default(ikgpu_dls_params *p);

/* ---- model loading: stands in for pinocchio::urdf::buildModelFromXML as called at reference
 * ik_ros/src/cassie.cpp:34-35 and ik_ros/src/rviz_model_loader.cpp:24-27.  Reproduces Pinocchio's
 * conventions (joint order, idx_q/idx_v, fixed joints folded into placements, frame table). */
int ikgpu_model_from_urdf(const char *xml, size_t len, int root_joint /* ikgpu_root_joint */, ikgpu_model **out);
/* From arrays (the hook for a caller that already holds a pinocchio::Model). Deep-copies. */
int ikgpu_model_create(const ikgpu_flat_model *flat, ikgpu_model **out);
void ikgpu_model_destroy(ikgpu_model *m);
int ikgpu_model_get_flat(const ikgpu_model *m, ikgpu_flat_model *out);
/* model.getFrameId(name) (reference ik/ik/common.hpp:50): returns nframes when absent. */
int32_t ikgpu_model_frame_id(const ikgpu_model *m, const char *name);
int32_t ikgpu_model_joint_id(const ikgpu_model *m, const char *name);

/* ---- problem: stands in for InverseKinematicsProblem + dls_data construction
 * (reference ik/ik/problem.hpp:17-22, ik/ik/dls.hpp:36-52, ik/ik/data.cpp:8-23).  Analyses the
 * support of every task, picks the kernel specialisation and uploads the constant tables.
 * device: HIP device ordinal. */
int ikgpu_problem_create(const ikgpu_model *m, const ikgpu_task *tasks, int32_t ntasks, int32_t device, ikgpu_problem **out);
void ikgpu_problem_destroy(ikgpu_problem *p);
/* Host-only dry run of the analysis ikgpu_problem_create performs: writes the name of the kernel
 * specialisation that would run (NUL-terminated, truncated to cap) or fails with
 * IKGPU_ERR_UNSUPPORTED / IKGPU_ERR_INVALID and the reason in ikgpu_last_error(). Touches no device. */
int ikgpu_problem_plan(const ikgpu_model *m, const ikgpu_task *tasks, int32_t ntasks, char *out, size_t cap);
/* The same with ik::FrameConstraint entries (reference ik/ik/frame.hpp:325-449; problem.hpp:68-77): each is an ikgpu_task
 * record of which frame, reference and type (Position / Orientation / Full) are read.  ik::dls keeps its step in the null
 * space of the stacked constraint Jacobian, dq = -N J^T (JJ^T + damping^2 I)^-1 e with N = I - pinv(Jc) Jc (reference
 * ik/ik/dls.cpp:26-34,43-53); ik::pik does not read constraints (reference ik/ik/pik.cpp).  A problem with constraints
 * runs on the generic kernel. */
int ikgpu_problem_create_constrained(const ikgpu_model *m, const ikgpu_task *tasks, int32_t ntasks, const ikgpu_task *constraints,
                                     int32_t nconstraints, int32_t device, ikgpu_problem **out);
int ikgpu_problem_plan_constrained(const ikgpu_model *m, const ikgpu_task *tasks, int32_t ntasks, const ikgpu_task *constraints,
                                   int32_t nconstraints, char *out, size_t cap);
/* Host-only, touches no device: compiles ahead of time whatever ikgpu_problem_create would compile at run time for this problem
 * -- today the structure-specialised chain kernel (`dls_chain<..,hot-rtc>`) of a fixed-base chain whose placement structure has no
 * pre-built instantiation, through hipRTC -- and leaves the code object in the on-disk cache ($IKGPU_CACHE_DIR, else
 * $XDG_CACHE_HOME/ikgpu, else ~/.cache/ikgpu), so that the first ikgpu_problem_create on the robot does not pay the few seconds
 * of compilation.  There is no counterpart in the reference (its kernels are the CPU's); it belongs to the cold path that stands
 * in for InverseKinematicsProblem + dls_data construction (reference ik/ik/problem.hpp:17-22, ik/ik/dls.hpp:36-52).
 * Returns IKGPU_OK when there was nothing to compile or the compilation succeeded (the name of the kernel that will run is written
 * to `out` as by ikgpu_problem_plan), IKGPU_ERR_UNSUPPORTED when hipRTC is unavailable or the compilation failed (the problem then
 * runs on the general build; the compiler's log is in ikgpu_last_error()). */
int ikgpu_problem_precompile(const ikgpu_model *m, const ikgpu_task *tasks, int32_t ntasks, const ikgpu_task *constraints,
                             int32_t nconstraints, char *out, size_t cap);
/* The run-time compiler never runs inside the caller's process (a compiler can abort(); ikgpu_problem_create must not): a cache
 * miss in ikgpu_problem_create / ikgpu_problem_precompile spawns the program `ikgpu_precompile` that is installed next to
 * libikgpu.so (tools/ikgpu_precompile.cpp; $IKGPU_PRECOMPILE_EXE overrides the path) as a child process with the request file
 * below, waits for it (IKGPU_RTC_TIMEOUT_S, default 600), and loads what it left in the cache.  A worker that fails, crashes or
 * hangs costs the caller nothing but the specialised build: the problem is created on its general kernel.  On a cache hit no
 * compiler and no child process run at all -- deployments run `ikgpu_precompile --urdf ...` once (or ikgpu_problem_precompile from
 * a process of their own) and the control process never meets a compiler.
 * ikgpu_rtc_worker_compile is that child's entry point: it reads one request written by the library (format private to the
 * library), compiles it in the calling process and stores the code object in the cache; 0 on success.  It initialises no
 * device.  Nothing else should call it.  No counterpart in the reference (its kernels are compiled with the library). */
int ikgpu_rtc_worker_compile(const char *request_path);
int32_t ikgpu_problem_rows(const ikgpu_problem *p);        /* M = sum of task dimensions */
const char *ikgpu_problem_kernel(const ikgpu_problem *p);  /* name of the chosen specialisation */
/* Which entries of q a solve can move: support[i] = 1 when q[i] is integrated by the kernel (a joint in the support of some
 * task row, or the floating base), 0 when the loop only ever clips it to its limits (reference ik/ik/dls.cpp:67-71 steps the
 * whole q with dq = 0 there, then ik/ik/common.hpp:53-56 clips it).  A consumer that already holds q0 needs only the
 * support entries of the result -- what the compact multi-GPU gather ships.  support: HOST pointer to nq bytes. */
int ikgpu_problem_support(const ikgpu_problem *p, uint8_t *support);

/* ---- the hot path: B independent calls of ik::dls() (reference ik/ik/dls.cpp:5-78) in lockstep.
 * All array arguments are DEVICE pointers on the problem's device:
 *   q0      [nq x B]            initial configurations
 *   targets [ntasks x 12 x B]   FrameTask::target per task, w.r.t. the task's reference frame
 *   q_out   [nq x B]            returned configuration (dls.cpp:63 / :77)
 *   success [B] (uint8, may be NULL)  data.success (dls.cpp:62 / :76)
 *   iters   [B] (int32, may be NULL)  iteration index at which should_stop fired, else max_iterations
 * stream: a hipStream_t (NULL = default stream).  Asynchronous; re-entrant per stream. */
int ikgpu_dls_solve_batch(const ikgpu_problem *p, int64_t B, const double *q0, const double *targets,
                          const ikgpu_dls_params *params, double *q_out, uint8_t *success, int32_t *iters,
                          int layout /* ikgpu_layout */, void *stream);

/* FrameTask::target (an SE(3), reference ik/ik/frame.hpp:189) given as 7 doubles -- translation (x y z), quaternion (qx qy qz qw;
 * converted as Eigen's toRotationMatrix does, i.e. as the free-flyer's configuration is read) -- expanded into the 12-double slots
 * the solve entry points take.  pose7: [ntasks x 7 x B] (IKGPU_SOA) or [B x ntasks x 7] (IKGPU_AOS); targets12 likewise with 12.
 * DEVICE pointers; asynchronous on `stream`.  Slots of tasks that read a vector only (alignment direction, centre-of-mass point,
 * posture value: doubles 9..11) take it from the translation part. */
int ikgpu_targets_from_pose7(int64_t B, int32_t ntasks, const double *pose7, double *targets12, int layout /* ikgpu_layout */, void *stream);

/* Same with HOST pointers: copies in, solves on the device, copies out, synchronises. This is what
 * the single-problem ik::dls() shim calls with B = 1.  layout may carry IKGPU_TARGETS_POSE7.  Batches above 1 MiB of buffers
 * are pipelined in chunks (H2D of chunk k + 1, solve of chunk k, D2H of chunk k - 1 overlap; pinned caller buffers make every
 * copy asynchronous), through an arena the problem keeps: no allocation and no device-wide synchronisation per call. */
int ikgpu_dls_solve_batch_host(const ikgpu_problem *p, int64_t B, const double *q0, const double *targets,
                               const ikgpu_dls_params *params, double *q_out, uint8_t *success, int32_t *iters,
                               int layout);

/* ---- the hot path on several GPUs of one node, from ONE process (the reference's caller is a single C++ process,
 * ik_ros/src/cassie.cpp:95-112; it has no multi-device code of its own).  Problems are independent and the model + task table are
 * replicated, so a batch of `total` problems is split into contiguous shards -- rank r of n owns [lo, hi) as ikgpu_shard_range
 * gives it (sizes differ by at most one) -- each device solves its shard with ikgpu_dls_solve_batch into a packed slot
 *     [ q rows: nq x b float64 | iterations: b int32 | success: b uint8 ]        (ikgpu_shard_slot_layout, b = the shard's size)
 * and ONE RCCL all-gather over xGMI leaves every device with every rank's slot: [n][slot_bytes], slot_bytes = the largest
 * shard's layout rounded up to 16 (ikgpu_shard_slot_bytes); slot r is decoded with rank r's own shard size.  The Python
 * one-process-per-GPU path (ik_amd/distributed.py, torch.distributed) uses the same two layout functions. */
void ikgpu_shard_range(int64_t total, int32_t rank, int32_t nranks, int64_t *lo, int64_t *hi);
size_t ikgpu_shard_slot_layout(int32_t rows, int64_t b, size_t *off_q, size_t *off_iters, size_t *off_success); /* returns bytes used */
size_t ikgpu_shard_slot_bytes(int32_t rows, int64_t total, int32_t nranks);
typedef struct ikgpu_shard_group ikgpu_shard_group; /* one problem handle, one stream and one RCCL communicator per device */
/* devices: ndev distinct HIP device ordinals.  Creates the per-device problem handles (as ikgpu_problem_create_constrained) and
 * the communicators (ncclCommInitAll; librccl is opened at run time -- a group of one device works without it). */
int ikgpu_shard_group_create(const ikgpu_model *m, const ikgpu_task *tasks, int32_t ntasks, const ikgpu_task *constraints,
                             int32_t nconstraints, const int32_t *devices, int32_t ndev, ikgpu_shard_group **out);
void ikgpu_shard_group_destroy(ikgpu_shard_group *g);
int32_t ikgpu_shard_group_size(const ikgpu_shard_group *g);
const ikgpu_problem *ikgpu_shard_group_problem(const ikgpu_shard_group *g, int32_t rank);
int32_t ikgpu_shard_group_uses_rccl(const ikgpu_shard_group *g);
void *ikgpu_shard_group_stream(const ikgpu_shard_group *g, int32_t rank); /* the hipStream_t rank's work is enqueued on */
/* Wall time (microseconds) rank's worker thread spent ENQUEUEING its shard's launch in the last ikgpu_dls_solve_batch_sharded call
 * (the launches of the ranks are issued in parallel, one thread per device; the collective follows in one group call).  < 0: no call
 * yet / bad rank.  A measurement aid for deployments: with kernels of 0.1 ms, serial issue over 8 devices would cost as much as
 * the solve.  No counterpart in the reference (one process, one device-less solver). */
double ikgpu_shard_group_last_issue_us(const ikgpu_shard_group *g, int32_t rank);
/* One step.  (`gathered[r]` is written by work enqueued on rank r's own stream, ikgpu_shard_group_stream(g, r), which does not
 * synchronise with the null stream: anything the caller enqueued elsewhere on these buffers -- a fill, a previous reader -- must have
 * completed, or be ordered against that stream, before the call.)
 * q0[r], targets[r]: DEVICE pointers on device r to rank r's shard, component-major ([nq x b_r], [ntasks x 12 x b_r]);
 * gathered[r]: DEVICE pointer on device r to ndev * ikgpu_shard_slot_bytes(nq, total, ndev) bytes.  Asynchronous: the solve and
 * the collective are enqueued on each rank's own stream (inputs must be complete before the call);
 * ikgpu_shard_group_synchronize waits for all of them. */
int ikgpu_dls_solve_batch_sharded(ikgpu_shard_group *g, int64_t total, const double *const *q0, const double *const *targets,
                                  const ikgpu_dls_params *params, void *const *gathered);
int ikgpu_shard_group_synchronize(ikgpu_shard_group *g);

/* ---- the other solver of the reference: B independent calls of ik::pik(), prioritised IK (reference
 * ik/ik/pik.cpp:31-103, declared ik/ik/pik.hpp:56-59), in lockstep.  Arguments as ikgpu_dls_solve_batch.  Every priority
 * level l of the task table is solved in the null space of the levels before it, with the damped pseudo-inverse of
 * pik.cpp:5-22 (damping factor lambda[l]) and the projector update of pik.cpp:58-61.  Runs on the generic device kernel
 * for every problem shape (`pik_generic<...>`). */
#define IKGPU_MAX_PIK_LEVELS 8
#define IKGPU_MAX_PIK_DA 128
/* ik::pik_parameters (reference ik/ik/pik.hpp:11-16; its `damping` and `max_time` are never read by the loop) + the stop
 * rule as in ikgpu_dls_params + the two members of ik::pik_data a caller sets before the call (ik/ik/pik.hpp:41-44). */
typedef struct {
    int32_t max_iterations;               /* default 100 */
    double step_length;                   /* default 1.0 */
    double stop_sq_tol;                   /* default 1e-4; < 0 never stops */
    int32_t num_levels;                   /* the problem's max_priority_level + 1: at least the tasks' max priority + 1 (a level
                                           * without tasks is a no-op, as in the reference's loop pik.cpp:47), <= IKGPU_MAX_PIK_LEVELS */
    double lambda[IKGPU_MAX_PIK_LEVELS];  /* pik_data::lambda, default 1.0 each (pik.hpp:24) */
    const double *da;                     /* pik_data::da, HOST pointer to nv doubles, or NULL for zero (the default);
                                           * read during the call; nv <= IKGPU_MAX_PIK_DA when not NULL */
} ikgpu_pik_params;
void ikgpu_pik_params_default(ikgpu_pik_params *p, int32_t num_levels);
int ikgpu_pik_solve_batch(const ikgpu_problem *p, int64_t B, const double *q0, const double *targets,
                          const ikgpu_pik_params *params, double *q_out, uint8_t *success, int32_t *iters,
                          int layout /* ikgpu_layout */, void *stream);
int ikgpu_pik_solve_batch_host(const ikgpu_problem *p, int64_t B, const double *q0, const double *targets,
                               const ikgpu_pik_params *params, double *q_out, uint8_t *success, int32_t *iters,
                               int layout);
/* Name of the kernel ikgpu_pik_solve_batch runs with these parameters.  With ONE priority level, lambda[0] > 0, da == NULL
 * and no constraints in the problem, ik::pik's step  -damp_pinv(J, lambda) e  (reference ik/ik/pik.cpp:5-21,47-61) IS the DLS
 * step  -J^T (J J^T + lambda^2 I)^-1 e  (ik/ik/dls.cpp:39-53) and the loop around it is the same, so the problem's DLS
 * kernel (ikgpu_problem_kernel) runs it; otherwise the generic PIK kernel does. */
const char *ikgpu_pik_kernel(const ikgpu_problem *p, const ikgpu_pik_params *params);

/* ---- stage kernels (device pointers), for stage-by-stage parity and for building targets:
 * evaluate_problem_data + stacking (reference ik/ik/data.cpp:25-58, ik/ik/dls.cpp:18-24):
 *   e_out [M x B], J_out [M x nv x B] (row-major M x nv per problem; J_out may be NULL). */
int ikgpu_evaluate_batch(const ikgpu_problem *p, int64_t B, const double *q, const double *targets,
                         double *e_out, double *J_out, int layout, void *stream);
/* framesForwardKinematics restricted to the task frames (reference ik/ik/data.cpp:28-29):
 *   oMf_out [ntasks x 12 x B] world placement of each task's frame. */
int ikgpu_task_frames_fk_batch(const ikgpu_problem *p, int64_t B, const double *q, double *oMf_out,
                               int layout, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* IKGPU_H */
