/*
 * ik_oracle_ext.c -- the SAME restatement as ik_oracle.c, compiled with an extended-precision scalar.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see ik_oracle.h).
 *
 * Why: the 50-step DLS map amplifies rounding on lanes stalled against a joint limit, so the double oracle and the device can
 * both be "right" and still differ by more than the 1e-6 rad bar there.  Which side is nearer the exact trajectory is a number,
 * not an opinion: this file includes ik_oracle.c with `double` re-defined to _Float128 (113-bit significand; -DIKO_EXT_LONG_DOUBLE:
 * x87 long double, 64-bit) -- every statement, branch threshold (DBL_EPSILON-based Taylor switches, rank tests) and operation
 * order of the double oracle is kept, only the arithmetic is wider -- and exports the batch solvers again under their usual names
 * with the usual double ABI (inputs widened exactly, results rounded once at the end).  The tests use it to arbitrate every lane
 * the perturbation probes exclude:  |q_gpu - q_ext| <= 10 max(|q_oracle - q_ext|, 1e-9).
 *
 * Reference path restated: as ik_oracle.c (ik/ik/dls.cpp:5-78, ik/ik/data.cpp:25-58, ik/ik/frame.hpp:37-62,152-182,
 * ik/ik/common.hpp:47-56, ik/ik/visitor.hpp:15-21, ik/ik/pik.cpp:5-103).
 */
#define __STDC_WANT_IEC_60559_TYPES_EXT__ 1
#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <tgmath.h>

#ifdef IKO_EXT_LONG_DOUBLE
typedef long double iko_real;
#define IKO_REAL_PI 3.141592653589793238462643383279502884L
#else
typedef _Float128 iko_real;
#define IKO_REAL_PI 3.141592653589793238462643383279502884f128
#endif

#define IKX_API __attribute__((visibility("default")))

/* the double-ABI records (identical to ik_oracle.h's in the double build) */
typedef struct {
    int njoints, nq, nv, nframes;
    const int *jtype, *parent, *idx_q, *idx_v;
    const double *placement, *axis, *lower, *upper;
    const int *frame_parent;
    const double *frame_placement, *mass, *lever;
} ikx_model_d;
typedef struct { int frame, reference, type, priority; double weight[6]; } ikx_task_d;
typedef struct { int max_iterations; double damping, step_length, stop_sq_tol; } ikx_params_d;
typedef struct { int max_iterations; double step_length, stop_sq_tol; int nlevels; const double *lambda, *da; } ikx_pik_params_d;

/* the wide copy of the whole oracle: its public names get an ikx_ prefix and stay hidden (-fvisibility=hidden) */
#define iko_log6 ikx_log6
#define iko_Jlog6 ikx_Jlog6
#define iko_exp6 ikx_exp6
#define iko_fk ikx_fk
#define iko_task_rows ikx_task_rows
#define iko_evaluate ikx_evaluate
#define iko_integrate ikx_integrate
#define iko_constraint_jacobian ikx_constraint_jacobian
#define iko_dls_constrained ikx_dls_constrained
#define iko_dls ikx_dls
#define iko_dls_batch ikx_dls_batch
#define iko_dls_batch_constrained ikx_dls_batch_constrained
#define iko_damp_pseudoinverse ikx_damp_pseudoinverse
#define iko_rowspace_projector ikx_rowspace_projector
#define iko_pik ikx_pik
#define iko_pik_batch ikx_pik_batch
#define iko_fk_batch ikx_fk_batch
#undef M_PI
#define M_PI IKO_REAL_PI
#define double iko_real
#include "ik_oracle.c"
#undef double
#undef iko_task_rows
#undef iko_dls_batch
#undef iko_dls_batch_constrained
#undef iko_pik_batch

static iko_real *widen(const double *src, long n) {
    iko_real *dst = (iko_real *)malloc(sizeof(iko_real) * (size_t)(n > 0 ? n : 1));
    for (long i = 0; i < n; ++i) dst[i] = (iko_real)src[i];   /* exact */
    return dst;
}

typedef struct {
    iko_model m;
    iko_real *placement, *axis, *lower, *upper, *frame_placement, *mass, *lever;
    iko_task *tasks, *cons;
} wide_problem;

static iko_task *widen_tasks(const ikx_task_d *t, int n) {
    iko_task *w = (iko_task *)calloc((size_t)(n > 0 ? n : 1), sizeof(iko_task));
    for (int i = 0; i < n; ++i) {
        w[i].frame = t[i].frame; w[i].reference = t[i].reference; w[i].type = t[i].type; w[i].priority = t[i].priority;
        for (int k = 0; k < 6; ++k) w[i].weight[k] = (iko_real)t[i].weight[k];
    }
    return w;
}

static void wide_init(wide_problem *w, const ikx_model_d *m, const ikx_task_d *tasks, int ntasks, const ikx_task_d *cons, int ncons) {
    memset(w, 0, sizeof *w);
    w->placement = widen(m->placement, 12L * m->njoints);
    w->axis = widen(m->axis, 3L * m->njoints);
    w->lower = widen(m->lower, m->nq);
    w->upper = widen(m->upper, m->nq);
    w->frame_placement = widen(m->frame_placement, 12L * m->nframes);
    w->mass = m->mass ? widen(m->mass, m->njoints) : NULL;
    w->lever = m->lever ? widen(m->lever, 3L * m->njoints) : NULL;
    w->m.njoints = m->njoints; w->m.nq = m->nq; w->m.nv = m->nv; w->m.nframes = m->nframes;
    w->m.jtype = m->jtype; w->m.parent = m->parent; w->m.idx_q = m->idx_q; w->m.idx_v = m->idx_v;
    w->m.placement = w->placement; w->m.axis = w->axis; w->m.lower = w->lower; w->m.upper = w->upper;
    w->m.frame_parent = m->frame_parent; w->m.frame_placement = w->frame_placement; w->m.mass = w->mass; w->m.lever = w->lever;
    w->tasks = widen_tasks(tasks, ntasks);
    w->cons = widen_tasks(cons, ncons);
}

static void wide_free(wide_problem *w) {
    free(w->placement); free(w->axis); free(w->lower); free(w->upper); free(w->frame_placement); free(w->mass); free(w->lever);
    free(w->tasks); free(w->cons);
}

static void narrow(const iko_real *src, double *dst, long n) {
    for (long i = 0; i < n; ++i) dst[i] = (double)src[i];   /* one rounding */
}

/* significand bits of the scalar this build computes in */
IKX_API int iko_ext_bits(void) {
#ifdef IKO_EXT_LONG_DOUBLE
    return LDBL_MANT_DIG;
#else
    return 113;
#endif
}

IKX_API int iko_task_rows(const ikx_task_d *tasks, int ntasks) {
    iko_task *w = widen_tasks(tasks, ntasks);
    const int M = ikx_task_rows(w, ntasks);
    free(w);
    return M;
}

IKX_API int iko_dls_batch_constrained(const ikx_model_d *m, const ikx_task_d *tasks, int ntasks, const ikx_task_d *cons, int ncons, long B,
                                      const double *targets, const double *q0, const ikx_params_d *p, double *q_out,
                                      unsigned char *success, int *iters, int nthreads) {
    wide_problem w;
    wide_init(&w, m, tasks, ntasks, cons, ncons);
    iko_real *tg = widen(targets, B * ntasks * 12), *q = widen(q0, B * m->nq);
    iko_real *out = (iko_real *)malloc(sizeof(iko_real) * (size_t)(B * m->nq > 0 ? B * m->nq : 1));
    iko_params wp;
    wp.max_iterations = p->max_iterations; wp.damping = p->damping; wp.step_length = p->step_length; wp.stop_sq_tol = p->stop_sq_tol;
    const int rc = ncons > 0 ? ikx_dls_batch_constrained(&w.m, w.tasks, ntasks, w.cons, ncons, B, tg, q, &wp, out, success, iters, nthreads)
                             : ikx_dls_batch(&w.m, w.tasks, ntasks, B, tg, q, &wp, out, success, iters, nthreads);
    narrow(out, q_out, B * m->nq);
    free(tg); free(q); free(out);
    wide_free(&w);
    return rc;
}

IKX_API int iko_dls_batch(const ikx_model_d *m, const ikx_task_d *tasks, int ntasks, long B, const double *targets, const double *q0,
                          const ikx_params_d *p, double *q_out, unsigned char *success, int *iters, int nthreads) {
    return iko_dls_batch_constrained(m, tasks, ntasks, NULL, 0, B, targets, q0, p, q_out, success, iters, nthreads);
}

IKX_API int iko_pik_batch(const ikx_model_d *m, const ikx_task_d *tasks, int ntasks, long B, const double *targets, const double *q0,
                          const ikx_pik_params_d *p, double *q_out, unsigned char *success, int *iters, int nthreads) {
    wide_problem w;
    wide_init(&w, m, tasks, ntasks, NULL, 0);
    iko_real *tg = widen(targets, B * ntasks * 12), *q = widen(q0, B * m->nq);
    iko_real *out = (iko_real *)malloc(sizeof(iko_real) * (size_t)(B * m->nq > 0 ? B * m->nq : 1));
    iko_real *lam = widen(p->lambda, p->nlevels), *da = p->da ? widen(p->da, m->nv) : NULL;
    iko_pik_params wp;
    wp.max_iterations = p->max_iterations; wp.step_length = p->step_length; wp.stop_sq_tol = p->stop_sq_tol;
    wp.nlevels = p->nlevels; wp.lambda = lam; wp.da = da;
    const int rc = ikx_pik_batch(&w.m, w.tasks, ntasks, B, tg, q, &wp, out, success, iters, nthreads);
    narrow(out, q_out, B * m->nq);
    free(tg); free(q); free(out); free(lam); free(da);
    wide_free(&w);
    return rc;
}
