/*
 * tinympc_batch64.h — C-ABI of the batched TinyMPC ADMM solver for `typedef double tinytype`.
 *
 * The reference ships with tinytype = double (src/tinympc/glob_opts.hpp:3: NSTATES 12, NINPUTS 4, NHORIZON 10 as checked
 * in); its code generator switches to float for microcontrollers (codegen.cpp:152).  tinympc_batch.h is the float
 * library; this header is the same drop-in boundary — tiny_solve() (src/tinympc/admm.cpp:111-152) over a batch of
 * independent instances of one problem class — in double, for callers that run the reference as shipped.
 *
 * Same conventions as tinympc_batch.h: plain C, host pointers, column-major matrices (types.hpp:13-21), batched arrays
 * in the array-of-reference-instances layout ([B][N][nx] / [B][N-1][nu]), device-resident workspace that persists
 * between solves (the warm start), every function returns 0 or a negative TinyBatchError (tinympc_batch.h),
 * tiny_batch_last_error() holds the message.  Arithmetic: every product and sum separately rounded in the order of the
 * reference's SSE2 Eigen build with 2-double packets; results are BITWISE equal to the compiled reference (all twelve
 * work arrays, the residuals, status, iter).
 *
 * Scope: tiny_solve, the six step functions, a closed-loop step and the wrapper-style accessors.  Problem classes with a compiled
 * instantiation: (nx, nu) = (12, 4), (4, 1), (8, 4), (12, 2), (4, 2), (4, 4), (16, 4), any horizon N (TINY_FOR_EACH_F64DIMS).
 * Two kernels with identical results: sixteen lanes per instance with the state on chip for the whole solve (nx + nu <= 16 and N <= 64: unrolled instantiations for the
 * reference's horizons, a launch-parameter horizon otherwise; the default where it applies) and one thread per instance with the state in HBM (any N).
 */
#ifndef TINYMPC_BATCH64_H
#define TINYMPC_BATCH64_H

#include "tinympc_batch.h"

#ifdef __cplusplus
extern "C"
{
#endif

    typedef struct TinyBatch64 TinyBatch64;

    /* Replaces the caller-owned TinyCache/TinyWorkspace/TinySettings/TinySolver (types.hpp:26-107) with tinytype = double
     * (glob_opts.hpp:3).  The workspace starts all zero (quadrotor_hovering.cpp:49-71). */
    const char *tiny_batch64_last_error(void);
    int tiny_batch64_create(TinyBatch64 **out, int nx, int nu, int N, int batch, int device);
    void tiny_batch64_destroy(TinyBatch64 *tb);

    /* TinyCache (types.hpp:26-34) and the per-problem members of TinyWorkspace (types.hpp:82-85), column-major doubles. */
    int tiny_batch64_set_cache(TinyBatch64 *tb, double rho, const double *Kinf, const double *Pinf, const double *Quu_inv, const double *AmBKt);
    int tiny_batch64_set_dynamics(TinyBatch64 *tb, const double *Adyn, const double *Bdyn, const double *Q);
    /* TinySettings (types.hpp:39-47) */
    int tiny_batch64_set_settings(TinyBatch64 *tb, double abs_pri_tol, double abs_dua_tol, int max_iter, int check_termination,
                                  int en_state_bound, int en_input_bound);

    /* wrapper twins (tiny_wrapper.hpp:14-23), batched, double: `shared` != 0 means one [N][nx] / [N-1][nu] array for the batch */
    int tiny_batch64_set_x0(TinyBatch64 *tb, const double *x0);                 /* [B][nx] -> x.col(0) */
    int tiny_batch64_set_xref(TinyBatch64 *tb, const double *xref, int shared); /* [B or 1][N][nx] */
    int tiny_batch64_set_xmin(TinyBatch64 *tb, const double *v, int shared);
    int tiny_batch64_set_xmax(TinyBatch64 *tb, const double *v, int shared);
    int tiny_batch64_set_umin(TinyBatch64 *tb, const double *v, int shared);
    int tiny_batch64_set_umax(TinyBatch64 *tb, const double *v, int shared);
    int tiny_batch64_reset_dual_variables(TinyBatch64 *tb);                     /* y = 0, g = 0 (tiny_wrapper.cpp:131) */
    /* tiny_solve() for every instance (admm.cpp:111-152): 0 = all converged, 1 = some hit max_iter, < 0 = error */
    int tiny_batch64_solve(TinyBatch64 *tb);

    /* The six functions tiny_solve() is made of (admm.hpp:12-18, admm.cpp:15-109), each over the whole batch: one launch that reads
     * and writes the members the reference function does.  termination_condition evaluates `work->iter % check_termination`
     * with the iter field of the workspace (set_status), writes the four residual fields when that holds and returns the
     * function's bool per instance in converged[batch]. */
    int tiny_batch64_forward_pass(TinyBatch64 *tb);
    int tiny_batch64_update_slack(TinyBatch64 *tb);
    int tiny_batch64_update_dual(TinyBatch64 *tb);
    int tiny_batch64_update_linear_cost(TinyBatch64 *tb);
    int tiny_batch64_backward_pass_grad(TinyBatch64 *tb);
    int tiny_batch64_termination_condition(TinyBatch64 *tb, int *converged);

    /* One step of the examples' closed loop (quadrotor_hovering.cpp:95-111) on the device: y = g = 0, tiny_solve, then the plant
     * update x.col(0) <- Adyn * x.col(0) + Bdyn * u.col(0) in the order Eigen evaluates that expression (bitwise equal to
     * the compiled example).  Returns what tiny_batch64_solve returns.  tiny_batch64_get_first_columns reads x.col(0) (after the
     * plant update: the next x0) and u.col(0) (the control just computed) without moving the whole horizon; either may be NULL. */
    int tiny_batch64_mpc_step(TinyBatch64 *tb);
    int tiny_batch64_get_first_columns(TinyBatch64 *tb, double *x0, double *u0);

    /* Implementation: 0 = automatic (the second where (nx, nu, N) has an instantiation, else the first), 1 = one thread per
     * instance with the state in HBM (any N), 2 = sixteen lanes per instance with the state in registers for the whole solve
     * (instantiated (nx, nu, N): (12,4,10) as shipped, (12,4,20), (12,4,30), (4,1,10), (8,4,9)).  Identical results. */
    int tiny_batch64_select_kernel(TinyBatch64 *tb, int which);
    const char *tiny_batch64_kernel_name(TinyBatch64 *tb);

    /* whole workspace, ids of TinyBatchArray */
    int tiny_batch64_set_array(TinyBatch64 *tb, int id, const double *src);
    int tiny_batch64_get_array(TinyBatch64 *tb, int id, double *dst);
    /* work->iter, work->status and the four residual fields (pri_state, pri_input, dua_state, dua_input); NULL = skip */
    int tiny_batch64_get_status(TinyBatch64 *tb, int *iter, int *status, double *residuals);
    int tiny_batch64_set_status(TinyBatch64 *tb, const int *iter, const int *status, const double *residuals);

#ifdef __cplusplus
}
#endif
#endif
