/*
 * tinympc_wrapper.h — same-name replacement of the reference's generated wrapper library.
 *
 * The reference's code generator emits libtinympcShared.so exporting ten flat-float* functions over ONE process-global
 * solver (src/tinympc/tiny_wrapper.hpp:14-23, tiny_wrapper.cpp:5-176) for callers in Python/Julia/MATLAB.
 * libtinympc_wrapper.so (accelerated-tinympc_amd/lib/) exports the same ten symbols with the same signatures and flat
 * orders, backed by the HIP solver with a batch of one, so such a caller only changes the library it loads.
 *
 * What the reference bakes into the generated tiny_data_workspace.cpp at code-generation time (dimensions, cache,
 * dynamics, settings: codegen.cpp:322-470) is supplied once at run time through tiny_wrapper_setup().
 * The four DATA symbols of the generated library are exported too — `settings`, `cache`, `work`, `tiny_data_solver`
 * (codegen.cpp:470, :513), declared below with this repository's plain-array struct types (tinympc_admm.h) — and, as in the
 * reference, the ten functions work on exactly that global solver: set_x0 writes work.x.col(0), call_tiny_solve runs
 * tiny_solve(&tiny_data_solver), get_u copies work.u out.
 * Differences: functions are no-ops returning silently (like the reference, they return void) if setup has not been
 * called or a HIP error occurred — tiny_wrapper_last_status() reports it; `verbose != 0` prints a one-line summary
 * instead of every element.  Not re-entrant / not thread-safe, exactly like the reference's global tiny_data_solver.
 */
#ifndef TINYMPC_WRAPPER_H
#define TINYMPC_WRAPPER_H
#ifdef __cplusplus
extern "C"
{
#endif

    /* Matrices column-major (Eigen storage).  Returns 0 or a negative TinyBatchError (see tinympc_batch.h). */
    int tiny_wrapper_setup(int nx, int nu, int N, float rho, const float *Kinf, const float *Pinf, const float *Quu_inv,
                           const float *AmBKt, const float *Adyn, const float *Bdyn, const float *Q, float abs_pri_tol,
                           float abs_dua_tol, int max_iter, int check_termination, int en_state_bound, int en_input_bound,
                           int device);
    void tiny_wrapper_teardown(void);
    /* 0 if the last wrapper call succeeded, else the negative error code; *iter / *status = work->iter / work->status */
    int tiny_wrapper_last_status(int *iter, int *status);

    /* the generated library's global solver (tiny_data_workspace.cpp emitted by codegen.cpp:322-470); members valid after
     * tiny_wrapper_setup().  Declared here only when tinympc_admm.h (the struct types) has been included before. */
#ifdef TINYMPC_ADMM_H
    extern TinySettings settings;
    extern TinyCache cache;
    extern TinyWorkspace work;
    extern TinySolver tiny_data_solver;
#endif

    /* tiny_wrapper.hpp:14-23, identical names, argument meaning and flat orders (x0[i]; xref[j*NSTATES+i]; ...) */
    void set_x0(float *x0, int verbose);
    void set_xref(float *xref, int verbose);
    void set_umin(float *umin, int verbose);
    void set_umax(float *umax, int verbose);
    void set_xmin(float *xmin, int verbose);
    void set_xmax(float *xmax, int verbose);
    void reset_dual_variables(int verbose);
    void call_tiny_solve(int verbose);
    void get_x(float *x_soln, int verbose);
    void get_u(float *u_soln, int verbose);

#ifdef __cplusplus
}
#endif
#endif
