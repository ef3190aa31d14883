// wrapper_compat.cpp — libtinympc_wrapper.so: the ten functions of the reference's generated wrapper
// (src/tinympc/tiny_wrapper.cpp:5-176) under their own names, over one global batch-of-one HIP solver.
#include "../../include/tinympc_wrapper.h"
#include "../../include/tinympc_batch.h"

#include <cstdio>

namespace
{
TinyBatch *g_tb = nullptr; // the reference's `tiny_data_solver` (codegen.cpp:470)
int g_last = TINY_BATCH_ENOTREADY;

template <class F>
void call(const char *what, int verbose, F f)
{
    g_last = g_tb ? f() : TINY_BATCH_ENOTREADY;
    if (g_last < 0) std::fprintf(stderr, "tinympc wrapper: %s failed (%d): %s\n", what, g_last, g_tb ? tiny_batch_last_error() : "tiny_wrapper_setup() not called");
    else if (verbose) std::printf("%s finished\n", what);
}
} // namespace

extern "C"
{

int tiny_wrapper_setup(int nx, int nu, int N, float rho, const float *Kinf, const float *Pinf, const float *Quu_inv,
                       const float *AmBKt, const float *Adyn, const float *Bdyn, const float *Q, float abs_pri_tol,
                       float abs_dua_tol, int max_iter, int check_termination, int en_state_bound, int en_input_bound, int device)
{
    tiny_wrapper_teardown();
    int rc = tiny_batch_create(&g_tb, nx, nu, N, 1, device);
    if (rc == 0) rc = tiny_batch_set_cache(g_tb, rho, Kinf, Pinf, Quu_inv, AmBKt);
    if (rc == 0) rc = tiny_batch_set_dynamics(g_tb, Adyn, Bdyn, Q);
    if (rc == 0) rc = tiny_batch_set_settings(g_tb, abs_pri_tol, abs_dua_tol, max_iter, check_termination, en_state_bound, en_input_bound);
    if (rc != 0) tiny_wrapper_teardown();
    return g_last = rc;
}

void tiny_wrapper_teardown(void)
{
    if (g_tb) tiny_batch_destroy(g_tb);
    g_tb = nullptr;
    g_last = TINY_BATCH_ENOTREADY;
}

int tiny_wrapper_last_status(int *iter, int *status)
{
    if (g_tb && (iter || status)) tiny_batch_get_status(g_tb, iter, status, nullptr);
    return g_last < 0 ? g_last : 0;
}

void set_x0(float *x0, int verbose) { call("set_x0", verbose, [&] { return tiny_batch_set_x0(g_tb, x0); }); }
void set_xref(float *xref, int verbose) { call("set_xref", verbose, [&] { return tiny_batch_set_xref(g_tb, xref, 0); }); }
void set_umin(float *umin, int verbose) { call("set_umin", verbose, [&] { return tiny_batch_set_umin(g_tb, umin, 1); }); }
void set_umax(float *umax, int verbose) { call("set_umax", verbose, [&] { return tiny_batch_set_umax(g_tb, umax, 1); }); }
void set_xmin(float *xmin, int verbose) { call("set_xmin", verbose, [&] { return tiny_batch_set_xmin(g_tb, xmin, 1); }); }
void set_xmax(float *xmax, int verbose) { call("set_xmax", verbose, [&] { return tiny_batch_set_xmax(g_tb, xmax, 1); }); }
void reset_dual_variables(int verbose) { call("reset duals", verbose, [&] { return tiny_batch_reset_dual_variables(g_tb); }); }
void call_tiny_solve(int verbose)
{
    // the reference discards tiny_solve's return code here too (tiny_wrapper.cpp:144)
    call("tiny solve", verbose, [&] { int rc = tiny_batch_solve(g_tb); return rc < 0 ? rc : 0; });
}
void get_x(float *x_soln, int verbose) { call("get_x", verbose, [&] { return tiny_batch_get_x(g_tb, x_soln); }); }
void get_u(float *u_soln, int verbose) { call("get_u", verbose, [&] { return tiny_batch_get_u(g_tb, u_soln); }); }

} // extern "C"
