// wrapper_compat.cpp — libtinympc_wrapper.so: the ten functions of the reference's generated wrapper
// (src/tinympc/tiny_wrapper.cpp:5-176) under their own names AND the four data symbols the generated library exports next to
// them — `settings`, `cache`, `work`, `tiny_data_solver` (src/tinympc/codegen.cpp:470, :513; SURVEY.md section 8(b), `nm -D` row).
//
// Like the reference's wrapper, every function works on the ONE process-global solver `tiny_data_solver`: set_x0 writes
// work.x.col(0), set_xref work.Xref, ... call_tiny_solve runs tiny_solve(&tiny_data_solver) (admm_compat.cpp: the HIP kernels on
// a batch of one), get_x / get_u copy work.x / work.u out.  A caller that pokes tiny_data_solver.work->... between wrapper
// calls therefore sees, and changes, exactly the state the wrapper functions use — as with the generated library.
// The members are this repository's plain column-major arrays (include/tinympc_admm.h), not Eigen matrices: the dimensions
// the reference bakes in at code-generation time arrive at run time through tiny_wrapper_setup().
#include "../../include/tinympc_wrapper.h"
#include "../../include/tinympc_admm.h"
#include "../../include/tinympc_batch.h"

#include <cstdio>
#include <cstring>
#include <vector>

extern "C"
{
// the generated library's data symbols (tiny_data_workspace.cpp as emitted by codegen.cpp:322-470)
TinySettings settings = {};
TinyCache cache = {};
TinyWorkspace work = {};
TinySolver tiny_data_solver = {&settings, &cache, &work};
}

namespace
{
std::vector<float> g_store; // backing storage of every array member, sized by tiny_wrapper_setup()
bool g_ready = false;
int g_last = TINY_BATCH_ENOTREADY;

void done(const char *what, int rc, int verbose)
{
    g_last = rc;
    if (rc < 0) std::fprintf(stderr, "tinympc wrapper: %s failed (%d): %s\n", what, rc, g_ready ? tiny_batch_last_error() : "tiny_wrapper_setup() not called");
    else if (verbose) std::printf("%s finished\n", what);
}

template <class F>
void with_setup(const char *what, int verbose, F f)
{
    if (!g_ready) { done(what, TINY_BATCH_ENOTREADY, verbose); return; }
    done(what, f(), verbose);
}
} // namespace

extern "C"
{

int tiny_wrapper_setup(int nx, int nu, int N, float rho, const float *Kinf, const float *Pinf, const float *Quu_inv,
                       const float *AmBKt, const float *Adyn, const float *Bdyn, const float *Q, float abs_pri_tol,
                       float abs_dua_tol, int max_iter, int check_termination, int en_state_bound, int en_input_bound, int device)
{
    tiny_wrapper_teardown();
    if (nx < 1 || nu < 1 || N < 2 || !Kinf || !Pinf || !Quu_inv || !AmBKt || !Adyn || !Bdyn || !Q) return g_last = TINY_BATCH_EINVAL;
    if (check_termination < 1) return g_last = TINY_BATCH_EINVAL;
    const size_t xs = (size_t)nx * N, us = (size_t)nu * (N - 1);
    // cache: Kinf, Pinf, Quu_inv, AmBKt, coeff_d2p; work: 5 state-type + 1 more (v, vnew, g...) ... laid out below
    const size_t total = (size_t)nu * nx + (size_t)nx * nx + (size_t)nu * nu + (size_t)nx * nx + (size_t)nx * nu // cache
                         + 6 * xs + 6 * us                                                                      // x q p v vnew g | u r d z znew y
                         + nx + nu + (size_t)nx * nx + (size_t)nx * nu                                          // Q R Adyn Bdyn
                         + 2 * us + 2 * xs + xs + us + nu;                                                      // u_min u_max x_min x_max Xref Uref Qu
    g_store.assign(total, 0.f);
    float *p = g_store.data();
    auto take = [&](size_t n) { float *q = p; p += n; return q; };
    cache.rho = rho;
    cache.Kinf = take((size_t)nu * nx); cache.Pinf = take((size_t)nx * nx); cache.Quu_inv = take((size_t)nu * nu);
    cache.AmBKt = take((size_t)nx * nx); cache.coeff_d2p = take((size_t)nx * nu);
    std::memcpy(cache.Kinf, Kinf, sizeof(float) * nu * nx); std::memcpy(cache.Pinf, Pinf, sizeof(float) * nx * nx);
    std::memcpy(cache.Quu_inv, Quu_inv, sizeof(float) * nu * nu); std::memcpy(cache.AmBKt, AmBKt, sizeof(float) * nx * nx);
    settings.abs_pri_tol = abs_pri_tol; settings.abs_dua_tol = abs_dua_tol; settings.max_iter = max_iter;
    settings.check_termination = check_termination; settings.en_state_bound = en_state_bound; settings.en_input_bound = en_input_bound;
    work = TinyWorkspace{};
    work.nx = nx; work.nu = nu; work.N = N;
    work.x = take(xs); work.u = take(us); work.q = take(xs); work.r = take(us); work.p = take(xs); work.d = take(us);
    work.v = take(xs); work.vnew = take(xs); work.z = take(us); work.znew = take(us); work.g = take(xs); work.y = take(us);
    work.Q = take(nx); work.R = take(nu); work.Adyn = take((size_t)nx * nx); work.Bdyn = take((size_t)nx * nu);
    work.u_min = take(us); work.u_max = take(us); work.x_min = take(xs); work.x_max = take(xs); work.Xref = take(xs);
    work.Uref = take(us); work.Qu = take(nu);
    std::memcpy(work.Q, Q, sizeof(float) * nx); std::memcpy(work.Adyn, Adyn, sizeof(float) * nx * nx);
    std::memcpy(work.Bdyn, Bdyn, sizeof(float) * nx * nu);
    work.status = 0; work.iter = 0;
    g_last = tiny_admm_set_device(device);
    g_ready = g_last >= 0;
    return g_last;
}

void tiny_wrapper_teardown(void)
{
    g_ready = false;
    g_last = TINY_BATCH_ENOTREADY;
    cache = TinyCache{};
    work = TinyWorkspace{};
    g_store.clear();
}

int tiny_wrapper_last_status(int *iter, int *status)
{
    if (g_ready)
    {
        if (iter) *iter = work.iter;
        if (status) *status = work.status;
    }
    return g_last < 0 ? g_last : 0;
}

// tiny_wrapper.cpp:25-133: the inputs are copied into the global workspace
void set_x0(float *x0, int verbose)
{
    with_setup("set_x0", verbose, [&] { std::memcpy(work.x, x0, sizeof(float) * work.nx); return 0; }); // work.x.col(0)
}
void set_xref(float *xref, int verbose)
{
    with_setup("set_xref", verbose, [&] { std::memcpy(work.Xref, xref, sizeof(float) * work.nx * work.N); return 0; });
}
void set_umin(float *umin, int verbose)
{
    with_setup("set_umin", verbose, [&] { std::memcpy(work.u_min, umin, sizeof(float) * work.nu * (work.N - 1)); return 0; });
}
void set_umax(float *umax, int verbose)
{
    with_setup("set_umax", verbose, [&] { std::memcpy(work.u_max, umax, sizeof(float) * work.nu * (work.N - 1)); return 0; });
}
void set_xmin(float *xmin, int verbose)
{
    with_setup("set_xmin", verbose, [&] { std::memcpy(work.x_min, xmin, sizeof(float) * work.nx * work.N); return 0; });
}
void set_xmax(float *xmax, int verbose)
{
    with_setup("set_xmax", verbose, [&] { std::memcpy(work.x_max, xmax, sizeof(float) * work.nx * work.N); return 0; });
}
void reset_dual_variables(int verbose) // tiny_wrapper.cpp:135-140: y = 0, g = 0
{
    with_setup("reset duals", verbose, [&] {
        std::memset(work.y, 0, sizeof(float) * work.nu * (work.N - 1));
        std::memset(work.g, 0, sizeof(float) * work.nx * work.N);
        return 0;
    });
}
void call_tiny_solve(int verbose)
{
    // the reference discards tiny_solve's return code here too (tiny_wrapper.cpp:144)
    with_setup("tiny solve", verbose, [&] { const int rc = tiny_solve(&tiny_data_solver); return rc < 0 ? rc : 0; });
}
void get_x(float *x_soln, int verbose)
{
    with_setup("get_x", verbose, [&] { std::memcpy(x_soln, work.x, sizeof(float) * work.nx * work.N); return 0; });
}
void get_u(float *u_soln, int verbose)
{
    with_setup("get_u", verbose, [&] { std::memcpy(u_soln, work.u, sizeof(float) * work.nu * (work.N - 1)); return 0; });
}

} // extern "C"
