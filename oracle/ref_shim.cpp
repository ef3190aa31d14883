/*
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * Thin C shim around the *compiled reference* (ucb-bar/Accelerated-TinyMPC,
 * src/tinympc/admm.cpp) so that tests can drive the real Eigen solver through
 * ctypes with flat arrays.  Nothing of the reference is copied: its translation
 * unit is #included from where it lies (REF_ROOT_ADMM, set by oracle/Makefile)
 * and the result is written to oracle/_ref/ (git-ignored).
 *
 * The reference fixes scalar type and dimensions at compile time in
 * src/tinympc/glob_opts.hpp:3-9, and types.hpp:4 includes that header with
 * quotes, so -D overrides cannot reach it.  We include glob_opts.hpp FIRST (its
 * `#pragma once` then makes the later include a no-op), choosing the scalar by
 * temporarily mapping the token `double`, and then re-#define the dimension
 * macros — the same knobs the reference's own codegen rewrites
 * (src/tinympc/codegen.cpp:131-160).
 */
#if defined(REF_SCALAR_f32)
#define double float
#include REF_ROOT_GLOB
#undef double
#define REF_IS_DOUBLE 0
#elif defined(REF_SCALAR_f64)
#include REF_ROOT_GLOB
#define REF_IS_DOUBLE 1
#else
#error "define REF_SCALAR_f32 or REF_SCALAR_f64"
#endif

#undef NSTATES
#undef NINPUTS
#undef NHORIZON
#define NSTATES REF_NX
#define NINPUTS REF_NU
#define NHORIZON REF_N

#include REF_ROOT_ADMM /* defines tiny_solve() and the six step functions over Eigen structs */

#include <cstring>

namespace
{
TinyCache g_cache;
TinyWorkspace g_work;
TinySettings g_settings;
TinySolver g_solver{&g_settings, &g_cache, &g_work};

template <class M>
void load(M &m, const tinytype *src) { std::memcpy(m.data(), src, sizeof(tinytype) * m.size()); }
template <class M>
void store(const M &m, tinytype *dst) { std::memcpy(dst, m.data(), sizeof(tinytype) * m.size()); }
} // namespace

extern "C"
{
    void ref_dims(int *nx, int *nu, int *N, int *is_double)
    {
        *nx = NSTATES; *nu = NINPUTS; *N = NHORIZON; *is_double = REF_IS_DOUBLE;
    }

    /* All matrices column-major (Eigen's native storage, types.hpp:13-21). */
    void ref_set_problem(tinytype rho, const tinytype *Kinf, const tinytype *Pinf, const tinytype *Quu_inv,
                         const tinytype *AmBKt, const tinytype *Adyn, const tinytype *Bdyn, const tinytype *Q)
    {
        g_cache.rho = rho;
        load(g_cache.Kinf, Kinf); load(g_cache.Pinf, Pinf); load(g_cache.Quu_inv, Quu_inv); load(g_cache.AmBKt, AmBKt);
        g_cache.coeff_d2p.setZero();
        load(g_work.Adyn, Adyn); load(g_work.Bdyn, Bdyn); load(g_work.Q, Q);
        g_work.R.setZero(); g_work.Qu.setZero(); g_work.Uref.setZero();
    }

    void ref_set_settings(tinytype abs_pri_tol, tinytype abs_dua_tol, int max_iter, int check_termination,
                          int en_state_bound, int en_input_bound)
    {
        g_settings.abs_pri_tol = abs_pri_tol; g_settings.abs_dua_tol = abs_dua_tol;
        g_settings.max_iter = max_iter; g_settings.check_termination = check_termination;
        g_settings.en_state_bound = en_state_bound; g_settings.en_input_bound = en_input_bound;
    }

    /*
     * One tiny_solve() on one instance.  `st` = 12 state arrays in the order
     * x,u,q,r,p,d,v,vnew,z,znew,g,y (in/out); `in` = u_min,u_max,x_min,x_max,Xref;
     * res4 = pri_state, pri_input, dua_state, dua_input (in/out); si = status, iter (in/out).
     * Returns tiny_solve's return code (admm.cpp:137,151).
     */
    int ref_solve(tinytype *x, tinytype *u, tinytype *q, tinytype *r, tinytype *p, tinytype *d, tinytype *v,
                  tinytype *vnew, tinytype *z, tinytype *znew, tinytype *g, tinytype *y, const tinytype *u_min,
                  const tinytype *u_max, const tinytype *x_min, const tinytype *x_max, const tinytype *Xref,
                  tinytype *res4, int *si)
    {
        TinyWorkspace &w = g_work;
        load(w.x, x); load(w.u, u); load(w.q, q); load(w.r, r); load(w.p, p); load(w.d, d);
        load(w.v, v); load(w.vnew, vnew); load(w.z, z); load(w.znew, znew); load(w.g, g); load(w.y, y);
        load(w.u_min, u_min); load(w.u_max, u_max); load(w.x_min, x_min); load(w.x_max, x_max); load(w.Xref, Xref);
        w.primal_residual_state = res4[0]; w.primal_residual_input = res4[1];
        w.dual_residual_state = res4[2]; w.dual_residual_input = res4[3];
        w.status = si[0]; w.iter = si[1];
        int rc = tiny_solve(&g_solver);
        store(w.x, x); store(w.u, u); store(w.q, q); store(w.r, r); store(w.p, p); store(w.d, d);
        store(w.v, v); store(w.vnew, vnew); store(w.z, z); store(w.znew, znew); store(w.g, g); store(w.y, y);
        res4[0] = w.primal_residual_state; res4[1] = w.primal_residual_input;
        res4[2] = w.dual_residual_state; res4[3] = w.dual_residual_input;
        si[0] = w.status; si[1] = w.iter;
        return rc;
    }

    /*
     * Batched driver over `batch` independent instances in the host-visible layout
     * (B, N, nx)/(B, N-1, nu); strides (in elements) of 0 mean "shared by all instances".
     * Single-threaded, as the reference is.  Returns the number of rc==1 instances.
     */
    int ref_solve_batch(int batch, tinytype *x, tinytype *u, tinytype *q, tinytype *r, tinytype *p, tinytype *d,
                        tinytype *v, tinytype *vnew, tinytype *z, tinytype *znew, tinytype *g, tinytype *y,
                        const tinytype *u_min, const tinytype *u_max, const tinytype *x_min, const tinytype *x_max,
                        const tinytype *Xref, long long bound_stride_x, long long bound_stride_u,
                        long long xref_stride, tinytype *res4, int *status, int *iter)
    {
        const long long sx = (long long)NSTATES * NHORIZON, su = (long long)NINPUTS * (NHORIZON - 1);
        int unsolved = 0;
        for (int b = 0; b < batch; b++)
        {
            int si[2] = {status[b], iter[b]};
            unsolved += ref_solve(x + b * sx, u + b * su, q + b * sx, r + b * su, p + b * sx, d + b * su, v + b * sx,
                                  vnew + b * sx, z + b * su, znew + b * su, g + b * sx, y + b * su,
                                  u_min + b * bound_stride_u, u_max + b * bound_stride_u, x_min + b * bound_stride_x,
                                  x_max + b * bound_stride_x, Xref + b * xref_stride, res4 + 4 * b, si);
            status[b] = si[0]; iter[b] = si[1];
        }
        return unsolved;
    }

    /*
     * The plant step of the reference's closed-loop examples, examples/quadrotor_hovering.cpp:110-111 and
     * examples/quadrotor_tracking.cpp:116-117:
     *     x1 = work.Adyn * x0 + work.Bdyn * work.u.col(0);   x0 = x1;
     * evaluated by Eigen over the same types (tiny_VectorNx, the workspace's Adyn/Bdyn/u members), so that the order of
     * its sums is the compiled reference's, not a restatement.  Uses the problem loaded by ref_set_problem().
     */
    void ref_plant_step(const tinytype *x0_in, const tinytype *u0_in, tinytype *x1_out)
    {
        TinyWorkspace &work = g_work;
        tiny_VectorNx x0, x1;
        std::memcpy(x0.data(), x0_in, sizeof(tinytype) * NSTATES);
        std::memcpy(work.u.data(), u0_in, sizeof(tinytype) * NINPUTS); /* work.u.col(0) */
        x1 = work.Adyn * x0 + work.Bdyn * work.u.col(0);
        x0 = x1;
        std::memcpy(x1_out, x0.data(), sizeof(tinytype) * NSTATES);
    }

    void ref_plant_step_batch(int batch, const tinytype *x0, const tinytype *u0, tinytype *x1)
    {
        for (int b = 0; b < batch; b++) ref_plant_step(x0 + (long long)b * NSTATES, u0 + (long long)b * NINPUTS, x1 + (long long)b * NSTATES);
    }
}
