// admm_compat.cpp — the reference's native API (tiny_solve + the step functions of src/tinympc/admm.hpp:10-18) under its
// own names over plain-array structs (include/tinympc_admm.h), for one instance, on top of the batched HIP solver.
// Part of libtinympc_wrapper.so.  Every call: copy the live-in members to a batch-of-one device workspace, launch the
// kernel, copy back what the reference function writes.  No CPU arithmetic happens here.
//
// Compiled twice: as is for `typedef float tinytype` (what the reference's code generator emits, codegen.cpp:152) into
// libtinympc_wrapper.so, and with -DTINYMPC_TINYTYPE_DOUBLE for `typedef double tinytype` (the reference as checked in,
// glob_opts.hpp:3) into libtinympc_wrapper64.so, on top of the fp64 library (include/tinympc_batch64.h).
#include "../../include/tinympc_admm.h"
#include "../../include/tinympc_batch.h"
#ifdef TINYMPC_TINYTYPE_DOUBLE
#include "../../include/tinympc_batch64.h"
#define TB(name) tiny_batch64_##name
typedef TinyBatch64 Handle;
#else
#define TB(name) tiny_batch_##name
typedef TinyBatch Handle;
#endif
typedef tinytype real;

#include <cstdio>
#include <cstring>
#include <vector>

namespace
{

struct Ctx
{
    Handle *tb = nullptr;
    int nx = 0, nu = 0, N = 0, device = 0;
    // last uploaded problem class / inputs, to skip unchanged uploads
    std::vector<real> gains, bnd[4], xref, opt;
    real rho = 0;
};
Ctx g_ctx;
int g_device = 0;
int g_en_uref = 0, g_en_d2p = 0; // tiny_admm_set_optional_terms
int g_code = 0;

int report(const char *what, int rc)
{
    g_code = rc < 0 ? rc : 0;
    if (rc < 0) std::fprintf(stderr, "tinympc: %s failed (%d): %s\n", what, rc, TB(last_error)());
    return rc;
}

bool same(const std::vector<real> &a, const real *b, size_t n) { return a.size() == n && std::memcmp(a.data(), b, n * sizeof(real)) == 0; }

#define TRYRC(e)          \
    do                    \
    {                     \
        int rc_ = (e);    \
        if (rc_ < 0) return rc_; \
    } while (0)

int null_member(const char *name)
{
    std::fprintf(stderr, "tinympc: TinySolver member '%s' is NULL\n", name);
    return TINY_BATCH_EINVAL;
}
#define NEED(p) \
    if (!(p)) return null_member(#p)

// (re)create the handle for the solver's dimensions and bring the problem class, settings, bounds and Xref up to date
int prepare(const TinySolver *s)
{
    NEED(s); NEED(s->settings); NEED(s->cache); NEED(s->work);
    const TinyWorkspace *w = s->work;
    const TinyCache *c = s->cache;
    const TinySettings *st = s->settings;
    NEED(c->Kinf); NEED(c->Pinf); NEED(c->Quu_inv); NEED(c->AmBKt); NEED(w->Q); NEED(w->Adyn); NEED(w->Bdyn);
    NEED(w->x); NEED(w->u); NEED(w->q); NEED(w->r); NEED(w->p); NEED(w->d); NEED(w->v); NEED(w->vnew); NEED(w->z);
    NEED(w->znew); NEED(w->g); NEED(w->y); NEED(w->u_min); NEED(w->u_max); NEED(w->x_min); NEED(w->x_max); NEED(w->Xref);
    Ctx &C = g_ctx;
    const int nx = w->nx, nu = w->nu, N = w->N;
    if (!C.tb || C.nx != nx || C.nu != nu || C.N != N || C.device != g_device)
    {
        if (C.tb) TB(destroy)(C.tb);
        C = Ctx();
        TRYRC(TB(create)(&C.tb, nx, nu, N, 1, g_device));
        C.nx = nx; C.nu = nu; C.N = N; C.device = g_device;
    }
    // problem class: one flat key {Kinf, Pinf, Quu_inv, AmBKt, Adyn, Bdyn, Q}
    std::vector<real> key;
    auto app = [&](const real *p, int n) { key.insert(key.end(), p, p + n); };
    app(c->Kinf, nu * nx); app(c->Pinf, nx * nx); app(c->Quu_inv, nu * nu); app(c->AmBKt, nx * nx);
    app(w->Adyn, nx * nx); app(w->Bdyn, nx * nu); app(w->Q, nx);
    if (key != C.gains || c->rho != C.rho)
    {
        TRYRC(TB(set_cache)(C.tb, c->rho, c->Kinf, c->Pinf, c->Quu_inv, c->AmBKt));
        TRYRC(TB(set_dynamics)(C.tb, w->Adyn, w->Bdyn, w->Q));
        C.gains.swap(key);
        C.rho = c->rho;
    }
    TRYRC(TB(set_settings)(C.tb, st->abs_pri_tol, st->abs_dua_tol, st->max_iter, st->check_termination,
                                  st->en_state_bound, st->en_input_bound));
    const real *bsrc[4] = {w->x_min, w->x_max, w->u_min, w->u_max};
    int (*bset[4])(Handle *, const real *, int) = {TB(set_xmin), TB(set_xmax), TB(set_umin), TB(set_umax)};
    for (int k = 0; k < 4; k++)
    {
        const size_t n = k < 2 ? (size_t)nx * N : (size_t)nu * (N - 1);
        if (same(C.bnd[k], bsrc[k], n)) continue;
        TRYRC(bset[k](C.tb, bsrc[k], 1));
        C.bnd[k].assign(bsrc[k], bsrc[k] + n);
    }
    if (!same(C.xref, w->Xref, (size_t)nx * N))
    {
        TRYRC(TB(set_xref)(C.tb, w->Xref, 1));
        C.xref.assign(w->Xref, w->Xref + (size_t)nx * N);
    }
#ifndef TINYMPC_TINYTYPE_DOUBLE
    // the two terms the reference ships commented out (admm.cpp:20, :79): members read only when switched on
    if (g_en_uref) { NEED(w->R); NEED(w->Uref); }
    if (g_en_d2p) NEED(c->coeff_d2p);
    std::vector<float> opt;
    if (g_en_uref) { opt.insert(opt.end(), w->R, w->R + nu); opt.insert(opt.end(), w->Uref, w->Uref + (size_t)nu * (N - 1)); }
    if (g_en_d2p) opt.insert(opt.end(), c->coeff_d2p, c->coeff_d2p + (size_t)nx * nu);
    if (opt != C.opt)
    {
        if (g_en_uref)
        {
            TRYRC(tiny_batch_set_input_cost(C.tb, w->R));
            TRYRC(tiny_batch_set_uref(C.tb, w->Uref, 1));
        }
        if (g_en_d2p) TRYRC(tiny_batch_set_coeff_d2p(C.tb, c->coeff_d2p));
        C.opt.swap(opt);
    }
    TRYRC(tiny_batch_set_optional_terms(C.tb, g_en_uref, g_en_d2p));
#endif
    return 0;
}

real *member(TinyWorkspace *w, int id)
{
    real *m[TINY_ARR_COUNT] = {w->x, w->u, w->q, w->r, w->p, w->d, w->v, w->vnew, w->z, w->znew, w->g, w->y};
    return m[id];
}

int upload(TinySolver *s, std::initializer_list<int> ids)
{
    TinyWorkspace *w = s->work;
    for (int id : ids) TRYRC(TB(set_array)(g_ctx.tb, id, member(w, id)));
    const real res[4] = {w->primal_residual_state, w->primal_residual_input, w->dual_residual_state, w->dual_residual_input};
    return TB(set_status)(g_ctx.tb, &w->iter, &w->status, res);
}

int download(TinySolver *s, std::initializer_list<int> ids, bool scalars)
{
    TinyWorkspace *w = s->work;
    for (int id : ids) TRYRC(TB(get_array)(g_ctx.tb, id, member(w, id)));
    if (scalars)
    {
        real res[4];
        TRYRC(TB(get_status)(g_ctx.tb, &w->iter, &w->status, res));
        w->primal_residual_state = res[0]; w->primal_residual_input = res[1];
        w->dual_residual_state = res[2]; w->dual_residual_input = res[3];
    }
    return 0;
}

// one step function: members read -> device, kernel, members written -> host
int step(TinySolver *s, int (*fn)(Handle *), std::initializer_list<int> in, std::initializer_list<int> out)
{
    TRYRC(prepare(s));
    TRYRC(upload(s, in));
    TRYRC(fn(g_ctx.tb));
    return download(s, out, false);
}

} // namespace

extern "C"
{

int tiny_admm_set_device(int device)
{
    g_device = device;
    return 0;
}
int tiny_admm_last_error_code(void) { return g_code; }
int tiny_admm_set_optional_terms(int en_uref, int en_coeff_d2p)
{
#ifdef TINYMPC_TINYTYPE_DOUBLE
    if (en_uref || en_coeff_d2p)
    {
        std::fprintf(stderr, "tinympc: the optional Uref / coeff_d2p terms are implemented by the float library only\n");
        return g_code = TINY_BATCH_EUNSUPPORTED;
    }
#endif
    g_en_uref = en_uref != 0;
    g_en_d2p = en_coeff_d2p != 0;
    g_ctx.opt.clear();
    return 0;
}

int tiny_solve(TinySolver *s) // admm.cpp:111-152
{
    int rc = prepare(s);
    // live-in: x.col(0), d, y, g, v, z (+ p, which a solve that converges in its first iteration leaves as it was)
    if (rc >= 0) rc = upload(s, {TINY_ARR_X, TINY_ARR_P, TINY_ARR_D, TINY_ARR_V, TINY_ARR_Z, TINY_ARR_G, TINY_ARR_Y});
    int ret = 0;
    if (rc >= 0) rc = ret = TB(solve)(g_ctx.tb);
    if (rc >= 0 && s->settings->max_iter > 0) // max_iter <= 0 touches only status and iter (admm.cpp:114-117,151)
        rc = download(s, {TINY_ARR_X, TINY_ARR_U, TINY_ARR_Q, TINY_ARR_R, TINY_ARR_P, TINY_ARR_D, TINY_ARR_V, TINY_ARR_VNEW,
                          TINY_ARR_Z, TINY_ARR_ZNEW, TINY_ARR_G, TINY_ARR_Y}, true);
    else if (rc >= 0)
        rc = TB(get_status)(g_ctx.tb, &s->work->iter, &s->work->status, nullptr);
    report("tiny_solve", rc);
    return rc < 0 ? rc : ret;
}

void forward_pass(TinySolver *s) // admm.cpp:27-37: reads x.col(0), d; writes u, x
{
    report("forward_pass", step(s, TB(forward_pass), {TINY_ARR_X, TINY_ARR_D}, {TINY_ARR_X, TINY_ARR_U}));
}
void update_slack(TinySolver *s) // admm.cpp:45-61: reads u, y, x, g, bounds; writes znew, vnew
{
    report("update_slack", step(s, TB(update_slack), {TINY_ARR_X, TINY_ARR_U, TINY_ARR_G, TINY_ARR_Y}, {TINY_ARR_VNEW, TINY_ARR_ZNEW}));
}
void update_dual(TinySolver *s) // admm.cpp:67-71: reads y, u, znew, g, x, vnew; writes y, g
{
    report("update_dual", step(s, TB(update_dual), {TINY_ARR_X, TINY_ARR_U, TINY_ARR_VNEW, TINY_ARR_ZNEW, TINY_ARR_G, TINY_ARR_Y},
                               {TINY_ARR_G, TINY_ARR_Y}));
}
void update_linear_cost(TinySolver *s) // admm.cpp:77-85: reads znew, y, Xref, vnew, g; writes r, q, p.col(N-1)
{
    report("update_linear_cost", step(s, TB(update_linear_cost), {TINY_ARR_P, TINY_ARR_VNEW, TINY_ARR_ZNEW, TINY_ARR_G, TINY_ARR_Y},
                                      {TINY_ARR_Q, TINY_ARR_R, TINY_ARR_P}));
}
void backward_pass_grad(TinySolver *s) // admm.cpp:15-22: reads p.col(N-1), q, r; writes d, p
{
    report("backward_pass_grad", step(s, TB(backward_pass_grad), {TINY_ARR_P, TINY_ARR_Q, TINY_ARR_R}, {TINY_ARR_P, TINY_ARR_D}));
}
bool termination_condition(TinySolver *s) // admm.cpp:91-109: reads x, vnew, v, u, znew, z, iter; writes the residual fields
{
    int conv = 0;
    int rc = prepare(s);
    if (rc >= 0) rc = upload(s, {TINY_ARR_X, TINY_ARR_U, TINY_ARR_V, TINY_ARR_VNEW, TINY_ARR_Z, TINY_ARR_ZNEW});
    if (rc >= 0) rc = TB(termination_condition)(g_ctx.tb, &conv);
    if (rc >= 0)
    {
        const int it = s->work->iter, stt = s->work->status; // not written by the reference function
        rc = download(s, {}, true);
        s->work->iter = it; s->work->status = stt;
    }
    report("termination_condition", rc);
    return rc >= 0 && conv != 0;
}
void update_primal(TinySolver *)
{
    std::fprintf(stderr, "tinympc: update_primal is declared in the reference (admm.hpp:12) but has no definition there; nothing to run\n");
    g_code = TINY_BATCH_EUNSUPPORTED;
}

} // extern "C"
