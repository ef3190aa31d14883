/*
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * Pins the arithmetic of the two terms the reference ships COMMENTED OUT (src/tinympc/admm.cpp:20 "+ coeff_d2p *
 * d.col(i)" and admm.cpp:79, the Uref term): the reference has no executable form of them, so what can be pinned is how
 * Eigen (the vendored 3.4.90, same flags as oracle/_ref) evaluates those expressions over the reference's own matrix
 * types (src/tinympc/types.hpp, included from where it lies; nothing is copied).  The two functions below spell the
 * expressions out — admm.cpp:19-20 with the trailing comment of :20 removed, and for the input cost the twin of
 * admm.cpp:81-82 that upstream TinyMPC uses (the text commented out at :79 does not compile as written) — and
 * tests/test_oracle.py compares oracle_backward_pass_grad / oracle_update_linear_cost with them bit for bit.
 */
#if defined(REF_SCALAR_f32)
#define double float
#include REF_ROOT_GLOB
#undef double
#elif defined(REF_SCALAR_f64)
#include REF_ROOT_GLOB
#else
#error "define REF_SCALAR_f32 or REF_SCALAR_f64"
#endif
#undef NSTATES
#undef NINPUTS
#undef NHORIZON
#define NSTATES REF_NX
#define NINPUTS REF_NU
#define NHORIZON REF_N
#include REF_ROOT_TYPES

#include <cstring>

namespace
{
TinyCache C_;
TinyWorkspace W_;
template <class M>
void load(M &m, const tinytype *src) { std::memcpy(m.data(), src, sizeof(tinytype) * m.size()); }
template <class M>
void store(const M &m, tinytype *dst) { std::memcpy(dst, m.data(), sizeof(tinytype) * m.size()); }
} // namespace

extern "C"
{
    void terms_dims(int *nx, int *nu, int *N) { *nx = NSTATES; *nu = NINPUTS; *N = NHORIZON; }

    /* backward sweep with the coeff_d2p term: p (in: column N-1; out: all), d (out), q, r (in) */
    void terms_backward_pass_grad(const tinytype *Kinf, const tinytype *Quu_inv, const tinytype *AmBKt, const tinytype *Bdyn,
                                  const tinytype *coeff_d2p, const tinytype *q, const tinytype *r, tinytype *p, tinytype *d)
    {
        load(C_.Kinf, Kinf); load(C_.Quu_inv, Quu_inv); load(C_.AmBKt, AmBKt); load(C_.coeff_d2p, coeff_d2p);
        load(W_.Bdyn, Bdyn); load(W_.q, q); load(W_.r, r); load(W_.p, p); load(W_.d, d);
        for (int i = NHORIZON - 2; i >= 0; i--)
        {
            (W_.d.col(i)).noalias() = C_.Quu_inv * (W_.Bdyn.transpose() * W_.p.col(i + 1) + W_.r.col(i));
            (W_.p.col(i)).noalias() = W_.q.col(i) + C_.AmBKt.lazyProduct(W_.p.col(i + 1)) -
                                      (C_.Kinf.transpose()).lazyProduct(W_.r.col(i)) + C_.coeff_d2p * W_.d.col(i);
        }
        store(W_.p, p); store(W_.d, d);
    }

    /* input cost with a reference: r = -(Uref o R) - rho*(znew - y) */
    void terms_input_cost(tinytype rho, const tinytype *Uref, const tinytype *R, const tinytype *znew, const tinytype *y, tinytype *r)
    {
        load(W_.Uref, Uref); load(W_.R, R); load(W_.znew, znew); load(W_.y, y);
        W_.r = -(W_.Uref.array().colwise() * W_.R.array());
        (W_.r).noalias() -= rho * (W_.znew - W_.y);
        store(W_.r, r);
    }
}
