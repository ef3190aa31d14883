// riccati.cpp — host-side fp64 precompute of the TinyMPC cache (Kinf, Pinf, Quu_inv, AmBKt).
//
// Restates the arithmetic of the reference's code generator, src/tinympc/codegen.cpp:254-292
// (the part of tiny_codegen() that is an input producer for the hot path; the source emitter
// around it is out of scope):
//     Q1 = diag(Q + rho), R1 = diag(R + rho)                                  (:255-258)
//     P <- rho*I ; repeat up to 1000 times                                     (:268-285)
//         K = (R1 + B'PB)^-1 B'PA ;  Pn = Q1 + A'P(A - BK)
//         stop when max|K - Kprev| < 1e-5  (the reference keeps K,Pn of that iteration)
//         P <- Pn
//     Quu_inv = (R1 + B'Pn B)^-1 ; AmBKt = (A - BK)' ; coeff_d2p = K'R1 - AmBKt Pn B   (:290-292)
// It runs once per problem class on the host in double precision ("otherwise Riccati may fail",
// examples/codegen_cartpole.cpp:9-11); it is not a GPU kernel.
#include "../../include/tinympc_batch.h"

#include <cmath>
#include <utility>
#include <vector>

namespace
{

// Minimal dense column-major matrix
struct Mat
{
    int r = 0, c = 0;
    std::vector<double> a;
    Mat() {}
    Mat(int r_, int c_) : r(r_), c(c_), a((size_t)r_ * c_, 0.0) {}
    double &operator()(int i, int j) { return a[(size_t)j * r + i]; }
    double operator()(int i, int j) const { return a[(size_t)j * r + i]; }
};

Mat mul(const Mat &A, const Mat &B)
{
    Mat C(A.r, B.c);
    for (int j = 0; j < B.c; j++)
        for (int k = 0; k < A.c; k++)
        {
            const double b = B(k, j);
            for (int i = 0; i < A.r; i++) C(i, j) += A(i, k) * b;
        }
    return C;
}

Mat transpose(const Mat &A)
{
    Mat T(A.c, A.r);
    for (int j = 0; j < A.c; j++)
        for (int i = 0; i < A.r; i++) T(j, i) = A(i, j);
    return T;
}

Mat add(const Mat &A, const Mat &B, double sb = 1.0)
{
    Mat C(A.r, A.c);
    for (size_t e = 0; e < A.a.size(); e++) C.a[e] = A.a[e] + sb * B.a[e];
    return C;
}

// Solve G X = RHS by LU with partial pivoting (G is nu x nu, small).  Returns false if singular.
bool lu_solve(Mat G, Mat RHS, Mat &X)
{
    const int n = G.r;
    for (int c = 0; c < n; c++)
    {
        int piv = c;
        for (int r = c + 1; r < n; r++)
            if (std::fabs(G(r, c)) > std::fabs(G(piv, c))) piv = r;
        if (G(piv, c) == 0.0) return false;
        if (piv != c)
        {
            for (int j = 0; j < n; j++) std::swap(G(c, j), G(piv, j));
            for (int j = 0; j < RHS.c; j++) std::swap(RHS(c, j), RHS(piv, j));
        }
        for (int r = c + 1; r < n; r++)
        {
            const double f = G(r, c) / G(c, c);
            if (f == 0.0) continue;
            for (int j = c; j < n; j++) G(r, j) -= f * G(c, j);
            for (int j = 0; j < RHS.c; j++) RHS(r, j) -= f * RHS(c, j);
        }
    }
    X = Mat(n, RHS.c);
    for (int j = 0; j < RHS.c; j++)
        for (int i = n - 1; i >= 0; i--)
        {
            double s = RHS(i, j);
            for (int k = i + 1; k < n; k++) s -= G(i, k) * X(k, j);
            X(i, j) = s / G(i, i);
        }
    return true;
}

Mat eye(int n)
{
    Mat I(n, n);
    for (int i = 0; i < n; i++) I(i, i) = 1.0;
    return I;
}

} // namespace

extern "C" int tiny_riccati(int nx, int nu, const double *A_, const double *B_, const double *Q, const double *R,
                            double rho, double *Kinf, double *Pinf, double *Quu_inv, double *AmBKt,
                            double *coeff_d2p, int *iters)
{
    if (nx < 1 || nu < 1 || !A_ || !B_ || !Q || !R || !Kinf || !Pinf || !Quu_inv || !AmBKt) return TINY_BATCH_EINVAL;
    Mat A(nx, nx), B(nx, nu), Q1(nx, nx), R1(nu, nu), P(nx, nx);
    A.a.assign(A_, A_ + (size_t)nx * nx);
    B.a.assign(B_, B_ + (size_t)nx * nu);
    for (int i = 0; i < nx; i++) { Q1(i, i) = Q[i] + rho; P(i, i) = rho; }
    for (int i = 0; i < nu; i++) R1(i, i) = R[i] + rho;
    const Mat At = transpose(A), Bt = transpose(B);
    Mat K(nu, nx), Kprev(nu, nx), Pn(nx, nx);
    int n_it = 1000;
    for (int it = 0; it < 1000; it++)
    {
        const Mat BtP = mul(Bt, P);
        const Mat G = add(R1, mul(BtP, B));
        if (!lu_solve(G, mul(BtP, A), K)) return TINY_BATCH_EINVAL;
        Pn = add(Q1, mul(mul(At, P), add(A, mul(B, K), -1.0)));
        double md = 0.0;
        for (size_t e = 0; e < K.a.size(); e++) md = std::fmax(md, std::fabs(K.a[e] - Kprev.a[e]));
        if (md < 1e-5) { n_it = it + 1; break; }
        Kprev = K;
        P = Pn;
    }
    Mat Qi;
    if (!lu_solve(add(R1, mul(mul(Bt, Pn), B)), eye(nu), Qi)) return TINY_BATCH_EINVAL;
    const Mat AmBK_t = transpose(add(A, mul(B, K), -1.0));
    for (size_t e = 0; e < K.a.size(); e++) Kinf[e] = K.a[e];
    for (size_t e = 0; e < Pn.a.size(); e++) Pinf[e] = Pn.a[e];
    for (size_t e = 0; e < Qi.a.size(); e++) Quu_inv[e] = Qi.a[e];
    for (size_t e = 0; e < AmBK_t.a.size(); e++) AmBKt[e] = AmBK_t.a[e];
    if (coeff_d2p)
    {
        const Mat C = add(mul(transpose(K), R1), mul(mul(AmBK_t, Pn), B), -1.0);
        for (size_t e = 0; e < C.a.size(); e++) coeff_d2p[e] = C.a[e];
    }
    if (iters) *iters = n_it;
    return TINY_BATCH_OK;
}
