/*
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.  See tinympc_oracle_impl.h.
 *
 * Builds the fp32 and fp64 instantiations of the CPU restatement of
 * src/tinympc/admm.cpp:15-152 plus the Riccati cache precompute of
 * src/tinympc/codegen.cpp:254-292 (fp64 only, as the reference requires:
 * examples/codegen_cartpole.cpp:9-11).
 *
 * Build: see oracle/Makefile  (gcc -O3 -ffp-contract=off -fopenmp -shared).
 */
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define ORACLE_MAX_DIM 256

/* Flush-to-zero / denormals-are-zero control for CPU timing runs (x86 MXCSR is per
 * thread, so the batch driver applies the requested mode inside its parallel region). */
static int g_oracle_ftz = 0;
#if defined(__x86_64__)
#include <xmmintrin.h>
#include <pmmintrin.h>
static inline void oracle_apply_fp_mode(void)
{
    _MM_SET_FLUSH_ZERO_MODE(g_oracle_ftz ? _MM_FLUSH_ZERO_ON : _MM_FLUSH_ZERO_OFF);
    _MM_SET_DENORMALS_ZERO_MODE(g_oracle_ftz ? _MM_DENORMALS_ZERO_ON : _MM_DENORMALS_ZERO_OFF);
}
#else
static inline void oracle_apply_fp_mode(void) {}
#endif
void oracle_set_ftz_daz(int on) { g_oracle_ftz = on; oracle_apply_fp_mode(); }


/* float -> IEEE binary16 (round to nearest even, subnormals kept, overflow to infinity) -> float: the value an fp16
 * store followed by a load leaves behind.  Bit-level, so it does not depend on compiler support for _Float16. */
float oracle_round_h16(float f)
{
    union { float f; unsigned u; } v = {f};
    const unsigned sign = v.u & 0x80000000u, a = v.u & 0x7fffffffu;
    if (a >= 0x7f800000u) return f;                     /* inf / nan */
    if (a >= 0x477ff000u) { v.u = sign | 0x7f800000u; return v.f; } /* >= 65520 rounds to inf */
    if (a < 0x38800000u)                                /* below 2^-14: fp16 subnormal, quantum 2^-24 */
    {
        union { float f; unsigned u; } m = {0};
        m.u = a;
        const float q = m.f * 16777216.0f;              /* exact scaling */
        const float r = __builtin_rintf(q);             /* current rounding mode = to nearest even */
        m.f = r * (1.0f / 16777216.0f);
        v.u = sign | m.u;
        return v.f;
    }
    /* normal: keep 10 mantissa bits */
    const unsigned rem = a & 0x1fffu, base = a & ~0x1fffu;
    unsigned res = base;
    if (rem > 0x1000u || (rem == 0x1000u && (base & 0x2000u))) res += 0x2000u;
    v.u = sign | res;
    return v.f;
}

#define ST(x) (x)
#define STD(x) (x)
#define REAL float
#define PS 4
#define SUF(name) name##_f32
#include "tinympc_oracle_impl.h"
#undef REAL
#undef SUF
#undef PS
#undef ST
#undef STD

/* fp16 storage / fp32 arithmetic (see the header of tinympc_oracle_impl.h) */
#define ST(x) oracle_round_h16(x)
#define STD(x) oracle_round_h16(x)
#define REAL float
#define PS 4
#define SUF(name) name##_h16
#include "tinympc_oracle_impl.h"
#undef REAL
#undef SUF
#undef PS
#undef STD
/* ... with the duals y, g kept in fp32 */
#define STD(x) (x)
#define REAL float
#define PS 4
#define SUF(name) name##_h16d
#include "tinympc_oracle_impl.h"
#undef REAL
#undef SUF
#undef PS
#undef ST
#undef STD
#define ST(x) (x)
#define STD(x) (x)

#define REAL double
#define PS 2
#define SUF(name) name##_f64
#include "tinympc_oracle_impl.h"
#undef REAL
#undef SUF
#undef PS

/* ---- small dense helpers (column-major, fp64) for the Riccati precompute ---- */

/* C(m x n) = op(A) * op(B); ta/tb = 1 means use the transpose.  A is (ta? k x m : m x k). */
static void mm(int m, int n, int k, const double *A, int ta, const double *B, int tb, double *C)
{
    for (int j = 0; j < n; j++)
        for (int i = 0; i < m; i++)
        {
            double acc = 0.0;
            for (int l = 0; l < k; l++)
            {
                double a = ta ? A[(size_t)i * k + l] : A[(size_t)l * m + i];
                double b = tb ? B[(size_t)l * n + j] : B[(size_t)j * k + l];
                acc += a * b;
            }
            C[(size_t)j * m + i] = acc;
        }
}

/* in-place inverse by Gauss-Jordan with partial pivoting; returns 0 on success */
static int inv_inplace(int n, double *A)
{
    double *W = (double *)malloc(sizeof(double) * (size_t)n * 2 * n);
    if (!W) return -1;
    /* W is row-major n x 2n: [A | I] */
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++)
        {
            W[(size_t)i * 2 * n + j] = A[(size_t)j * n + i];
            W[(size_t)i * 2 * n + n + j] = (i == j) ? 1.0 : 0.0;
        }
    for (int c = 0; c < n; c++)
    {
        int piv = c;
        double best = fabs(W[(size_t)c * 2 * n + c]);
        for (int r = c + 1; r < n; r++)
            if (fabs(W[(size_t)r * 2 * n + c]) > best) { best = fabs(W[(size_t)r * 2 * n + c]); piv = r; }
        if (best == 0.0) { free(W); return -2; }
        if (piv != c)
            for (int j = 0; j < 2 * n; j++)
            {
                double t = W[(size_t)c * 2 * n + j];
                W[(size_t)c * 2 * n + j] = W[(size_t)piv * 2 * n + j];
                W[(size_t)piv * 2 * n + j] = t;
            }
        double ip = 1.0 / W[(size_t)c * 2 * n + c];
        for (int j = 0; j < 2 * n; j++) W[(size_t)c * 2 * n + j] *= ip;
        for (int r = 0; r < n; r++)
        {
            if (r == c) continue;
            double f = W[(size_t)r * 2 * n + c];
            if (f == 0.0) continue;
            for (int j = 0; j < 2 * n; j++) W[(size_t)r * 2 * n + j] -= f * W[(size_t)c * 2 * n + j];
        }
    }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) A[(size_t)j * n + i] = W[(size_t)i * 2 * n + n + j];
    free(W);
    return 0;
}

/*
 * src/tinympc/codegen.cpp:254-292.
 *   Q1 = diag(Q + rho), R1 = diag(R + rho)                     (:255-258)
 *   Ptp1 = rho*I, Ktp1 = 0                                     (:268-269)
 *   repeat (<=1000):  Kinf = (R1 + B'PB)^-1 B'PA ; Pinf = Q1 + A'P(A - B Kinf)
 *                     stop when max|Kinf - Ktp1| < 1e-5        (:273-285)
 *   Quu_inv = (R1 + B' Pinf B)^-1 ; AmBKt = (A - B Kinf)' ; coeff_d2p = Kinf' R1 - AmBKt Pinf B   (:290-292)
 * Outputs are column-major.  Returns the number of Riccati iterations executed
 * (the reference prints i+1 on convergence; 1000 means "did not converge", the
 * reference then silently keeps the last iterate), or a negative value on a
 * singular matrix.
 */
int oracle_riccati_f64(int nx, int nu, const double *A, const double *B, const double *Q, const double *R,
                       double rho, double *Kinf, double *Pinf, double *Quu_inv, double *AmBKt, double *coeff_d2p)
{
    size_t nn = (size_t)nx * nx, nm = (size_t)nx * nu, mm_ = (size_t)nu * nu;
    double *Q1 = calloc(nn, 8), *R1 = calloc(mm_, 8), *P = calloc(nn, 8), *K0 = calloc(nm, 8);
    double *BtP = malloc(nm * 8), *G = malloc(mm_ * 8), *BtPA = malloc(nm * 8), *AmBK = malloc(nn * 8);
    double *AtP = malloc(nn * 8), *T = malloc(nn * 8), *BK = malloc(nn * 8), *T2 = malloc(nm * 8), *T3 = malloc(nm * 8);
    int iters = 1000, rc = 0;
    for (int i = 0; i < nx; i++) { Q1[(size_t)i * nx + i] = Q[i] + rho; P[(size_t)i * nx + i] = rho; }
    for (int i = 0; i < nu; i++) R1[(size_t)i * nu + i] = R[i] + rho;
    for (int it = 0; it < 1000; it++)
    {
        mm(nu, nx, nx, B, 1, P, 0, BtP);    /* B' P      (nu x nx) */
        mm(nu, nu, nx, BtP, 0, B, 0, G);    /* B' P B    */
        for (size_t e = 0; e < mm_; e++) G[e] += R1[e];
        if ((rc = inv_inplace(nu, G)) != 0) break;
        mm(nu, nx, nx, BtP, 0, A, 0, BtPA); /* B' P A    */
        mm(nu, nx, nu, G, 0, BtPA, 0, Kinf);
        mm(nx, nx, nu, B, 0, Kinf, 0, BK);
        for (size_t e = 0; e < nn; e++) AmBK[e] = A[e] - BK[e];
        mm(nx, nx, nx, A, 1, P, 0, AtP);
        mm(nx, nx, nx, AtP, 0, AmBK, 0, T);
        for (size_t e = 0; e < nn; e++) Pinf[e] = Q1[e] + T[e];
        double md = 0.0;
        for (size_t e = 0; e < nm; e++) { double a = fabs(Kinf[e] - K0[e]); if (a > md) md = a; }
        if (md < 1e-5) { iters = it + 1; break; }
        memcpy(K0, Kinf, nm * 8);
        memcpy(P, Pinf, nn * 8);
    }
    if (rc == 0)
    {
        mm(nu, nx, nx, B, 1, Pinf, 0, BtP);
        mm(nu, nu, nx, BtP, 0, B, 0, G);
        for (size_t e = 0; e < mm_; e++) G[e] += R1[e];
        rc = inv_inplace(nu, G);
        memcpy(Quu_inv, G, mm_ * 8);
        mm(nx, nx, nu, B, 0, Kinf, 0, BK);
        for (int j = 0; j < nx; j++)
            for (int i = 0; i < nx; i++) AmBKt[(size_t)j * nx + i] = A[(size_t)i * nx + j] - BK[(size_t)i * nx + j];
        if (coeff_d2p)
        {
            mm(nx, nu, nu, Kinf, 1, R1, 0, T2);  /* Kinf' R1  (nx x nu) */
            mm(nx, nx, nx, AmBKt, 0, Pinf, 0, T);
            mm(nx, nu, nx, T, 0, B, 0, T3);
            for (size_t e = 0; e < nm; e++) coeff_d2p[e] = T2[e] - T3[e];
        }
    }
    free(Q1); free(R1); free(P); free(K0); free(BtP); free(G); free(BtPA); free(AmBK);
    free(AtP); free(T); free(BK); free(T2); free(T3);
    return rc ? rc : iters;
}

