/*
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 * AddressSanitizer / UndefinedBehaviorSanitizer run of the CPU restatement (sanitizers run on the CPU
 * build only; GPU sanitizers are not available on this pool).  Built and executed by `make -C oracle sanitize`:
 * a few batched solves of a small stable random system in fp32, fp64 and the fp16-storage instantiation, every array
 * exactly sized so that any out-of-bounds access of the restatement trips ASan.
 */
#include "tinympc_oracle.c"
#include <stdio.h>

#define NX 8
#define NU 4
#define NH 7
#define NB 5

static double urand(unsigned *s) { *s = *s * 1664525u + 1013904223u; return (double)(*s >> 8) / (double)(1u << 24) - 0.5; }

#define RUN(REAL, SUF)                                                                                              \
    static int run##SUF(void)                                                                                        \
    {                                                                                                                \
        unsigned seed = 12345u;                                                                                      \
        REAL *K = malloc(sizeof(REAL) * NU * NX), *P = malloc(sizeof(REAL) * NX * NX), *Qi = malloc(sizeof(REAL) * NU * NU),  \
             *Am = malloc(sizeof(REAL) * NX * NX), *A = malloc(sizeof(REAL) * NX * NX), *B = malloc(sizeof(REAL) * NX * NU),  \
             *Q = malloc(sizeof(REAL) * NX);                                                                         \
        for (int i = 0; i < NU * NX; i++) K[i] = (REAL)(0.1 * urand(&seed));                                         \
        for (int i = 0; i < NX * NX; i++) { P[i] = (REAL)urand(&seed); Am[i] = (REAL)(0.2 * urand(&seed)); A[i] = (REAL)(0.2 * urand(&seed)); } \
        for (int i = 0; i < NU * NU; i++) Qi[i] = (REAL)(0.1 * urand(&seed));                                        \
        for (int i = 0; i < NX * NU; i++) B[i] = (REAL)(0.3 * urand(&seed));                                         \
        for (int i = 0; i < NX; i++) Q[i] = (REAL)(1.0 + i);                                                         \
        REAL Rr[NU], *Cd = malloc(sizeof(REAL) * NX * NU), *uref = malloc(sizeof(REAL) * (NH - 1) * NU);             \
        for (int i = 0; i < NU; i++) Rr[i] = (REAL)(0.5 + i);                                                        \
        for (int i = 0; i < NX * NU; i++) Cd[i] = (REAL)(0.01 * urand(&seed));                                       \
        for (int i = 0; i < (NH - 1) * NU; i++) uref[i] = (REAL)(0.05 * urand(&seed));                               \
        /* the two optional terms (admm.cpp:20,79) are switched on so that their loops run under the sanitizers too */ \
        OracleProblem##SUF pr = {NX, NU, NH, (REAL)1.0, K, P, Qi, Am, A, B, Q, (REAL)1e-3, (REAL)1e-3, 25, 2, 1, 1, 1, 1, Rr, Cd}; \
        const size_t sx = (size_t)NB * NH * NX, su = (size_t)NB * (NH - 1) * NU;                                     \
        REAL *xs[6], *us[6];                                                                                         \
        for (int k = 0; k < 6; k++) { xs[k] = calloc(sx, sizeof(REAL)); us[k] = calloc(su, sizeof(REAL)); }          \
        REAL *umin = malloc(sizeof(REAL) * (NH - 1) * NU), *umax = malloc(sizeof(REAL) * (NH - 1) * NU),           \
             *xmin = malloc(sizeof(REAL) * NH * NX), *xmax = malloc(sizeof(REAL) * NH * NX), *xref = malloc(sizeof(REAL) * sx); \
        for (int i = 0; i < (NH - 1) * NU; i++) { umin[i] = (REAL)-0.2; umax[i] = (REAL)0.2; }                       \
        for (int i = 0; i < NH * NX; i++) { xmin[i] = (REAL)-1; xmax[i] = (REAL)1; }                                 \
        for (size_t i = 0; i < sx; i++) xref[i] = (REAL)(0.3 * urand(&seed));                                        \
        for (int b = 0; b < NB; b++) for (int i = 0; i < NX; i++) xs[0][(size_t)b * NH * NX + i] = (REAL)urand(&seed); \
        REAL *res = calloc((size_t)NB * 4, sizeof(REAL));                                                            \
        int *status = calloc(NB, sizeof(int)), *iter = calloc(NB, sizeof(int));                                      \
        OracleBatch##SUF bt = {NB, xs[0], us[0], xs[1], us[1], xs[2], us[2], xs[3], xs[4], us[3], us[4], xs[5], us[5],  \
                              umin, umax, xmin, xmax, xref, 0, 0, (long long)NH * NX, res, status, iter, uref, 0};   \
        int unsolved = 0;                                                                                            \
        for (int rep = 0; rep < 3; rep++) unsolved = oracle_solve_batch##SUF(&pr, &bt, 2);                           \
        int total = 0;                                                                                               \
        for (int b = 0; b < NB; b++) total += iter[b];                                                               \
        printf("%-5s unsolved %d, iterations %d\n", #SUF, unsolved, total);                                          \
        for (int k = 0; k < 6; k++) { free(xs[k]); free(us[k]); }                                                    \
        free(K); free(P); free(Qi); free(Am); free(A); free(B); free(Q); free(umin); free(umax); free(xmin); free(xmax); \
        free(xref); free(res); free(status); free(iter); free(Cd); free(uref);                                       \
        return total > 0 ? 0 : 1;                                                                                    \
    }
RUN(float, _f32)
RUN(double, _f64)
RUN(float, _h16)

int main(void)
{
    double A[4] = {1.0, 0.0, 0.1, 1.0}, B[2] = {0.005, 0.1}, Q[2] = {1, 1}, R[1] = {1};
    double K[2], P[4], Qi[1], Am[4], cd[2];
    const int it = oracle_riccati_f64(2, 1, A, B, Q, R, 1.0, K, P, Qi, Am, cd);
    printf("riccati iterations %d\n", it);
    return run_f32() | run_f64() | run_h16() | (it > 0 && it < 1000 ? 0 : 1);
}
