/*
 * tinympc_admm.h — the reference's native solver API under its own names, for ONE instance, backed by the HIP solver.
 *
 * Replaces src/tinympc/admm.hpp:10-18 + src/tinympc/types.hpp:26-107 for callers that own a TinySolver (the reference's
 * example programs): the same seven `extern "C"` functions, the same struct and member names, the same member order and
 * the same column-major storage — but the members are plain caller-owned float arrays instead of fixed-size Eigen
 * matrices (whose binary layout depends on Eigen's alignment macros and is not a stable ABI), and the dimensions the
 * reference fixes at compile time (glob_opts.hpp:3-9) are three ints in TinyWorkspace.
 *
 * Every call copies the structs' live-in members to the device, runs the corresponding kernel of libtinympc_hip.so on a
 * batch of one and copies every member the reference function writes back.  It is a compatibility path (a few hundred
 * microseconds of copies per call); throughput comes from the batched API in tinympc_batch.h.  There is no CPU
 * implementation behind these names: without a GPU they report TINY_BATCH_EHIP.
 *
 * Exported by accelerated-tinympc_amd/lib/libtinympc_wrapper.so together with the ten wrapper functions
 * (tinympc_wrapper.h) — the same 17 function symbols the reference's generated libtinympcShared.so exports.
 */
#ifndef TINYMPC_ADMM_H
#define TINYMPC_ADMM_H
#ifndef __cplusplus
#include <stdbool.h>
#endif
#ifdef __cplusplus
extern "C"
{
#endif

    /* codegen.cpp:152: generated code is always float — the default here, served by libtinympc_wrapper.so.  Define
     * TINYMPC_TINYTYPE_DOUBLE before including this header for the reference as it is checked in (glob_opts.hpp:3:
     * typedef double tinytype) and link libtinympc_wrapper64.so instead: same names, same structs with double members,
     * results bitwise equal to the reference's fp64 build (classes (12,4), (4,1), (8,4); the optional terms are float only). */
#ifdef TINYMPC_TINYTYPE_DOUBLE
    typedef double tinytype;
#else
    typedef float tinytype;
#endif

    /* types.hpp:26-34.  Matrices column-major: Kinf nu x nx, Pinf nx x nx, Quu_inv nu x nu, AmBKt nx x nx.
     * coeff_d2p (nx x nu) is not read by the solver (admm.cpp:20 has the term commented out) and may be NULL, unless
     * tiny_admm_set_optional_terms() switches the term on. */
    typedef struct
    {
        tinytype rho;
        tinytype *Kinf;
        tinytype *Pinf;
        tinytype *Quu_inv;
        tinytype *AmBKt;
        tinytype *coeff_d2p;
    } TinyCache;

    /* types.hpp:39-47 */
    typedef struct
    {
        tinytype abs_pri_tol;
        tinytype abs_dua_tol;
        int max_iter;
        int check_termination;
        int en_state_bound;
        int en_input_bound;
    } TinySettings;

    /* types.hpp:52-97, same member names and order.  State-type members are nx x N, input-type members nu x (N-1),
     * column-major (element (i,j) at j*rows + i), caller-allocated.  R, Uref, Qu are not read (admm.cpp:79) and may be NULL;
     * tiny_admm_set_optional_terms() makes R and Uref live. */
    typedef struct
    {
        int nx, nu, N; /* NSTATES, NINPUTS, NHORIZON of glob_opts.hpp */

        tinytype *x;
        tinytype *u;
        tinytype *q;
        tinytype *r;
        tinytype *p;
        tinytype *d;
        tinytype *v;
        tinytype *vnew;
        tinytype *z;
        tinytype *znew;
        tinytype *g;
        tinytype *y;

        tinytype primal_residual_state;
        tinytype primal_residual_input;
        tinytype dual_residual_state;
        tinytype dual_residual_input;
        int status;
        int iter;

        tinytype *Q;    /* nx */
        tinytype *R;    /* nu, unused */
        tinytype *Adyn; /* nx x nx */
        tinytype *Bdyn; /* nx x nu */

        tinytype *u_min;
        tinytype *u_max;
        tinytype *x_min;
        tinytype *x_max;
        tinytype *Xref;
        tinytype *Uref; /* unused */

        tinytype *Qu; /* unused */
    } TinyWorkspace;

    /* types.hpp:102-107 */
    typedef struct
    {
        TinySettings *settings;
        TinyCache *cache;
        TinyWorkspace *work;
    } TinySolver;

    /* admm.hpp:10 / admm.cpp:111-152.  0 = converged, 1 = max_iter reached (as the reference); negative TinyBatchError
     * if the device path failed (the structs are then left untouched). */
    int tiny_solve(TinySolver *solver);

    /* admm.hpp:12-18 / admm.cpp:15-109: each reads and writes exactly the members the reference function does.
     * They need a row-kernel instantiation (nx + nu <= 16, TINY_FOR_EACH_ROWDIMS). */
    void update_primal(TinySolver *solver); /* declared in admm.hpp:12 but defined nowhere in the reference; reports EUNSUPPORTED */
    void backward_pass_grad(TinySolver *solver);
    void forward_pass(TinySolver *solver);
    void update_slack(TinySolver *solver);
    void update_dual(TinySolver *solver);
    void update_linear_cost(TinySolver *solver);
    bool termination_condition(TinySolver *solver);

    /* Additions: HIP device used by the calls above (default 0) and the code of the last call (0 or TinyBatchError;
     * message via tiny_batch_last_error()). */
    int tiny_admm_set_device(int device);
    int tiny_admm_last_error_code(void);
    /* Switch on the two terms the reference ships commented out, for every later call: en_coeff_d2p adds
     * "+ coeff_d2p * d.col(i)" in backward_pass_grad (admm.cpp:20), en_uref makes update_linear_cost compute
     * r = -(Uref o R) - rho*(znew - y) (what admm.cpp:79 stands for).  Both off by default = the reference as shipped. */
    int tiny_admm_set_optional_terms(int en_uref, int en_coeff_d2p);

#ifdef __cplusplus
}
#endif
#endif
