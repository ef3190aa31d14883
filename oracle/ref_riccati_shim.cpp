/*
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * Shim around the compiled reference's tiny_codegen() (src/tinympc/codegen.cpp:218-696),
 * used in THIS container only, by tests/golden/make_golden.py, to obtain the reference's
 * own Riccati cache (Kinf, Pinf, Quu_inv, AmBKt) for problems that ship no precomputed
 * gains (cartpole, the synthetic nx=32 system).  tiny_codegen() also emits a source tree
 * into `output_dir` (it copies the reference into it), so the caller must point it at a
 * scratch directory under /tmp that is deleted afterwards; only the numbers parsed from
 * the generated tiny_data_workspace.cpp are kept, as fixtures under tests/golden/.
 * The translation unit is #included from where it lies; nothing is copied into the repo.
 */
#include REF_ROOT_CODEGEN

extern "C" int ref_codegen(int nx, int nu, int N, double *A, double *B, double *Q, double *R, double *x_min,
                           double *x_max, double *u_min, double *u_max, double rho, double abs_pri_tol,
                           double abs_dua_tol, int max_iters, int check_termination, const char *tinympc_dir,
                           const char *output_dir)
{
    static_assert(sizeof(tinytype) == sizeof(double), "codegen must run in double (examples/codegen_cartpole.cpp:9-11)");
    return tiny_codegen(nx, nu, N, A, B, Q, R, x_min, x_max, u_min, u_max, rho, abs_pri_tol, abs_dua_tol, max_iters,
                        check_termination, 0, tinympc_dir, output_dir);
}
