/*
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * CPU restatement of the TinyMPC ADMM hot path, used only as the parity checker
 * (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).  The shipped
 * solver is the HIP library under accelerated-tinympc_amd/csrc and never calls
 * into this file.
 *
 * This header is a "template": it is included three times by tinympc_oracle.c, with
 *   REAL=float  SUF(x)=x##_f32,   REAL=double SUF(x)=x##_f64   (ST(x) = x: the reference's arithmetic), and
 *   REAL=float  SUF(x)=x##_h16 with ST(x) = round-to-nearest-even to IEEE binary16 and back: "fp16 storage, fp32
 *   arithmetic" (BASELINE.json configs[4]) — every ASSIGNMENT to a TinyWorkspace array member rounds the stored value,
 *   products, sums, Eigen temporaries and the four residual reductions stay fp32.  The reference has no such mode, so
 *   the _h16 instantiation is pinned only through the _f32 one it is textually identical to apart from ST().
 *   REAL=float  SUF(x)=x##_h16d: the same with the DUALS y, g kept in fp32 (STD(x) = x): "fp16 states, fp32 residual
 *   accumulation" read literally — the duals are the running sums of the primal residuals.
 *
 * Every function names the reference lines it restates (paths relative to
 * /root/reference).  All matrices are column-major, exactly like the Eigen
 * fixed-size matrices of src/tinympc/types.hpp:13-21: element (i,j) of an
 * R x C matrix lives at flat index j*R + i.
 *
 * Arithmetic order: every mat-vec is a plain inner product accumulated in
 * ascending k with separate multiply and add (compile with -ffp-contract=off),
 * which is what Eigen's coefficient-based lazyProduct does on an SSE2 target
 * (include/Eigen/Eigen/src/Core/GeneralProduct.h:444 and the coeff-based
 * product evaluator).  Parity with the compiled reference is pinned by
 * tests/test_oracle_vs_ref.py and by the fixtures under tests/golden/.
 */

typedef struct SUF(OracleProblem)
{
    int nx, nu, N;
    /* TinyCache (types.hpp:26-34) */
    REAL rho;
    const REAL *Kinf;    /* nu x nx */
    const REAL *Pinf;    /* nx x nx */
    const REAL *Quu_inv; /* nu x nu */
    const REAL *AmBKt;   /* nx x nx */
    /* per-problem members of TinyWorkspace (types.hpp:82-85) */
    const REAL *Adyn; /* nx x nx */
    const REAL *Bdyn; /* nx x nu */
    const REAL *Q;    /* nx */
    /* TinySettings (types.hpp:39-47) */
    REAL abs_pri_tol, abs_dua_tol;
    int max_iter, check_termination, en_state_bound, en_input_bound;
    /* The two terms the reference ships commented out (admm.cpp:20 "+ coeff_d2p * d.col(i)", admm.cpp:79 Uref), off by
     * default.  en_uref: r = -(Uref o R) - rho*(znew - y), the input-side twin of admm.cpp:81-82 (this is what upstream
     * TinyMPC computes; the line commented out at admm.cpp:79 does not compile as written).  en_coeff_d2p: the
     * expression of admm.cpp:20 with its trailing comment removed.  The reference has no executable form of either, so
     * they are pinned against Eigen evaluating these expressions over the reference's types (oracle/ref_terms_shim.cpp). */
    int en_uref, en_coeff_d2p;
    const REAL *R;         /* nu */
    const REAL *coeff_d2p; /* nx x nu (types.hpp:33) */
} SUF(OracleProblem);

typedef struct SUF(OracleWork)
{
    /* TinyWorkspace (types.hpp:52-97); nx x N or nu x (N-1), column-major */
    REAL *x, *u, *q, *r, *p, *d, *v, *vnew, *z, *znew, *g, *y;
    const REAL *u_min, *u_max, *x_min, *x_max, *Xref;
    REAL primal_residual_state, primal_residual_input;
    REAL dual_residual_state, dual_residual_input;
    int status, iter;
    const REAL *Uref; /* nu x (N-1) (types.hpp:93); read only when en_uref */
} SUF(OracleWork);

/*
 * Inner products in the summation order the compiled reference uses.
 *
 * The reference's arithmetic is Eigen 3.4.90 expression templates (vendored under
 * /root/reference/include/Eigen).  Which reduction order Eigen picks is a compile-time
 * decision that depends on storage order, sizes and the SSE2 packet size PS (4 floats /
 * 2 doubles); the rules below restate it (Eigen/src/Core/Redux.h: redux_novec_unroller,
 * redux_vec_unroller, redux_impl; Eigen/src/Core/ProductEvaluators.h: product_evaluator
 * CanVectorizeLhs/CanVectorizeRhs, etor_product_packet_impl; Eigen/src/Core/GeneralProduct.h:
 * product_type_selector) and tests/test_oracle_vs_ref.py pins them bit-for-bit against the
 * compiled reference for every configuration built under oracle/_ref.
 *
 *   seq    : ((a0b0 + a1b1) + a2b2) + ...          packet-wise evaluated lazy products
 *   novec  : halving tree  T(lo,n) = T(lo,n/2) + T(lo+n/2, n-n/2)   (complete unrolling,
 *            taken while 3n-1 <= EIGEN_UNROLLING_LIMIT=110; verified empirically up to n=32), else seq
 *   vec    : products grouped in packets of PS, packets summed by the same halving tree,
 *            then predux = (s0+s2)+(s1+s3) [PS=4] or s0+s1 [PS=2], then the n%PS leftover
 *            (halving tree) is added; falls back to novec when n < PS
 */
#define ORACLE_UNROLL_LIMIT 110

static REAL SUF(tree_sum)(const REAL *v, int n)
{
    if (n == 1) return v[0];
    int h = n / 2;
    return SUF(tree_sum)(v, h) + SUF(tree_sum)(v + h, n - h);
}

static void SUF(ptree_sum)(const REAL *v, int npk, REAL *out) /* v: npk packets of PS */
{
    if (npk == 1) { for (int l = 0; l < PS; l++) out[l] = v[l]; return; }
    int h = npk / 2;
    REAL a[PS], b[PS];
    SUF(ptree_sum)(v, h, a);
    SUF(ptree_sum)(v + (size_t)h * PS, npk - h, b);
    for (int l = 0; l < PS; l++) out[l] = a[l] + b[l];
}

/* strided/contiguous products -> tmp, then reduce */
static REAL SUF(dot_seq)(const REAL *a, int sa, const REAL *b, int n)
{
    REAL acc = a[0] * b[0];
    for (int k = 1; k < n; k++) acc = acc + a[(size_t)k * sa] * b[k];
    return acc;
}

static REAL SUF(dot_novec)(const REAL *a, int sa, const REAL *b, int n)
{
    if (3 * n - 1 > ORACLE_UNROLL_LIMIT || n > ORACLE_MAX_DIM) return SUF(dot_seq)(a, sa, b, n);
    REAL t[ORACLE_MAX_DIM];
    for (int k = 0; k < n; k++) t[k] = a[(size_t)k * sa] * b[k];
    return SUF(tree_sum)(t, n);
}

static REAL SUF(dot_vec)(const REAL *a, const REAL *b, int n) /* both contiguous */
{
    if (n < PS) return SUF(dot_novec)(a, 1, b, n);
    if (3 * n - 1 > ORACLE_UNROLL_LIMIT * PS || n > ORACLE_MAX_DIM) return SUF(dot_seq)(a, 1, b, n); /* not emulated */
    REAL t[ORACLE_MAX_DIM], s[PS];
    for (int k = 0; k < n; k++) t[k] = a[k] * b[k];
    int npk = n / PS, vs = npk * PS;
    SUF(ptree_sum)(t, npk, s);
    REAL res = (PS == 4) ? ((s[0] + s[2 % PS]) + (s[1] + s[3 % PS])) : (s[0] + s[1]);
    if (vs != n) res = res + SUF(tree_sum)(t + vs, n - vs);
    return res;
}

/* row-major GEMV kernel order (Eigen general_matrix_vector_product, RowMajor lhs), used by the
 * reference only when both dims are >= 8: packets accumulated sequentially from zero, predux,
 * scalar leftover, then res = 0 + 1*acc. */
static REAL SUF(dot_gemv_rm)(const REAL *a, const REAL *b, int n)
{
    REAL c[PS];
    for (int l = 0; l < PS; l++) c[l] = 0;
    int vs = (n / PS) * PS;
    for (int k = 0; k < vs; k += PS)
        for (int l = 0; l < PS; l++) c[l] = c[l] + a[k + l] * b[k + l];
    REAL res = (PS == 4) ? ((c[0] + c[2 % PS]) + (c[1] + c[3 % PS])) : (c[0] + c[1]);
    for (int k = vs; k < n; k++) res = res + a[k] * b[k];
    return (REAL)0 + (REAL)1 * res;
}

/* (row i of a column-major rows x cols matrix) . xin, for a lazy product whose result has `rows`
 * rows: packet-evaluated (seq) when rows is a multiple of PS, else coefficient-wise (novec).
 * NOT RESTATED: rows > PS with rows % PS != 0 (e.g. nu = 7 in fp32).  There Eigen's LinearVectorized assignment
 * packet-evaluates (seq) only a window of rows starting at the first 16-byte aligned element of the destination column
 * and coefficient-evaluates (novec) the rest, so the order of u.col(i) depends on i mod PS and on the offset of the
 * member inside TinyWorkspace (measured against the compiled reference for nx = 20, nu = 7).  oracle.py refuses such
 * dimensions; every configuration of the reference's examples and of BASELINE.json has nx, nu in {1, 3, 4, 8, 12, 16, 32}. */
static inline REAL SUF(row_dot)(const REAL *M, int rows, int cols, int i, const REAL *xin)
{
    if (rows > 1 && rows % PS == 0) return SUF(dot_seq)(M + i, rows, xin, cols);
    if (rows == 1) return SUF(dot_vec)(M, xin, cols); /* 1 x cols matrices are stored row-major: contiguous */
    return SUF(dot_novec)(M + i, rows, xin, cols);
}

/* src/tinympc/admm.cpp:27-37 */
void SUF(oracle_forward_pass)(const SUF(OracleProblem) * P, SUF(OracleWork) * W)
{
    const int nx = P->nx, nu = P->nu, N = P->N;
    for (int i = 0; i < N - 1; i++)
    {
        const REAL *xi = W->x + (size_t)i * nx;
        REAL *ui = W->u + (size_t)i * nu;
        const REAL *di = W->d + (size_t)i * nu;
        REAL *xn = W->x + (size_t)(i + 1) * nx;
        /* u_i = -Kinf*x_i - d_i   (admm.cpp:31) */
        for (int j = 0; j < nu; j++)
            ui[j] = ST(-SUF(row_dot)(P->Kinf, nu, nx, j, xi) - di[j]);
        /* x_{i+1} = Adyn*x_i + Bdyn*u_i   (admm.cpp:35) */
        for (int j = 0; j < nx; j++)
            xn[j] = ST(SUF(row_dot)(P->Adyn, nx, nx, j, xi) + SUF(row_dot)(P->Bdyn, nx, nu, j, ui));
    }
}

/* src/tinympc/admm.cpp:45-61 */
void SUF(oracle_update_slack)(const SUF(OracleProblem) * P, SUF(OracleWork) * W)
{
    const int nxt = P->nx * P->N, nut = P->nu * (P->N - 1);
    for (int e = 0; e < nut; e++)
        W->znew[e] = ST(W->u[e] + W->y[e]); /* :47 */
    for (int e = 0; e < nxt; e++)
        W->vnew[e] = ST(W->x[e] + W->g[e]); /* :48 */
    if (P->en_input_bound)              /* :51-54  u_max.cwiseMin(u_min.cwiseMax(znew)) */
        for (int e = 0; e < nut; e++)
        {
            REAL t = W->znew[e];
            t = (W->u_min[e] < t) ? t : W->u_min[e]; /* cwiseMax(u_min, znew) */
            t = (t < W->u_max[e]) ? t : W->u_max[e]; /* cwiseMin(u_max, .)    */
            W->znew[e] = t;
        }
    if (P->en_state_bound) /* :57-60 */
        for (int e = 0; e < nxt; e++)
        {
            REAL t = W->vnew[e];
            t = (W->x_min[e] < t) ? t : W->x_min[e];
            t = (t < W->x_max[e]) ? t : W->x_max[e];
            W->vnew[e] = t;
        }
}

/* src/tinympc/admm.cpp:67-71 */
void SUF(oracle_update_dual)(const SUF(OracleProblem) * P, SUF(OracleWork) * W)
{
    const int nxt = P->nx * P->N, nut = P->nu * (P->N - 1);
    /* STD: the storage rounding of the DUALS; == ST except in the _h16d instantiation (fp16 primal arrays, fp32 duals) */
    for (int e = 0; e < nut; e++)
        W->y[e] = STD(W->y[e] + W->u[e] - W->znew[e]);
    for (int e = 0; e < nxt; e++)
        W->g[e] = STD(W->g[e] + W->x[e] - W->vnew[e]);
}

/* src/tinympc/admm.cpp:77-85 */
void SUF(oracle_update_linear_cost)(const SUF(OracleProblem) * P, SUF(OracleWork) * W)
{
    const int nx = P->nx, nu = P->nu, N = P->N;
    const int nxt = nx * N, nut = nu * (N - 1);
    const REAL rho = P->rho;
    if (P->en_uref)
    {
        /* r = -(Uref.array().colwise() * R.array());  r.noalias() -= rho * (znew - y);   (cf. :81-82 for q) */
        for (int j = 0; j < N - 1; j++)
            for (int i = 0; i < nu; i++)
                W->r[(size_t)j * nu + i] = ST(-(W->Uref[(size_t)j * nu + i] * P->R[i]));
        for (int e = 0; e < nut; e++)
            W->r[e] = ST(W->r[e] - rho * (W->znew[e] - W->y[e]));
    }
    else
        for (int e = 0; e < nut; e++)
            W->r[e] = ST(-rho * (W->znew[e] - W->y[e])); /* :80 */
    for (int j = 0; j < N; j++)                  /* :81  q(i,j) = -(Xref(i,j)*Q(i)) */
        for (int i = 0; i < nx; i++)
            W->q[(size_t)j * nx + i] = ST(-(W->Xref[(size_t)j * nx + i] * P->Q[i]));
    for (int e = 0; e < nxt; e++)
        W->q[e] = ST(W->q[e] - rho * (W->vnew[e] - W->g[e])); /* :82 */
    /* :83  p.col(N-1) = -(Xref.col(N-1)^T * Pinf)  => p_j = -(sum_k Xref_k * Pinf(k,j)) */
    {
        const REAL *xr = W->Xref + (size_t)(N - 1) * nx;
        REAL *pN = W->p + (size_t)(N - 1) * nx;
        for (int j = 0; j < nx; j++)
            pN[j] = ST(-SUF(dot_vec)(xr, P->Pinf + (size_t)j * nx, nx)); /* row-vector lazy product: coefficient-wise */
        /* :84 */
        for (int j = 0; j < nx; j++)
            pN[j] = ST(pN[j] - rho * (W->vnew[(size_t)(N - 1) * nx + j] - W->g[(size_t)(N - 1) * nx + j]));
    }
}

/* cwiseAbs clears the sign bit: |-0| = +0 (the sign of a zero residual is observable in the workspace) */
static inline REAL SUF(absr)(REAL a) { return (REAL)__builtin_fabs((double)a); }

/* src/tinympc/admm.cpp:91-109 */
int SUF(oracle_termination_condition)(const SUF(OracleProblem) * P, SUF(OracleWork) * W)
{
    if (W->iter % P->check_termination == 0) /* :93 */
    {
        const int nxt = P->nx * P->N, nut = P->nu * (P->N - 1);
        REAL m;
        m = SUF(absr)(W->x[0] - W->vnew[0]);
        for (int e = 1; e < nxt; e++) { REAL a = SUF(absr)(W->x[e] - W->vnew[e]); if (a > m) m = a; }
        W->primal_residual_state = m; /* :95 */
        m = SUF(absr)(W->v[0] - W->vnew[0]);
        for (int e = 1; e < nxt; e++) { REAL a = SUF(absr)(W->v[e] - W->vnew[e]); if (a > m) m = a; }
        W->dual_residual_state = m * P->rho; /* :96 */
        m = SUF(absr)(W->u[0] - W->znew[0]);
        for (int e = 1; e < nut; e++) { REAL a = SUF(absr)(W->u[e] - W->znew[e]); if (a > m) m = a; }
        W->primal_residual_input = m; /* :97 */
        m = SUF(absr)(W->z[0] - W->znew[0]);
        for (int e = 1; e < nut; e++) { REAL a = SUF(absr)(W->z[e] - W->znew[e]); if (a > m) m = a; }
        W->dual_residual_input = m * P->rho; /* :98 */
        if (W->primal_residual_state < P->abs_pri_tol && W->primal_residual_input < P->abs_pri_tol &&
            W->dual_residual_state < P->abs_dua_tol && W->dual_residual_input < P->abs_dua_tol) /* :100-103 */
            return 1;
    }
    return 0;
}

/* src/tinympc/admm.cpp:15-22 */
void SUF(oracle_backward_pass_grad)(const SUF(OracleProblem) * P, SUF(OracleWork) * W)
{
    const int nx = P->nx, nu = P->nu, N = P->N;
    REAL tmp[ORACLE_MAX_DIM];
    const int gemv = (nu >= 8 && nx >= 8); /* product_type_selector<Large,1,Large> = GemvProduct */
    const int p_packet = (nu == 1 && nx % PS == 0);
    for (int i = N - 2; i >= 0; i--)
    {
        const REAL *pn = W->p + (size_t)(i + 1) * nx;
        const REAL *ri = W->r + (size_t)i * nu;
        REAL *di = W->d + (size_t)i * nu;
        REAL *pi = W->p + (size_t)i * nx;
        const REAL *qi = W->q + (size_t)i * nx;
        /* d_i = Quu_inv * (Bdyn^T * p_{i+1} + r_i)   (admm.cpp:19): the inner product is evaluated
         * into a temporary first (column j of Bdyn is contiguous), then the nu x nu product. */
        for (int j = 0; j < nu; j++)
        {
            const REAL *bj = P->Bdyn + (size_t)j * nx;
            tmp[j] = (gemv ? SUF(dot_gemv_rm)(bj, pn, nx) : SUF(dot_vec)(bj, pn, nx)) + ri[j];
        }
        for (int j = 0; j < nu; j++)
        {
            if (gemv) di[j] = ST((REAL)0 + (REAL)1 * ((REAL)0 + SUF(dot_seq)(P->Quu_inv + j, nu, tmp, nu)));
            else      di[j] = ST(SUF(row_dot)(P->Quu_inv, nu, nu, j, tmp));
        }
        /* p_i = q_i + AmBKt*p_{i+1} - Kinf^T*r_i   (admm.cpp:20; the coeff_d2p term is commented out there).
         * Kinf^T is a row-major view, so unless nu == 1 the expression is evaluated coefficient-wise. */
        for (int j = 0; j < nx; j++)
        {
            REAL a = p_packet ? SUF(dot_seq)(P->AmBKt + j, nx, pn, nx) : SUF(dot_novec)(P->AmBKt + j, nx, pn, nx);
            REAL k = SUF(dot_vec)(P->Kinf + (size_t)j * nu, ri, nu);
            pi[j] = ST(qi[j] + a - k);
        }
        /* "+ coeff_d2p * d.col(i)" (the trailing comment of admm.cpp:20): Eigen assigns the expression above to p.col(i)
         * first and then adds the product, evaluated into a temporary in sequential order (pinned by ref_terms_shim) */
        if (P->en_coeff_d2p)
            for (int j = 0; j < nx; j++)
                pi[j] = ST(pi[j] + SUF(dot_seq)(P->coeff_d2p + j, nx, di, nu));
    }
}

/* src/tinympc/admm.cpp:111-152 */
int SUF(oracle_tiny_solve)(const SUF(OracleProblem) * P, SUF(OracleWork) * W)
{
    const int nxt = P->nx * P->N, nut = P->nu * (P->N - 1);
    W->status = 11; /* TINY_UNSOLVED  :114 */
    W->iter = 1;    /* :115 */
    for (int i = 0; i < P->max_iter; i++)
    {
        W->iter = i + 1;                          /* :120 */
        SUF(oracle_forward_pass)(P, W);           /* :123 */
        SUF(oracle_update_slack)(P, W);           /* :126 */
        SUF(oracle_update_dual)(P, W);            /* :129 */
        SUF(oracle_update_linear_cost)(P, W);     /* :132 */
        if (SUF(oracle_termination_condition)(P, W)) /* :135 */
        {
            W->status = 1; /* TINY_SOLVED */
            return 0;      /* returns BEFORE the v/z copy and the backward pass */
        }
        for (int e = 0; e < nxt; e++) W->v[e] = W->vnew[e]; /* :141 */
        for (int e = 0; e < nut; e++) W->z[e] = W->znew[e]; /* :142 */
        SUF(oracle_backward_pass_grad)(P, W);                /* :144 */
    }
    return 1; /* :151 */
}

/*
 * Plant step of the reference's closed-loop examples (examples/quadrotor_hovering.cpp:110-111,
 * examples/quadrotor_tracking.cpp:116-117):   x1 = work.Adyn * x0 + work.Bdyn * work.u.col(0);
 * Eigen evaluates "dst = prod1 + prod2" as  dst = prod1;  dst += prod2  (ProductEvaluators.h,
 * assignment_from_xpr_op_product), and each product by its own product_type_selector (GeneralProduct.h): rows >= 8 and
 * depth >= 8 (EIGEN_CACHEFRIENDLY_PRODUCT_THRESHOLD) -> GemvProduct = the column-major kernel of
 * products/GeneralMatrixVector.h:108-260: per result row an accumulator that starts at ZERO, the products added to it in
 * ascending column order (pmadd without FMA on SSE2 = multiply, then add), finally res = acc*alpha + res with alpha = 1 and
 * res = 0 (dst.setZero() of evalTo) or the value already in dst (addTo).  Otherwise the coefficient-based lazy product
 * of row_dot() above.  Pinned bit for bit against oracle/ref_shim.cpp: ref_plant_step (tests/test_oracle.py).
 */
static REAL SUF(dot_gemv_cm)(const REAL *M, int rows, int cols, int i, const REAL *xin, REAL res)
{
    REAL c = 0;
    for (int j = 0; j < cols; j++) c = M[(size_t)j * rows + i] * xin[j] + c;
    return c * (REAL)1 + res;
}

void SUF(oracle_plant_step)(const SUF(OracleProblem) * P, const REAL *x0, const REAL *u0, REAL *x1)
{
    const int nx = P->nx, nu = P->nu;
    REAL out[ORACLE_MAX_DIM];
    for (int i = 0; i < nx; i++)
    {
        const REAL a = (nx >= 8) ? SUF(dot_gemv_cm)(P->Adyn, nx, nx, i, x0, (REAL)0) : SUF(row_dot)(P->Adyn, nx, nx, i, x0);
        out[i] = (nx >= 8 && nu >= 8) ? SUF(dot_gemv_cm)(P->Bdyn, nx, nu, i, u0, a) : a + SUF(row_dot)(P->Bdyn, nx, nu, i, u0);
    }
    for (int i = 0; i < nx; i++) x1[i] = out[i];
}

void SUF(oracle_plant_step_batch)(const SUF(OracleProblem) * P, int batch, const REAL *x0, const REAL *u0, REAL *x1)
{
    for (int b = 0; b < batch; b++)
        SUF(oracle_plant_step)(P, x0 + (size_t)b * P->nx, u0 + (size_t)b * P->nu, x1 + (size_t)b * P->nx);
}

/*
 * Batched convenience driver (the batch is our addition; the reference has none).
 * Host-visible layout = array of reference-layout instances: (B, N, nx) / (B, N-1, nu),
 * instance-major.  Bounds / Xref may be per-instance or shared (stride 0).
 * Exactly one tiny_solve per instance.
 * Returns the number of instances that hit max_iter.
 */
typedef struct SUF(OracleBatch)
{
    int batch;
    REAL *x, *u, *q, *r, *p, *d, *v, *vnew, *z, *znew, *g, *y; /* (B, N, nx) or (B, N-1, nu) */
    const REAL *u_min, *u_max, *x_min, *x_max, *Xref;
    long long bound_stride_x, bound_stride_u, xref_stride; /* element stride between instances, 0 = shared */
    REAL *residuals; /* (B,4): pri_state, pri_input, dua_state, dua_input */
    int *status, *iter;
    const REAL *Uref; /* (B, N-1, nu) or shared; may be NULL unless en_uref */
    long long uref_stride;
} SUF(OracleBatch);

int SUF(oracle_solve_batch)(const SUF(OracleProblem) * P, SUF(OracleBatch) * Bt, int nthreads)
{
    const long long sx = (long long)P->nx * P->N, su = (long long)P->nu * (P->N - 1);
    int unsolved = 0;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads) reduction(+ : unsolved)
    {
    oracle_apply_fp_mode();
#pragma omp for schedule(static)
    for (int b = 0; b < Bt->batch; b++)
    {
        SUF(OracleWork) W;
        W.x = Bt->x + b * sx; W.q = Bt->q + b * sx; W.p = Bt->p + b * sx;
        W.v = Bt->v + b * sx; W.vnew = Bt->vnew + b * sx; W.g = Bt->g + b * sx;
        W.u = Bt->u + b * su; W.r = Bt->r + b * su; W.d = Bt->d + b * su;
        W.z = Bt->z + b * su; W.znew = Bt->znew + b * su; W.y = Bt->y + b * su;
        W.u_min = Bt->u_min + b * Bt->bound_stride_u; W.u_max = Bt->u_max + b * Bt->bound_stride_u;
        W.x_min = Bt->x_min + b * Bt->bound_stride_x; W.x_max = Bt->x_max + b * Bt->bound_stride_x;
        W.Xref = Bt->Xref + b * Bt->xref_stride;
        W.Uref = Bt->Uref ? Bt->Uref + b * Bt->uref_stride : 0;
        W.primal_residual_state = Bt->residuals[4 * b + 0];
        W.primal_residual_input = Bt->residuals[4 * b + 1];
        W.dual_residual_state = Bt->residuals[4 * b + 2];
        W.dual_residual_input = Bt->residuals[4 * b + 3];
        W.status = Bt->status[b]; W.iter = Bt->iter[b];
        unsolved += SUF(oracle_tiny_solve)(P, &W);
        Bt->residuals[4 * b + 0] = W.primal_residual_state;
        Bt->residuals[4 * b + 1] = W.primal_residual_input;
        Bt->residuals[4 * b + 2] = W.dual_residual_state;
        Bt->residuals[4 * b + 3] = W.dual_residual_input;
        Bt->status[b] = W.status; Bt->iter[b] = W.iter;
    }
    }
    return unsolved;
}
