// tinympc_batch.hip — host side of the C-ABI declared in include/tinympc_batch.h:
// device-resident batched workspace, layout conversion kernels, MFMA operand packing and
// kernel dispatch.  The solver kernels live in admm_stream.hip / admm_resident.hip.
//
// There is deliberately NO CPU fallback in this library: every entry point that computes
// launches a HIP kernel and reports HIP failures as TINY_BATCH_EHIP.
#include "../../include/tinympc_batch.h"
#include "tinympc_internal.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

using namespace tinympc;

namespace
{

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do                                                                                             \
    {                                                                                              \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(TINY_BATCH_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

#define CHECK_TB(tb) \
    if (!(tb)) return fail(TINY_BATCH_EINVAL, "%s: NULL TinyBatch handle", __func__)
#define CHECK_PTR(p) \
    if (!(p)) return fail(TINY_BATCH_EINVAL, "%s: NULL pointer argument '%s'", __func__, #p)

// ---------------------------------------------------------------------------------------------
// layout conversion: host-visible [B][steps][dim]  <->  tile layout [ntiles][tsteps][64][NC]
// ---------------------------------------------------------------------------------------------
// dst tile array gets steps [step0, step0+nsteps) from src ([Bsrc][nsteps][dim]; Bsrc==1 => shared).
__global__ void pack_kernel(const float *__restrict__ src, float *__restrict__ dst, int batch, int shared, int dim,
                            int NC, int ntiles_dst, int tsteps, int step0, int nsteps)
{
    const long long total = (long long)ntiles_dst * nsteps * WAVE * NC;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x)
    {
        int ch = (int)(e % NC);
        long long t = e / NC;
        int lane = (int)(t % WAVE); t /= WAVE;
        int s = (int)(t % nsteps);
        int tile = (int)(t / nsteps);
        int b = tile * TILE + (lane & 15);
        int row = 4 * ch + (lane >> 4);
        float val = 0.f;
        if (row < dim && (shared || b < batch))
            val = src[((long long)(shared ? 0 : b) * nsteps + s) * dim + row];
        dst[(((long long)tile * tsteps + step0 + s) * WAVE + lane) * NC + ch] = val;
    }
}

__global__ void unpack_kernel(const float *__restrict__ src, float *__restrict__ dst, int batch, int dim, int NC,
                              int tsteps, int step0, int nsteps)
{
    const long long total = (long long)batch * nsteps * dim;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x)
    {
        int row = (int)(e % dim);
        long long t = e / dim;
        int s = (int)(t % nsteps);
        int b = (int)(t / nsteps);
        int tile = b / TILE, c = b % TILE;
        int lane = (row & 3) * 16 + c, ch = row >> 2;
        dst[e] = src[(((long long)tile * tsteps + step0 + s) * WAVE + lane) * NC + ch];
    }
}

// x0 <- Adyn*x0 + Bdyn*u.col(0)   (quadrotor_hovering.cpp:110-111), and x.col(0) <- x0 (:95).
// One thread per instance; matrices column-major in global memory (tiny, cache resident).
__global__ void plant_step_kernel(float *__restrict__ x0buf, float *__restrict__ xtile, const float *__restrict__ utile,
                                  const float *__restrict__ A, const float *__restrict__ Bm, int *__restrict__ wstart,
                                  int window_advance, int batch, int nx, int nu, int NXC, int NUC, int N)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    int tile = b / TILE, c = b % TILE;
    const float *x0 = x0buf + (long long)b * nx;
    float xn[64];
    for (int i = 0; i < nx; i++)
    {
        float acc = 0.f;
        for (int k = 0; k < nx; k++) acc += A[k * nx + i] * x0[k];
        float acc2 = 0.f;
        for (int m = 0; m < nu; m++)
        {
            float um = utile[(((long long)tile * (N - 1) + 0) * WAVE + ((m & 3) * 16 + c)) * NUC + (m >> 2)];
            acc2 += Bm[m * nx + i] * um;
        }
        xn[i] = acc + acc2;
    }
    for (int i = 0; i < nx; i++)
    {
        x0buf[(long long)b * nx + i] = xn[i];
        xtile[(((long long)tile * N + 0) * WAVE + ((i & 3) * 16 + c)) * NXC + (i >> 2)] = xn[i];
    }
    if (wstart && window_advance) wstart[b] += window_advance;
}

__global__ void gather_u0_kernel(const float *__restrict__ utile, float *__restrict__ u0, int batch, int nu, int NUC, int N)
{
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= batch * nu) return;
    int b = e / nu, m = e % nu;
    int tile = b / TILE, c = b % TILE;
    u0[e] = utile[(((long long)tile * (N - 1)) * WAVE + ((m & 3) * 16 + c)) * NUC + (m >> 2)];
}

int grid_for(long long total, int block = 256)
{
    long long g = (total + block - 1) / block;
    if (g > 8192) g = 8192;
    if (g < 1) g = 1;
    return (int)g;
}

} // namespace

// ---------------------------------------------------------------------------------------------
struct TinyBatch
{
    int nx = 0, nu = 0, N = 0, batch = 0, device = 0;
    int NXC = 0, NUC = 0, ntiles = 0;
    hipStream_t stream = nullptr;
    // problem class
    bool have_cache = false, have_dyn = false, have_settings = false, operands_dirty = true;
    float rho = 0.f;
    std::vector<float> Kinf, Pinf, Quu_inv, AmBKt, Adyn, Bdyn, Q;
    float abs_pri_tol = 0.f, abs_dua_tol = 0.f;
    int max_iter = 0, check_termination = 1, en_state_bound = 0, en_input_bound = 0;
    // device memory
    float *arr[TINY_ARR_COUNT] = {};     // tile layout
    float *xmin = nullptr, *xmax = nullptr, *umin = nullptr, *umax = nullptr, *xref = nullptr;
    bool xb_shared[2] = {true, true}, ub_shared[2] = {true, true}, xref_shared = true;
    size_t xfam_floats = 0, ufam_floats = 0; // per full-batch tile array
    float *xref_table = nullptr;
    int table_rows = 0;
    int *xref_start = nullptr;
    int xref_mode = 0;
    float *res = nullptr;
    int *status = nullptr, *iter = nullptr, *n_unsolved = nullptr;
    float *opnd = nullptr, *qvec = nullptr;
    float *dA = nullptr, *dB = nullptr; // column-major copies for the plant step
    float *x0buf = nullptr;             // [B][nx] host-layout current state (closed loop)
    float *staging = nullptr;           // host-layout staging for pack/unpack
    size_t staging_floats = 0;
    bool duals_zero_pending = false;
    bool cold_pending = false; // reset_workspace() folded into the next solve (d,v,z,y,g read as zero)
    int variant = 0;
    bool timing = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool ev_valid = false;
    std::string kname;
};

namespace
{

bool is_xfam(int id)
{
    return id == TINY_ARR_X || id == TINY_ARR_Q || id == TINY_ARR_P || id == TINY_ARR_V || id == TINY_ARR_VNEW ||
           id == TINY_ARR_G;
}

int set_device(TinyBatch *tb)
{
    HIP_TRY(hipSetDevice(tb->device));
    return 0;
}

// upload a host-layout array ([Bsrc][nsteps][dim]) into steps [step0, step0+nsteps) of a tile array
int upload_packed(TinyBatch *tb, const float *host, float *dst, bool xfam, bool shared, int step0, int nsteps)
{
    const int dim = xfam ? tb->nx : tb->nu, NC = xfam ? tb->NXC : tb->NUC;
    const int tsteps = xfam ? tb->N : tb->N - 1;
    const size_t n = (size_t)(shared ? 1 : tb->batch) * nsteps * dim;
    if (n > tb->staging_floats) return fail(TINY_BATCH_EINVAL, "internal: staging too small");
    HIP_TRY(hipMemcpyAsync(tb->staging, host, n * sizeof(float), hipMemcpyHostToDevice, tb->stream));
    const int ntd = shared ? 1 : tb->ntiles;
    const long long total = (long long)ntd * nsteps * WAVE * NC;
    hipLaunchKernelGGL(pack_kernel, dim3(grid_for(total)), dim3(256), 0, tb->stream, tb->staging, dst, tb->batch,
                       shared ? 1 : 0, dim, NC, ntd, tsteps, step0, nsteps);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(tb->stream)); // staging and `host` are reusable on return
    return 0;
}

int download_unpacked(TinyBatch *tb, const float *src, float *host, bool xfam, int step0, int nsteps)
{
    const int dim = xfam ? tb->nx : tb->nu, NC = xfam ? tb->NXC : tb->NUC;
    const int tsteps = xfam ? tb->N : tb->N - 1;
    const size_t n = (size_t)tb->batch * nsteps * dim;
    if (n > tb->staging_floats) return fail(TINY_BATCH_EINVAL, "internal: staging too small");
    hipLaunchKernelGGL(unpack_kernel, dim3(grid_for((long long)n)), dim3(256), 0, tb->stream, src, tb->staging,
                       tb->batch, dim, NC, tsteps, step0, nsteps);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(host, tb->staging, n * sizeof(float), hipMemcpyDeviceToHost, tb->stream));
    HIP_TRY(hipStreamSynchronize(tb->stream));
    return 0;
}

// Materialise a pending reset_dual_variables() (needed before anything other than a solve looks at y/g).
int flush_pending(TinyBatch *tb)
{
    if (tb->cold_pending)
    {
        // every work array reads as zero after reset_workspace(), except x.col(0) which carries x0
        for (int id = 0; id < TINY_ARR_COUNT; id++)
            if (id != TINY_ARR_X)
                HIP_TRY(hipMemsetAsync(tb->arr[id], 0, (is_xfam(id) ? tb->xfam_floats : tb->ufam_floats) * sizeof(float), tb->stream));
        {
            const size_t step_bytes = (size_t)WAVE * tb->NXC * sizeof(float);
            HIP_TRY(hipMemset2DAsync(tb->arr[TINY_ARR_X] + (size_t)WAVE * tb->NXC, step_bytes * tb->N, 0,
                                     step_bytes * (tb->N - 1), tb->ntiles, tb->stream));
        }
        tb->cold_pending = false;
        tb->duals_zero_pending = false;
    }
    if (tb->duals_zero_pending)
    {
        HIP_TRY(hipMemsetAsync(tb->arr[TINY_ARR_Y], 0, tb->ufam_floats * sizeof(float), tb->stream));
        HIP_TRY(hipMemsetAsync(tb->arr[TINY_ARR_G], 0, tb->xfam_floats * sizeof(float), tb->stream));
        tb->duals_zero_pending = false;
    }
    return 0;
}

// ---- MFMA A-operand packing -----------------------------------------------------------------
// Stacked vector s = [x ; u] in chunks of 4 rows; chunk ch, in-chunk row g  <->  x row 4ch+g (ch < NXC)
// or u row 4(ch-NXC)+g.  For v_mfma_f32_16x16x4_f32 the A operand of lane l is A[i = l&15][k = l>>4];
// the D row i of output tile t is held by lane group i>>2 in register i&3, which we DEFINE to be
// chunk 4t + (i&3), in-chunk row i>>2.  So:  A_{t,ch_in}[l] = M[ out(4t + (i&3), i>>2) ][ in(ch_in, l>>4) ].
struct RowRef { int fam; int idx; }; // fam 0 = x row, 1 = u row, -1 = padding

RowRef stacked_row(const TinyBatch *tb, int ch, int g)
{
    if (ch < tb->NXC) { int r = 4 * ch + g; return r < tb->nx ? RowRef{0, r} : RowRef{-1, 0}; }
    int r = 4 * (ch - tb->NXC) + g;
    return (ch < tb->NXC + tb->NUC && r < tb->nu) ? RowRef{1, r} : RowRef{-1, 0};
}

template <class F>
void pack_one(const TinyBatch *tb, std::vector<float> &out, int t_out, int ch_in, F entry)
{
    for (int l = 0; l < WAVE; l++)
    {
        int i = l & 15, kk = l >> 4;
        RowRef o = stacked_row(tb, 4 * t_out + (i & 3), i >> 2);
        RowRef in = stacked_row(tb, ch_in, kk);
        float v = 0.f;
        if (o.fam >= 0 && in.fam >= 0) v = entry(o, in);
        out.push_back(v);
    }
}

int pack_operands(TinyBatch *tb)
{
    const int nx = tb->nx, nu = tb->nu, NXC = tb->NXC, NUC = tb->NUC;
    const int NCH = NXC + NUC, NT = (NCH + 3) / 4, NTX = (NXC + 3) / 4, TU0 = NXC / 4;
    const float *K = tb->Kinf.data(), *Pf = tb->Pinf.data(), *Qi = tb->Quu_inv.data(), *Am = tb->AmBKt.data();
    const float *A = tb->Adyn.data(), *B = tb->Bdyn.data();
    // column-major accessors
    auto Kat = [&](int m, int k) { return K[k * nu + m]; };
    auto Aat = [&](int j, int k) { return A[k * nx + j]; };
    auto Bat = [&](int j, int m) { return B[m * nx + j]; };
    auto Amat = [&](int j, int k) { return Am[k * nx + j]; };
    auto Qiat = [&](int a, int b) { return Qi[b * nu + a]; };
    auto Pat = [&](int k, int j) { return Pf[j * nx + k]; };
    std::vector<float> o;
    // A1: fwd [A ; -K] * x
    for (int t = 0; t < NT; t++)
        for (int k = 0; k < NXC; k++)
            pack_one(tb, o, t, k, [&](RowRef out, RowRef in) {
                if (in.fam != 0) return 0.f;
                return out.fam == 0 ? Aat(out.idx, in.idx) : -Kat(out.idx, in.idx);
            });
    // A2: fwd [B] * u
    for (int t = 0; t < NTX; t++)
        for (int m = 0; m < NUC; m++)
            pack_one(tb, o, t, NXC + m, [&](RowRef out, RowRef in) {
                return (out.fam == 0 && in.fam == 1) ? Bat(out.idx, in.idx) : 0.f;
            });
    // A3: bwd [AmBKt ; B^T] * p
    for (int t = 0; t < NT; t++)
        for (int k = 0; k < NXC; k++)
            pack_one(tb, o, t, k, [&](RowRef out, RowRef in) {
                if (in.fam != 0) return 0.f;
                return out.fam == 0 ? Amat(out.idx, in.idx) : Bat(in.idx, out.idx);
            });
    // A4: bwd [-K^T] * r
    for (int t = 0; t < NTX; t++)
        for (int m = 0; m < NUC; m++)
            pack_one(tb, o, t, NXC + m, [&](RowRef out, RowRef in) {
                return (out.fam == 0 && in.fam == 1) ? -Kat(in.idx, out.idx) : 0.f;
            });
    // A5: bwd [Quu_inv] * (B^T p + r), output tiles TU0..NT-1
    for (int t = TU0; t < NT; t++)
        for (int m = 0; m < NUC; m++)
            pack_one(tb, o, t, NXC + m, [&](RowRef out, RowRef in) {
                return (out.fam == 1 && in.fam == 1) ? Qiat(out.idx, in.idx) : 0.f;
            });
    // AP: terminal  p_j = -(sum_k Xref_k Pinf(k,j))
    for (int t = 0; t < NTX; t++)
        for (int k = 0; k < NXC; k++)
            pack_one(tb, o, t, k, [&](RowRef out, RowRef in) {
                return (out.fam == 0 && in.fam == 0) ? -Pat(in.idx, out.idx) : 0.f;
            });
    std::vector<float> qv((size_t)WAVE * NXC, 0.f);
    for (int l = 0; l < WAVE; l++)
        for (int ch = 0; ch < NXC; ch++)
        {
            int r = 4 * ch + (l >> 4);
            if (r < nx) qv[(size_t)l * NXC + ch] = tb->Q[r];
        }
    if (!tb->opnd) HIP_TRY(hipMalloc(&tb->opnd, o.size() * sizeof(float)));
    if (!tb->qvec) HIP_TRY(hipMalloc(&tb->qvec, qv.size() * sizeof(float)));
    if (!tb->dA) HIP_TRY(hipMalloc(&tb->dA, (size_t)nx * nx * sizeof(float)));
    if (!tb->dB) HIP_TRY(hipMalloc(&tb->dB, (size_t)nx * nu * sizeof(float)));
    HIP_TRY(hipMemcpyAsync(tb->opnd, o.data(), o.size() * sizeof(float), hipMemcpyHostToDevice, tb->stream));
    HIP_TRY(hipMemcpyAsync(tb->qvec, qv.data(), qv.size() * sizeof(float), hipMemcpyHostToDevice, tb->stream));
    HIP_TRY(hipMemcpyAsync(tb->dA, A, (size_t)nx * nx * sizeof(float), hipMemcpyHostToDevice, tb->stream));
    HIP_TRY(hipMemcpyAsync(tb->dB, B, (size_t)nx * nu * sizeof(float), hipMemcpyHostToDevice, tb->stream));
    HIP_TRY(hipStreamSynchronize(tb->stream));
    tb->operands_dirty = false;
    return 0;
}

bool dims_supported(int nxc, int nuc)
{
#define TINY_CHECK_DIMS(NXC, NUC) \
    if (nxc == NXC && nuc == NUC) return true;
    TINY_FOR_EACH_DIMS(TINY_CHECK_DIMS)
    return false;
}

void fill_params(TinyBatch *tb, SolveParams &P)
{
    P.nx = tb->nx; P.nu = tb->nu; P.N = tb->N; P.batch = tb->batch; P.ntiles = tb->ntiles;
    P.rho = tb->rho; P.abs_pri_tol = tb->abs_pri_tol; P.abs_dua_tol = tb->abs_dua_tol;
    P.max_iter = tb->max_iter; P.check_termination = tb->check_termination;
    P.en_state_bound = tb->en_state_bound; P.en_input_bound = tb->en_input_bound;
    P.duals_zero = tb->duals_zero_pending ? 1 : 0;
    P.cold_start = tb->cold_pending ? 1 : 0;
    P.xref_mode = tb->xref_mode;
    P.x = tb->arr[TINY_ARR_X]; P.q = tb->arr[TINY_ARR_Q]; P.p = tb->arr[TINY_ARR_P];
    P.v = tb->arr[TINY_ARR_V]; P.vnew = tb->arr[TINY_ARR_VNEW]; P.g = tb->arr[TINY_ARR_G];
    P.u = tb->arr[TINY_ARR_U]; P.r = tb->arr[TINY_ARR_R]; P.d = tb->arr[TINY_ARR_D];
    P.z = tb->arr[TINY_ARR_Z]; P.znew = tb->arr[TINY_ARR_ZNEW]; P.y = tb->arr[TINY_ARR_Y];
    P.xmin = tb->xmin; P.xmax = tb->xmax; P.umin = tb->umin; P.umax = tb->umax; P.xref = tb->xref;
    const long long xt = (long long)tb->N * WAVE * tb->NXC, ut = (long long)(tb->N - 1) * WAVE * tb->NUC;
    P.xb_tile_stride = (tb->xb_shared[0] && tb->xb_shared[1]) ? 0 : xt;
    P.ub_tile_stride = (tb->ub_shared[0] && tb->ub_shared[1]) ? 0 : ut;
    P.xref_tile_stride = tb->xref_shared ? 0 : xt;
    P.xref_table = tb->xref_table; P.xref_start = tb->xref_start; P.table_rows = tb->table_rows;
    P.res = tb->res; P.status = tb->status; P.iter = tb->iter; P.n_unsolved = tb->n_unsolved;
    P.opnd = tb->opnd; P.qvec = tb->qvec;
}

int launch_solve(TinyBatch *tb)
{
    if (!tb->have_cache || !tb->have_dyn || !tb->have_settings)
        return fail(TINY_BATCH_ENOTREADY, "tiny_batch_solve: set_cache, set_dynamics and set_settings must be called first");
    if (tb->check_termination <= 0)
        return fail(TINY_BATCH_EINVAL, "check_termination must be >= 1 (the reference divides by it, admm.cpp:93)");
    if (set_device(tb)) return TINY_BATCH_EHIP;
    if (tb->operands_dirty)
        if (int rc = pack_operands(tb)) return rc;
    if (tb->max_iter <= 0)
        if (int rc = flush_pending(tb)) return rc;
    // a bound array that is per-instance while its partner is shared: expand the shared one
    SolveParams P;
    fill_params(tb, P);
    HIP_TRY(hipMemsetAsync(tb->n_unsolved, 0, sizeof(int), tb->stream));
    if (tb->timing) HIP_TRY(hipEventRecord(tb->ev0, tb->stream));
    hipError_t e = launch_admm_stream(tb->NXC, tb->NUC, P, tb->stream);
    if (e != hipSuccess) return fail(TINY_BATCH_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
    if (tb->timing)
    {
        HIP_TRY(hipEventRecord(tb->ev1, tb->stream));
        tb->ev_valid = true;
    }
    if (tb->max_iter > 0) tb->duals_zero_pending = tb->cold_pending = false; // consumed by the kernel's first iteration
    return 0;
}

int set_bound(TinyBatch *tb, const float *src, int shared, float **slot, bool xfam, bool *shared_flags, int which)
{
    CHECK_TB(tb);
    CHECK_PTR(src);
    if (set_device(tb)) return TINY_BATCH_EHIP;
    // Both arrays of a pair (min,max) must end up in the same sharing mode (checked at solve time):
    // the kernel walks them with one tile stride.
    const int steps = xfam ? tb->N : tb->N - 1;
    int rc = upload_packed(tb, src, *slot, xfam, shared != 0, 0, steps);
    if (rc) return rc;
    shared_flags[which] = shared != 0;
    return 0;
}

} // namespace

extern "C"
{

const char *tiny_batch_last_error(void) { return g_err.c_str(); }

int tiny_batch_create(TinyBatch **out, int nx, int nu, int N, int batch, int device)
{
    CHECK_PTR(out);
    *out = nullptr;
    if (nx < 1 || nu < 1 || N < 2 || batch < 1)
        return fail(TINY_BATCH_EINVAL, "tiny_batch_create: need nx>=1, nu>=1, N>=2, batch>=1 (got %d,%d,%d,%d)", nx, nu, N, batch);
    const int nxc = (nx + 3) / 4, nuc = (nu + 3) / 4;
    if (!dims_supported(nxc, nuc))
        return fail(TINY_BATCH_EUNSUPPORTED,
                    "no kernel instantiation for nx=%d nu=%d (chunks %d,%d); add it to TINY_FOR_EACH_DIMS", nx, nu, nxc, nuc);
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(TINY_BATCH_EINVAL, "device %d out of range (have %d)", device, ndev);
    TinyBatch *tb = new TinyBatch();
    tb->nx = nx; tb->nu = nu; tb->N = N; tb->batch = batch; tb->device = device;
    tb->NXC = nxc; tb->NUC = nuc; tb->ntiles = (batch + TILE - 1) / TILE;
    tb->xfam_floats = (size_t)tb->ntiles * N * WAVE * nxc;
    tb->ufam_floats = (size_t)tb->ntiles * (N - 1) * WAVE * nuc;
    auto cleanup = [&](int rc) { tiny_batch_destroy(tb); return rc; };
    if (hipSetDevice(device) != hipSuccess) return cleanup(fail(TINY_BATCH_EHIP, "hipSetDevice(%d) failed", device));
#define ALLOC(ptr, nfloats)                                                                           \
    do                                                                                                \
    {                                                                                                 \
        hipError_t e_ = hipMalloc((void **)&(ptr), (size_t)(nfloats) * sizeof(float));                \
        if (e_ == hipSuccess) e_ = hipMemset((ptr), 0, (size_t)(nfloats) * sizeof(float));            \
        if (e_ != hipSuccess) return cleanup(fail(TINY_BATCH_EHIP, "hipMalloc/hipMemset(%zu B) failed: %s", \
                                                  (size_t)(nfloats) * sizeof(float), hipGetErrorString(e_))); \
    } while (0)
    for (int id = 0; id < TINY_ARR_COUNT; id++) ALLOC(tb->arr[id], is_xfam(id) ? tb->xfam_floats : tb->ufam_floats);
    ALLOC(tb->xmin, tb->xfam_floats); ALLOC(tb->xmax, tb->xfam_floats); ALLOC(tb->xref, tb->xfam_floats);
    ALLOC(tb->umin, tb->ufam_floats); ALLOC(tb->umax, tb->ufam_floats);
    ALLOC(tb->res, (size_t)batch * 4);
    ALLOC(tb->status, batch); ALLOC(tb->iter, batch); ALLOC(tb->n_unsolved, 1);
    ALLOC(tb->xref_start, batch);
    ALLOC(tb->x0buf, (size_t)batch * nx);
    tb->staging_floats = (size_t)batch * N * (nx > nu ? nx : nu);
    ALLOC(tb->staging, tb->staging_floats);
#undef ALLOC
    if (hipEventCreate(&tb->ev0) != hipSuccess || hipEventCreate(&tb->ev1) != hipSuccess)
        return cleanup(fail(TINY_BATCH_EHIP, "hipEventCreate failed"));
    char nm[64];
    snprintf(nm, sizeof nm, "stream<%d,%d>", nxc, nuc);
    tb->kname = nm;
    *out = tb;
    return TINY_BATCH_OK;
}

void tiny_batch_destroy(TinyBatch *tb)
{
    if (!tb) return;
    (void)hipSetDevice(tb->device);
    for (int id = 0; id < TINY_ARR_COUNT; id++) (void)hipFree(tb->arr[id]);
    (void)hipFree(tb->xmin); (void)hipFree(tb->xmax); (void)hipFree(tb->umin); (void)hipFree(tb->umax);
    (void)hipFree(tb->xref); (void)hipFree(tb->xref_table); (void)hipFree(tb->xref_start);
    (void)hipFree(tb->res); (void)hipFree(tb->status); (void)hipFree(tb->iter); (void)hipFree(tb->n_unsolved);
    (void)hipFree(tb->opnd); (void)hipFree(tb->qvec); (void)hipFree(tb->dA); (void)hipFree(tb->dB);
    (void)hipFree(tb->x0buf); (void)hipFree(tb->staging);
    if (tb->ev0) (void)hipEventDestroy(tb->ev0);
    if (tb->ev1) (void)hipEventDestroy(tb->ev1);
    delete tb;
}

int tiny_batch_set_stream(TinyBatch *tb, void *hip_stream)
{
    CHECK_TB(tb);
    tb->stream = (hipStream_t)hip_stream;
    return 0;
}

int tiny_batch_synchronize(TinyBatch *tb)
{
    CHECK_TB(tb);
    if (set_device(tb)) return TINY_BATCH_EHIP;
    HIP_TRY(hipStreamSynchronize(tb->stream));
    return 0;
}

int tiny_batch_set_cache(TinyBatch *tb, float rho, const float *Kinf, const float *Pinf, const float *Quu_inv,
                         const float *AmBKt)
{
    CHECK_TB(tb); CHECK_PTR(Kinf); CHECK_PTR(Pinf); CHECK_PTR(Quu_inv); CHECK_PTR(AmBKt);
    const int nx = tb->nx, nu = tb->nu;
    tb->rho = rho;
    tb->Kinf.assign(Kinf, Kinf + (size_t)nu * nx);
    tb->Pinf.assign(Pinf, Pinf + (size_t)nx * nx);
    tb->Quu_inv.assign(Quu_inv, Quu_inv + (size_t)nu * nu);
    tb->AmBKt.assign(AmBKt, AmBKt + (size_t)nx * nx);
    tb->have_cache = true;
    tb->operands_dirty = true;
    return 0;
}

int tiny_batch_set_dynamics(TinyBatch *tb, const float *Adyn, const float *Bdyn, const float *Q)
{
    CHECK_TB(tb); CHECK_PTR(Adyn); CHECK_PTR(Bdyn); CHECK_PTR(Q);
    const int nx = tb->nx, nu = tb->nu;
    tb->Adyn.assign(Adyn, Adyn + (size_t)nx * nx);
    tb->Bdyn.assign(Bdyn, Bdyn + (size_t)nx * nu);
    tb->Q.assign(Q, Q + nx);
    tb->have_dyn = true;
    tb->operands_dirty = true;
    return 0;
}

int tiny_batch_set_settings(TinyBatch *tb, float abs_pri_tol, float abs_dua_tol, int max_iter, int check_termination,
                            int en_state_bound, int en_input_bound)
{
    CHECK_TB(tb);
    if (check_termination < 1)
        return fail(TINY_BATCH_EINVAL, "check_termination must be >= 1 (the reference computes iter %% check_termination, admm.cpp:93)");
    tb->abs_pri_tol = abs_pri_tol; tb->abs_dua_tol = abs_dua_tol;
    tb->max_iter = max_iter; tb->check_termination = check_termination;
    tb->en_state_bound = en_state_bound; tb->en_input_bound = en_input_bound;
    tb->have_settings = true;
    return 0;
}

int tiny_batch_set_x0(TinyBatch *tb, const float *x0)
{
    CHECK_TB(tb); CHECK_PTR(x0);
    if (set_device(tb)) return TINY_BATCH_EHIP;
    HIP_TRY(hipMemcpyAsync(tb->x0buf, x0, (size_t)tb->batch * tb->nx * sizeof(float), hipMemcpyHostToDevice, tb->stream));
    return upload_packed(tb, x0, tb->arr[TINY_ARR_X], true, false, 0, 1);
}

int tiny_batch_set_x0_device(TinyBatch *tb, const float *d_x0)
{
    CHECK_TB(tb); CHECK_PTR(d_x0);
    if (set_device(tb)) return TINY_BATCH_EHIP;
    HIP_TRY(hipMemcpyAsync(tb->x0buf, d_x0, (size_t)tb->batch * tb->nx * sizeof(float), hipMemcpyDeviceToDevice, tb->stream));
    const long long total = (long long)tb->ntiles * WAVE * tb->NXC;
    hipLaunchKernelGGL(pack_kernel, dim3(grid_for(total)), dim3(256), 0, tb->stream, d_x0, tb->arr[TINY_ARR_X],
                       tb->batch, 0, tb->nx, tb->NXC, tb->ntiles, tb->N, 0, 1);
    HIP_TRY(hipGetLastError());
    return 0;
}

int tiny_batch_set_xref(TinyBatch *tb, const float *xref, int shared)
{
    CHECK_TB(tb); CHECK_PTR(xref);
    if (set_device(tb)) return TINY_BATCH_EHIP;
    int rc = upload_packed(tb, xref, tb->xref, true, shared != 0, 0, tb->N);
    if (rc) return rc;
    tb->xref_shared = shared != 0;
    tb->xref_mode = 0;
    return 0;
}

int tiny_batch_set_xref_window(TinyBatch *tb, const float *table, int rows, const int *start)
{
    CHECK_TB(tb); CHECK_PTR(table); CHECK_PTR(start);
    if (rows < tb->N) return fail(TINY_BATCH_EINVAL, "trajectory table has %d rows, need at least N=%d", rows, tb->N);
    for (int b = 0; b < tb->batch; b++)
        if (start[b] < 0 || start[b] + tb->N > rows)
            return fail(TINY_BATCH_EINVAL, "window start[%d]=%d out of range for %d rows, N=%d", b, start[b], rows, tb->N);
    if (set_device(tb)) return TINY_BATCH_EHIP;
    // table layout on device: [rows][4 gq][NXC]
    std::vector<float> t((size_t)rows * 4 * tb->NXC, 0.f);
    for (int r = 0; r < rows; r++)
        for (int gq = 0; gq < 4; gq++)
            for (int ch = 0; ch < tb->NXC; ch++)
            {
                int row = 4 * ch + gq;
                if (row < tb->nx) t[((size_t)r * 4 + gq) * tb->NXC + ch] = table[(size_t)r * tb->nx + row];
            }
    if (tb->xref_table && tb->table_rows != rows) { (void)hipFree(tb->xref_table); tb->xref_table = nullptr; }
    if (!tb->xref_table) HIP_TRY(hipMalloc((void **)&tb->xref_table, t.size() * sizeof(float)));
    HIP_TRY(hipMemcpyAsync(tb->xref_table, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice, tb->stream));
    HIP_TRY(hipMemcpyAsync(tb->xref_start, start, (size_t)tb->batch * sizeof(int), hipMemcpyHostToDevice, tb->stream));
    HIP_TRY(hipStreamSynchronize(tb->stream));
    tb->table_rows = rows;
    tb->xref_mode = 1;
    return 0;
}

int tiny_batch_set_umin(TinyBatch *tb, const float *s, int shared) { CHECK_TB(tb); return set_bound(tb, s, shared, &tb->umin, false, tb->ub_shared, 0); }
int tiny_batch_set_umax(TinyBatch *tb, const float *s, int shared) { CHECK_TB(tb); return set_bound(tb, s, shared, &tb->umax, false, tb->ub_shared, 1); }
int tiny_batch_set_xmin(TinyBatch *tb, const float *s, int shared) { CHECK_TB(tb); return set_bound(tb, s, shared, &tb->xmin, true, tb->xb_shared, 0); }
int tiny_batch_set_xmax(TinyBatch *tb, const float *s, int shared) { CHECK_TB(tb); return set_bound(tb, s, shared, &tb->xmax, true, tb->xb_shared, 1); }

int tiny_batch_reset_dual_variables(TinyBatch *tb)
{
    CHECK_TB(tb);
    tb->duals_zero_pending = true; // folded into the next solve's first iteration; flushed by any other reader
    return 0;
}

int tiny_batch_solve_async(TinyBatch *tb)
{
    CHECK_TB(tb);
    if (tb->xb_shared[0] != tb->xb_shared[1] || tb->ub_shared[0] != tb->ub_shared[1])
        return fail(TINY_BATCH_EINVAL, "min and max bounds must both be shared or both be per-instance");
    return launch_solve(tb);
}

int tiny_batch_wait(TinyBatch *tb, int *n_unsolved)
{
    CHECK_TB(tb);
    if (set_device(tb)) return TINY_BATCH_EHIP;
    int n = 0;
    HIP_TRY(hipMemcpyAsync(&n, tb->n_unsolved, sizeof(int), hipMemcpyDeviceToHost, tb->stream));
    HIP_TRY(hipStreamSynchronize(tb->stream));
    if (n_unsolved) *n_unsolved = n;
    return 0;
}

int tiny_batch_solve(TinyBatch *tb)
{
    int rc = tiny_batch_solve_async(tb);
    if (rc) return rc;
    int n = 0;
    rc = tiny_batch_wait(tb, &n);
    if (rc) return rc;
    return n > 0 ? 1 : 0;
}

int tiny_batch_get_x(TinyBatch *tb, float *x) { return tiny_batch_get_array(tb, TINY_ARR_X, x); }
int tiny_batch_get_u(TinyBatch *tb, float *u) { return tiny_batch_get_array(tb, TINY_ARR_U, u); }

int tiny_batch_get_status(TinyBatch *tb, int *iter, int *status, float *residuals)
{
    CHECK_TB(tb);
    if (set_device(tb)) return TINY_BATCH_EHIP;
    if (iter) HIP_TRY(hipMemcpyAsync(iter, tb->iter, (size_t)tb->batch * sizeof(int), hipMemcpyDeviceToHost, tb->stream));
    if (status) HIP_TRY(hipMemcpyAsync(status, tb->status, (size_t)tb->batch * sizeof(int), hipMemcpyDeviceToHost, tb->stream));
    if (residuals) HIP_TRY(hipMemcpyAsync(residuals, tb->res, (size_t)tb->batch * 4 * sizeof(float), hipMemcpyDeviceToHost, tb->stream));
    HIP_TRY(hipStreamSynchronize(tb->stream));
    return 0;
}

int tiny_batch_set_status(TinyBatch *tb, const int *iter, const int *status, const float *residuals)
{
    CHECK_TB(tb);
    if (set_device(tb)) return TINY_BATCH_EHIP;
    if (iter) HIP_TRY(hipMemcpyAsync(tb->iter, iter, (size_t)tb->batch * sizeof(int), hipMemcpyHostToDevice, tb->stream));
    if (status) HIP_TRY(hipMemcpyAsync(tb->status, status, (size_t)tb->batch * sizeof(int), hipMemcpyHostToDevice, tb->stream));
    if (residuals) HIP_TRY(hipMemcpyAsync(tb->res, residuals, (size_t)tb->batch * 4 * sizeof(float), hipMemcpyHostToDevice, tb->stream));
    HIP_TRY(hipStreamSynchronize(tb->stream));
    return 0;
}

int tiny_batch_set_array(TinyBatch *tb, int id, const float *src)
{
    CHECK_TB(tb); CHECK_PTR(src);
    if (id < 0 || id >= TINY_ARR_COUNT) return fail(TINY_BATCH_EINVAL, "bad array id %d", id);
    if (set_device(tb)) return TINY_BATCH_EHIP;
    if (int rc = flush_pending(tb)) return rc;
    const bool xf = is_xfam(id);
    int rc = upload_packed(tb, src, tb->arr[id], xf, false, 0, xf ? tb->N : tb->N - 1);
    if (rc) return rc;
    if (id == TINY_ARR_X) // keep the closed-loop state buffer coherent with x.col(0)
    {
        hipLaunchKernelGGL(unpack_kernel, dim3(grid_for((long long)tb->batch * tb->nx)), dim3(256), 0, tb->stream,
                           tb->arr[TINY_ARR_X], tb->x0buf, tb->batch, tb->nx, tb->NXC, tb->N, 0, 1);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

int tiny_batch_get_array(TinyBatch *tb, int id, float *dst)
{
    CHECK_TB(tb); CHECK_PTR(dst);
    if (id < 0 || id >= TINY_ARR_COUNT) return fail(TINY_BATCH_EINVAL, "bad array id %d", id);
    if (set_device(tb)) return TINY_BATCH_EHIP;
    if (int rc = flush_pending(tb)) return rc;
    const bool xf = is_xfam(id);
    return download_unpacked(tb, tb->arr[id], dst, xf, 0, xf ? tb->N : tb->N - 1);
}

int tiny_batch_reset_workspace(TinyBatch *tb)
{
    CHECK_TB(tb);
    if (set_device(tb)) return TINY_BATCH_EHIP;
    // x.col(0) (the x0 the caller sets next) is zeroed now; everything else is zeroed lazily: the next solve reads d,v,z,y,g as zero in its first iteration and overwrites the rest, any other
    // reader triggers the memsets (flush_pending).
    {
        const size_t step_bytes = (size_t)WAVE * tb->NXC * sizeof(float);
        HIP_TRY(hipMemset2DAsync(tb->arr[TINY_ARR_X], step_bytes * tb->N, 0, step_bytes, tb->ntiles, tb->stream)); // x.col(0)
    }
    HIP_TRY(hipMemsetAsync(tb->res, 0, (size_t)tb->batch * 4 * sizeof(float), tb->stream));
    HIP_TRY(hipMemsetAsync(tb->status, 0, (size_t)tb->batch * sizeof(int), tb->stream));
    HIP_TRY(hipMemsetAsync(tb->iter, 0, (size_t)tb->batch * sizeof(int), tb->stream));
    HIP_TRY(hipMemsetAsync(tb->x0buf, 0, (size_t)tb->batch * tb->nx * sizeof(float), tb->stream));
    tb->cold_pending = true;
    tb->duals_zero_pending = false;
    return 0;
}

int tiny_batch_get_u0_device(TinyBatch *tb, float *d_u0)
{
    CHECK_TB(tb); CHECK_PTR(d_u0);
    if (set_device(tb)) return TINY_BATCH_EHIP;
    const int n = tb->batch * tb->nu;
    hipLaunchKernelGGL(gather_u0_kernel, dim3((n + 255) / 256), dim3(256), 0, tb->stream, tb->arr[TINY_ARR_U], d_u0,
                       tb->batch, tb->nu, tb->NUC, tb->N);
    HIP_TRY(hipGetLastError());
    return 0;
}

int tiny_batch_mpc_step_async(TinyBatch *tb, int window_advance)
{
    CHECK_TB(tb);
    if (tb->nx > 64) return fail(TINY_BATCH_EUNSUPPORTED, "mpc_step supports nx <= 64");
    // x.col(0) already holds x0 (set_x0 / previous plant step); reset duals, solve, then simulate forward.
    tb->duals_zero_pending = true;
    int rc = tiny_batch_solve_async(tb);
    if (rc) return rc;
    hipLaunchKernelGGL(plant_step_kernel, dim3((tb->batch + 127) / 128), dim3(128), 0, tb->stream, tb->x0buf,
                       tb->arr[TINY_ARR_X], tb->arr[TINY_ARR_U], tb->dA, tb->dB,
                       tb->xref_mode == 1 ? tb->xref_start : nullptr, window_advance, tb->batch, tb->nx, tb->nu, tb->NXC,
                       tb->NUC, tb->N);
    HIP_TRY(hipGetLastError());
    return 0;
}

int tiny_batch_get_x0(TinyBatch *tb, float *x0)
{
    CHECK_TB(tb); CHECK_PTR(x0);
    if (set_device(tb)) return TINY_BATCH_EHIP;
    HIP_TRY(hipMemcpyAsync(x0, tb->x0buf, (size_t)tb->batch * tb->nx * sizeof(float), hipMemcpyDeviceToHost, tb->stream));
    HIP_TRY(hipStreamSynchronize(tb->stream));
    return 0;
}

int tiny_batch_enable_timing(TinyBatch *tb, int on)
{
    CHECK_TB(tb);
    tb->timing = on != 0;
    tb->ev_valid = false;
    return 0;
}

int tiny_batch_last_solve_ms(TinyBatch *tb, float *ms)
{
    CHECK_TB(tb); CHECK_PTR(ms);
    if (!tb->ev_valid) return fail(TINY_BATCH_ENOTREADY, "no timed solve recorded (call tiny_batch_enable_timing first)");
    if (set_device(tb)) return TINY_BATCH_EHIP;
    HIP_TRY(hipEventSynchronize(tb->ev1));
    HIP_TRY(hipEventElapsedTime(ms, tb->ev0, tb->ev1));
    return 0;
}

const char *tiny_batch_kernel_name(TinyBatch *tb) { return tb ? tb->kname.c_str() : ""; }

int tiny_batch_select_kernel(TinyBatch *tb, int variant)
{
    CHECK_TB(tb);
    if (variant < 0 || variant > 2) return fail(TINY_BATCH_EINVAL, "variant must be 0 (auto), 1 (stream) or 2 (resident)");
    if (variant == 2) return fail(TINY_BATCH_EUNSUPPORTED, "resident kernel not available for nx=%d nu=%d N=%d", tb->nx, tb->nu, tb->N);
    tb->variant = variant;
    return 0;
}

} // extern "C"
