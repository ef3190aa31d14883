// tinympc_batch.hip — host side of the C-ABI declared in include/tinympc_batch.h:
// device-resident batched workspace, layout conversion kernels, gain packing and kernel dispatch.
// The solver kernels live in admm_rowlane.hip (state on chip) and admm_stream.hip (state in HBM).
//
// There is deliberately NO CPU fallback in this library: every entry point that computes
// launches a HIP kernel and reports HIP failures as TINY_BATCH_EHIP.
#include "../../include/tinympc_batch.h"
#include "tinympc_internal.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
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
#define TRY(expr)                 \
    do                            \
    {                             \
        int rc_ = (expr);         \
        if (rc_) return rc_;      \
    } while (0)
#define CHECK_TB(tb) \
    if (!(tb)) return fail(TINY_BATCH_EINVAL, "%s: NULL TinyBatch handle", __func__)
#define CHECK_PTR(p) \
    if (!(p)) return fail(TINY_BATCH_EINVAL, "%s: NULL pointer argument '%s'", __func__, #p)

// ---------------------------------------------------------------------------------------------
// Debug guard zones ("a debug bounds-checked kernel variant" — GPU AddressSanitizer does not exist on this pool).
// With tiny_batch_debug_guards(1) every device allocation of this library is made kGuard floats larger at both ends; the guards are filled
// with a quiet-NaN pattern.  An out-of-bounds WRITE of any kernel then lands in a guard and tiny_batch_debug_check() counts the
// damaged words; an out-of-bounds READ returns NaN, which no parity test survives.  It is a mode of the same library — the kernels
// are the shipped ones, unchanged — used by tests/test_parity_gpu.py::test_debug_guard_zones_stay_intact over every kernel family.
// ---------------------------------------------------------------------------------------------
constexpr size_t kGuard = 256;               // floats on each side (1 KB: keeps the 256-byte alignment of hipMalloc)
constexpr unsigned kGuardWord = 0x7fc0dea1u; // a quiet NaN
std::mutex g_guard_mu;
std::map<void *, size_t> g_guarded;          // user pointer -> user bytes
bool g_guards_on = false;

__global__ void guard_fill_kernel(unsigned *p, size_t nwords)
{
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < nwords; e += (size_t)gridDim.x * blockDim.x) p[e] = kGuardWord;
}
__global__ void guard_count_kernel(const unsigned *p, size_t nwords, unsigned long long *bad)
{
    unsigned long long n = 0;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < nwords; e += (size_t)gridDim.x * blockDim.x) n += p[e] != kGuardWord;
    if (n) atomicAdd(bad, n);
}

hipError_t guarded_malloc(void **out, size_t bytes)
{
    if (!g_guards_on) return hipMalloc(out, bytes);
    const size_t user = (bytes + 3) / 4 * 4;
    char *raw = nullptr;
    hipError_t e = hipMalloc((void **)&raw, user + 2 * kGuard * sizeof(float));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(guard_fill_kernel, dim3(1), dim3(256), 0, nullptr, (unsigned *)raw, kGuard);
    hipLaunchKernelGGL(guard_fill_kernel, dim3(1), dim3(256), 0, nullptr, (unsigned *)(raw + kGuard * sizeof(float) + user), kGuard);
    e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) { (void)hipFree(raw); return e; }
    *out = raw + kGuard * sizeof(float);
    std::lock_guard<std::mutex> lk(g_guard_mu);
    g_guarded[*out] = user;
    return hipSuccess;
}
hipError_t guarded_free(void *p)
{
    if (!p) return hipSuccess;
    {
        std::lock_guard<std::mutex> lk(g_guard_mu);
        auto it = g_guarded.find(p);
        if (it != g_guarded.end())
        {
            g_guarded.erase(it);
            return hipFree((char *)p - kGuard * sizeof(float));
        }
    }
    return hipFree(p);
}

enum { LAYOUT_TILE = 0, LAYOUT_ROW = 1 };
enum { VAR_AUTO = 0, VAR_STREAM = 1, VAR_ROW_EXACT = 2, VAR_ROW_FAST = 3, VAR_GENERIC = 4 }; // 4: admm_generic.hip, exact arithmetic for any eligible class
inline bool tile_variant(int v) { return v == VAR_STREAM || v == VAR_GENERIC; } // the variants that work on the TILE layout

// ---------------------------------------------------------------------------------------------
// Element addressing of the two device layouts.  `fam` 0 = state-type (nx rows, N steps),
// 1 = input-type (nu rows, N-1 steps).
//   TILE (admm_stream.hip):  x-family [ntiles][N][64][NXC], u-family [ntiles][N-1][64][NUC]
//   ROW  (row / wave kernels): pair array [batch_pad4][N][rw], rows [0,nx) state member, [nx,nx+nu) input member;
//        rw = 16 (one DPP row per instance) or 64 (one wavefront per instance, admm_wave.hip)
// ---------------------------------------------------------------------------------------------
struct Geo
{
    int nx, nu, N, NXC, NUC;
    int rw; // lanes per instance-step of the ROW layout: 16 (one DPP row) or 64 (one wavefront, 16 < nx + nu <= 64)
};

__device__ __forceinline__ long long idx_tile(const Geo g, int fam, int b, int step, int row)
{
    const int tile = b / TILE, c = b % TILE, lane = (row & 3) * 16 + c, ch = row >> 2;
    const int steps = fam ? g.N - 1 : g.N, NC = fam ? g.NUC : g.NXC;
    return (((long long)tile * steps + step) * WAVE + lane) * NC + ch;
}
__device__ __forceinline__ long long idx_row(const Geo g, int fam, int b, int step, int row)
{
    return ((long long)b * g.N + step) * g.rw + (fam ? g.nx + row : row);
}
__device__ __forceinline__ long long idx_of(int layout, const Geo g, int fam, int b, int step, int row)
{
    return layout == LAYOUT_ROW ? idx_row(g, fam, b, step, row) : idx_tile(g, fam, b, step, row);
}

// element access of a device-layout array: fp32, or IEEE binary16 when the handle stores fp16 (ROW layout only)
__device__ __forceinline__ void put_elem(float *dst, long long i, float v, int h16)
{
    if (h16) reinterpret_cast<_Float16 *>(dst)[i] = (_Float16)v; // round to nearest even
    else dst[i] = v;
}
__device__ __forceinline__ float get_elem(const float *src, long long i, int h16)
{
    return h16 ? (float)reinterpret_cast<const _Float16 *>(src)[i] : src[i];
}

// host-layout src [cnt][nsteps][dim] (cnt = 1: shared by all instances)  ->  device layout, steps [step0, step0+nsteps)
// of instances [0, nb).  Rows/instances beyond the source are left untouched (they were zeroed at allocation).
// `mirror` (may be NULL): a plain copy of src, element for element — set_x0_device fills x.col(0) and the [B][nx] state buffer of the closed loop in
// ONE launch (round 4: it was a device-to-device copy plus this kernel)
__global__ void pack_kernel(const float *__restrict__ src, float *__restrict__ dst, int layout, Geo g, int fam, int nb,
                            int shared, int step0, int nsteps, int h16, float *__restrict__ mirror = nullptr)
{
    const int dim = fam ? g.nu : g.nx;
    const long long total = (long long)nb * nsteps * dim;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x)
    {
        const int row = (int)(e % dim);
        const long long t = e / dim;
        const int s = (int)(t % nsteps), b = (int)(t / nsteps);
        const float val = src[((long long)(shared ? 0 : b) * nsteps + s) * dim + row];
        put_elem(dst, idx_of(layout, g, fam, b, step0 + s, row), val, h16);
        if (mirror) mirror[e] = val;
    }
}

__global__ void unpack_kernel(const float *__restrict__ src, float *__restrict__ dst, int layout, Geo g, int fam, int nb,
                              int step0, int nsteps, int h16)
{
    const int dim = fam ? g.nu : g.nx;
    const long long total = (long long)nb * nsteps * dim;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x)
    {
        const int row = (int)(e % dim);
        const long long t = e / dim;
        const int s = (int)(t % nsteps), b = (int)(t / nsteps);
        dst[e] = get_elem(src, idx_of(layout, g, fam, b, step0 + s, row), h16);
    }
}

// zero steps [step0, step0+nsteps) of one member of a device array
__global__ void zero_kernel(float *__restrict__ dst, int layout, Geo g, int fam, int nb, int step0, int nsteps, int h16)
{
    const int dim = fam ? g.nu : g.nx;
    const long long total = (long long)nb * nsteps * dim;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x)
    {
        const int row = (int)(e % dim);
        const long long t = e / dim;
        put_elem(dst, idx_of(layout, g, fam, (int)(t / nsteps), step0 + (int)(t % nsteps), row), 0.f, h16);
    }
}

// the duals pair g | y between fp32 and binary16 (tiny_batch_set_storage(tb, 16) is a preference: the width follows the kernel a call resolves to)
__global__ void dual_width_kernel(const float *__restrict__ src, float *__restrict__ dst, long long n, int to32)
{
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x)
    {
        if (to32) dst[e] = (float)reinterpret_cast<const _Float16 *>(src)[e];
        else reinterpret_cast<_Float16 *>(dst)[e] = (_Float16)src[e]; // round to nearest even, like every store of the 16-bit mode
    }
}

// x0 <- Adyn*x0 + Bdyn*u.col(0)   (quadrotor_hovering.cpp:110-111), and x.col(0) <- x0 (:95).
// One thread per instance; matrices column-major in global memory (tiny, cache resident).
// Summation order = Eigen's for that expression (pinned bit for bit against that expression compiled from the reference, tests/test_oracle.py:
// test_plant_step_bit_exact_vs_compiled_reference): dst = Adyn*x0, then dst += Bdyn*u0; a product whose rows and depth are both >= 8 takes the column-major
// GEMV kernel (row accumulator starting at +0, products added in ascending column order), a smaller one the lazy product
// (sequential from the first product for the nx % 4 == 0 classes this library serves).
__global__ void plant_step_kernel(float *__restrict__ x0buf, float *__restrict__ xarr, const float *__restrict__ uarr,
                                  const float *__restrict__ A, const float *__restrict__ Bm, int *__restrict__ wstart,
                                  int window_advance, int batch, int layout, Geo g, int h16)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const int nx = g.nx, nu = g.nu;
    const float *x0 = x0buf + (long long)b * nx;
    const bool gemv_a = nx >= 8, gemv_b = nx >= 8 && nu >= 8;
    float xn[64];
    for (int i = 0; i < nx; i++)
    {
        float acc = A[i] * x0[0];
        if (gemv_a) acc = 0.f + acc;
        for (int k = 1; k < nx; k++) acc += A[k * nx + i] * x0[k];
        float acc2 = Bm[i] * get_elem(uarr, idx_of(layout, g, 1, b, 0, 0), h16);
        if (gemv_b) acc2 = 0.f + acc2;
        for (int m = 1; m < nu; m++) acc2 += Bm[m * nx + i] * get_elem(uarr, idx_of(layout, g, 1, b, 0, m), h16);
        xn[i] = acc + acc2;
    }
    for (int i = 0; i < nx; i++)
    {
        x0buf[(long long)b * nx + i] = xn[i];
        put_elem(xarr, idx_of(layout, g, 0, b, 0, i), xn[i], h16);
    }
    if (wstart && window_advance) wstart[b] += window_advance;
}

// per-instance bounds table of the ROW layout: dst[b][step][r] = {min(lo, hi), hi}, +-inf where a bound is disabled or the
// row carries nothing.  Sources are the canonical inputs ([B or 1][steps][dim], NULL = not set = 0).
__global__ void bounds_table_kernel(const float *__restrict__ xmin, const float *__restrict__ xmax, const float *__restrict__ umin,
                                    const float *__restrict__ umax, int sh_x, int sh_u, float *__restrict__ dst, Geo g, int nb,
                                    int en_state, int en_input, int h16)
{
    const long long total = (long long)nb * g.N * g.rw;
    const float inf = __builtin_inff();
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x)
    {
        const int r = (int)(e % g.rw);
        const long long t = e / g.rw;
        const int i = (int)(t % g.N), b = (int)(t / g.N);
        float lo = -inf, hi = inf;
        if (r < g.nx && en_state)
        {
            const long long o = ((long long)(sh_x ? 0 : b) * g.N + i) * g.nx + r;
            lo = xmin ? xmin[o] : 0.f; hi = xmax ? xmax[o] : 0.f;
        }
        else if (r >= g.nx && r < g.nx + g.nu && i < g.N - 1 && en_input)
        {
            const long long o = ((long long)(sh_u ? 0 : b) * (g.N - 1) + i) * g.nu + (r - g.nx);
            lo = umin ? umin[o] : 0.f; hi = umax ? umax[o] : 0.f;
        }
        lo = lo < hi ? lo : hi;
        if (h16)
        {
            reinterpret_cast<_Float16 *>(dst)[2 * e] = (_Float16)lo;
            reinterpret_cast<_Float16 *>(dst)[2 * e + 1] = (_Float16)hi;
        }
        else { dst[2 * e] = lo; dst[2 * e + 1] = hi; }
    }
}

// Tile images of the ROW tables for the LDS-DMA rings of admm_tile16_pi.hip: dst[tile][step][piece][lane = 16 g + c] (float4) = words 4 (P g + piece) .. + 3 of
// row `step` of instance 16 tile + c, P = pieces per lane (1: the 16-float reference rows, 2: the 32-float {lo, hi} rows) — byte for byte what a ring slot
// holds, so that one DMA reads 1 KB of consecutive memory.  Columns past the batch repeat its last instance (they are computed and never stored).
__global__ void tile16_image_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst, int nb, int N, int pieces, long long inst_stride_f4)
{
    const long long total = (long long)((nb + 15) / 16) * N * pieces * 64;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x)
    {
        const int lane = (int)(e & 63), piece = (int)((e >> 6) % pieces);
        const long long t = e / (64 * pieces);
        const int step = (int)(t % N), g = lane >> 4, c = lane & 15;
        long long inst = (t / N) * 16 + c;
        inst = inst < nb ? inst : nb - 1;
        dst[e] = src[inst * inst_stride_f4 + (long long)step * 4 * pieces + pieces * g + piece];
    }
}

// Do the per-instance ROW tables change along the horizon?  vary[0]: the {lo, hi} table [nb][N][16] (x rows over all N steps, u rows over the
// N - 1 steps that have an input), vary[1]: the reference [nb][N][16].  Bit patterns are compared.  admm_tile16_pi.hip keeps a table that does
// not as ONE resident row per instance instead of streaming N of them through its ring every iteration (RowParams::pi_flags).
__global__ void rows_vary_kernel(const float2 *__restrict__ bounds, const float *__restrict__ xref, int nb, int N, int nx, int nu, int *__restrict__ vary)
{
    const long long total = (long long)nb * 16;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x)
    {
        const int r = (int)(e & 15);
        const long long b = e >> 4;
        if (bounds && r < nx + nu)
        {
            const int steps = r < nx ? N : N - 1;
            const unsigned long long *p = reinterpret_cast<const unsigned long long *>(bounds) + (b * N) * 16 + r;
            bool differ = false;
            for (int i = 1; i < steps; i++) differ |= p[(size_t)i * 16] != p[0];
            if (differ) vary[0] = 1;
        }
        if (xref && r < nx)
        {
            const unsigned *p = reinterpret_cast<const unsigned *>(xref) + (b * N) * 16 + r;
            bool differ = false;
            for (int i = 1; i < N; i++) differ |= p[(size_t)i * 16] != p[0];
            if (differ) vary[1] = 1;
        }
    }
}

int grid_for(long long total, int block = 256)
{
    long long gsz = (total + block - 1) / block;
    if (gsz > 8192) gsz = 8192;
    if (gsz < 1) gsz = 1;
    return (int)gsz;
}

// pair index of each work array in the ROW layout: x,u | q,r | p,d | v,z | vnew,znew | g,y
const int kPairOf[TINY_ARR_COUNT] = {0, 0, 1, 1, 2, 2, 3, 4, 3, 4, 5, 5};
bool is_xfam(int id)
{
    return id == TINY_ARR_X || id == TINY_ARR_Q || id == TINY_ARR_P || id == TINY_ARR_V || id == TINY_ARR_VNEW ||
           id == TINY_ARR_G;
}

struct InputArr // canonical copy of a caller-provided input, host layout [cnt][steps][dim] on the device
{
    float *dev = nullptr;
    bool shared = true, set = false;
    std::vector<float> host; // kept only when shared (small)
};

} // namespace

// ---------------------------------------------------------------------------------------------
struct TinyBatch
{
    int nx = 0, nu = 0, N = 0, batch = 0, device = 0;
    int NXC = 0, NUC = 0, ntiles = 0, bpad4 = 0;
    int rw = 16; // lanes per instance-step of the ROW layout
    bool row_dims_ok = false, tile_dims_ok = false, rowmath_ok = false, rowloop_ok = false, wave_ok = false, quad_ok = false;
    bool tile16_ok = false; // admm_tile16.hip has an instantiation for (nx, nu, N)
    bool waveres_ok = false; // admm_waveres.hip serves (nx, nu, N): wave class with N <= 50
    bool tile48_ok = false;  // admm_tile48.hip does (nx = 32, nu = 16, N = 50: sixteen instances per workgroup on the matrix cores)
    bool generic_ok = false; // admm_generic.hip: exact arithmetic with run-time dimensions (nx, nu each <= 4 or a multiple of 4)
    float *gen_mats = nullptr; // its gains: Kinf | Pinf | Quu_inv | AmBKt | Adyn | Bdyn | Q, column-major
    int *conv_dev = nullptr; // [batch] result of termination_condition
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr; // created by tiny_batch_group_solve / mpc_run for handles left on the null stream
    hipGraphExec_t graph_exec = nullptr; // tiny_batch_mpc_run_async: captured (solve + plant) x steps
    std::string graph_sig;
    // problem class
    bool have_cache = false, have_dyn = false, have_settings = false, gains_dirty = true;
    float rho = 0.f;
    std::vector<float> Kinf, Pinf, Quu_inv, AmBKt, Adyn, Bdyn, Q;
    float abs_pri_tol = 0.f, abs_dua_tol = 0.f;
    int max_iter = 0, check_termination = 1, en_state_bound = 0, en_input_bound = 0;
    // work arrays: exactly one layout is allocated at a time
    int layout = LAYOUT_TILE;
    float *arr[TINY_ARR_COUNT] = {}; // TILE
    float *pair[6] = {};             // ROW
    size_t xfam_floats = 0, ufam_floats = 0, pair_floats = 0;
    // caller inputs (canonical) and their per-layout derived forms
    InputArr in_xref, in_bnd[4]; // bnd: xmin, xmax, umin, umax
    // the two terms the reference ships commented out (admm.cpp:20, :79); tiny_batch_set_optional_terms
    bool en_uref = false, en_d2p = false, have_rcost = false;
    std::vector<float> Rcost, coeff_d2p; // R [nu], coeff_d2p [nx x nu] column-major
    InputArr in_uref;                    // [batch or 1][N-1][nu]
    float *r_uref = nullptr;             // ROW derived: [batch_pad4 or 1][N][rw], Uref on the u rows
    const int *order_dev = nullptr;      // caller-owned dispatch order of the instance groups (tiny_batch_set_dispatch_order_device)
    int dispatch_mode = -1;              // tiny_batch_set_dispatch: 0 in index order, 1 longest first by the predicted iteration count, -1 (default) automatic:
                                         // longest first for a launch that starts from a reset workspace (dispatch_effective)
    float *u0_stage = nullptr;           // [batch][nu] staging of u.col(0) for the peer copy of tiny_batch_group_gather_u0
    float *key_buf = nullptr;            // [groups] predictor of dispatch_order.hip
    int *order_buf = nullptr;            // [groups] its sorted order
    bool derived_dirty[2] = {true, true};
    float *t_xref = nullptr, *t_bnd[4] = {};              // TILE derived
    float *r_xref = nullptr, *r_bounds = nullptr;         // ROW derived
    float *r_bounds_img = nullptr, *r_xref_img = nullptr; // tile images of the two tables when they go through the rings of admm_tile16_pi.hip
    size_t r_bounds_img_n = 0, r_xref_img_n = 0;
    int *rows_vary_dev = nullptr;                         // [2] rows_vary_kernel's answer for the per-instance ROW tables ...
    unsigned rows_vary = 3u;                              // ... bit 0: bounds, bit 1: reference (set = changes along the horizon)
    size_t r_xref_n = 0, r_bounds_n = 0, r_uref_n = 0;    // their allocated sizes in floats (sized_buffer)
    int graph_captures = 0;                               // closed-loop graphs captured so far (tiny_batch_debug_graph_captures)
    int n_cu = 256;                                       // compute units of the handle's device
    int tile_queue = -1;                                  // tiny_batch_set_tile_queue: -1 automatic, 0 plain counter, k every k-th wave from the short end
    float *tab_tile = nullptr, *tab_row = nullptr;        // trajectory table in both forms
    float *tab_row_h = nullptr;                           // ... and [rows][16] binary16 for fp16 storage
    int table_rows = 0;
    int *xref_start = nullptr;
    int xref_mode = 0;
    float *res = nullptr;
    int *status = nullptr, *iter = nullptr, *n_unsolved = nullptr;
    float *opnd = nullptr, *qvec = nullptr;               // TILE gains (MFMA operands)
    float *mats_exact = nullptr, *mats_fast = nullptr;    // ROW gains
    float *dA = nullptr, *dB = nullptr;                   // column-major copies for the plant step
    float *x0buf = nullptr;                               // [B][nx] current state (closed loop)
    float *staging = nullptr;                             // host-layout staging
    size_t staging_floats = 0;
    bool duals_zero_pending = false;
    bool cold_pending = false;
    bool x0_zero_pending = false; // reset_workspace(): x.col(0) and x0buf read as zero until a set_x0 overwrites them
    int variant = VAR_AUTO;
    int row_family_forced = -1; // tiny_batch_set_row_kernel
    int last_dispatch = 0;        // what the most recent solve launch did: 0 index order, 1 predicted longest first, 2 the caller's order, 3 longest first by the previous solve's counts
    bool iter_history = false;    // iter[] holds the counts of a solve of THIS workspace's instances (not a reset, not an upload): the history order's key
    bool closed_loop_run = false; // inside tiny_batch_mpc_run_*(steps > 1): the auto choice keeps the kernel with the on-chip loop
    bool h16 = false; // ROW-layout arrays, Xref and bounds stored as IEEE binary16 (tiny_batch_set_storage)
    bool dual32 = false; // with h16: the duals pair gy IS fp32 right now (the state of the array)
    bool dual32_pref = false;   // tiny_batch_set_storage(tb, 16): fp32 duals wherever the kernel a call resolves to implements them, 16-bit duals elsewhere
    bool dual32_forced = false; // tiny_batch_set_storage_ex(tb, 16, 32): fp32 duals or TINY_BATCH_EUNSUPPORTED
    bool timing = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool ev_valid = false;
    std::string kname;
};

namespace
{

Geo geo(const TinyBatch *tb) { return Geo{tb->nx, tb->nu, tb->N, tb->NXC, tb->NUC, tb->rw}; }

// The hipGraph cached by tiny_batch_mpc_run_traj_async bakes in every kernel ARGUMENT of its launches (RowParams /
// SolveParams are passed by value): rho, tolerances, bound flags, array pointers and strides, window advance.  Anything
// that can change one of them drops the graph; the next mpc_run captures a fresh one.
void invalidate_graph(TinyBatch *tb)
{
    if (tb->graph_exec)
    {
        (void)hipSetDevice(tb->device);
        (void)hipStreamSynchronize(tb->stream); // a replay may still be in flight
        (void)hipGraphExecDestroy(tb->graph_exec);
        tb->graph_exec = nullptr;
    }
    tb->graph_sig.clear();
}

int set_device(TinyBatch *tb)
{
    HIP_TRY(hipSetDevice(tb->device));
    return 0;
}

int dev_alloc_zero(float **p, size_t nfloats)
{
    HIP_TRY(guarded_malloc((void **)p, nfloats * sizeof(float)));
    HIP_TRY(hipMemset(*p, 0, nfloats * sizeof(float)));
    // hipMemset of device memory is enqueued on the null stream and may return early; a handle on a non-blocking
    // stream would not be ordered behind it
    HIP_TRY(hipStreamSynchronize(nullptr));
    return 0;
}

// A derived input buffer of the ROW layout (reference, bounds table, Uref) is re-filled whenever its source changes; it is re-ALLOCATED only when its
// size does: its address is a kernel argument baked into the captured closed-loop graph, and `set_xref; mpc_run(k)` in a loop must replay one graph
// (round-3 advisor: with free + malloc per refill that held only while the allocator happened to return the same address)
int sized_buffer(float **p, size_t *have, size_t nfloats)
{
    if (*p && *have == nfloats) return 0;
    if (*p) { (void)guarded_free(*p); *p = nullptr; }
    *have = 0;
    TRY(dev_alloc_zero(p, nfloats));
    *have = nfloats;
    return 0;
}

float *work_ptr(TinyBatch *tb, int id) { return tb->layout == LAYOUT_ROW ? tb->pair[kPairOf[id]] : tb->arr[id]; }
int h16_of(const TinyBatch *tb, int layout) { return (tb->h16 && layout == LAYOUT_ROW) ? 1 : 0; }
// element type of one device array: the duals pair is fp32 under tiny_batch_set_storage_ex(16, 32)
int h16_at(const TinyBatch *tb, int layout, const float *arr) { return (h16_of(tb, layout) && !(tb->dual32 && arr == tb->pair[5])) ? 1 : 0; }

int alloc_layout(TinyBatch *tb, int layout)
{
    if (layout == LAYOUT_ROW)
    {
        for (int p = 0; p < 6; p++)
            if (!tb->pair[p]) TRY(dev_alloc_zero(&tb->pair[p], (tb->h16 && !(tb->dual32 && p == 5)) ? (tb->pair_floats + 1) / 2 : tb->pair_floats));
    }
    else
    {
        for (int id = 0; id < TINY_ARR_COUNT; id++)
            if (!tb->arr[id]) TRY(dev_alloc_zero(&tb->arr[id], is_xfam(id) ? tb->xfam_floats : tb->ufam_floats));
    }
    return 0;
}

void free_layout(TinyBatch *tb, int layout)
{
    if (layout == LAYOUT_ROW)
        for (int p = 0; p < 6; p++) { (void)guarded_free(tb->pair[p]); tb->pair[p] = nullptr; }
    else
        for (int id = 0; id < TINY_ARR_COUNT; id++) { (void)guarded_free(tb->arr[id]); tb->arr[id] = nullptr; }
}

int launch_pack(TinyBatch *tb, const float *src, float *dst, int layout, int fam, int nb, bool shared, int step0, int nsteps, float *mirror = nullptr)
{
    const long long total = (long long)nb * nsteps * (fam ? tb->nu : tb->nx);
    hipLaunchKernelGGL(pack_kernel, dim3(grid_for(total)), dim3(256), 0, tb->stream, src, dst, layout, geo(tb), fam, nb,
                       shared ? 1 : 0, step0, nsteps, h16_at(tb, layout, dst), mirror);
    HIP_TRY(hipGetLastError());
    return 0;
}
int launch_unpack(TinyBatch *tb, const float *src, float *dst, int layout, int fam, int nb, int step0, int nsteps)
{
    const long long total = (long long)nb * nsteps * (fam ? tb->nu : tb->nx);
    hipLaunchKernelGGL(unpack_kernel, dim3(grid_for(total)), dim3(256), 0, tb->stream, src, dst, layout, geo(tb), fam, nb,
                       step0, nsteps, h16_at(tb, layout, src));
    HIP_TRY(hipGetLastError());
    return 0;
}
int launch_zero(TinyBatch *tb, float *dst, int layout, int fam, int step0, int nsteps)
{
    const long long total = (long long)tb->batch * nsteps * (fam ? tb->nu : tb->nx);
    hipLaunchKernelGGL(zero_kernel, dim3(grid_for(total)), dim3(256), 0, tb->stream, dst, layout, geo(tb), fam, tb->batch,
                       step0, nsteps, h16_at(tb, layout, dst));
    HIP_TRY(hipGetLastError());
    return 0;
}

// reset_workspace() zeroes x.col(0) lazily too: the usual next call is set_x0, which overwrites all of it
int flush_x0_zero(TinyBatch *tb)
{
    if (!tb->x0_zero_pending) return 0;
    TRY(launch_zero(tb, work_ptr(tb, TINY_ARR_X), tb->layout, 0, 0, 1));
    HIP_TRY(hipMemsetAsync(tb->x0buf, 0, (size_t)tb->batch * tb->nx * sizeof(float), tb->stream));
    tb->x0_zero_pending = false;
    return 0;
}

// Materialise pending lazy resets (needed before anything other than a solve looks at the arrays).
int flush_pending(TinyBatch *tb)
{
    TRY(flush_x0_zero(tb));
    if (tb->cold_pending)
    {
        // every work array reads as zero after reset_workspace(), except x.col(0) which carries x0
        for (int id = 0; id < TINY_ARR_COUNT; id++)
        {
            const int fam = is_xfam(id) ? 0 : 1;
            const int steps = fam ? tb->N - 1 : tb->N;
            if (id == TINY_ARR_X) TRY(launch_zero(tb, work_ptr(tb, id), tb->layout, 0, 1, tb->N - 1));
            else TRY(launch_zero(tb, work_ptr(tb, id), tb->layout, fam, 0, steps));
        }
        HIP_TRY(hipMemsetAsync(tb->res, 0, (size_t)tb->batch * 4 * sizeof(float), tb->stream));
        HIP_TRY(hipMemsetAsync(tb->status, 0, (size_t)tb->batch * sizeof(int), tb->stream));
        HIP_TRY(hipMemsetAsync(tb->iter, 0, (size_t)tb->batch * sizeof(int), tb->stream));
        tb->cold_pending = false;
        tb->duals_zero_pending = false;
    }
    if (tb->duals_zero_pending)
    {
        TRY(launch_zero(tb, work_ptr(tb, TINY_ARR_Y), tb->layout, 1, 0, tb->N - 1));
        TRY(launch_zero(tb, work_ptr(tb, TINY_ARR_G), tb->layout, 0, 0, tb->N));
        tb->duals_zero_pending = false;
    }
    return 0;
}

// Move the twelve work arrays to another device layout (through the host-layout staging buffer, on the device).
int ensure_layout(TinyBatch *tb, int want)
{
    if (tb->layout == want) return 0;
    TRY(flush_pending(tb));
    const int from = tb->layout;
    TRY(alloc_layout(tb, want));
    for (int id = 0; id < TINY_ARR_COUNT; id++)
    {
        const int fam = is_xfam(id) ? 0 : 1, steps = fam ? tb->N - 1 : tb->N;
        const float *src = from == LAYOUT_ROW ? tb->pair[kPairOf[id]] : tb->arr[id];
        float *dst = want == LAYOUT_ROW ? tb->pair[kPairOf[id]] : tb->arr[id];
        TRY(launch_unpack(tb, src, tb->staging, from, fam, tb->batch, 0, steps));
        TRY(launch_pack(tb, tb->staging, dst, want, fam, tb->batch, false, 0, steps));
    }
    HIP_TRY(hipStreamSynchronize(tb->stream));
    free_layout(tb, from);
    tb->layout = want;
    return 0;
}

// upload a host array ([batch][nsteps][dim]) into steps [step0, ...) of a work array of the current layout
int upload_work(TinyBatch *tb, const float *host, int id, int step0, int nsteps)
{
    const int fam = is_xfam(id) ? 0 : 1;
    const size_t n = (size_t)tb->batch * nsteps * (fam ? tb->nu : tb->nx);
    if (n > tb->staging_floats) return fail(TINY_BATCH_EINVAL, "internal: staging too small");
    HIP_TRY(hipMemcpyAsync(tb->staging, host, n * sizeof(float), hipMemcpyHostToDevice, tb->stream));
    TRY(launch_pack(tb, tb->staging, work_ptr(tb, id), tb->layout, fam, tb->batch, false, step0, nsteps));
    HIP_TRY(hipStreamSynchronize(tb->stream)); // staging and `host` are reusable on return
    return 0;
}

int download_work(TinyBatch *tb, int id, float *host, int step0, int nsteps)
{
    const int fam = is_xfam(id) ? 0 : 1;
    const size_t n = (size_t)tb->batch * nsteps * (fam ? tb->nu : tb->nx);
    if (n > tb->staging_floats) return fail(TINY_BATCH_EINVAL, "internal: staging too small");
    TRY(launch_unpack(tb, work_ptr(tb, id), tb->staging, tb->layout, fam, tb->batch, step0, nsteps));
    HIP_TRY(hipMemcpyAsync(host, tb->staging, n * sizeof(float), hipMemcpyDeviceToHost, tb->stream));
    HIP_TRY(hipStreamSynchronize(tb->stream));
    return 0;
}

int store_input(TinyBatch *tb, InputArr &in, const float *host, bool shared, int steps, int dim)
{
    const size_t n = (size_t)(shared ? 1 : tb->batch) * steps * dim;
    if (in.dev && in.shared != shared) { (void)guarded_free(in.dev); in.dev = nullptr; }
    if (!in.dev) HIP_TRY(guarded_malloc((void **)&in.dev, n * sizeof(float)));
    HIP_TRY(hipMemcpyAsync(in.dev, host, n * sizeof(float), hipMemcpyHostToDevice, tb->stream));
    HIP_TRY(hipStreamSynchronize(tb->stream));
    in.shared = shared;
    in.set = true;
    if (shared) in.host.assign(host, host + n);
    else in.host.clear();
    tb->derived_dirty[0] = tb->derived_dirty[1] = true; // (a captured closed-loop graph carries the mode in its signature)
    return 0;
}

// tiny_batch_set_dispatch(-1), the default: the predictor sweep and the sort pay where the iteration counts of a launch spread widely — a launch that starts from a
// reset workspace (65 536 tracking instances, wall time per solve: 2.20 -> 1.76 ms; 16 384: 0.82 -> 0.68) — and cost 2 - 4 % on warm-started steps (1.62 -> 1.66 ms)
// (second session of round 4) ... and a warm-started launch is ordered by what the predictor cannot see and the workspace already holds: the iteration counts of the
// PREVIOUS solve of the same instances (dispatch_order.hip, history order; mode 2): consecutive MPC steps are strongly correlated.  65 536 tracking instances,
// warm-started step: makespan 108 -> 78 iterations on the true counts (tests/fuzz/sim_history_dispatch.py).
constexpr int kDispatchMinGroups = 4096; // two rounds of waves on 256 CUs x 4 SIMDs x 2 waves
int dispatch_effective(const TinyBatch *tb)
{
    if (tb->dispatch_mode == 2) return tb->iter_history && !tb->cold_pending ? 2 : 0;
    if (tb->dispatch_mode >= 0) return tb->dispatch_mode;
    return tb->cold_pending ? 1 : (tb->iter_history ? 2 : 0);
}

bool bounds_all_shared(const TinyBatch *tb)
{
    for (int k = 0; k < 4; k++)
        if (tb->in_bnd[k].set && !tb->in_bnd[k].shared) return false;
    return true;
}

// ---- gains for the streaming kernel: MFMA A operands ---------------------------------------------
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

int upload_vec(TinyBatch *tb, float **dst, const std::vector<float> &v)
{
    if (!*dst) HIP_TRY(guarded_malloc((void **)dst, v.size() * sizeof(float)));
    HIP_TRY(hipMemcpyAsync(*dst, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice, tb->stream));
    HIP_TRY(hipStreamSynchronize(tb->stream)); // `v` may be a temporary
    return 0;
}

int pack_gains(TinyBatch *tb)
{
    const int nx = tb->nx, nu = tb->nu, NXC = tb->NXC, NUC = tb->NUC;
    const float *K = tb->Kinf.data(), *Pf = tb->Pinf.data(), *Qi = tb->Quu_inv.data(), *Am = tb->AmBKt.data();
    const float *A = tb->Adyn.data(), *B = tb->Bdyn.data();
    auto Kat = [&](int m, int k) { return K[k * nu + m]; };
    auto Aat = [&](int j, int k) { return A[k * nx + j]; };
    auto Bat = [&](int j, int m) { return B[m * nx + j]; };
    auto Amat = [&](int j, int k) { return Am[k * nx + j]; };
    auto Qiat = [&](int a, int b) { return Qi[b * nu + a]; };
    auto Pat = [&](int k, int j) { return Pf[j * nx + k]; };
    if (tb->tile_dims_ok)
    {
        const int NCH = NXC + NUC, NT = (NCH + 3) / 4, NTX = (NXC + 3) / 4, TU0 = NXC / 4;
        std::vector<float> o;
        for (int t = 0; t < NT; t++) // A1: fwd [A ; -K] * x
            for (int k = 0; k < NXC; k++)
                pack_one(tb, o, t, k, [&](RowRef out, RowRef in) {
                    if (in.fam != 0) return 0.f;
                    return out.fam == 0 ? Aat(out.idx, in.idx) : -Kat(out.idx, in.idx);
                });
        for (int t = 0; t < NTX; t++) // A2: fwd [B] * u
            for (int m = 0; m < NUC; m++)
                pack_one(tb, o, t, NXC + m, [&](RowRef out, RowRef in) { return (out.fam == 0 && in.fam == 1) ? Bat(out.idx, in.idx) : 0.f; });
        for (int t = 0; t < NT; t++) // A3: bwd [AmBKt ; B^T] * p
            for (int k = 0; k < NXC; k++)
                pack_one(tb, o, t, k, [&](RowRef out, RowRef in) {
                    if (in.fam != 0) return 0.f;
                    return out.fam == 0 ? Amat(out.idx, in.idx) : Bat(in.idx, out.idx);
                });
        for (int t = 0; t < NTX; t++) // A4: bwd [-K^T] * r
            for (int m = 0; m < NUC; m++)
                pack_one(tb, o, t, NXC + m, [&](RowRef out, RowRef in) { return (out.fam == 0 && in.fam == 1) ? -Kat(in.idx, out.idx) : 0.f; });
        for (int t = TU0; t < NT; t++) // A5: bwd [Quu_inv] * (B^T p + r)
            for (int m = 0; m < NUC; m++)
                pack_one(tb, o, t, NXC + m, [&](RowRef out, RowRef in) { return (out.fam == 1 && in.fam == 1) ? Qiat(out.idx, in.idx) : 0.f; });
        for (int t = 0; t < NTX; t++) // AP: terminal  p_j = -(sum_k Xref_k Pinf(k,j))
            for (int k = 0; k < NXC; k++)
                pack_one(tb, o, t, k, [&](RowRef out, RowRef in) { return (out.fam == 0 && in.fam == 0) ? -Pat(in.idx, out.idx) : 0.f; });
        std::vector<float> qv((size_t)WAVE * NXC, 0.f);
        for (int l = 0; l < WAVE; l++)
            for (int ch = 0; ch < NXC; ch++)
            {
                int r = 4 * ch + (l >> 4);
                if (r < nx) qv[(size_t)l * NXC + ch] = tb->Q[r];
            }
        TRY(upload_vec(tb, &tb->opnd, o));
        TRY(upload_vec(tb, &tb->qvec, qv));
    }
    if (tb->rowmath_ok || tb->wave_ok)
    {
        const int RW = tb->rw;
        // ---- gains for the rowlane kernel: [3nx + 2nu + 1][16], entry (reg, r) = the value lane r of a row holds.
        //   M1[k]  (k<nx): x rows A(r,k)      | u rows K(m,k) (exact: the kernel negates the SUM, like the reference) / -K(m,k) (fast)
        //   M2[m]  (m<nu): x rows B(r,m)      | u rows  0
        //   M3[k]  (k<nx): x rows AmBKt(r,k)  | u rows  B(k,m)          [= Bdyn^T]
        //   M45[m] (m<nu): x rows K(m,r) (exact) / -K(m,r) (fast)  | u rows  Quu_inv(mr,m)
        //   Q             : x rows Q(r)
        //   PT[k]  (k<nx): x rows Pinf(k,r)
        for (int fast = 0; fast < 2; fast++)
        {
            const float sg = fast ? -1.f : 1.f;
            const int nreg = 3 * nx + 3 * nu + 2; // ... + R (u rows) + coeff_d2p(r, m) (x rows): the optional terms
            std::vector<float> m((size_t)nreg * RW, 0.f);
            for (int r = 0; r < RW; r++)
            {
                const bool isx = r < nx, isu = r >= nx && r < nx + nu;
                const int mr = r - nx;
                for (int k = 0; k < nx; k++)
                {
                    m[(size_t)k * RW + r] = isx ? Aat(r, k) : (isu ? sg * Kat(mr, k) : 0.f); // u rows: K (exact), -K (fast)
                    m[(size_t)(nx + nu + k) * RW + r] = isx ? Amat(r, k) : (isu ? Bat(k, mr) : 0.f);
                    m[(size_t)(2 * nx + 2 * nu + 1 + k) * RW + r] = isx ? Pat(k, r) : 0.f;
                }
                for (int mm = 0; mm < nu; mm++)
                {
                    m[(size_t)(nx + mm) * RW + r] = isx ? Bat(r, mm) : 0.f;
                    m[(size_t)(2 * nx + nu + mm) * RW + r] = isx ? sg * Kat(mm, r) : (isu ? Qiat(mr, mm) : 0.f);
                }
                m[(size_t)(2 * nx + 2 * nu) * RW + r] = isx ? tb->Q[r] : 0.f;
                m[(size_t)(3 * nx + 2 * nu + 1) * RW + r] = (isu && tb->have_rcost) ? tb->Rcost[mr] : 0.f;
                for (int mm = 0; mm < nu; mm++)
                    m[(size_t)(3 * nx + 2 * nu + 2 + mm) * RW + r] = (isx && !tb->coeff_d2p.empty()) ? tb->coeff_d2p[(size_t)mm * nx + r] : 0.f;
            }
            TRY(upload_vec(tb, fast ? &tb->mats_fast : &tb->mats_exact, m));
        }
    }
    if (tb->generic_ok)
    {
        std::vector<float> gm;
        for (const std::vector<float> *m : {&tb->Kinf, &tb->Pinf, &tb->Quu_inv, &tb->AmBKt, &tb->Adyn, &tb->Bdyn, &tb->Q}) gm.insert(gm.end(), m->begin(), m->end());
        TRY(upload_vec(tb, &tb->gen_mats, gm));
    }
    if (!tb->dA) HIP_TRY(guarded_malloc((void **)&tb->dA, (size_t)nx * nx * sizeof(float)));
    if (!tb->dB) HIP_TRY(guarded_malloc((void **)&tb->dB, (size_t)nx * nu * sizeof(float)));
    HIP_TRY(hipMemcpyAsync(tb->dA, A, (size_t)nx * nx * sizeof(float), hipMemcpyHostToDevice, tb->stream));
    HIP_TRY(hipMemcpyAsync(tb->dB, B, (size_t)nx * nu * sizeof(float), hipMemcpyHostToDevice, tb->stream));
    HIP_TRY(hipStreamSynchronize(tb->stream));
    tb->gains_dirty = false;
    return 0;
}

// (re)build the layout-specific forms of the caller's inputs (Xref, bounds)
int prepare_inputs(TinyBatch *tb, int layout)
{
    if (!tb->derived_dirty[layout]) return 0;
    const int N = tb->N, nx = tb->nx, nu = tb->nu;
    if (layout == LAYOUT_TILE)
    {
        for (int k = 0; k < 4; k++)
        {
            const int fam = k < 2 ? 0 : 1;
            const size_t nf = fam ? tb->ufam_floats : tb->xfam_floats;
            if (!tb->t_bnd[k]) TRY(dev_alloc_zero(&tb->t_bnd[k], nf));
            const InputArr &in = tb->in_bnd[k];
            if (in.set) // shared inputs fill tile 0 only (tile stride 0 in the kernel)
                TRY(launch_pack(tb, in.dev, tb->t_bnd[k], LAYOUT_TILE, fam, in.shared ? TILE : tb->batch, in.shared, 0, fam ? N - 1 : N));
        }
        if (!tb->t_xref) TRY(dev_alloc_zero(&tb->t_xref, tb->xfam_floats));
        if (tb->in_xref.set)
            TRY(launch_pack(tb, tb->in_xref.dev, tb->t_xref, LAYOUT_TILE, 0, tb->in_xref.shared ? TILE : tb->batch, tb->in_xref.shared, 0, N));
    }
    else
    {
        // bounds table [N][16]{lo,hi}: +-inf where a bound is disabled or the row carries nothing; lo := min(lo, hi)
        // (min(hi, max(lo, t)) == med3(t, min(lo,hi), hi) for every t, also for the infeasible lo > hi case)
        const float inf = std::numeric_limits<float>::infinity();
        const int RW = tb->rw;
        if (!bounds_all_shared(tb)) // per-instance bounds: [bpad4][N][rw]{lo,hi}, built on the device from the canonical inputs
        {
            const size_t nf = (size_t)tb->bpad4 * N * RW * 2;
            TRY(sized_buffer(&tb->r_bounds, &tb->r_bounds_n, tb->h16 ? (nf + 1) / 2 : nf));
            const InputArr *in = tb->in_bnd;
            hipLaunchKernelGGL(bounds_table_kernel, dim3(grid_for((long long)tb->batch * N * RW)), dim3(256), 0, tb->stream,
                               in[0].set ? in[0].dev : nullptr, in[1].set ? in[1].dev : nullptr, in[2].set ? in[2].dev : nullptr,
                               in[3].set ? in[3].dev : nullptr, (in[0].set ? in[0].shared : in[1].shared) ? 1 : 0,
                               (in[2].set ? in[2].shared : in[3].shared) ? 1 : 0, tb->r_bounds, geo(tb), tb->batch, tb->en_state_bound,
                               tb->en_input_bound, tb->h16 ? 1 : 0);
            HIP_TRY(hipGetLastError());
        }
        else
        {
        std::vector<float> tab((size_t)N * RW * 2);
        for (int i = 0; i < N; i++)
            for (int r = 0; r < RW; r++)
            {
                float lo = -inf, hi = inf;
                if (r < nx && tb->en_state_bound)
                {
                    lo = tb->in_bnd[0].set ? tb->in_bnd[0].host[(size_t)i * nx + r] : 0.f;
                    hi = tb->in_bnd[1].set ? tb->in_bnd[1].host[(size_t)i * nx + r] : 0.f;
                }
                else if (r >= nx && r < nx + nu && i < N - 1 && tb->en_input_bound)
                {
                    lo = tb->in_bnd[2].set ? tb->in_bnd[2].host[(size_t)i * nu + (r - nx)] : 0.f;
                    hi = tb->in_bnd[3].set ? tb->in_bnd[3].host[(size_t)i * nu + (r - nx)] : 0.f;
                }
                tab[((size_t)i * RW + r) * 2 + 0] = lo < hi ? lo : hi;
                tab[((size_t)i * RW + r) * 2 + 1] = hi;
            }
        if (tb->h16) // same table in binary16 (bounds round to nearest; +-inf stays +-inf); half the floats
        {
            std::vector<_Float16> th(tab.size() + 1, (_Float16)0.f);
            for (size_t e = 0; e < tab.size(); e++) th[e] = (_Float16)tab[e];
            std::vector<float> packed((tab.size() + 1) / 2);
            std::memcpy(packed.data(), th.data(), packed.size() * sizeof(float));
            TRY(sized_buffer(&tb->r_bounds, &tb->r_bounds_n, packed.size()));
            TRY(upload_vec(tb, &tb->r_bounds, packed));
        }
        else
        {
            TRY(sized_buffer(&tb->r_bounds, &tb->r_bounds_n, tab.size()));
            TRY(upload_vec(tb, &tb->r_bounds, tab));
        }
        }
        const size_t nf = (size_t)(tb->in_xref.set && !tb->in_xref.shared ? tb->bpad4 : 1) * N * RW;
        TRY(sized_buffer(&tb->r_xref, &tb->r_xref_n, tb->h16 ? (nf + 1) / 2 : nf));
        if (!tb->in_xref.set) HIP_TRY(hipMemsetAsync(tb->r_xref, 0, tb->r_xref_n * sizeof(float), tb->stream)); // (a kept buffer of a reference since withdrawn)
        if (tb->in_xref.set)
            TRY(launch_pack(tb, tb->in_xref.dev, tb->r_xref, LAYOUT_ROW, 0, tb->in_xref.shared ? 1 : tb->batch, tb->in_xref.shared, 0, N));
        if (tb->in_uref.set) // Uref on the u rows of an [inst][N][rw] array of its own (row N-1 and the x rows stay zero)
        {
            const size_t nu_f = (size_t)(tb->in_uref.shared ? 1 : tb->bpad4) * N * RW;
            TRY(sized_buffer(&tb->r_uref, &tb->r_uref_n, tb->h16 ? (nu_f + 1) / 2 : nu_f));
            TRY(launch_pack(tb, tb->in_uref.dev, tb->r_uref, LAYOUT_ROW, 1, tb->in_uref.shared ? 1 : tb->batch, tb->in_uref.shared, 0, N - 1));
        }
        // per-instance tables of the class the sixteen-instances-per-wave kernel serves: do they change along the horizon? (one pass, at set time)
        tb->rows_vary = 3u;
        const bool xper = tb->xref_mode != 1 && tb->in_xref.set && !tb->in_xref.shared;
        if (tb->tile16_ok && !tb->h16 && tb->rw == 16 && (!bounds_all_shared(tb) || xper))
        {
            if (!tb->rows_vary_dev) TRY(dev_alloc_zero((float **)&tb->rows_vary_dev, 2));
            HIP_TRY(hipMemsetAsync(tb->rows_vary_dev, 0, 2 * sizeof(int), tb->stream));
            hipLaunchKernelGGL(rows_vary_kernel, dim3(grid_for((long long)tb->batch * 16)), dim3(256), 0, tb->stream,
                               bounds_all_shared(tb) ? nullptr : reinterpret_cast<const float2 *>(tb->r_bounds), xper ? tb->r_xref : nullptr, tb->batch, N, nx, nu,
                               tb->rows_vary_dev);
            HIP_TRY(hipGetLastError());
            int h[2] = {1, 1};
            HIP_TRY(hipMemcpyAsync(h, tb->rows_vary_dev, sizeof h, hipMemcpyDeviceToHost, tb->stream));
            HIP_TRY(hipStreamSynchronize(tb->stream));
            tb->rows_vary = (h[0] ? 1u : 0u) | (h[1] ? 2u : 0u);
            const long long ntl = (tb->batch + 15) / 16;
            if (!bounds_all_shared(tb) && (tb->rows_vary & 1u))
            {
                TRY(sized_buffer(&tb->r_bounds_img, &tb->r_bounds_img_n, (size_t)ntl * N * 2 * 64 * 4));
                hipLaunchKernelGGL(tile16_image_kernel, dim3(grid_for(ntl * N * 128)), dim3(256), 0, tb->stream, reinterpret_cast<const float4 *>(tb->r_bounds),
                                   reinterpret_cast<float4 *>(tb->r_bounds_img), tb->batch, N, 2, (long long)N * 8);
                HIP_TRY(hipGetLastError());
            }
            if (xper && (tb->rows_vary & 2u))
            {
                TRY(sized_buffer(&tb->r_xref_img, &tb->r_xref_img_n, (size_t)ntl * N * 64 * 4));
                hipLaunchKernelGGL(tile16_image_kernel, dim3(grid_for(ntl * N * 64)), dim3(256), 0, tb->stream, reinterpret_cast<const float4 *>(tb->r_xref),
                                   reinterpret_cast<float4 *>(tb->r_xref_img), tb->batch, N, 1, (long long)N * 4);
                HIP_TRY(hipGetLastError());
            }
        }
    }
    HIP_TRY(hipStreamSynchronize(tb->stream));
    tb->derived_dirty[layout] = false;
    return 0;
}

int row_family(const TinyBatch *tb);
// Automatic choice of the sixteen-instances-per-wave kernel (round 4, re-measured on the final binaries, 65 536-instance tracking workload cut to size, kernel ms,
// tile16 / 16-lane kernel, both longest first): 32 768: 1.03 / 0.97, 36 864: 1.11 / 1.08, 40 960: 1.09 / 1.19, 49 152: 1.24 / 1.40, 65 536: 1.54 / 1.83 — from 160
// instances per CU on; in index order the 16-lane kernel wins or ties at every size (65 536: 2.00 / 1.99), so the choice also asks for the longest-first dispatch
constexpr int kTile16AutoPerCu = 160;
int dispatch_effective(const TinyBatch *tb);
bool tile16_auto_size(const TinyBatch *tb) { return dispatch_effective(tb) == 1 && tb->batch >= kTile16AutoPerCu * tb->n_cu; }
// ... and for a closed-loop run (tiny_batch_mpc_run_async: all MPC steps of a tile inside one launch).  Since the tiles of a run are dispatched by the iteration
// counts of the solve before it (history order, dispatch_order.hip: a tile's total over the steps of a run spreads widely, and four tiles per wave slot in index
// order ended a third above even slots) the warm-started tracking loop measures, ms per MPC step of the second of two 20-step runs, tile16 / 16-lane kernel:
// 24 576: 0.45 / 0.41, 32 768: 0.53 / 0.52, 40 960: 0.59 / 0.65, 49 152: 0.64 / 0.76, 65 536: 0.84 / 0.94, 98 304: 1.13 / 1.38, 131 072: 1.51 / 1.83 — from 160
// instances per CU on (tools/closed_loop_threshold.py; in index order the cross-over was 240: 65 536: 0.96 / 1.02)
constexpr int kTile16ClosedLoopPerCu = 160;
bool tile16_closed_loop_size(const TinyBatch *tb) { return tb->batch >= kTile16ClosedLoopPerCu * tb->n_cu; }

// fp16 storage: bring the duals pair to the width the coming launch implements.  Under tiny_batch_set_storage(tb, 16) fp32 duals are a
// PREFERENCE (the register-resident 16-lane and quad kernels keep them, every other kernel — streamed state, per-instance bounds under
// fp16, the optional terms, the six single-function kernels — stores binary16 duals): the array is converted in place of being refused.
// An explicit tiny_batch_set_storage_ex(tb, 16, 32) stays a requirement (the caller is told when a kernel cannot honour it).
int settle_dual_width(TinyBatch *tb, bool kernel_keeps_fp32_duals)
{
    if (!tb->h16 || tb->dual32_forced) return 0;
    const bool want32 = tb->dual32_pref && kernel_keeps_fp32_duals;
    if (tb->dual32 == want32) return 0;
    if (tb->layout == LAYOUT_ROW && tb->pair[5])
    {
        float *nw = nullptr;
        TRY(dev_alloc_zero(&nw, want32 ? tb->pair_floats : (tb->pair_floats + 1) / 2));
        hipLaunchKernelGGL(dual_width_kernel, dim3(grid_for((long long)tb->pair_floats)), dim3(256), 0, tb->stream, tb->pair[5], nw, (long long)tb->pair_floats, want32 ? 1 : 0);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(tb->stream));
        (void)guarded_free(tb->pair[5]);
        tb->pair[5] = nw;
    }
    tb->dual32 = want32;
    invalidate_graph(tb); // the array pointer is a kernel argument
    return 0;
}
bool family_keeps_fp32_duals(int fam) { return fam == 0 || fam == 4; }

int resolve_variant(TinyBatch *tb, int *out)
{
    int v = tb->variant;
    // row variants: register-resident kernel when (nx,nu,N) is instantiated, else the any-N row kernel with the state in HBM
    // per-instance bounds: the streaming row kernel and the wave kernel read them per instance; the register-resident
    // kernels stage ONE table in LDS and need batch-shared bounds
    const bool row_ok = tb->row_dims_ok || tb->rowmath_ok || tb->wave_ok;
    if (tb->variant == VAR_ROW_FAST && tb->wave_ok && row_family(tb) != 6 && row_family(tb) != 7)
        return fail(TINY_BATCH_EUNSUPPORTED, "fma arithmetic for 16 < nx + nu <= 64 needs the state-on-chip wave kernel (N <= 50); beyond that it is the streaming MFMA kernel (variant 1)");
    if (v == VAR_AUTO)
    {
        // the automatic choice is exact arithmetic or nothing: a class without a compiled exact kernel runs in fma arithmetic on the padded
        // MFMA kernel only when the caller has asked for it by name (round 4: it used to be selected silently)
        if (!row_ok && !tb->generic_ok)
            return fail(TINY_BATCH_EUNSUPPORTED, "nx=%d nu=%d has no exact-arithmetic kernel (the reference's own summation order depends on column alignment "
                                                 "unless nx and nu are each <= 4 or a multiple of 4); tiny_batch_select_kernel(tb, 1) opts into fma arithmetic "
                                                 "on the MFMA streaming kernel", tb->nx, tb->nu);
        v = row_ok ? VAR_ROW_EXACT : VAR_GENERIC; // a class outside the compiled lists: exact arithmetic with run-time dimensions (admm_generic.hip)
    }
    if (v == VAR_GENERIC && (!tb->generic_ok || tb->h16 || tb->en_uref || tb->en_d2p))
        return fail(TINY_BATCH_EUNSUPPORTED, "the run-time-dimension exact kernel (variant 4) needs nx, nu each <= 4 or a multiple of 4 (nx <= 64, nu <= 32), "
                                             "fp32 storage and no optional terms (nx=%d nu=%d)", tb->nx, tb->nu);
    if ((tb->en_uref || tb->en_d2p) && (!tb->rowmath_ok || tile_variant(v)))
        return fail(TINY_BATCH_EUNSUPPORTED, "the optional Uref / coeff_d2p terms (tiny_batch_set_optional_terms) are implemented by the row "
                                             "kernels for nx + nu <= 16 only (nx=%d nu=%d, variant %d)", tb->nx, tb->nu, v);
    if ((v == VAR_ROW_EXACT || v == VAR_ROW_FAST) && !row_ok)
    {
        return fail(TINY_BATCH_EUNSUPPORTED, "no row kernel instantiation for nx=%d nu=%d (needs nx + nu <= 16)", tb->nx, tb->nu);
    }
    if (v == VAR_STREAM && !tb->tile_dims_ok)
        return fail(TINY_BATCH_EUNSUPPORTED, "no streaming kernel instantiation for nx=%d nu=%d", tb->nx, tb->nu);
    if (v == VAR_STREAM && tb->h16)
        return fail(TINY_BATCH_EUNSUPPORTED, "fp16 storage is implemented by the row kernels only (nx + nu <= 16)");
    *out = v;
    return 0;
}

// which of the three row kernels a row variant launches: 0 = unrolled register-resident (rowlane, fastest, one
// instantiation per (nx, nu, N)), 1 = rolled-loop register-resident (rowloop, any N <= 64), 2 = any N with the state in
// HBM (rowstream).  tiny_batch_set_row_kernel() can force one of them.
// admm_tile16.hip needs fp32 storage, a reference it does not have to keep resident (a window of a trajectory table that
// fits its LDS share, or one shared reference) and — checked by the caller — batch-shared bounds and no optional terms
// Round 4: per-instance bounds and a per-instance reference are served by the PI instantiations (rows fetched by LDS-DMA into per-wave rings,
// admm_tile16_pi.hip), one solve per launch; a closed-loop run with either keeps the 16-lane kernel.
bool tile16_per_instance(const TinyBatch *tb)
{
    return !bounds_all_shared(tb) || (tb->xref_mode != 1 && tb->in_xref.set && !tb->in_xref.shared);
}
// which tables of a pi launch go through the per-wave LDS-DMA slots, and which of those change along the horizon (RowParams::pi_flags)
struct Tile16Pi { bool bounds_ring, xref_ring, fits; unsigned flags; }; // flags: RowParams::pi_flags
Tile16Pi tile16_pi_plan(const TinyBatch *tb)
{
    Tile16Pi p;
    p.bounds_ring = !bounds_all_shared(tb);
    p.xref_ring = tb->xref_mode != 1 && tb->in_xref.set && !tb->in_xref.shared;
    p.flags = (p.bounds_ring ? (tb->rows_vary & 1u) : 0u) | (p.xref_ring ? (tb->rows_vary & 2u) : 0u);
    const int rows = tb->xref_mode == 1 ? tb->table_rows : tb->N;
    p.fits = tile16_pi_lds_bytes(tb->N, p.bounds_ring, p.xref_ring, p.flags, rows) <= 160 * 1024; // a staged table too long for the LDS share beside the slots
    return p;
}
// automatic choice with per-instance tables: the same rule as with shared ones (65 536 tracking instances, longest first, kernel ms, tile16 pi against the 16-lane
// kernel: bounds constant along the horizon 1.75 / 2.13, reference per step 1.81 / 1.96, both per step 1.95 / 2.15; in index order the rings lose, 2.45 / 2.22)
bool tile16_pi_auto(const TinyBatch *tb)
{
    return !tb->order_dev && tile16_auto_size(tb);
}
bool tile16_applies(const TinyBatch *tb)
{
    if (!tb->tile16_ok || tb->h16) return false;
    if (tile16_per_instance(tb)) return !tb->closed_loop_run && tile16_pi_plan(tb).fits;
    if (tb->xref_mode == 1) return tb->table_rows <= tile16_max_table_rows();
    return true;
}

// auto between the two state-on-chip kernels of the nx = 32 class, by rounds of the launch (measured on the 256 CUs of an MI355X,
// bench workload: one round of the wave kernel = 2 048 instances = 5.0 ms, one round of the tile kernel = 4 096 instances = 7.2 ms;
// 2 048: 5.4 against 7.1 ms, 2 304: 8.9 / 7.1, 4 096: 10.1 / 7.4, 4 352: 13.4 / 14.9, 6 144: 15.0 / 14.4, 16 384: 37.5 / 28.6; round 4: tile48 27.6)
bool tile48_pays(int batch, int n_cu)
{
    const long slots_w = 8L * n_cu, slots_t = 16L * n_cu; // resident instances: two waves per SIMD / one workgroup of sixteen per CU
    if (batch <= slots_w) return false;
    const long rounds_w = (batch + slots_w - 1) / slots_w, rounds_t = (batch + slots_t - 1) / slots_t;
    // a launch costs its rounds; ONE ratio carries the comparison (round 4: it used to be two absolute times of one workload): a round of the tile kernel
    // (sixteen instances per CU) takes 1.42 rounds of the wave kernel (eight per CU) — 6.9 against 4.85 ms at N = 50 on the MI355X, and the two scale
    // alike with the horizon and the iteration count; the wave kernel's last, partly filled round overlaps the one before (x 0.93)
    constexpr double kTileRoundInWaveRounds = 1.42;
    return rounds_t * kTileRoundInWaveRounds <= rounds_w * 0.93;
}

int row_family(const TinyBatch *tb)
{
    // one wavefront per instance: state on chip where the horizon fits (admm_waveres.hip, 6), else streamed through HBM (admm_wave.hip, 3)
    // (7: sixteen instances per workgroup on the matrix cores, admm_tile48.hip: fp32 storage)
    if (tb->wave_ok)
    {
        if (tb->row_family_forced == 3) return 3;
        if (tb->tile48_ok && !tb->h16 && (tb->row_family_forced == 7 || (tb->row_family_forced < 0 && tile48_pays(tb->batch, tb->n_cu)))) return 7;
        return tb->waveres_ok ? 6 : 3;
    }
    // per-instance bounds: the unrolled register-resident kernel (fp32 storage) and the rolled-loop ones (N <= 64, either storage)
    // read them from the [B][N][16] table, the streaming row kernel serves every other case; the quad kernel stages one shared table
    // the optional terms (Uref, coeff_d2p): on the unrolled register-resident kernel since round 4 (fp32 storage, batch-shared bounds, one solve per
    // launch); every other combination — fp16 storage, per-instance bounds, a closed-loop run, another forced family — on the streaming row kernel
    if ((tb->en_uref || tb->en_d2p) && tb->row_dims_ok && !tb->h16 && bounds_all_shared(tb) && !tb->closed_loop_run &&
        (tb->row_family_forced < 0 || tb->row_family_forced == 0))
        return 0;
    if (!bounds_all_shared(tb))
    {
        if (tb->en_uref || tb->en_d2p) return 2;
        if (tile16_applies(tb) && (tb->row_family_forced == 5 || (tb->row_family_forced < 0 && tile16_pi_auto(tb)))) return 5;
        const int ff = tb->row_family_forced == 5 ? -1 : tb->row_family_forced; // tile16 asked for but not applicable (closed-loop run): like auto
        if (tb->row_dims_ok && !tb->h16 && (ff < 0 || ff == 0)) return 0;
        if (tb->rowloop_ok && (ff < 0 || ff == 1)) return 1; // one step ahead from global memory
        return 2;
    }
    if (tb->en_uref || tb->en_d2p) return 2; // the optional terms live in the streaming row kernel (c's u rows hold d elsewhere)
    // 5 = sixteen instances per wave, products on the matrix cores (admm_tile16.hip): on request only; needs fp32 storage and
    // a reference it does not have to keep resident (window of a table, or one shared reference)
    if (tb->row_family_forced == 5)
        return tile16_applies(tb) ? 5 : (tb->row_dims_ok ? 0 : (tb->rowloop_ok ? 1 : 2));
    if (tb->row_family_forced >= 0) return tb->row_family_forced;
    if (tb->quad_ok) return 4; // four lanes per instance (admm_quadlane.hip): nx = 4, nu = 1
    // auto (round 3): sixteen instances per wave on the matrix cores where the launch is at least two rounds deep for its one
    // wave per SIMD (2 048 tiles) — measured 1.79 against 1.91 ms on 65 536 tracking instances; smaller launches fill the chip
    // better with four instances per wave, and a closed-loop run keeps the kernel whose MPC loop stays on chip
    // (a caller that hands over its own dispatch order lists groups of four instances: the automatic choice then stays with the
    // kernel that order is for)
    // (round 4: tile16's MPC loop stays on chip too — tiny_batch_set_row_kernel(tb, 5) — but the warm-started solves of a closed loop are short and
    //  uneven, and sixteen instances in lock step lose more there than the matrix cores gain: measured 1.02 ms per MPC step of 65 536 tracking
    //  instances against 0.97 ms on the 16-lane kernel, so the automatic choice of a closed-loop run stays with the latter)
    if (tile16_applies(tb) && !tb->order_dev && (tb->closed_loop_run ? tile16_closed_loop_size(tb) : tile16_auto_size(tb))) return 5;
    if (tb->row_dims_ok) return 0;
    if (tb->rowloop_ok) return 1;
    return 2;
}

hipError_t launch_tile16_pi(const TinyBatch *tb, bool exact, RowParams &P)
{
    const Tile16Pi pl = tile16_pi_plan(tb);
    P.pi_flags = pl.flags;
    // ring tables are read from their tile images (a window of the trajectory table through the ring: from the table's rows)
    if (pl.bounds_ring && (pl.flags & 1u))
    {
        if (!tb->r_bounds_img) return hipErrorInvalidValue;
        P.bounds = tb->r_bounds_img;
    }
    if (pl.xref_ring && (pl.flags & 2u))
    {
        if (!tb->r_xref_img) return hipErrorInvalidValue;
        P.xref = tb->r_xref_img;
    }
    return launch_admm_tile16_pi(tb->N, exact, pl.bounds_ring, pl.xref_ring, P, tb->stream, tb->n_cu, tb->tile_queue);
}

void update_kname(TinyBatch *tb)
{
    int v = 0;
    char nm[96];
    const std::string keep = g_err;
    if (resolve_variant(tb, &v)) { tb->kname = "unsupported"; g_err = keep; return; }
    const bool d32 = tb->h16 && (tb->dual32_forced ? tb->dual32 : (tb->dual32_pref && !tile_variant(v) && family_keeps_fp32_duals(row_family(tb))));
    const char *ar = v == VAR_ROW_EXACT ? "exact" : "fast", *sto = tb->h16 ? (d32 ? ",h16d" : ",h16") : "";
    if (v == VAR_STREAM) snprintf(nm, sizeof nm, "stream<%d,%d>", tb->NXC, tb->NUC);
    else if (v == VAR_GENERIC) snprintf(nm, sizeof nm, "generic<%d,%d,exact>", tb->nx, tb->nu);
    else if (row_family(tb) == 0) snprintf(nm, sizeof nm, "rowlane<%d,%d,%d,%s%s>", tb->nx, tb->nu, tb->N, ar, sto);
    else if (row_family(tb) == 1) snprintf(nm, sizeof nm, "rowloop<%d,%d,%s%s>", tb->nx, tb->nu, ar, sto);
    else if (row_family(tb) == 3) snprintf(nm, sizeof nm, "wavestream<%d,%d,%s>", tb->nx, tb->nu, ar);
    else if (row_family(tb) == 6) snprintf(nm, sizeof nm, "waveres<%d,%d,%s>", tb->nx, tb->nu, ar);
    else if (row_family(tb) == 7) snprintf(nm, sizeof nm, "tile48<%d,%d,%d,%s>", tb->nx, tb->nu, tb->N, ar);
    else if (row_family(tb) == 4) snprintf(nm, sizeof nm, "quadlane<%d,%d,%d,%s%s>", tb->nx, tb->nu, tb->N, ar, sto);
    else if (row_family(tb) == 5) snprintf(nm, sizeof nm, "tile16<%d,%d,%d,%s%s>", tb->nx, tb->nu, tb->N, ar, tile16_per_instance(tb) ? ",pi" : "");
    else snprintf(nm, sizeof nm, "rowstream<%d,%d,%s%s>", tb->nx, tb->nu, ar, sto);
    tb->kname = nm;
}

void fill_row_params(TinyBatch *tb, RowParams &P, bool exact)
{
    P.nx = tb->nx; P.nu = tb->nu; P.N = tb->N; P.batch = tb->batch;
    P.rho = tb->rho; P.abs_pri_tol = tb->abs_pri_tol; P.abs_dua_tol = tb->abs_dua_tol;
    P.max_iter = tb->max_iter; P.check_termination = tb->check_termination;
    P.duals_zero = tb->duals_zero_pending ? 1 : 0;
    P.cold_start = tb->cold_pending ? 1 : 0;
    P.xref_mode = tb->xref_mode;
    P.xu = tb->pair[0]; P.qr = tb->pair[1]; P.pd = tb->pair[2]; P.vz = tb->pair[3]; P.vzn = tb->pair[4]; P.gy = tb->pair[5];
    P.xref = tb->r_xref;
    P.xref_inst_stride = (tb->in_xref.set && !tb->in_xref.shared) ? (unsigned)(tb->N * tb->rw) : 0u;
    P.pi_flags = 0u;
    P.xref_table = tb->h16 ? tb->tab_row_h : tb->tab_row; P.xref_start = tb->xref_start; P.table_rows = tb->table_rows;
    P.bounds = tb->r_bounds;
    P.bounds_inst_stride = bounds_all_shared(tb) ? 0u : (unsigned)(tb->N * tb->rw);
    P.mats = exact ? tb->mats_exact : tb->mats_fast;
    P.uref = tb->en_uref ? tb->r_uref : nullptr;
    P.uref_inst_stride = (tb->in_uref.set && !tb->in_uref.shared) ? (unsigned)(tb->N * tb->rw) : 0u;
    P.en_d2p = tb->en_d2p ? 1 : 0;
    P.order = tb->order_dev;
    P.res = tb->res; P.status = tb->status; P.iter = tb->iter; P.n_unsolved = tb->n_unsolved;
    P.mpc_steps = 1; P.window_advance = 0; P.u0_traj = nullptr; P.x0buf = tb->x0buf;
    P.dual32 = (tb->h16 && tb->dual32) ? 1 : 0;
}

int check_optional_terms(const TinyBatch *tb)
{
    if (tb->en_uref && (!tb->have_rcost || !tb->in_uref.set))
        return fail(TINY_BATCH_ENOTREADY, "the Uref term is enabled: tiny_batch_set_input_cost and tiny_batch_set_uref must be called first");
    if (tb->en_d2p && tb->coeff_d2p.empty())
        return fail(TINY_BATCH_ENOTREADY, "the coeff_d2p term is enabled: tiny_batch_set_coeff_d2p must be called first");
    return 0;
}

// One of the six step functions of admm.hpp:10-18 over the whole batch (admm_steps.hip).
int run_step(TinyBatch *tb, int fn, int *converged_host, int *n_true)
{
    if (!tb->have_cache || !tb->have_dyn || !tb->have_settings)
        return fail(TINY_BATCH_ENOTREADY, "set_cache, set_dynamics and set_settings must be called first");
    if (!tb->rowmath_ok)
        return fail(TINY_BATCH_EUNSUPPORTED, "the single-function kernels need nx + nu <= 16 and an entry in TINY_FOR_EACH_ROWDIMS (nx=%d nu=%d)", tb->nx, tb->nu);
    if (tb->dual32_forced) return fail(TINY_BATCH_EUNSUPPORTED, "the single-function kernels do not implement fp16 storage with fp32 duals (tiny_batch_set_storage_ex(tb, 16, 32))");
    TRY(check_optional_terms(tb));
    TRY(set_device(tb));
    if (tb->gains_dirty) TRY(pack_gains(tb));
    TRY(ensure_layout(tb, LAYOUT_ROW));
    TRY(settle_dual_width(tb, false));
    TRY(prepare_inputs(tb, LAYOUT_ROW));
    TRY(flush_pending(tb));
    if (!tb->conv_dev) TRY(dev_alloc_zero((float **)&tb->conv_dev, tb->batch));
    RowParams P;
    fill_row_params(tb, P, tb->variant != VAR_ROW_FAST);
    HIP_TRY(hipMemsetAsync(tb->n_unsolved, 0, 2 * sizeof(int), tb->stream)); // [0] unsolved count, [1] tile queue of admm_tile16.hip
    hipError_t e = launch_admm_step(tb->nx, tb->nu, tb->variant != VAR_ROW_FAST, tb->h16, fn, P, tb->conv_dev, tb->stream);
    if (e != hipSuccess) return fail(TINY_BATCH_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
    if (fn == STEP_TERMINATION_CONDITION)
    {
        int n_false = 0;
        HIP_TRY(hipMemcpyAsync(&n_false, tb->n_unsolved, sizeof(int), hipMemcpyDeviceToHost, tb->stream));
        if (converged_host)
            HIP_TRY(hipMemcpyAsync(converged_host, tb->conv_dev, (size_t)tb->batch * sizeof(int), hipMemcpyDeviceToHost, tb->stream));
        HIP_TRY(hipStreamSynchronize(tb->stream));
        if (n_true) *n_true = tb->batch - n_false;
    }
    return 0;
}

// everything a solve needs that may allocate, copy or synchronise (not capturable in a hipGraph)
int prepare_solve(TinyBatch *tb, int *variant)
{
    if (!tb->have_cache || !tb->have_dyn || !tb->have_settings)
        return fail(TINY_BATCH_ENOTREADY, "tiny_batch_solve: set_cache, set_dynamics and set_settings must be called first");
    for (int k = 0; k < 4; k += 2)
    {
        const InputArr &lo = tb->in_bnd[k], &hi = tb->in_bnd[k + 1];
        const bool lo_inst = lo.set && !lo.shared, hi_inst = hi.set && !hi.shared;
        if (lo_inst != hi_inst && (lo.set || hi.set))
            return fail(TINY_BATCH_EINVAL, "min and max bounds must both be shared or both be per-instance");
    }
    TRY(check_optional_terms(tb));
    TRY(set_device(tb));
    int v = 0;
    TRY(resolve_variant(tb, &v));
    if (tb->gains_dirty) TRY(pack_gains(tb));
    // (whatever the mode says NOW: a launch sequence enqueued after one prepare_solve — the captured graph of tiny_batch_mpc_run_async — gains a history with its
    //  first solve, and its second one is then dispatched by it)
    if (!tb->order_buf && tb->bpad4 / 4 >= kDispatchMinGroups)
    {
        TRY(dev_alloc_zero(&tb->key_buf, (size_t)tb->bpad4 / 4 + (size_t)tb->bpad4 / 16 + 16)); // group keys, then tile keys
        TRY(dev_alloc_zero((float **)&tb->order_buf, (size_t)tb->bpad4 / 4));
    }
    {
        const int layout = tile_variant(v) ? LAYOUT_TILE : LAYOUT_ROW;
        TRY(ensure_layout(tb, layout));
        TRY(settle_dual_width(tb, layout == LAYOUT_ROW && family_keeps_fp32_duals(row_family(tb))));
        TRY(prepare_inputs(tb, layout));
    }
    TRY(flush_x0_zero(tb)); // every solve reads x.col(0)
    if (tb->max_iter <= 0) TRY(flush_pending(tb));
    // A cold start that converges in its first iteration (x0 at the origin) runs no backward sweep, which is what writes p, d, v, z.
    // reset_workspace() is folded into the launch by every fused kernel: the register-resident ones start from zero registers,
    // the streaming ones (MFMA, rowstream, wavestream) read zeros in their first iteration and zero-fill p, d, v, z of such an
    // instance in their epilogue
    update_kname(tb);
    *variant = v;
    return 0;
}

// the stream operations of one solve: counter reset + kernel launch (capturable)

int enqueue_solve(TinyBatch *tb, int v, bool record_events)
{
    const int layout = tile_variant(v) ? LAYOUT_TILE : LAYOUT_ROW;
    // longest-first dispatch (dispatch_order.hip): predictor sweep + bucket sort ahead of the register-resident 16-lane kernels;
    // pays off only when the launch is several rounds of waves deep
    const int fam_l = layout == LAYOUT_ROW ? row_family(tb) : -1;
    const bool predicted_order = layout == LAYOUT_ROW && dispatch_effective(tb) == 1 && !tb->order_dev && (fam_l == 0 || fam_l == 1 || fam_l == 5) && !tb->dual32 &&
                                 tb->bpad4 / 4 >= kDispatchMinGroups && tb->max_iter > 1 && tb->order_buf;
    // [0] unsolved count, [1] tile queue of admm_tile16.hip: zeroed by the sort kernel of the predicted order where that runs (one stream node less)
    const bool history_order = layout == LAYOUT_ROW && dispatch_effective(tb) == 2 && !tb->order_dev && (fam_l == 0 || fam_l == 1 || fam_l == 5) &&
                               tb->bpad4 / 4 >= kDispatchMinGroups && tb->max_iter > 1 && tb->order_buf;
    if (history_order)
    {
        hipError_t ek = launch_dispatch_order_history(tb->iter, tb->batch, fam_l == 5 ? 16 : 4, tb->order_buf, tb->n_unsolved, tb->stream);
        if (ek != hipSuccess) return fail(TINY_BATCH_EHIP, "kernel launch failed: %s", hipGetErrorString(ek));
    }
    else if (!predicted_order) HIP_TRY(hipMemsetAsync(tb->n_unsolved, 0, 2 * sizeof(int), tb->stream));
    if (predicted_order)
    {
        RowParams K;
        fill_row_params(tb, K, false); // fma gains
        hipError_t ek = launch_dispatch_order(tb->nx, tb->nu, tb->h16, K, tb->key_buf, tb->order_buf, tb->stream, fam_l == 5);
        if (ek != hipSuccess) return fail(TINY_BATCH_EHIP, "kernel launch failed: %s", hipGetErrorString(ek));
    }
    if (record_events) HIP_TRY(hipEventRecord(tb->ev0, tb->stream)); // the events bracket the solve kernel itself
    hipError_t e;
    if (layout == LAYOUT_TILE)
    {
        SolveParams P;
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
        P.xmin = tb->t_bnd[0]; P.xmax = tb->t_bnd[1]; P.umin = tb->t_bnd[2]; P.umax = tb->t_bnd[3]; P.xref = tb->t_xref;
        const long long xt = (long long)tb->N * WAVE * tb->NXC, ut = (long long)(tb->N - 1) * WAVE * tb->NUC;
        P.xb_tile_stride = (tb->in_bnd[0].set && !tb->in_bnd[0].shared) ? xt : 0;
        P.ub_tile_stride = (tb->in_bnd[2].set && !tb->in_bnd[2].shared) ? ut : 0;
        P.xref_tile_stride = (tb->in_xref.set && !tb->in_xref.shared) ? xt : 0;
        P.xref_table = tb->tab_tile; P.xref_start = tb->xref_start; P.table_rows = tb->table_rows;
        P.res = tb->res; P.status = tb->status; P.iter = tb->iter; P.n_unsolved = tb->n_unsolved;
        P.opnd = tb->opnd; P.qvec = tb->qvec;
        e = v == VAR_GENERIC ? launch_admm_generic(P, tb->gen_mats, tb->NXC, tb->NUC, tb->stream) : launch_admm_stream(tb->NXC, tb->NUC, P, tb->stream);
    }
    else
    {
        RowParams P;
        fill_row_params(tb, P, v == VAR_ROW_EXACT);
        if (predicted_order || history_order) P.order = tb->order_buf;
        const int fam = row_family(tb);
        if (fam == 5 && !predicted_order && !history_order) P.order = nullptr; // a caller's order lists groups of four instances, not tiles of sixteen
        tb->last_dispatch = predicted_order ? 1 : history_order ? 3 : (P.order ? 2 : 0);
        if (P.dual32 && fam != 0 && fam != 4)
            return fail(TINY_BATCH_EUNSUPPORTED, "fp16 storage with fp32 duals runs on the register-resident 16-lane and quad kernels only "
                                                 "(batch-shared bounds, no optional terms, no forced row kernel)");
        e = fam == 0   ? launch_admm_rowlane(tb->nx, tb->nu, tb->N, v == VAR_ROW_EXACT, tb->h16, P, tb->stream)
            : fam == 1 ? launch_admm_rowloop(tb->nx, tb->nu, v == VAR_ROW_EXACT, tb->h16, P, tb->stream)
            : fam == 3 ? launch_admm_wavestream(tb->nx, tb->nu, P, tb->stream)
            : fam == 4 ? launch_admm_quadlane(tb->N, v == VAR_ROW_EXACT, tb->h16, P, tb->stream)
            : fam == 5 ? (tile16_per_instance(tb) ? launch_tile16_pi(tb, v == VAR_ROW_EXACT, P)
                                                  : launch_admm_tile16(tb->N, v == VAR_ROW_EXACT, P, tb->stream, tb->n_cu, tb->tile_queue))
            : fam == 6 ? launch_admm_waveres(tb->nx, tb->nu, v == VAR_ROW_EXACT, P, tb->stream)
            : fam == 7 ? launch_admm_tile48(tb->N, v == VAR_ROW_EXACT, P, tb->stream)
                       : launch_admm_rowstream(tb->nx, tb->nu, v == VAR_ROW_EXACT, tb->h16, P, tb->stream);
    }
    if (e != hipSuccess) return fail(TINY_BATCH_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
    if (record_events)
    {
        HIP_TRY(hipEventRecord(tb->ev1, tb->stream));
        tb->ev_valid = true;
    }
    if (tb->max_iter > 0) tb->duals_zero_pending = tb->cold_pending = false; // consumed by the kernel's first iteration
    if (tb->max_iter > 0) tb->iter_history = true;                            // iter[] now holds this launch's counts
    return 0;
}

int launch_solve(TinyBatch *tb)
{
    int v = 0;
    TRY(prepare_solve(tb, &v));
    return enqueue_solve(tb, v, tb->timing);
}

int enqueue_plant_step(TinyBatch *tb, int window_advance)
{
    hipLaunchKernelGGL(plant_step_kernel, dim3((tb->batch + 127) / 128), dim3(128), 0, tb->stream, tb->x0buf,
                       work_ptr(tb, TINY_ARR_X), work_ptr(tb, TINY_ARR_U), tb->dA, tb->dB,
                       tb->xref_mode == 1 ? tb->xref_start : nullptr, window_advance, tb->batch, tb->layout, geo(tb),
                       h16_of(tb, tb->layout));
    HIP_TRY(hipGetLastError());
    return 0;
}

int set_bound(TinyBatch *tb, const float *src, int shared, int which)
{
    CHECK_TB(tb);
    CHECK_PTR(src);
    TRY(set_device(tb));
    const bool xf = which < 2;
    return store_input(tb, tb->in_bnd[which], src, shared != 0, xf ? tb->N : tb->N - 1, xf ? tb->nx : tb->nu);
}

// The MFMA streaming kernel is instantiated per (chunks of 4 state rows, chunks of 4 input rows).  A class without its own
// instantiation runs on the smallest one that contains it: the extra chunks are rows that do not exist (zero gains, zero
// state, no bounds — exactly like the unused rows of a partly filled chunk), so any nx <= 64, nu <= 32 is served.
bool stream_dims_supported(int nxc, int nuc, int *pxc, int *puc)
{
    int best = 1 << 30;
    bool found = false;
#define TINY_CHECK_DIMS(NXC, NUC)                                        \
    if (nxc <= NXC && nuc <= NUC && (NXC + NUC) * 64 + NXC < best)       \
    {                                                                    \
        best = (NXC + NUC) * 64 + NXC; *pxc = NXC; *puc = NUC; found = true; \
    }
    TINY_FOR_EACH_DIMS(TINY_CHECK_DIMS)
    return found;
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
    int nxc = (nx + 3) / 4, nuc = (nu + 3) / 4;
    // the rowlane kernel addresses its arrays with 32-bit element offsets
    const bool tile_ok = stream_dims_supported(nxc, nuc, &nxc, &nuc),
               row_ok = rowlane_supported(nx, nu, N) && ((long long)(batch + 3) * N * 16 < (1ll << 30));
    const bool wave_ok = !rowdims_supported(nx, nu) && wavedims_supported(nx, nu) && ((long long)(batch + 3) * N * 64 < (1ll << 30));
    if (!tile_ok && !row_ok && !rowdims_supported(nx, nu) && !wave_ok)
        return fail(TINY_BATCH_EUNSUPPORTED,
                    "no kernel for nx=%d nu=%d N=%d: exact arithmetic needs a compiled class (TINY_FOR_EACH_ROWDIMS / _WAVEDIMS), fma arithmetic nx <= 64 and nu <= 32",
                    nx, nu, N);
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(TINY_BATCH_EINVAL, "device %d out of range (have %d)", device, ndev);
    TinyBatch *tb = new TinyBatch();
    tb->nx = nx; tb->nu = nu; tb->N = N; tb->batch = batch; tb->device = device;
    tb->NXC = nxc; tb->NUC = nuc; tb->ntiles = (batch + TILE - 1) / TILE; tb->bpad4 = (batch + 3) / 4 * 4;
    tb->tile_dims_ok = tile_ok; tb->row_dims_ok = row_ok;
    tb->rowmath_ok = rowdims_supported(nx, nu) && ((long long)(batch + 3) * N * 16 < (1ll << 30));
    tb->rowloop_ok = tb->rowmath_ok && rowloop_supported(nx, nu, N);
    tb->wave_ok = wave_ok;
    tb->quad_ok = tb->rowmath_ok && quadlane_supported(nx, nu, N);
    tb->tile16_ok = tb->rowmath_ok && tile16_supported(nx, nu, N);
    tb->waveres_ok = wave_ok && waveres_supported(nx, nu, N);
    tb->tile48_ok = wave_ok && tile48_supported(nx, nu, N);
    tb->generic_ok = tile_ok && generic_exact_supported(nx, nu);
    tb->rw = wave_ok ? 64 : 16;
    tb->xfam_floats = (size_t)tb->ntiles * N * WAVE * nxc;
    tb->ufam_floats = (size_t)tb->ntiles * (N - 1) * WAVE * nuc;
    tb->pair_floats = (size_t)tb->bpad4 * N * tb->rw;
    tb->layout = (row_ok || wave_ok || !tile_ok) ? LAYOUT_ROW : LAYOUT_TILE;
    auto cleanup = [&](int rc) { tiny_batch_destroy(tb); return rc; };
    if (hipSetDevice(device) != hipSuccess) return cleanup(fail(TINY_BATCH_EHIP, "hipSetDevice(%d) failed", device));
    {
        int ncu = 0; // of THIS handle's device (round-3 advisor: a process-wide static used to cache whichever device was current first)
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && ncu > 0) tb->n_cu = ncu;
    }
    if (int rc = alloc_layout(tb, tb->layout)) return cleanup(rc);
    if (int rc = dev_alloc_zero(&tb->res, (size_t)batch * 4)) return cleanup(rc);
    if (int rc = dev_alloc_zero((float **)&tb->status, batch)) return cleanup(rc);
    if (int rc = dev_alloc_zero((float **)&tb->iter, batch)) return cleanup(rc);
    if (int rc = dev_alloc_zero((float **)&tb->n_unsolved, 2)) return cleanup(rc);
    if (int rc = dev_alloc_zero((float **)&tb->xref_start, batch)) return cleanup(rc);
    if (int rc = dev_alloc_zero(&tb->x0buf, (size_t)batch * nx)) return cleanup(rc);
    tb->staging_floats = (size_t)batch * N * (nx > nu ? nx : nu);
    if (int rc = dev_alloc_zero(&tb->staging, tb->staging_floats)) return cleanup(rc);
    if (hipEventCreate(&tb->ev0) != hipSuccess || hipEventCreate(&tb->ev1) != hipSuccess)
        return cleanup(fail(TINY_BATCH_EHIP, "hipEventCreate failed"));
    tb->cold_pending = true; // a fresh workspace IS a reset one (every array was allocated zero): its first solve is a cold start, like one after tiny_batch_reset_workspace
    update_kname(tb);
    *out = tb;
    return TINY_BATCH_OK;
}

void tiny_batch_destroy(TinyBatch *tb)
{
    if (!tb) return;
    (void)hipSetDevice(tb->device);
    free_layout(tb, LAYOUT_TILE);
    free_layout(tb, LAYOUT_ROW);
    (void)guarded_free(tb->in_xref.dev);
    (void)guarded_free(tb->in_uref.dev); (void)guarded_free(tb->r_uref);
    (void)guarded_free(tb->key_buf); (void)guarded_free(tb->order_buf); (void)guarded_free(tb->u0_stage);
    for (int k = 0; k < 4; k++) { (void)guarded_free(tb->in_bnd[k].dev); (void)guarded_free(tb->t_bnd[k]); }
    (void)guarded_free(tb->t_xref); (void)guarded_free(tb->r_xref); (void)guarded_free(tb->r_bounds); (void)guarded_free((float *)tb->rows_vary_dev);
    (void)guarded_free(tb->r_bounds_img); (void)guarded_free(tb->r_xref_img);
    (void)guarded_free(tb->tab_tile); (void)guarded_free(tb->tab_row); (void)guarded_free(tb->tab_row_h); (void)guarded_free(tb->xref_start);
    (void)guarded_free(tb->res); (void)guarded_free(tb->status); (void)guarded_free(tb->iter); (void)guarded_free(tb->n_unsolved);
    (void)guarded_free(tb->opnd); (void)guarded_free(tb->qvec); (void)guarded_free(tb->gen_mats); (void)guarded_free(tb->mats_exact); (void)guarded_free(tb->mats_fast);
    (void)guarded_free(tb->dA); (void)guarded_free(tb->dB); (void)guarded_free(tb->x0buf); (void)guarded_free(tb->staging); (void)guarded_free(tb->conv_dev);
    if (tb->graph_exec) (void)hipGraphExecDestroy(tb->graph_exec);
    if (tb->own_stream) (void)hipStreamDestroy(tb->own_stream);
    if (tb->ev0) (void)hipEventDestroy(tb->ev0);
    if (tb->ev1) (void)hipEventDestroy(tb->ev1);
    delete tb;
}

int tiny_batch_set_stream(TinyBatch *tb, void *hip_stream)
{
    CHECK_TB(tb);
    invalidate_graph(tb); // synchronises the stream a replay may still be running on: before that stream is replaced
    tb->stream = (hipStream_t)hip_stream;
    return 0;
}

int tiny_batch_synchronize(TinyBatch *tb)
{
    CHECK_TB(tb);
    TRY(set_device(tb));
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
    tb->gains_dirty = true;
    invalidate_graph(tb); // rho is a kernel argument
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
    tb->gains_dirty = true;
    invalidate_graph(tb);
    return 0;
}

// ---- the two terms the reference ships commented out (admm.cpp:20 and :79), off by default ------------------------------
int tiny_batch_set_optional_terms(TinyBatch *tb, int en_uref, int en_coeff_d2p)
{
    CHECK_TB(tb);
    tb->en_uref = en_uref != 0;
    tb->en_d2p = en_coeff_d2p != 0;
    invalidate_graph(tb);
    update_kname(tb);
    return 0;
}

int tiny_batch_set_input_cost(TinyBatch *tb, const float *R)
{
    CHECK_TB(tb); CHECK_PTR(R);
    tb->Rcost.assign(R, R + tb->nu);
    tb->have_rcost = true;
    tb->gains_dirty = true;
    invalidate_graph(tb);
    return 0;
}

int tiny_batch_set_coeff_d2p(TinyBatch *tb, const float *coeff_d2p)
{
    CHECK_TB(tb); CHECK_PTR(coeff_d2p);
    tb->coeff_d2p.assign(coeff_d2p, coeff_d2p + (size_t)tb->nx * tb->nu);
    tb->gains_dirty = true;
    invalidate_graph(tb);
    return 0;
}

int tiny_batch_set_uref(TinyBatch *tb, const float *uref, int shared)
{
    CHECK_TB(tb); CHECK_PTR(uref);
    TRY(set_device(tb));
    TRY(store_input(tb, tb->in_uref, uref, shared != 0, tb->N - 1, tb->nu));
    return 0;
}

int tiny_batch_set_dispatch(TinyBatch *tb, int mode)
{
    CHECK_TB(tb);
    if (mode < -1 || mode > 2) return fail(TINY_BATCH_EINVAL, "tiny_batch_set_dispatch: mode must be 0 (index order), 1 (longest first, predicted), 2 (longest first by the previous solve's iteration counts) or -1 (automatic)");
    tb->dispatch_mode = mode;
    invalidate_graph(tb);
    return 0;
}

int tiny_batch_set_tile_queue(TinyBatch *tb, int stride)
{
    CHECK_TB(tb);
    if (stride < -1 || stride > 255) return fail(TINY_BATCH_EINVAL, "tiny_batch_set_tile_queue: stride must be -1 (automatic), 0 (one counter) or 1 .. 255");
    tb->tile_queue = stride;
    invalidate_graph(tb);
    return TINY_BATCH_OK;
}

int tiny_batch_dispatch_applied(TinyBatch *tb)
{
    CHECK_TB(tb);
    return tb->last_dispatch;
}

int tiny_batch_set_dispatch_order_device(TinyBatch *tb, const int *d_order)
{
    CHECK_TB(tb);
    tb->order_dev = d_order;
    invalidate_graph(tb);
    update_kname(tb); // a caller's order keeps the 16-lane kernel it is written for
    return 0;
}

int tiny_batch_set_settings(TinyBatch *tb, float abs_pri_tol, float abs_dua_tol, int max_iter, int check_termination,
                            int en_state_bound, int en_input_bound)
{
    CHECK_TB(tb);
    if (check_termination < 1)
        return fail(TINY_BATCH_EINVAL, "check_termination must be >= 1 (the reference computes iter %% check_termination, admm.cpp:93)");
    if (tb->en_state_bound != en_state_bound || tb->en_input_bound != en_input_bound) tb->derived_dirty[LAYOUT_ROW] = true;
    tb->abs_pri_tol = abs_pri_tol; tb->abs_dua_tol = abs_dua_tol;
    tb->max_iter = max_iter; tb->check_termination = check_termination;
    tb->en_state_bound = en_state_bound; tb->en_input_bound = en_input_bound;
    tb->have_settings = true; // tolerances, max_iter, check_termination and the bound flags are in the graph signature
    return 0;
}

int tiny_batch_set_x0(TinyBatch *tb, const float *x0)
{
    CHECK_TB(tb); CHECK_PTR(x0);
    TRY(set_device(tb));
    tb->x0_zero_pending = false; // all of x.col(0) and x0buf is overwritten
    HIP_TRY(hipMemcpyAsync(tb->x0buf, x0, (size_t)tb->batch * tb->nx * sizeof(float), hipMemcpyHostToDevice, tb->stream));
    return upload_work(tb, x0, TINY_ARR_X, 0, 1);
}

int tiny_batch_set_x0_device(TinyBatch *tb, const float *d_x0)
{
    CHECK_TB(tb); CHECK_PTR(d_x0);
    TRY(set_device(tb));
    tb->x0_zero_pending = false; // all of x.col(0) and x0buf is overwritten
    return launch_pack(tb, d_x0, work_ptr(tb, TINY_ARR_X), tb->layout, 0, tb->batch, false, 0, 1, tb->x0buf); // x.col(0) and the state buffer in one launch
}

int tiny_batch_set_xref(TinyBatch *tb, const float *xref, int shared)
{
    CHECK_TB(tb); CHECK_PTR(xref);
    TRY(set_device(tb));
    TRY(store_input(tb, tb->in_xref, xref, shared != 0, tb->N, tb->nx));
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
    TRY(set_device(tb));
    // table on the device in both forms: [rows][4 gq][NXC] (streaming kernel) and [rows][rw] (row / wave kernels)
    std::vector<float> tt((size_t)rows * 4 * tb->NXC, 0.f), tr((size_t)rows * tb->rw, 0.f);
    for (int r = 0; r < rows; r++)
        for (int row = 0; row < tb->nx; row++)
        {
            const float v = table[(size_t)r * tb->nx + row];
            tt[((size_t)r * 4 + (row & 3)) * tb->NXC + (row >> 2)] = v;
            if (row < tb->rw) tr[(size_t)r * tb->rw + row] = v;
        }
    if (tb->table_rows != rows)
    {
        (void)guarded_free(tb->tab_tile); (void)guarded_free(tb->tab_row); (void)guarded_free(tb->tab_row_h);
        tb->tab_tile = tb->tab_row = tb->tab_row_h = nullptr;
    }
    TRY(upload_vec(tb, &tb->tab_tile, tt));
    TRY(upload_vec(tb, &tb->tab_row, tr));
    {
        std::vector<_Float16> th(tr.size());
        for (size_t e = 0; e < tr.size(); e++) th[e] = (_Float16)tr[e];
        std::vector<float> packed(tr.size() / 2); // rows * rw halves
        std::memcpy(packed.data(), th.data(), packed.size() * sizeof(float));
        TRY(upload_vec(tb, &tb->tab_row_h, packed));
    }
    HIP_TRY(hipMemcpyAsync(tb->xref_start, start, (size_t)tb->batch * sizeof(int), hipMemcpyHostToDevice, tb->stream));
    HIP_TRY(hipStreamSynchronize(tb->stream));
    tb->table_rows = rows;
    tb->xref_mode = 1;
    invalidate_graph(tb);
    return 0;
}

int tiny_batch_set_xmin(TinyBatch *tb, const float *s, int shared) { return set_bound(tb, s, shared, 0); }
int tiny_batch_set_xmax(TinyBatch *tb, const float *s, int shared) { return set_bound(tb, s, shared, 1); }
int tiny_batch_set_umin(TinyBatch *tb, const float *s, int shared) { return set_bound(tb, s, shared, 2); }
int tiny_batch_set_umax(TinyBatch *tb, const float *s, int shared) { return set_bound(tb, s, shared, 3); }

int tiny_batch_reset_dual_variables(TinyBatch *tb)
{
    CHECK_TB(tb);
    tb->duals_zero_pending = true; // folded into the next solve's first iteration; flushed by any other reader
    return 0;
}

int tiny_batch_solve_async(TinyBatch *tb)
{
    CHECK_TB(tb);
    return launch_solve(tb);
}

int tiny_batch_wait(TinyBatch *tb, int *n_unsolved)
{
    CHECK_TB(tb);
    TRY(set_device(tb));
    int n = 0;
    HIP_TRY(hipMemcpyAsync(&n, tb->n_unsolved, sizeof(int), hipMemcpyDeviceToHost, tb->stream));
    HIP_TRY(hipStreamSynchronize(tb->stream));
    if (n_unsolved) *n_unsolved = n;
    return 0;
}

int tiny_batch_solve(TinyBatch *tb)
{
    TRY(tiny_batch_solve_async(tb));
    int n = 0;
    TRY(tiny_batch_wait(tb, &n));
    return n > 0 ? 1 : 0;
}

// Mixed problem classes in one call (BASELINE.json configs[4]): every handle is one class (nx, nu, N, storage); all
// solves are enqueued before any is waited for, each on its handle's stream, so they overlap on the device.
int tiny_batch_group_solve(TinyBatch **tbs, int n, int *n_unsolved)
{
    CHECK_PTR(tbs);
    if (n < 1) return fail(TINY_BATCH_EINVAL, "tiny_batch_group_solve: need at least one handle");
    for (int i = 0; i < n; i++)
    {
        CHECK_TB(tbs[i]);
        for (int j = 0; j < i; j++)
            if (tbs[j] == tbs[i]) return fail(TINY_BATCH_EINVAL, "tiny_batch_group_solve: handle %d appears twice", i);
    }
    for (int i = 0; i < n; i++)
    {
        TinyBatch *tb = tbs[i];
        if (!tb->stream) // the null stream would serialise the group
        {
            TRY(set_device(tb));
            HIP_TRY(hipStreamSynchronize(nullptr)); // work already queued on the null stream (uploads) comes first
            if (!tb->own_stream) HIP_TRY(hipStreamCreateWithFlags(&tb->own_stream, hipStreamNonBlocking));
            tb->stream = tb->own_stream;
        }
        TRY(launch_solve(tb));
    }
    int total = 0;
    for (int i = 0; i < n; i++)
    {
        int u = 0;
        TRY(tiny_batch_wait(tbs[i], &u));
        total += u;
    }
    if (n_unsolved) *n_unsolved = total;
    return total > 0 ? 1 : 0;
}

// Multi-device epilogue: handles that each own a block of the instance index (one handle per GPU, one
// stream per handle) deliver u.col(0) of all their instances into ONE device buffer — handle order = block order.  Each
// handle unpacks its first input column into a buffer of its own device and, where that is not the destination device, the
// block travels device-to-device (hipMemcpyPeerAsync: xGMI between the GPUs of a node, no host staging), all copies in flight
// together, one wait per handle at the end.  No collective: the path has none (the batch shards with no exchange step).
int tiny_batch_group_gather_u0(TinyBatch **tbs, int n, int dst_device, float *d_dst)
{
    CHECK_PTR(tbs); CHECK_PTR(d_dst);
    if (n < 1) return fail(TINY_BATCH_EINVAL, "tiny_batch_group_gather_u0: need at least one handle");
    for (int i = 0; i < n; i++)
    {
        CHECK_TB(tbs[i]);
        if (tbs[i]->nu != tbs[0]->nu) return fail(TINY_BATCH_EINVAL, "tiny_batch_group_gather_u0: handle %d has nu=%d, handle 0 nu=%d", i, tbs[i]->nu, tbs[0]->nu);
    }
    size_t off = 0;
    for (int i = 0; i < n; i++)
    {
        TinyBatch *tb = tbs[i];
        const size_t cnt = (size_t)tb->batch * tb->nu;
        TRY(set_device(tb));
        if (tb->device == dst_device)
            TRY(launch_unpack(tb, work_ptr(tb, TINY_ARR_U), d_dst + off, tb->layout, 1, tb->batch, 0, 1)); // straight into its block
        else
        {
            if (!tb->u0_stage) TRY(dev_alloc_zero(&tb->u0_stage, cnt));
            TRY(launch_unpack(tb, work_ptr(tb, TINY_ARR_U), tb->u0_stage, tb->layout, 1, tb->batch, 0, 1));
            HIP_TRY(hipMemcpyPeerAsync(d_dst + off, dst_device, tb->u0_stage, tb->device, cnt * sizeof(float), tb->stream));
        }
        off += cnt;
    }
    for (int i = 0; i < n; i++)
    {
        TRY(set_device(tbs[i]));
        HIP_TRY(hipStreamSynchronize(tbs[i]->stream));
    }
    return 0;
}

// the same into host memory: gathered on the device of handle 0, then ONE device-to-host copy
int tiny_batch_group_get_u0(TinyBatch **tbs, int n, float *u0_host)
{
    CHECK_PTR(tbs); CHECK_PTR(u0_host);
    if (n < 1) return fail(TINY_BATCH_EINVAL, "tiny_batch_group_get_u0: need at least one handle");
    size_t total = 0;
    for (int i = 0; i < n; i++) { CHECK_TB(tbs[i]); total += (size_t)tbs[i]->batch * tbs[i]->nu; }
    TRY(set_device(tbs[0]));
    float *d = nullptr;
    HIP_TRY(guarded_malloc((void **)&d, total * sizeof(float)));
    int rc = tiny_batch_group_gather_u0(tbs, n, tbs[0]->device, d);
    if (rc == 0)
    {
        (void)hipSetDevice(tbs[0]->device);
        hipError_t e = hipMemcpy(u0_host, d, total * sizeof(float), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(TINY_BATCH_EHIP, "tiny_batch_group_get_u0: %s", hipGetErrorString(e));
    }
    (void)hipSetDevice(tbs[0]->device);
    (void)guarded_free(d);
    return rc;
}

int tiny_batch_forward_pass(TinyBatch *tb) { CHECK_TB(tb); return run_step(tb, STEP_FORWARD_PASS, nullptr, nullptr); }
int tiny_batch_update_slack(TinyBatch *tb) { CHECK_TB(tb); return run_step(tb, STEP_UPDATE_SLACK, nullptr, nullptr); }
int tiny_batch_update_dual(TinyBatch *tb) { CHECK_TB(tb); return run_step(tb, STEP_UPDATE_DUAL, nullptr, nullptr); }
int tiny_batch_update_linear_cost(TinyBatch *tb) { CHECK_TB(tb); return run_step(tb, STEP_UPDATE_LINEAR_COST, nullptr, nullptr); }
int tiny_batch_backward_pass_grad(TinyBatch *tb) { CHECK_TB(tb); return run_step(tb, STEP_BACKWARD_PASS_GRAD, nullptr, nullptr); }
int tiny_batch_termination_condition(TinyBatch *tb, int *converged)
{
    CHECK_TB(tb);
    int n_true = 0;
    TRY(run_step(tb, STEP_TERMINATION_CONDITION, converged, &n_true));
    return n_true;
}

int tiny_batch_get_x(TinyBatch *tb, float *x) { return tiny_batch_get_array(tb, TINY_ARR_X, x); }
int tiny_batch_get_u(TinyBatch *tb, float *u) { return tiny_batch_get_array(tb, TINY_ARR_U, u); }

int tiny_batch_get_status(TinyBatch *tb, int *iter, int *status, float *residuals)
{
    CHECK_TB(tb);
    TRY(set_device(tb));
    TRY(flush_pending(tb));
    if (iter) HIP_TRY(hipMemcpyAsync(iter, tb->iter, (size_t)tb->batch * sizeof(int), hipMemcpyDeviceToHost, tb->stream));
    if (status) HIP_TRY(hipMemcpyAsync(status, tb->status, (size_t)tb->batch * sizeof(int), hipMemcpyDeviceToHost, tb->stream));
    if (residuals) HIP_TRY(hipMemcpyAsync(residuals, tb->res, (size_t)tb->batch * 4 * sizeof(float), hipMemcpyDeviceToHost, tb->stream));
    HIP_TRY(hipStreamSynchronize(tb->stream));
    return 0;
}

int tiny_batch_set_status(TinyBatch *tb, const int *iter, const int *status, const float *residuals)
{
    CHECK_TB(tb);
    TRY(set_device(tb));
    TRY(flush_pending(tb));
    if (iter) { HIP_TRY(hipMemcpyAsync(tb->iter, iter, (size_t)tb->batch * sizeof(int), hipMemcpyHostToDevice, tb->stream)); tb->iter_history = false; }
    if (status) HIP_TRY(hipMemcpyAsync(tb->status, status, (size_t)tb->batch * sizeof(int), hipMemcpyHostToDevice, tb->stream));
    if (residuals) HIP_TRY(hipMemcpyAsync(tb->res, residuals, (size_t)tb->batch * 4 * sizeof(float), hipMemcpyHostToDevice, tb->stream));
    HIP_TRY(hipStreamSynchronize(tb->stream));
    return 0;
}

int tiny_batch_set_array(TinyBatch *tb, int id, const float *src)
{
    CHECK_TB(tb); CHECK_PTR(src);
    if (id < 0 || id >= TINY_ARR_COUNT) return fail(TINY_BATCH_EINVAL, "bad array id %d", id);
    TRY(set_device(tb));
    TRY(flush_pending(tb));
    TRY(upload_work(tb, src, id, 0, is_xfam(id) ? tb->N : tb->N - 1));
    if (id == TINY_ARR_X) // keep the closed-loop state buffer coherent with x.col(0)
        TRY(launch_unpack(tb, work_ptr(tb, TINY_ARR_X), tb->x0buf, tb->layout, 0, tb->batch, 0, 1));
    return 0;
}

int tiny_batch_get_array(TinyBatch *tb, int id, float *dst)
{
    CHECK_TB(tb); CHECK_PTR(dst);
    if (id < 0 || id >= TINY_ARR_COUNT) return fail(TINY_BATCH_EINVAL, "bad array id %d", id);
    TRY(set_device(tb));
    TRY(flush_pending(tb));
    return download_work(tb, id, dst, 0, is_xfam(id) ? tb->N : tb->N - 1);
}

int tiny_batch_reset_workspace(TinyBatch *tb)
{
    CHECK_TB(tb);
    TRY(set_device(tb));
    // Everything is zeroed lazily: the next solve reads d,v,z,y,g (and p) as zero in its first iteration and overwrites
    // the rest, any other reader triggers the zero fill (flush_pending); x.col(0) is zeroed before the next solve unless
    // a set_x0 overwrites it first (flush_x0_zero).
    tb->x0_zero_pending = true;
    tb->cold_pending = true;
    tb->duals_zero_pending = false;
    tb->iter_history = false;
    return 0;
}

// Device-pointer forms of set_array / get_array / set_xref: same host-layout arrays ([B][N][nx] / [B][N-1][nu], fp32),
// but resident in device memory on the handle's device; converted to/from the private layout by one kernel on the
// handle's stream, no host round trip and no synchronisation.
int tiny_batch_set_array_device(TinyBatch *tb, int id, const float *d_src)
{
    CHECK_TB(tb); CHECK_PTR(d_src);
    if (id < 0 || id >= TINY_ARR_COUNT) return fail(TINY_BATCH_EINVAL, "bad array id %d", id);
    TRY(set_device(tb));
    TRY(flush_pending(tb));
    const int fam = is_xfam(id) ? 0 : 1;
    TRY(launch_pack(tb, d_src, work_ptr(tb, id), tb->layout, fam, tb->batch, false, 0, fam ? tb->N - 1 : tb->N));
    if (id == TINY_ARR_X) TRY(launch_unpack(tb, work_ptr(tb, TINY_ARR_X), tb->x0buf, tb->layout, 0, tb->batch, 0, 1));
    return 0;
}

int tiny_batch_get_array_device(TinyBatch *tb, int id, float *d_dst)
{
    CHECK_TB(tb); CHECK_PTR(d_dst);
    if (id < 0 || id >= TINY_ARR_COUNT) return fail(TINY_BATCH_EINVAL, "bad array id %d", id);
    TRY(set_device(tb));
    TRY(flush_pending(tb));
    const int fam = is_xfam(id) ? 0 : 1;
    return launch_unpack(tb, work_ptr(tb, id), d_dst, tb->layout, fam, tb->batch, 0, fam ? tb->N - 1 : tb->N);
}

int tiny_batch_set_xref_device(TinyBatch *tb, const float *d_xref, int shared)
{
    CHECK_TB(tb); CHECK_PTR(d_xref);
    TRY(set_device(tb));
    InputArr &in = tb->in_xref;
    const bool sh = shared != 0;
    const size_t n = (size_t)(sh ? 1 : tb->batch) * tb->N * tb->nx;
    if (in.dev && in.shared != sh) { (void)guarded_free(in.dev); in.dev = nullptr; }
    if (!in.dev) HIP_TRY(guarded_malloc((void **)&in.dev, n * sizeof(float)));
    HIP_TRY(hipMemcpyAsync(in.dev, d_xref, n * sizeof(float), hipMemcpyDeviceToDevice, tb->stream));
    in.shared = sh;
    in.set = true;
    in.host.clear(); // only the bounds are ever read on the host
    tb->derived_dirty[0] = tb->derived_dirty[1] = true;
    tb->xref_mode = 0;
    return 0;
}

int tiny_batch_get_u0_device(TinyBatch *tb, float *d_u0)
{
    CHECK_TB(tb); CHECK_PTR(d_u0);
    TRY(set_device(tb));
    return launch_unpack(tb, work_ptr(tb, TINY_ARR_U), d_u0, tb->layout, 1, tb->batch, 0, 1);
}

int tiny_batch_mpc_step_async(TinyBatch *tb, int window_advance)
{
    CHECK_TB(tb);
    if (tb->nx > 64) return fail(TINY_BATCH_EUNSUPPORTED, "mpc_step supports nx <= 64");
    if (window_advance < 0) return fail(TINY_BATCH_EINVAL, "tiny_batch_mpc_step_async: window_advance must be >= 0 (got %d): the window gather clamps at the last table row only", window_advance);
    // x.col(0) already holds x0 (set_x0 / previous plant step); reset duals, solve, then simulate forward.
    tb->duals_zero_pending = true;
    TRY(tiny_batch_solve_async(tb));
    return enqueue_plant_step(tb, window_advance);
}

// `steps` closed-loop MPC steps back to back; d_u0_traj (device, [steps][B][nu], may be NULL) receives u.col(0) of every
// step.  Two implementations with identical results:
//  * on chip (the unrolled row kernel and the quad kernel, fp32 storage): ONE launch runs all the steps, the state never leaves registers/LDS
//    between solves (admm_rowlane.hip, MPC = true); the host only adds the plant step of the last solve;
//  * otherwise the launch sequence (counter reset, solve kernel, plant kernel) x steps is captured ONCE into a hipGraph and
//    replayed, which removes the per-launch overhead that dominates small batches with short warm-started solves.
int tiny_batch_mpc_run_traj_async(TinyBatch *tb, int steps, int window_advance, float *d_u0_traj)
{
    CHECK_TB(tb);
    if (steps < 1) return fail(TINY_BATCH_EINVAL, "tiny_batch_mpc_run_async: steps must be >= 1");
    if (window_advance < 0) return fail(TINY_BATCH_EINVAL, "tiny_batch_mpc_run_async: window_advance must be >= 0 (got %d): the window gather clamps at the last table row only", window_advance);
    if (tb->nx > 64) return fail(TINY_BATCH_EUNSUPPORTED, "mpc_run supports nx <= 64");
    if (tb->max_iter <= 0) return fail(TINY_BATCH_EINVAL, "tiny_batch_mpc_run_async needs max_iter > 0");
    TRY(set_device(tb));
    const bool from_reset = tb->cold_pending; // (flush_pending materialises the zeros and clears the flag: the run's first solve is the cold one all the same)
    TRY(flush_pending(tb)); // every solve of the run starts from "duals reset, workspace warm"
    tb->duals_zero_pending = true;
    int v = 0;
    struct Scope { TinyBatch *t; ~Scope() { t->closed_loop_run = false; } } scope{tb};
    tb->closed_loop_run = steps > 1;
    TRY(prepare_solve(tb, &v));
    const size_t u0n = (size_t)tb->batch * tb->nu;
    const int fam = !tile_variant(v) ? row_family(tb) : -1;
    if ((fam == 0 || fam == 4 || fam == 5) && !tb->h16 && steps > 1 && bounds_all_shared(tb))
    {
        RowParams P;
        fill_row_params(tb, P, v == VAR_ROW_EXACT);
        P.mpc_steps = steps; P.window_advance = window_advance; P.u0_traj = d_u0_traj;
        if (fam == 5) P.order = nullptr; // a caller's order lists groups of four instances, not tiles
        // the run's tiles / groups longest first by the counts of the solve before it (the last step of the previous run): a tile's total over the steps
        // of a run spreads 150 ... 700 iterations around a mean of 280 (65 536 tracking instances, 20 steps) and four tiles per wave slot in index order end
        // 33 % above even slots; ordered by the previous step's counts 15 % (tests/fuzz/sim_history_dispatch.py)
        const bool history_order = dispatch_effective(tb) == 2 && !tb->order_dev && (fam == 0 || fam == 5) && tb->bpad4 / 4 >= kDispatchMinGroups && tb->order_buf;
        // ... and a run that starts from a reset workspace by the predictor of its first, cold solve (which is also its longest: 22 iterations against 11): steps 0 - 19
        // of the tracking loop, makespan 1 522 iterations in index order, 1 299 by the predictor (by the true first-step counts 1 291; 16-lane kernel 2 336 -> 2 175)
        const bool predicted_order = (tb->dispatch_mode == 1 || (tb->dispatch_mode == -1 && from_reset)) && !history_order && !tb->order_dev && (fam == 0 || fam == 5) && !tb->dual32 && tb->bpad4 / 4 >= kDispatchMinGroups &&
                                     tb->order_buf && tb->max_iter > 1;
        if (history_order)
        {
            hipError_t ek = launch_dispatch_order_history(tb->iter, tb->batch, fam == 5 ? 16 : 4, tb->order_buf, tb->n_unsolved, tb->stream, fam == 5 /* a tile's total over the run: by the sum */);
            if (ek != hipSuccess) return fail(TINY_BATCH_EHIP, "kernel launch failed: %s", hipGetErrorString(ek));
            P.order = tb->order_buf;
        }
        else if (predicted_order)
        {
            RowParams K;
            fill_row_params(tb, K, false); // fma gains
            hipError_t ek = launch_dispatch_order(tb->nx, tb->nu, tb->h16, K, tb->key_buf, tb->order_buf, tb->stream, fam == 5); // (its sort zeroes the two counters)
            if (ek != hipSuccess) return fail(TINY_BATCH_EHIP, "kernel launch failed: %s", hipGetErrorString(ek));
            P.order = tb->order_buf;
        }
        else HIP_TRY(hipMemsetAsync(tb->n_unsolved, 0, 2 * sizeof(int), tb->stream)); // [0] unsolved count, [1] tile queue of admm_tile16.hip
        tb->last_dispatch = history_order ? 3 : predicted_order ? 1 : (P.order ? 2 : 0);
        hipError_t e = fam == 0   ? launch_admm_rowlane(tb->nx, tb->nu, tb->N, v == VAR_ROW_EXACT, false, P, tb->stream)
                       : fam == 5 ? launch_admm_tile16(tb->N, v == VAR_ROW_EXACT, P, tb->stream, tb->n_cu)
                                  : launch_admm_quadlane(tb->N, v == VAR_ROW_EXACT, false, P, tb->stream);
        if (e != hipSuccess) return fail(TINY_BATCH_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
        if (d_u0_traj) TRY(launch_unpack(tb, work_ptr(tb, TINY_ARR_U), d_u0_traj + (size_t)(steps - 1) * u0n, tb->layout, 1, tb->batch, 0, 1));
        TRY(enqueue_plant_step(tb, window_advance));
        tb->duals_zero_pending = tb->cold_pending = false;
        tb->iter_history = true; // iter[] = the counts of the run's last solve
        tb->ev_valid = false;
        return 0;
    }
    if (!tb->stream) // stream capture is not allowed on the null stream
    {
        HIP_TRY(hipStreamSynchronize(nullptr));
        if (!tb->own_stream) HIP_TRY(hipStreamCreateWithFlags(&tb->own_stream, hipStreamNonBlocking));
        tb->stream = tb->own_stream;
    }
    // the graph bakes in kernel arguments: rebuild it whenever anything they depend on may have changed
    // (round 3: the signature is complete — rho, the shared / per-instance modes that set the strides, the bound flags and the
    // dispatch order are in it — so setters that only change buffer CONTENTS no longer drop the graph: `set_xref; mpc_run(k)`
    // in a loop replays one captured graph instead of re-capturing it every step)
    char sig[480];
    snprintf(sig, sizeof sig, "%d|%d|%d|%s|%d|%d|%g|%g|%d|%d|%p|%p|%p|%d|%p|%p|%p|%p|%d%d|%a|%d%d%d|%d%d|%d|%p|%p|%p", steps, window_advance, v, tb->kname.c_str(), tb->max_iter,
             tb->check_termination, (double)tb->abs_pri_tol, (double)tb->abs_dua_tol, tb->xref_mode, tb->table_rows, (void *)tb->pair[0],
             (void *)tb->arr[0], (void *)tb->r_bounds, (int)tb->h16, (void *)tb->stream, (void *)d_u0_traj, (void *)tb->r_xref,
             (void *)tb->r_uref, (int)tb->en_uref, (int)tb->en_d2p, (double)tb->rho, (int)bounds_all_shared(tb),
             (int)(tb->in_xref.set && tb->in_xref.shared), (int)(tb->in_uref.set && tb->in_uref.shared), tb->en_state_bound, tb->en_input_bound,
             tb->dispatch_mode, (void *)tb->order_dev, (void *)tb->mats_exact, (void *)tb->xref_start);
    if (!tb->graph_exec || tb->graph_sig != sig)
    {
        if (tb->graph_exec) { (void)hipGraphExecDestroy(tb->graph_exec); tb->graph_exec = nullptr; }
        hipGraph_t graph = nullptr;
        HIP_TRY(hipStreamBeginCapture(tb->stream, hipStreamCaptureModeThreadLocal));
        int rc = 0;
        for (int k = 0; k < steps && rc == 0; k++)
        {
            tb->duals_zero_pending = true;
            rc = enqueue_solve(tb, v, false);
            if (rc == 0 && d_u0_traj) rc = launch_unpack(tb, work_ptr(tb, TINY_ARR_U), d_u0_traj + (size_t)k * u0n, tb->layout, 1, tb->batch, 0, 1);
            if (rc == 0) rc = enqueue_plant_step(tb, window_advance);
        }
        hipError_t ec = hipStreamEndCapture(tb->stream, &graph);
        if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        HIP_TRY(ec);
        hipError_t ei = hipGraphInstantiate(&tb->graph_exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        HIP_TRY(ei);
        tb->graph_sig = sig;
        tb->graph_captures++;
    }
    HIP_TRY(hipGraphLaunch(tb->graph_exec, tb->stream));
    tb->duals_zero_pending = tb->cold_pending = false;
    tb->ev_valid = false;
    return 0;
}

int tiny_batch_mpc_run_async(TinyBatch *tb, int steps, int window_advance)
{
    return tiny_batch_mpc_run_traj_async(tb, steps, window_advance, nullptr);
}

// host-copy variant: the trajectory buffer is allocated on the HANDLE's device (whatever device is current for the calling
// thread), the run is waited for and u.col(0) of every step lands in host memory
int tiny_batch_mpc_run_traj(TinyBatch *tb, int steps, int window_advance, float *u0_traj_host)
{
    CHECK_TB(tb); CHECK_PTR(u0_traj_host);
    if (steps < 1) return fail(TINY_BATCH_EINVAL, "tiny_batch_mpc_run_traj: steps must be >= 1");
    TRY(set_device(tb));
    const size_t n = (size_t)steps * tb->batch * tb->nu;
    float *d = nullptr;
    HIP_TRY(guarded_malloc((void **)&d, n * sizeof(float)));
    int rc = tiny_batch_mpc_run_traj_async(tb, steps, window_advance, d);
    if (rc == 0)
    {
        hipError_t e = hipStreamSynchronize(tb->stream);
        if (e == hipSuccess) e = hipMemcpy(u0_traj_host, d, n * sizeof(float), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(TINY_BATCH_EHIP, "tiny_batch_mpc_run_traj: %s", hipGetErrorString(e));
    }
    (void)guarded_free(d);
    return rc;
}

int tiny_batch_get_x0(TinyBatch *tb, float *x0)
{
    CHECK_TB(tb); CHECK_PTR(x0);
    TRY(set_device(tb));
    TRY(flush_x0_zero(tb));
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
    TRY(set_device(tb));
    HIP_TRY(hipEventSynchronize(tb->ev1));
    HIP_TRY(hipEventElapsedTime(ms, tb->ev0, tb->ev1));
    return 0;
}

const char *tiny_batch_kernel_name(TinyBatch *tb)
{
    if (!tb) return "";
    // per-instance tables of the class admm_tile16_pi.hip serves: whether they change along the horizon (one resident row per instance, or rings)
    // is found when the derived tables are built, and the automatic choice depends on it — build them now, as the next solve would
    if (tb->tile16_ok && !tb->h16 && tb->derived_dirty[LAYOUT_ROW] && tile16_per_instance(tb))
    {
        const std::string keep = g_err;
        int v = 0;
        (void)prepare_solve(tb, &v); // (not ready yet: the name is then the one the present knowledge gives)
        g_err = keep;
    }
    update_kname(tb);
    return tb->kname.c_str();
}

int tiny_batch_debug_graph_captures(TinyBatch *tb)
{
    CHECK_TB(tb);
    return tb->graph_captures; // how often tiny_batch_mpc_run_* had to capture its hipGraph (a replay does not count)
}

int tiny_batch_debug_guards(int on)
{
    std::lock_guard<std::mutex> lk(g_guard_mu);
    g_guards_on = on != 0; // applies to allocations made from now on (handles created earlier keep what they have)
    return 0;
}

long long tiny_batch_debug_check(void)
{
    std::lock_guard<std::mutex> lk(g_guard_mu);
    if (g_guarded.empty()) return 0;
    unsigned long long *bad = nullptr, host = 0;
    if (hipMalloc((void **)&bad, sizeof *bad) != hipSuccess || hipMemset(bad, 0, sizeof *bad) != hipSuccess) return fail(TINY_BATCH_EHIP, "tiny_batch_debug_check: allocation failed");
    if (hipDeviceSynchronize() != hipSuccess) { (void)hipFree(bad); return fail(TINY_BATCH_EHIP, "tiny_batch_debug_check: a kernel faulted: %s", hipGetErrorString(hipGetLastError())); }
    for (const auto &kv : g_guarded)
    {
        char *u = (char *)kv.first;
        hipLaunchKernelGGL(guard_count_kernel, dim3(1), dim3(256), 0, nullptr, (const unsigned *)(u - kGuard * sizeof(float)), kGuard, bad);
        hipLaunchKernelGGL(guard_count_kernel, dim3(1), dim3(256), 0, nullptr, (const unsigned *)(u + kv.second), kGuard, bad);
    }
    hipError_t e = hipMemcpy(&host, bad, sizeof host, hipMemcpyDeviceToHost);
    (void)hipFree(bad);
    if (e != hipSuccess) return fail(TINY_BATCH_EHIP, "tiny_batch_debug_check: %s", hipGetErrorString(e));
    return (long long)host;
}

int tiny_batch_debug_poke(TinyBatch *tb, int which)
{
    CHECK_TB(tb);
    TRY(set_device(tb));
    // test hook of the guard zones: deliberately writes ONE word just outside a work array of this handle (which = 0: in front, 1: behind)
    float *arr = tb->layout == LAYOUT_ROW ? tb->pair[0] : tb->arr[0];
    size_t bytes = 0;
    {
        std::lock_guard<std::mutex> lk(g_guard_mu);
        auto it = g_guarded.find(arr);
        if (it == g_guarded.end()) return fail(TINY_BATCH_ENOTREADY, "tiny_batch_debug_poke: the handle was not created under tiny_batch_debug_guards(1)");
        bytes = it->second;
    }
    const float v = 1.0f;
    HIP_TRY(hipMemcpy(which ? (char *)arr + bytes : (char *)arr - sizeof(float), &v, sizeof v, hipMemcpyHostToDevice));
    return 0;
}

int tiny_batch_arithmetic(TinyBatch *tb)
{
    CHECK_TB(tb);
    int v = 0;
    TRY(resolve_variant(tb, &v));
    return (v == VAR_ROW_EXACT || v == VAR_GENERIC) ? TINY_BATCH_ARITH_EXACT : TINY_BATCH_ARITH_FMA;
}

// the kernel a closed-loop run of several steps (tiny_batch_mpc_run_async) launches: it can differ from the kernel of a lone solve (the automatic
// choice keeps the 16-lane kernel's on-chip MPC loop where a lone solve of the same batch goes to the matrix-core kernel)
const char *tiny_batch_closed_loop_kernel_name(TinyBatch *tb)
{
    if (!tb) return "";
    static thread_local std::string name;
    const bool keep = tb->closed_loop_run;
    tb->closed_loop_run = true;
    update_kname(tb);
    name = tb->kname;
    tb->closed_loop_run = keep;
    update_kname(tb);
    return name.c_str();
}

int tiny_batch_set_row_kernel(TinyBatch *tb, int family)
{
    CHECK_TB(tb);
    if (family < 0 || family > 8)
        return fail(TINY_BATCH_EINVAL, "row kernel must be 0 (auto), 1 (rowlane), 2 (rowloop), 3 (rowstream), 4 (quadlane), 5 (tile16), 6 (wavestream), 7 (waveres) or 8 (tile48)");
    const bool ok = family == 0 || (family == 1 && tb->row_dims_ok) || (family == 2 && tb->rowloop_ok) || (family == 3 && tb->rowmath_ok) ||
                    (family == 4 && tb->quad_ok) || (family == 5 && tb->tile16_ok) || (family == 6 && tb->wave_ok) || (family == 7 && tb->waveres_ok) ||
                    (family == 8 && tb->tile48_ok);
    if (!ok)
        return fail(TINY_BATCH_EUNSUPPORTED, "row kernel %d has no instantiation for nx=%d nu=%d N=%d", family, tb->nx, tb->nu, tb->N);
    static const int kFam[9] = {-1, 0, 1, 2, 4, 5, 3, 6, 7};
    tb->row_family_forced = kFam[family];
    invalidate_graph(tb);
    return 0;
}

// bits = 16: the duals stay fp32 wherever the kernels implement that (round 3: with 16-bit duals a quarter of a cartpole batch and
// 8 % of a quadrotor batch stall above the tolerances; 16-bit duals remain available through tiny_batch_set_storage_ex(tb, 16, 16)
// and are what the kernels that stream their state implement)
int tiny_batch_set_storage(TinyBatch *tb, int bits)
{
    CHECK_TB(tb);
    const bool d32 = bits == 16 && (tb->row_dims_ok || tb->quad_ok);
    TRY(tiny_batch_set_storage_ex(tb, bits, d32 ? 32 : bits));
    tb->dual32_pref = d32;     // a preference: every solve / step call settles the width on the kernel it resolves to (settle_dual_width)
    tb->dual32_forced = false;
    return 0;
}

int tiny_batch_set_storage_ex(TinyBatch *tb, int bits, int dual_bits)
{
    CHECK_TB(tb);
    if (bits != 16 && bits != 32) return fail(TINY_BATCH_EINVAL, "storage must be 32 (fp32, default) or 16 (IEEE binary16)");
    if (dual_bits != bits && !(bits == 16 && dual_bits == 32)) return fail(TINY_BATCH_EINVAL, "dual storage must equal the storage, or be 32 with 16-bit storage");
    const bool want = bits == 16, want_d32 = want && dual_bits == 32;
    if (want_d32 && !(tb->row_dims_ok || tb->quad_ok))
        return fail(TINY_BATCH_EUNSUPPORTED, "fp16 storage with fp32 duals needs a register-resident kernel instantiation (nx=%d nu=%d N=%d has none)", tb->nx, tb->nu, tb->N);
    if (want && !(tb->row_dims_ok || tb->rowmath_ok) )
        return fail(TINY_BATCH_EUNSUPPORTED, "fp16 storage is implemented by the row kernels only (nx=%d nu=%d has none)", tb->nx, tb->nu);
    if (want && tb->variant == VAR_STREAM) return fail(TINY_BATCH_EUNSUPPORTED, "fp16 storage cannot be combined with the streaming kernel");
    tb->dual32_pref = false;      // (a refused call leaves the handle as it was)
    tb->dual32_forced = want_d32;
    if (want == tb->h16 && want_d32 == tb->dual32) return 0;
    TRY(set_device(tb));
    HIP_TRY(hipStreamSynchronize(tb->stream));
    invalidate_graph(tb);
    // the workspace restarts from zero in the new precision (like tiny_batch_create), inputs are re-derived
    free_layout(tb, LAYOUT_TILE);
    free_layout(tb, LAYOUT_ROW);
    tb->h16 = want;
    tb->dual32 = want_d32;
    tb->layout = (want || tb->row_dims_ok || tb->wave_ok || !tb->tile_dims_ok) ? LAYOUT_ROW : LAYOUT_TILE;
    TRY(alloc_layout(tb, tb->layout));
    HIP_TRY(hipMemsetAsync(tb->res, 0, (size_t)tb->batch * 4 * sizeof(float), tb->stream));
    HIP_TRY(hipMemsetAsync(tb->status, 0, (size_t)tb->batch * sizeof(int), tb->stream));
    HIP_TRY(hipMemsetAsync(tb->iter, 0, (size_t)tb->batch * sizeof(int), tb->stream));
    tb->iter_history = false;
    HIP_TRY(hipMemsetAsync(tb->x0buf, 0, (size_t)tb->batch * tb->nx * sizeof(float), tb->stream));
    tb->cold_pending = tb->duals_zero_pending = tb->x0_zero_pending = false;
    tb->derived_dirty[0] = tb->derived_dirty[1] = true;
    return 0;
}

int tiny_batch_select_kernel(TinyBatch *tb, int variant)
{
    CHECK_TB(tb);
    if (variant < VAR_AUTO || variant > VAR_GENERIC)
        return fail(TINY_BATCH_EINVAL, "variant must be 0 (auto), 1 (streaming, fma), 2 (row / wave kernels, exact), 3 (row / wave kernels, fma) or 4 (run-time dimensions, exact)");
    const int old = tb->variant;
    tb->variant = variant;
    int v = 0;
    if (int rc = resolve_variant(tb, &v)) { tb->variant = old; return rc; }
    invalidate_graph(tb);
    TRY(set_device(tb));
    TRY(ensure_layout(tb, tile_variant(v) ? LAYOUT_TILE : LAYOUT_ROW));
    return 0;
}

} // extern "C"
