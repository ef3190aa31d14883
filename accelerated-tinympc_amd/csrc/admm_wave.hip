// admm_wave.hip — exact-arithmetic batched TinyMPC ADMM kernel for problem classes with 16 < nx + nu <= 64
// (BASELINE.json configs[3]: nx = 32, nu = 16, N = 50): ONE WAVEFRONT = ONE INSTANCE.
//
// Restates tiny_solve() (src/tinympc/admm.cpp:111-152) like admm_rowstream_kernel, with the row mapping widened from a
// DPP row to the whole wave: lane r owns row r of the stacked vector [x ; u] (r < NX: x(r), NX <= r < NX+NU: u(r-NX)).
// A gain x state product broadcasts the state vector through LDS (one ds_write_b32, then 16-byte broadcast reads) and is
// one plain v_mul_f32 per column with the lane's own gain entry; the sums follow the orders the reference's SSE2 Eigen build uses for these sizes, including the row-major GEMV
// kernel Eigen switches to when nx >= 8 and nu >= 8 (WavePlans below; Eigen/src/Core/GeneralProduct.h: gemv_dense_selector).
// Results are BITWISE identical to the compiled reference.  The loop-carried state lives in the row-layout arrays
// (row width 64) and is streamed every iteration; the whole wave leaves the iteration loop when its instance converges,
// so there is no lock-step waste.  fma arithmetic for these sizes is the MFMA streaming kernel (admm_stream.hip).
#include "wave_math.h"

namespace tinympc
{

// PF = look-ahead of the streamed state in horizon steps, WPS = waves per SIMD the register budget is cut for.  Two
// instantiations per class: (1, 3) for launches that queue more waves than the chip holds (occupancy hides the latency),
// (4, 2) for launches of at most 2 048 instances, where every wave is resident anyway and a deeper look-ahead is worth
// more than the third wave: 7.1 -> 5.7 ms at 2 048 instances of the nx = 32 class, 40.6 vs 43.9 ms at 16 384.
template <int NX, int NU, int PF, int WPS>
__global__ __launch_bounds__(WAVE, WPS) void admm_wavestream_kernel(const RowParams P)
{
    using PL = WavePlans<NX, NU>;
    __shared__ __attribute__((aligned(16))) float vec[WAVE]; // broadcast buffer of lane_products
    const int lane = threadIdx.x;
    const int inst = blockIdx.x;
    const bool is_x = lane < NX, is_u = (lane >= NX) && (lane < NX + NU);
    const int N = P.N;
    const int rowbase = (inst * N) * WAVE + lane;
    const float rho = P.rho;
    const float2 *bnd = reinterpret_cast<const float2 *>(P.bounds) + (size_t)inst * P.bounds_inst_stride; // shared or per instance
    WaveGains<NX, NU> G;
    G.load(P.mats, lane);
    const float qrow = P.mats[(2 * NX + 2 * NU) * WAVE + lane];
    int wstart = 0;
    if (P.xref_mode == 1) wstart = P.xref_start[inst];
    const int xref_off = inst * (int)P.xref_inst_stride + lane;
    auto xref_at = [&](int i) {
        if (P.xref_mode == 1)
        {
            int row = wstart + i;
            row = row < P.table_rows ? row : P.table_rows - 1;
            return P.xref_table[row * WAVE + lane];
        }
        return P.xref[xref_off + i * WAVE];
    };
    const float x0 = P.xu[rowbase];
    float pterm;
    {
        // -(Xref_{N-1}^T Pinf) (admm.cpp:83), x rows; PT[k] = Pinf(k, r)
        float PT[NX], t[NX];
#pragma unroll
        for (int k = 0; k < NX; k++) PT[k] = P.mats[(2 * NX + 2 * NU + 1 + k) * WAVE + lane];
        lane_products<0, NX>(t, xref_at(N - 1), PT, vec, lane);
        pterm = -wreduce<PL::TERM>(t);
    }
    int st = TINY_STATUS_UNSOLVED_, itn = 1;
    float r_ps = 0.f, r_pi = 0.f, r_ds = 0.f, r_di = 0.f;
    if (!P.cold_start)
    {
        r_ps = P.res[4 * inst + 0]; r_pi = P.res[4 * inst + 1];
        r_ds = P.res[4 * inst + 2]; r_di = P.res[4 * inst + 3];
    }
    // max over the lanes of the wave, every lane gets the result
    auto wave_max = [](float v) {
        v = fmaxf(v, dpp_mov<0x128>(v)); v = fmaxf(v, dpp_mov<0x124>(v)); v = fmaxf(v, dpp_mov<0x122>(v)); v = fmaxf(v, dpp_mov<0x121>(v));
        float m = v; // row maxima -> wave maximum through SGPRs
        m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16)));
        m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32)));
        m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48)));
        return fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0)));
    };
    for (int it = 0; it < P.max_iter; ++it)
    {
        const bool last_iter = (it == P.max_iter - 1);
        const bool zero_state = (it == 0) && (P.cold_start != 0);
        const bool zero_duals = (it == 0) && ((P.cold_start | P.duals_zero) != 0);
        float s = x0, pri = 0.f, dua = 0.f, t1 = 0.f;
        // The sweep is a dependent chain and the compiler cannot hoist loads over the stores of a step (it cannot know the
        // arrays are distinct), so the streamed state is requested by hand, PF steps ahead: one step of arithmetic (~600
        // cycles) is shorter than a trip to HBM, a look-ahead of one step left every step waiting for its operands.
        struct Fwd { float d, a, b; float2 lh; };
        auto load_fwd = [&](int i) {
            const int o = rowbase + i * WAVE;
            Fwd f;
            f.d = zero_state ? 0.f : P.pd[o]; f.a = zero_duals ? 0.f : P.gy[o]; f.b = zero_state ? 0.f : P.vz[o];
            f.lh = bnd[i * WAVE + lane];
            return f;
        };
        Fwd fq[PF];
#pragma unroll
        for (int j = 0; j < PF; j++)
            if (j < N) fq[j] = load_fwd(j);
        for (int i0 = 0; i0 < N; i0 += PF)
        {
#pragma unroll
            for (int j = 0; j < PF; j++)
            {
                const int i = i0 + j;
                if (i >= N) break;
                const int o = rowbase + i * WAVE;
                const Fwd f = fq[j];
                if (i + PF < N) fq[j] = load_fwd(i + PF);
                const float di = f.d, a = f.a, bprev = f.b;
                const float2 lh = f.lh;
                float sv, xn = 0.f;
                if (i < N - 1) wave_lqr_step<NX, NU>(G, vec, lane, is_x, is_u, s, di, sv, xn);
                else sv = is_x ? s : 0.f;
                const float t = __builtin_amdgcn_fmed3f(sv + a, lh.x, lh.y); // admm.cpp:47-60 (lo := min(lo, hi) on the host)
                const float an = (a + sv) - t;                               // admm.cpp:69-70
                pri = fmaxf(pri, fabsf(sv - t));
                dua = fmaxf(dua, fabsf(bprev - t));
                P.vzn[o] = t;
                P.gy[o] = an;
                if (last_iter) P.xu[o] = sv;
                t1 = t - an;
                s = xn;
            }
        }
        const float pN = pterm - rho * t1; // admm.cpp:83-84
        P.pd[rowbase + (N - 1) * WAVE] = is_x ? pN : 0.f;
        const float pri_x = wave_max(is_x ? pri : 0.f), dua_x = wave_max(is_x ? dua : 0.f);
        const float pri_u = wave_max(is_u ? pri : 0.f), dua_u = wave_max(is_u ? dua : 0.f);
        itn = it + 1;
        bool conv = false;
        if ((it + 1) % P.check_termination == 0) // admm.cpp:91-109
        {
            r_ps = pri_x; r_ds = dua_x * rho; r_pi = pri_u; r_di = dua_u * rho;
            conv = (r_ps < P.abs_pri_tol) && (r_pi < P.abs_pri_tol) && (r_ds < P.abs_dua_tol) && (r_di < P.abs_dua_tol);
        }
        if (conv) // wave-uniform: the instance is the wave
        {
            st = TINY_STATUS_SOLVED_;
            break;
        }
        float p = pN;
        P.vz[rowbase + (N - 1) * WAVE] = P.vzn[rowbase + (N - 1) * WAVE]; // admm.cpp:141-142
        struct Bwd { float sn, g, xr; };
        auto load_bwd = [&](int i) {
            const int o = rowbase + i * WAVE;
            Bwd f;
            f.sn = P.vzn[o]; f.g = P.gy[o]; f.xr = xref_at(i);
            return f;
        };
        Bwd bq[PF];
#pragma unroll
        for (int j = 0; j < PF; j++)
            if (N - 2 - j >= 0) bq[j] = load_bwd(N - 2 - j);
        for (int i0 = N - 2; i0 >= 0; i0 -= PF)
        {
#pragma unroll
            for (int j = 0; j < PF; j++)
            {
                const int i = i0 - j;
                if (i < 0) break;
                const int o = rowbase + i * WAVE;
                const Bwd f = bq[j];
                if (i - PF >= 0) bq[j] = load_bwd(i - PF);
                const float sni = f.sn, gi = f.g, xri = f.xr;
                const float cq = is_x ? -(xri * qrow) : -0.f; // -0: r = -rho*(znew - y) keeps the sign of a zero difference
                float pn, dd;
                wave_riccati_step<NX, NU>(G, vec, lane, is_x, p, cq - rho * (sni - gi), pn, dd); // admm.cpp:19-20,80-82
                P.pd[o] = is_u ? dd : pn;
                P.vz[o] = sni;
                p = pn;
            }
        }
    }
    if (P.max_iter <= 0) // tiny_solve only sets status and iter (admm.cpp:114-117,151)
    {
        if (lane == 0)
        {
            P.status[inst] = TINY_STATUS_UNSOLVED_;
            P.iter[inst] = 1;
            atomicAdd(P.n_unsolved, 1);
        }
        return;
    }
    {
        // live-out: q, r (admm.cpp:80-82) and, for a converged instance, x,u regenerated from the d it converged with
        const bool solved = (st == TINY_STATUS_SOLVED_);
        // reset_workspace() folded into this launch (cold start): an instance that converged in its FIRST iteration ran no
        // backward sweep, which is what writes [p;d] and [v;z] — they are the zeros of the reset, materialised here
        const bool fresh = (P.cold_start != 0) && solved && itn == 1;
        float s = x0;
        for (int i = 0; i < N; i++)
        {
            const int o = rowbase + i * WAVE;
            float sv, xn = 0.f;
            if (i < N - 1) wave_lqr_step<NX, NU>(G, vec, lane, is_x, is_u, s, fresh ? 0.f : P.pd[o], sv, xn);
            else sv = is_x ? s : 0.f;
            if (fresh)
            {
                if (i < N - 1) P.pd[o] = 0.f;
                P.vz[o] = 0.f;
            }
            if (solved) P.xu[o] = sv;
            s = xn;
            const float cq = is_x ? -(xref_at(i) * qrow) : -0.f; // -0: r = -rho*(znew - y) keeps the sign of a zero difference
            const float lin = cq - rho * (P.vzn[o] - P.gy[o]);
            P.qr[o] = (i < N - 1 || is_x) ? lin : 0.f;
        }
        if (lane == 0)
        {
            P.res[4 * inst + 0] = r_ps; P.res[4 * inst + 1] = r_pi;
            P.res[4 * inst + 2] = r_ds; P.res[4 * inst + 3] = r_di;
            P.status[inst] = st;
            P.iter[inst] = itn;
            if (!solved) atomicAdd(P.n_unsolved, 1);
        }
    }
}

bool wavedims_supported(int nx, int nu)
{
#define TINY_WAVEDIMS_CHECK(NX, NU) \
    if (nx == NX && nu == NU) return true;
    TINY_FOR_EACH_WAVEDIMS(TINY_WAVEDIMS_CHECK)
    return false;
}

hipError_t launch_admm_wavestream(int nx, int nu, const RowParams &P, hipStream_t stream)
{
#define TINY_WAVE_DISPATCH(NX, NU)                                                                         \
    if (nx == NX && nu == NU)                                                                              \
    {                                                                                                      \
        if (P.batch <= 2048) hipLaunchKernelGGL((admm_wavestream_kernel<NX, NU, 4, 2>), dim3(P.batch), dim3(WAVE), 0, stream, P); \
        else hipLaunchKernelGGL((admm_wavestream_kernel<NX, NU, 1, 3>), dim3(P.batch), dim3(WAVE), 0, stream, P);                 \
        return hipGetLastError();                                                                          \
    }
    TINY_FOR_EACH_WAVEDIMS(TINY_WAVE_DISPATCH)
    return hipErrorInvalidValue;
}

} // namespace tinympc
