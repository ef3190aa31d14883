// dispatch_order.hip — longest-first dispatch for the register-resident row kernel (tiny_batch_set_dispatch).
//
// Iteration counts are uneven (a few instances need 3-4x the mean) and a launch of B/4 waves over 2 048 wave slots is only a
// handful of rounds, so waves that start late and run long leave most of the chip idle at the end of the launch.  Workgroups
// start in index order: letting workgroup b solve instance group order[b], with the groups sorted by a PREDICTED iteration
// count (longest first), shortens that tail.  The predictor needs no history: it is the largest primal residual
// |[x;u] - clip([x;u] + [g;y])| of ONE forward sweep from the current workspace (how far the unconstrained LQR rollout is
// from the box), computed here in fma arithmetic over the first steps of the horizon — a hint, the solve itself is
// untouched and its results do not depend on the order.  Cost: a fraction of an ADMM iteration per instance plus a bucket
// sort of B/4 keys.
#include "rowlane_math.h"

namespace tinympc
{

// The sweep stops after the first kPredictorSteps horizon steps: under a stabilising feedback the rollout is farthest from
// the box at the start of the horizon (on the bench workload the first two steps already order the groups as well as all
// thirty do, tools/launch_tail.py), and the sweep is pure overhead.
constexpr int kPredictorSteps = 4; // (round 4: 8 -> 4, the same order on the bench workload, 4 us less in the timed step)

// TILES: one workgroup of four waves per TILE of 16 instances (admm_tile16.hip) and key[] receives the tile's key — the largest of its four
// groups' — directly (round 4: a separate kernel used to reduce the group keys, 5 us and a launch gap of the headline step)
template <int NX, int NU, bool H16, bool TILES = false>
__global__ __launch_bounds__(TILES ? 4 * WAVE : WAVE) void dispatch_key_kernel(const RowParams P, float *__restrict__ key)
{
    const int lane = threadIdx.x & (WAVE - 1), r16 = lane & 15;
    const int grp = TILES ? (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6) : (int)blockIdx.x;
    const int inst = grp * 4 + (lane >> 4);
    const bool valid = inst < P.batch;
    const int inst_a = valid ? inst : P.batch - 1; // padding rows of the last group read a valid instance
    const bool is_x = r16 < NX, is_u = (r16 >= NX) && (r16 < NX + NU);
    const int N = P.N;
    const int rowbase = (inst_a * N) * 16 + r16;
    const bool cold = P.cold_start != 0, zdual = cold || (P.duals_zero != 0);
    RowGains<NX, NU> G;
    G.load(P.mats, r16); // fma gains
    float s = ldw<H16>(P.xu, rowbase), pri = 0.f;
    const int steps = N < kPredictorSteps ? N : kPredictorSteps;
    for (int i = 0; i < steps; i++)
    {
        const int o = rowbase + i * 16;
        float sv, xn = 0.f;
        if (i < N - 1) lqr_step<NX, NU, false, H16>(G, is_x, is_u, s, cold ? 0.f : ldw<H16>(P.pd, o), sv, xn);
        else sv = is_x ? s : 0.f;
        const float2 lh = ld_bounds<H16>(P.bounds, inst_a * (int)P.bounds_inst_stride + i * 16 + r16); // per-instance tables: this instance's own
        const float a = zdual ? 0.f : ldw<H16>(P.gy, o);
        pri = fmaxf(pri, fabsf(sv - __builtin_amdgcn_fmed3f(sv + a, lh.x, lh.y)));
        s = xn;
    }
    pri = (valid && (is_x || is_u)) ? pri : 0.f;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) pri = fmaxf(pri, __shfl_xor(pri, m));
    if constexpr (TILES)
    {
        __shared__ float wk[4];
        if (lane == 0) wk[threadIdx.x >> 6] = pri;
        __syncthreads();
        if (threadIdx.x == 0) key[blockIdx.x] = fmaxf(fmaxf(wk[0], wk[1]), fmaxf(wk[2], wk[3]));
    }
    else if (lane == 0) key[blockIdx.x] = pri;
}

// order[] = the n groups sorted by key, largest first: counting sort over 2 048 buckets = sign-less float bits >> 20 (exponent
// and three mantissa bits; monotonic for the non-negative keys).  One workgroup; the order inside a bucket is arbitrary.
constexpr int NBUCKET = 2048;
// one workgroup: order[] = 0 .. n-1 sorted by bucket, largest bucket first (counting sort; the order inside a bucket is arbitrary)
template <class BucketOf>
__device__ inline void bucket_sort_descending(BucketOf bucket_of, int *__restrict__ order, int n, int *__restrict__ counters)
{
    __shared__ int cnt[NBUCKET], sa[NBUCKET], sb[NBUCKET];
    const int t = threadIdx.x;
    if (t == 0 && counters) counters[0] = counters[1] = 0; // the solve kernel's unsolved count and tile queue (saves the memset node in front of this launch)
    for (int b = t; b < NBUCKET; b += 1024) cnt[b] = 0;
    __syncthreads();
    for (int g = t; g < n; g += 1024) atomicAdd(&cnt[bucket_of(g)], 1);
    __syncthreads();
    // inclusive scan over the buckets in DESCENDING key order (index r = NBUCKET-1-b)
    for (int r = t; r < NBUCKET; r += 1024) sa[r] = cnt[NBUCKET - 1 - r];
    __syncthreads();
    int *src = sa, *dst = sb;
    for (int d = 1; d < NBUCKET; d <<= 1)
    {
        for (int r = t; r < NBUCKET; r += 1024) dst[r] = src[r] + (r >= d ? src[r - d] : 0);
        __syncthreads();
        int *tmp = src; src = dst; dst = tmp;
    }
    // exclusive offset of bucket b, reusing cnt[] as the running cursor
    for (int b = t; b < NBUCKET; b += 1024)
    {
        const int r = NBUCKET - 1 - b;
        dst[b] = src[r] - cnt[b]; // dst is free now
    }
    __syncthreads();
    for (int g = t; g < n; g += 1024) order[atomicAdd(&dst[bucket_of(g)], 1)] = g;
}

__global__ __launch_bounds__(1024) void dispatch_order_kernel(const float *__restrict__ key, int *__restrict__ order, int n, int *__restrict__ counters)
{
    bucket_sort_descending([key](int g) { return (int)((__builtin_bit_cast(unsigned, key[g]) & 0x7fffffffu) >> 20); }, order, n, counters);
}

// Warm-started launches (round 4, second session).  The predictor above is blind there — one sweep from a warm workspace sees residuals of the
// size of the tolerance — but the instance's own PAST is not: iteration counts of consecutive MPC steps are strongly correlated (an instance at
// its bounds stays there for a while).  Key of a unit (a group of 4 instances, a tile of 16) = the largest iteration count the previous solve
// spent on one of its instances, read from the workspace's iter array; the bucket is the count itself.  Replayed on the true counts of the
// tracking loop (tests/fuzz/sim_history_dispatch.py): makespan of a warm-started step of 65 536 instances 108 -> 78 iterations on the
// 16-instances-per-wave kernel (order by the TRUE counts: 77.5), 148 -> 120 on the 16-lane kernel.
// use_sum: the key of a unit is the SUM of its instances' counts instead of the largest — the better predictor of a tile's TOTAL over the steps of an
// on-chip closed-loop run (makespan 1 262 against 1 298 iterations; the previous run's own total, which only the kernel could record, 1 236), the
// worse one of its next single solve (79.7 against 78.2) and for groups of four (2 244 against 2 175)
__global__ __launch_bounds__(1024) void dispatch_order_history_kernel(const int *__restrict__ iter, int unit, int batch, int use_sum, int *__restrict__ order, int n,
                                                                      int *__restrict__ counters)
{
    bucket_sort_descending(
        [=](int g) {
            int m = 0;
            for (int k = 0; k < unit; k++)
            {
                const int i = g * unit + k;
                if (i < batch) m = use_sum ? m + max(iter[i], 0) : max(m, iter[i]);
            }
            return min(max(m, 0), NBUCKET - 1);
        },
        order, n, counters);
}

hipError_t launch_dispatch_order_history(const int *iter, int batch, int unit, int *order, int *counters, hipStream_t stream, int use_sum)
{
    const int n = (batch + unit - 1) / unit;
    hipLaunchKernelGGL(dispatch_order_history_kernel, dim3(1), dim3(1024), 0, stream, iter, unit, batch, use_sum, order, n, counters);
    return hipGetLastError();
}

// tile != 0: order[] is a permutation of the ceil(batch/16) tiles of the 16-instances-per-wave kernel
hipError_t launch_dispatch_order(int nx, int nu, bool h16, const RowParams &P, float *key, int *order, hipStream_t stream, int tile)
{
    const int ngroups = (P.batch + 3) / 4;
    const int ntiles = (P.batch + 15) / 16;
    // (the sort also zeroes the two counters of the solve launch behind it: P.n_unsolved[0..1])
#define TINY_KEY_DISPATCH(NX, NU)                                                                                      \
    if (nx == NX && nu == NU)                                                                                          \
    {                                                                                                                  \
        if (tile)                                                                                                      \
        {                                                                                                              \
            if (h16) return hipErrorInvalidValue;                                                                      \
            hipLaunchKernelGGL((dispatch_key_kernel<NX, NU, false, true>), dim3(ntiles), dim3(4 * WAVE), 0, stream, P, key); \
            hipLaunchKernelGGL(dispatch_order_kernel, dim3(1), dim3(1024), 0, stream, key, order, ntiles, P.n_unsolved); \
            return hipGetLastError();                                                                                  \
        }                                                                                                              \
        if (h16) hipLaunchKernelGGL((dispatch_key_kernel<NX, NU, true>), dim3(ngroups), dim3(WAVE), 0, stream, P, key); \
        else hipLaunchKernelGGL((dispatch_key_kernel<NX, NU, false>), dim3(ngroups), dim3(WAVE), 0, stream, P, key);   \
        hipLaunchKernelGGL(dispatch_order_kernel, dim3(1), dim3(1024), 0, stream, key, order, ngroups, P.n_unsolved);  \
        return hipGetLastError();                                                                                      \
    }
    TINY_FOR_EACH_ROWDIMS(TINY_KEY_DISPATCH)
    return hipErrorInvalidValue;
}

} // namespace tinympc
