// Internal definitions shared by the HIP kernels and the C-ABI host code.
// gfx950 (MI355X / CDNA4) only.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace tinympc
{

// One wavefront solves TILE = 16 instances: they are the 16 columns of the
// v_mfma_f32_16x16x4_f32 tile, so every gain x state mat-vec of the horizon sweeps is a
// handful of MFMA issues whose A operand (a slice of the gain matrix) sits in ONE VGPR.
//
// Device-internal ("tile") layout of every per-instance array
//   x-family (x,q,p,v,vnew,g,Xref,x_min,x_max):  float [ntiles][N  ][64 lanes][NXC]
//   u-family (u,r,d,z,znew,y,u_min,u_max):       float [ntiles][N-1][64 lanes][NUC]
// lane = 16*gq + c holds, for instance 16*tile + c, the rows {4*ch + gq : ch < NXC} of the
// vector: "chunk" ch = four consecutive rows, one per lane group gq.  That is exactly the
// B-operand layout of the MFMA (lane (gq,c) supplies B[k=gq][col=c]) for the K-slice ch, and
// (with the gain matrices' rows permuted on the host) also its D layout, so a sweep never
// moves data between lanes.  NXC = ceil(nx/4), NUC = ceil(nu/4); padded rows are zero.
constexpr int TILE = 16;
constexpr int WAVE = 64;
constexpr int TINY_STATUS_SOLVED_ = 1;    // work->status, admm.cpp:136
constexpr int TINY_STATUS_UNSOLVED_ = 11; // admm.cpp:114

// (NXC, NUC) pairs with compiled kernels: quadrotor (12,4), cartpole (4,1), random (32,16), test dims (8,3)
#define TINY_FOR_EACH_DIMS(X) X(3, 1) X(1, 1) X(8, 4) X(2, 1) X(16, 8)

template <int NXC_, int NUC_>
struct Dims
{
    static constexpr int NXC = NXC_;                  // x chunks
    static constexpr int NUC = NUC_;                  // u chunks
    static constexpr int NCH = NXC_ + NUC_;           // stacked [x;u] chunks
    static constexpr int NT = (NCH + 3) / 4;          // 16-row output tiles over the stacked vector
    static constexpr int NTX = (NXC_ + 3) / 4;        // tiles 0..NTX-1 contain x chunks
    static constexpr int TU0 = NXC_ / 4;              // first tile that contains a u chunk
    static constexpr int NTU = NT - TU0;              // tiles TU0..NT-1 contain u chunks
    // MFMA A-operand registers (one VGPR each), in the order they are packed by the host:
    static constexpr int N_A1 = NT * NXC_;            // fwd  [A;-K] * x
    static constexpr int N_A2 = NTX * NUC_;           // fwd  [B]    * u
    static constexpr int N_A3 = NT * NXC_;            // bwd  [AmBKt; B^T] * p
    static constexpr int N_A4 = NTX * NUC_;           // bwd  [-K^T] * r
    static constexpr int N_A5 = NTU * NUC_;           // bwd  [Quu_inv] * (B^T p + r)
    static constexpr int N_AP = NTX * NXC_;           // terminal [-Pinf^T] * Xref_{N-1}
    static constexpr int N_OPND = N_A1 + N_A2 + N_A3 + N_A4 + N_A5 + N_AP;
};

struct SolveParams
{
    int nx, nu, N, batch, ntiles;
    float rho, abs_pri_tol, abs_dua_tol;
    int max_iter, check_termination, en_state_bound, en_input_bound;
    int duals_zero; // y = g = 0 on entry (reset_dual_variables() folded into the solve)
    int cold_start; // d = v = z = y = g = 0 on entry (reset_workspace() folded into the solve)
    int xref_mode;  // 0: tile array (per-instance or shared), 1: window gather from a trajectory table
    float *x, *q, *p, *v, *vnew, *g;      // x-family
    float *u, *r, *d, *z, *znew, *y;      // u-family
    const float *xmin, *xmax, *umin, *umax, *xref;
    long long xb_tile_stride, ub_tile_stride, xref_tile_stride; // floats between tiles; 0 = shared by the batch
    const float *xref_table; // [rows][4 gq][NXC]
    const int *xref_start;   // [batch]
    int table_rows;
    float *res;  // [batch][4]
    int *status; // [batch]
    int *iter;   // [batch]
    int *n_unsolved;
    const float *opnd; // [N_OPND][64]
    const float *qvec; // [64][NXC]  Q(row) per lane
};

// (nx, nu, N) triples with a compiled register-resident kernel (admm_rowlane.hip); needs nx + nu <= 16
#define TINY_FOR_EACH_ROWLANE(X) X(12, 4, 30) X(12, 4, 25) X(12, 4, 20) X(12, 4, 10) X(4, 1, 10) X(8, 3, 7) X(12, 4, 40) X(12, 4, 50)

// "Row" layout used by the rowlane kernel: the twelve work arrays are stored as six stacked pairs
//   xu = [x;u], qr = [q;r], pd = [p;d], vz = [v;z], vzn = [vnew;znew], gy = [g;y]
// each float [batch_pad4][N][16]: row r < nx is the state-type member, nx <= r < nx+nu the input-type
// member (zero at step N-1), the rest zero.  One DPP row (16 lanes) = one instance-step = 64 contiguous bytes.
struct RowParams
{
    int nx, nu, N, batch;
    float rho, abs_pri_tol, abs_dua_tol;
    int max_iter, check_termination;
    int duals_zero, cold_start;
    int xref_mode;               // 0: xref array (stride 0 = shared), 1: window gather
    float *xu, *qr, *pd, *vz, *vzn, *gy;
    const float *xref;           // [batch or 1][N][16]
    unsigned xref_inst_stride;   // floats between instances (0 = shared)
    unsigned pi_flags;           // admm_tile16*.hip only (fills a padding hole: no other kernel's argument layout moves).  bits 8-15: the launcher's tail stride of the
                                 // two-ended tile queue.  bit 0: the {lo, hi} rows differ
                                 // from step to step (`bounds` is then their tile image, read through a ring of step slots; clear = one row per instance, fetched
                                 // once per tile from the [B][N][16] table), bit 1: the same for the reference rows and `xref`
    const float *xref_table;     // [rows][16]
    int *xref_start;             // [batch] window start; advanced by the closed-loop kernels
    int table_rows;
    const float *bounds;         // [N][rw][2] = {lo, hi}; +-inf where a bound is disabled or the row is unused
    unsigned bounds_inst_stride; // {lo,hi} entries between instances: 0 = one table for the batch, N*rw = per-instance bounds
                                 // (streaming row kernel, step kernels and wave kernel only)
    const float *mats;           // [3nx + 3nu + 2][16] gain rows per lane (see pack_gains)
    // the two terms the reference ships commented out (admm.cpp:20, :79), off by default; read by the streaming row kernel
    // and the step kernels only (tiny_batch_set_optional_terms routes a handle that enables one of them there)
    const float *uref;           // NULL = off; [batch or 1][N][16], Uref on the u rows (row N-1 zero)
    unsigned uref_inst_stride;   // floats between instances (0 = shared)
    int en_d2p;                  // p_i += coeff_d2p * d_i
    const int *order;            // NULL, or a permutation of the ceil(batch/4) instance groups: workgroup b solves group order[b]
                                 // (admm_tile16.hip: of the ceil(batch/16) tiles, wave w of workgroup b solves tile order[4b + w])
                                 // (register-resident 16-lane kernels only; results do not depend on it)
    float *res;
    int *status, *iter, *n_unsolved;
    // closed loop on chip (admm_rowlane.hip): mpc_steps > 1 runs that many MPC steps (solve; x0 <- Adyn x0 + Bdyn u0;
    // window += window_advance; y = g = 0) inside one launch, the state never leaving registers/LDS in between
    int mpc_steps, window_advance;
    float *u0_traj;              // [mpc_steps][batch][nu] or NULL: u.col(0) of every step
    float *x0buf;                // [batch][nx]: x0 of the LAST solve of the launch (the host's plant step reads it)
    int dual32;                  // with fp16 storage: gy (the duals g, y) stays fp32 in HBM and is not rounded (rowlane and quadlane kernels only)
};

// (nx, nu) pairs with single-function kernels (admm_steps.hip), any N
// (2, 2): the reference's own examples/codegen_random.cpp:19-31 (n = 2, m = 2, N = 3, min > max bounds)
#define TINY_FOR_EACH_ROWDIMS(X) X(12, 4) X(4, 1) X(8, 3) X(8, 4) X(12, 2) X(4, 2) X(4, 4) X(2, 2)
enum { STEP_FORWARD_PASS = 0, STEP_UPDATE_SLACK, STEP_UPDATE_DUAL, STEP_UPDATE_LINEAR_COST, STEP_TERMINATION_CONDITION,
       STEP_BACKWARD_PASS_GRAD };

typedef float f32x4 __attribute__((ext_vector_type(4)));

bool rowlane_supported(int nx, int nu, int N);
// h16: the row-layout arrays, Xref and the bounds table are IEEE binary16 (see rowlane_math.h: rnd/ldw/stw)
hipError_t launch_admm_rowlane(int nx, int nu, int N, bool exact, bool h16, const RowParams &P, hipStream_t stream);
bool rowdims_supported(int nx, int nu);
hipError_t launch_admm_rowstream(int nx, int nu, bool exact, bool h16, const RowParams &P, hipStream_t stream);
hipError_t launch_admm_step(int nx, int nu, bool exact, bool h16, int fn, const RowParams &P, int *conv_out, hipStream_t stream);

// rolled-loop register-resident kernel (admm_rowloop.hip): any N <= 32 for the (nx, nu) pairs of TINY_FOR_EACH_ROWDIMS
bool rowloop_supported(int nx, int nu, int N);
hipError_t launch_admm_rowloop(int nx, int nu, bool exact, bool h16, const RowParams &P, hipStream_t stream);

// four-lanes-per-instance register-resident kernel (admm_quadlane.hip): nx = 4, nu = 1, instantiated horizons only
bool quadlane_supported(int nx, int nu, int N);
hipError_t launch_admm_quadlane(int N, bool exact, bool h16, const RowParams &P, hipStream_t stream);

// sixteen-instances-per-wave register-resident kernel with the products on the matrix cores (admm_tile16.hip): nx = 12, nu = 4,
// instantiated horizons; ROW layout and RowParams of the row kernels; shared bounds, window / shared reference, fp32 storage
bool tile16_supported(int nx, int nu, int N);
// tail: the kernel's two-ended tile queue — -1 automatic (t16_tail_stride), 0 plain counter, k: every k-th wave takes tiles from the short end of the order
hipError_t launch_admm_tile16(int N, bool exact, const RowParams &P, hipStream_t stream, int n_cu, int tail = -1);
// the same kernel with the bounds and / or the reference PER INSTANCE ([B][N][16] tables fetched by LDS-DMA into per-wave rings; a shared table or a
// window goes through the same rings): one solve per launch (admm_tile16_pi.hip)
// bounds_ring / xref_ring: the table goes through the per-wave LDS-DMA slots (per instance, or a window too long for the LDS share); otherwise it is
// the batch-shared table staged in LDS as in launch_admm_tile16.  P.pi_flags says which of the ring tables change along the horizon.
hipError_t launch_admm_tile16_pi(int N, bool exact, bool bounds_ring, bool xref_ring, const RowParams &P, hipStream_t stream, int n_cu, int tail = -1);
size_t tile16_pi_lds_bytes(int N, bool bounds_ring, bool xref_ring, unsigned pi_flags, int table_rows);
int tile16_max_table_rows(); // rows of a trajectory table that fit the kernel's LDS share

// wave-per-instance exact kernel (admm_wave.hip): 16 < nx + nu <= 64, any N, state in HBM, row width 64
#define TINY_FOR_EACH_WAVEDIMS(X) X(32, 16) X(16, 8) X(16, 4) X(20, 8) X(24, 4)
bool wavedims_supported(int nx, int nu);
hipError_t launch_admm_wavestream(int nx, int nu, const RowParams &P, hipStream_t stream);
// the same classes with the loop-carried state on chip (admm_waveres.hip): N <= 50
bool waveres_supported(int nx, int nu, int N);
hipError_t launch_admm_waveres(int nx, int nu, bool exact, const RowParams &P, hipStream_t stream);
// nx = 32, nu = 16, N <= 50 with sixteen instances per workgroup on the matrix cores (admm_tile48.hip): exact and fma arithmetic
bool tile48_supported(int nx, int nu, int N);
hipError_t launch_admm_tile48(int N, bool exact, const RowParams &P, hipStream_t stream);

hipError_t launch_admm_stream(int nxc, int nuc, const SolveParams &P, hipStream_t stream);
// exact arithmetic with RUN-TIME dimensions, one thread per instance, TILE layout (admm_generic.hip): any nx <= 64, nu <= 32 with nx, nu each <= 4 or a
// multiple of 4; gains = Kinf | Pinf | Quu_inv | AmBKt | Adyn | Bdyn | Q (column-major)
bool generic_exact_supported(int nx, int nu);
hipError_t launch_admm_generic(const SolveParams &P, const float *gains, int nxc, int nuc, hipStream_t stream);

// longest-first dispatch order of the instance groups for the register-resident row kernel (dispatch_order.hip);
// P.mats must be the fma gains
hipError_t launch_dispatch_order(int nx, int nu, bool h16, const RowParams &P, float *key, int *order, hipStream_t stream, int tile = 0);
// order[] = the ceil(batch/unit) units (unit = 4: groups, 16: tiles) sorted by the largest iter[] of their instances, largest first; zeroes counters[0..1]
hipError_t launch_dispatch_order_history(const int *iter, int batch, int unit, int *order, int *counters, hipStream_t stream, int use_sum = 0);

} // namespace tinympc
