/*
 * tinympc_batch.h — C-ABI of the MI355X-native batched TinyMPC ADMM solver.
 *
 * Drop-in boundary for ONE path of ucb-bar/Accelerated-TinyMPC: tiny_solve()
 * (src/tinympc/admm.cpp:111-152) and the flat-float* wrapper the reference generates
 * for foreign-language callers (src/tinympc/tiny_wrapper.hpp:14-23), re-expressed over a
 * batch of B independent problem instances of one problem class (nx, nu, N).
 *
 * Conventions
 *  - plain C, plain pointers and sizes; no C++/torch/Eigen types cross this boundary.
 *  - every function returns TINY_BATCH_OK (0) or a negative TinyBatchError unless noted;
 *    tiny_batch_last_error() returns a human-readable message for the calling thread.
 *  - matrices (Kinf, Adyn, ...) are COLUMN-MAJOR, the storage order of the reference's Eigen
 *    members (src/tinympc/types.hpp:13-21): element (i,j) of an R x C matrix is at j*R+i.
 *  - batched arrays use the array-of-reference-instances layout: state-type arrays are
 *    [B][N][nx], input-type arrays are [B][N-1][nu] (instance-major, then horizon step,
 *    then state index) = the flat order tiny_wrapper.cpp uses (xref[j*NSTATES+i], :27)
 *    repeated per instance.  The device-internal layout is private to the library.
 *  - pointers are HOST pointers unless the function name ends in _device.
 *  - the workspace is device-resident and persists between solves (that IS the warm start:
 *    examples/quadrotor_hovering.cpp:99-101 only resets y and g).
 *  - float only: the reference's wrapper is float-only too (tiny_wrapper.cpp:154,167) and its
 *    code generator always emits `typedef float tinytype` (src/tinympc/codegen.cpp:152).
 *
 * Not thread-safe per handle; distinct handles may be used from distinct threads.
 */
#ifndef TINYMPC_BATCH_H
#define TINYMPC_BATCH_H

#ifdef __cplusplus
extern "C"
{
#endif

    typedef struct TinyBatch TinyBatch;

    typedef enum
    {
        TINY_BATCH_OK = 0,
        TINY_BATCH_EINVAL = -1,      /* bad argument / NULL pointer / size mismatch */
        TINY_BATCH_EHIP = -2,        /* a HIP runtime call failed (message has the hipError string) */
        TINY_BATCH_EUNSUPPORTED = -3,/* (nx, nu) has no compiled kernel instantiation */
        TINY_BATCH_ENOTREADY = -4    /* cache / dynamics / settings not set before solve */
    } TinyBatchError;

    /* status codes stored per instance, as the reference's work->status (admm.cpp:114,136) */
#define TINY_STATUS_SOLVED 1
#define TINY_STATUS_UNSOLVED 11

    /* Workspace array ids, in the member order of TinyWorkspace (types.hpp:52-97). */
    typedef enum
    {
        TINY_ARR_X = 0, TINY_ARR_U = 1, TINY_ARR_Q = 2, TINY_ARR_R = 3, TINY_ARR_P = 4, TINY_ARR_D = 5,
        TINY_ARR_V = 6, TINY_ARR_VNEW = 7, TINY_ARR_Z = 8, TINY_ARR_ZNEW = 9, TINY_ARR_G = 10, TINY_ARR_Y = 11,
        TINY_ARR_COUNT = 12
    } TinyBatchArray;

    const char *tiny_batch_last_error(void);

    /* ---- lifetime --------------------------------------------------------------------------- */
    /* Allocates the device-resident workspaces of `batch` instances on HIP device `device`, all
     * zero (the state the reference examples start from, quadrotor_hovering.cpp:49-71).
     * Replaces: the caller-owned TinyCache/TinyWorkspace/TinySettings/TinySolver globals
     * (types.hpp:26-107, quadrotor_hovering.cpp:25-28).
     * Dimensions (glob_opts.hpp:5-7 fixes them at compile time; here they are arguments): any nx <= 64, nu <= 32, N >= 2.
     * Classes with a compiled exact kernel — (nx, nu) = (12,4), (4,1), (8,3), (8,4), (12,2), (4,2), (4,4), (32,16), (16,8),
     * (16,4), (20,8), (24,4) — have fast exact kernels and compute bitwise what the reference computes.  Any other class with
     * nx <= 36 and nx, nu each <= 4 or a multiple of 4 is served, also bitwise, by the run-time-dimension exact kernel (round 4,
     * "generic<nx,nu,exact>": no rebuild, slow).  The remaining classes (e.g. nu = 7, nx = 40: the reference's own summation order
     * depends on column alignment there, or lies beyond what is pinned) have the MFMA streaming kernel in FMA arithmetic only
     * (results within the reference's own fp64-vs-fp32 spread, not bitwise): the automatic kernel choice never lands there by
     * itself — a solve on such a handle returns TINY_BATCH_EUNSUPPORTED until the caller opts in with
     * tiny_batch_select_kernel(tb, 1); tiny_batch_arithmetic() says what a solve would compute in.  Beyond nx <= 64, nu <= 32
     * create itself returns TINY_BATCH_EUNSUPPORTED. */
    int tiny_batch_create(TinyBatch **out, int nx, int nu, int N, int batch, int device);
    void tiny_batch_destroy(TinyBatch *tb);
    /* Launch on this hipStream_t (passed as void*; NULL = the null stream).  Default: NULL. */
    int tiny_batch_set_stream(TinyBatch *tb, void *hip_stream);
    int tiny_batch_synchronize(TinyBatch *tb);

    /* ---- problem class (shared by all instances) ------------------------------------------- */
    /* TinyCache{rho,Kinf,Pinf,Quu_inv,AmBKt} (types.hpp:26-34); coeff_d2p is never read by tiny_solve. */
    int tiny_batch_set_cache(TinyBatch *tb, float rho, const float *Kinf /*nu x nx*/, const float *Pinf /*nx x nx*/,
                             const float *Quu_inv /*nu x nu*/, const float *AmBKt /*nx x nx*/);
    /* work->Adyn, work->Bdyn, work->Q (types.hpp:82-85); work->R/Qu/Uref are never read (admm.cpp:79). */
    int tiny_batch_set_dynamics(TinyBatch *tb, const float *Adyn /*nx x nx*/, const float *Bdyn /*nx x nu*/,
                                const float *Q /*nx*/);
    /* TinySettings (types.hpp:39-47), same field meaning. */
    int tiny_batch_set_settings(TinyBatch *tb, float abs_pri_tol, float abs_dua_tol, int max_iter,
                                int check_termination, int en_state_bound, int en_input_bound);

    /* ---- batched twins of the wrapper calls (tiny_wrapper.hpp:14-23) ------------------------- */
    int tiny_batch_set_x0(TinyBatch *tb, const float *x0 /*[B][nx]*/);                       /* set_x0   :14 */
    /* shared != 0: one [N][nx] reference for the whole batch */
    int tiny_batch_set_xref(TinyBatch *tb, const float *xref /*[B][N][nx]*/, int shared);    /* set_xref :15 */
    /* Xref_b = table[start[b] .. start[b]+N): the sliding window of quadrotor_tracking.cpp:84-85,101,
     * gathered on the device from one shared trajectory table. */
    int tiny_batch_set_xref_window(TinyBatch *tb, const float *table /*[rows][nx]*/, int rows,
                                   const int *start /*[B]*/);
    int tiny_batch_set_umin(TinyBatch *tb, const float *umin /*[B][N-1][nu]*/, int shared);  /* set_umin :16 */
    int tiny_batch_set_umax(TinyBatch *tb, const float *umax, int shared);                   /* set_umax :17 */
    int tiny_batch_set_xmin(TinyBatch *tb, const float *xmin /*[B][N][nx]*/, int shared);    /* set_xmin :18 */
    int tiny_batch_set_xmax(TinyBatch *tb, const float *xmax, int shared);                   /* set_xmax :19 */
    int tiny_batch_reset_dual_variables(TinyBatch *tb);                                      /* reset_dual_variables :20 */
    /* call_tiny_solve :21 / tiny_solve (admm.hpp:10).  Synchronous.  Returns 0 if every instance
     * converged (tiny_solve returned 0 for all), 1 if at least one hit max_iter (tiny_solve returned 1),
     * negative on error. */
    int tiny_batch_solve(TinyBatch *tb);
    int tiny_batch_get_x(TinyBatch *tb, float *x /*[B][N][nx]*/);                            /* get_x :22 */
    int tiny_batch_get_u(TinyBatch *tb, float *u /*[B][N-1][nu]*/);                          /* get_u :23 */
    /* per-instance work->iter, work->status and the four residual fields in the order
     * {primal_residual_state, primal_residual_input, dual_residual_state, dual_residual_input}.
     * Any pointer may be NULL. */
    int tiny_batch_get_status(TinyBatch *tb, int *iter /*[B]*/, int *status /*[B]*/, float *residuals /*[B][4]*/);

    /* ---- the six step functions the reference exports next to tiny_solve (src/tinympc/admm.hpp:12-18), batched ----
     * Each reads and writes the device-resident workspace exactly as the reference function reads and writes
     * TinyWorkspace (forward_pass admm.cpp:27-37, update_slack :45-61, update_dual :67-71, update_linear_cost :77-85,
     * backward_pass_grad :15-22).  tiny_solve's own bookkeeping (status, iter, v=vnew, z=znew) is NOT part of them,
     * as in the reference.  Available for problem classes with nx + nu <= 16 (any N). */
    int tiny_batch_forward_pass(TinyBatch *tb);
    int tiny_batch_update_slack(TinyBatch *tb);
    int tiny_batch_update_dual(TinyBatch *tb);
    int tiny_batch_update_linear_cost(TinyBatch *tb);
    int tiny_batch_backward_pass_grad(TinyBatch *tb);
    /* termination_condition (admm.cpp:91-109): updates the residual fields when iter % check_termination == 0 and
     * writes, per instance, 1 where the reference function would return true (converged may be NULL).
     * Returns the number of such instances (>= 0) or a negative error. */
    int tiny_batch_termination_condition(TinyBatch *tb, int *converged /*[B]*/);

    /* ---- asynchronous form ----------------------------------------------------------------- */
    int tiny_batch_solve_async(TinyBatch *tb);                 /* enqueue on the stream, do not wait */
    /* wait for the stream; *n_unsolved = number of instances whose tiny_solve returned 1 */
    int tiny_batch_wait(TinyBatch *tb, int *n_unsolved);

    /* ---- mixed problem classes in one call ---------------------------------------------------- */
    /* Solves n handles (each one problem class: its own nx, nu, N, batch, storage precision) as one group: all
     * launches are enqueued before any is waited for, each on its handle's stream (a handle still on the null stream
     * is moved to a stream of its own), so the classes overlap on the device.  Returns 0 if every instance of every
     * handle converged, 1 if some hit max_iter (*n_unsolved = how many, may be NULL), negative on error. */
    int tiny_batch_group_solve(TinyBatch **tbs, int n, int *n_unsolved);
    /* ---- one node, several GPUs: one handle per device, each owning a contiguous block of the instance index ----
     * (the batch shards with no data-path collective; one host thread drives all devices through the
     * handles' streams.)  tiny_batch_group_solve above launches every handle's solve on its own device before waiting for any.
     * The optional epilogue: u.col(0) (the control actually applied, nu floats per instance) of every handle, in handle order,
     * into ONE device buffer on dst_device ([sum of batches][nu]); blocks of other devices travel device to device
     * (hipMemcpyPeerAsync, xGMI inside a node), all in flight together.  Returns when the buffer is complete. */
    int tiny_batch_group_gather_u0(TinyBatch **tbs, int n, int dst_device, float *d_dst);
    /* the same into host memory (gathered on the device of handle 0, then one device-to-host copy) */
    int tiny_batch_group_get_u0(TinyBatch **tbs, int n, float *u0_host);

    /* ---- whole-workspace access (warm-start upload, parity tests) ---------------------------- */
    int tiny_batch_set_array(TinyBatch *tb, int array_id, const float *src);
    int tiny_batch_get_array(TinyBatch *tb, int array_id, float *dst);
    int tiny_batch_set_status(TinyBatch *tb, const int *iter, const int *status, const float *residuals);
    /* zero every work array, residuals, status and iter (cold start) */
    int tiny_batch_reset_workspace(TinyBatch *tb);

    /* ---- device-pointer forms (no host round trip) ------------------------------------------ */
    int tiny_batch_set_x0_device(TinyBatch *tb, const float *d_x0 /*[B][nx]*/);
    int tiny_batch_get_u0_device(TinyBatch *tb, float *d_u0 /*[B][nu] = u.col(0) of every instance*/);
    /* set_array / get_array / set_xref with DEVICE pointers (same array-of-instances fp32 layout, memory on the handle's
     * device): one conversion kernel on the handle's stream, asynchronous. */
    int tiny_batch_set_array_device(TinyBatch *tb, int array_id, const float *d_src);
    int tiny_batch_get_array_device(TinyBatch *tb, int array_id, float *d_dst);
    int tiny_batch_set_xref_device(TinyBatch *tb, const float *d_xref /*[B][N][nx] or [N][nx]*/, int shared);

    /* ---- closed loop on the device (quadrotor_hovering.cpp:90-114 / quadrotor_tracking.cpp:93-118) ------
     * One MPC step for every instance without touching the host:
     *   x.col(0) = x0;  [window start += window_advance];  y = 0, g = 0;  tiny_solve;  x0 = Adyn*x0 + Bdyn*u.col(0)
     * x0 lives in an internal device buffer seeded by tiny_batch_set_x0(). */
    int tiny_batch_mpc_step_async(TinyBatch *tb, int window_advance);
    /* `steps` such MPC steps back to back (same window_advance each), results identical to `steps` calls of
     * tiny_batch_mpc_step_async.  Where the unrolled row kernel applies (fp32 storage) ONE launch runs all the steps with
     * the state staying on chip between solves; otherwise the launch sequence is captured once into a hipGraph and
     * replayed (a handle still on the null stream is then moved to a stream of its own: capture needs one).
     * The _traj form also records u.col(0) of every step into a DEVICE buffer [steps][B][nu]. */
    int tiny_batch_mpc_run_async(TinyBatch *tb, int steps, int window_advance);
    int tiny_batch_mpc_run_traj_async(TinyBatch *tb, int steps, int window_advance, float *d_u0_traj);
    /* The same, blocking, with the trajectory in HOST memory ([steps][batch][nu]): the device buffer is allocated on the
     * handle's own device, whatever device is current for the calling thread. */
    int tiny_batch_mpc_run_traj(TinyBatch *tb, int steps, int window_advance, float *u0_traj_host);
    int tiny_batch_get_x0(TinyBatch *tb, float *x0 /*[B][nx]*/);

    /* ---- measurement ------------------------------------------------------------------------- */
    /* When enabled, every solve records hipEvents around its kernel launches on the stream. */
    int tiny_batch_enable_timing(TinyBatch *tb, int on);
    /* Synchronises and returns the device time in ms of the most recent solve's kernel (the predictor sweep and sort of
     * tiny_batch_set_dispatch(tb, 1) run before the first event and are not included). */
    int tiny_batch_last_solve_ms(TinyBatch *tb, float *ms);
    /* Name of the kernel variant the next solve will launch ("rowlane<12,4,30,exact>", "rowstream<12,4,fast>",
     * "stream<3,1>", with ",h16" appended under fp16 storage). */
    const char *tiny_batch_kernel_name(TinyBatch *tb);
    /* ... and of the kernel a closed-loop run of several steps (tiny_batch_mpc_run_async) launches, which can differ: both the 16-lane kernel and the
     * matrix-core kernel keep their MPC loop on chip; the automatic choice takes the latter from 160 instances per compute unit on (measured cross-over)
     * with batch-shared bounds, the former below that and with per-instance tables. */
    const char *tiny_batch_closed_loop_kernel_name(TinyBatch *tb);
    /* Debug guard zones (the stand-in for a GPU address sanitizer, which this platform does not offer).
     * tiny_batch_debug_guards(1): every device allocation this library makes FROM NOW ON carries 1 KB of quiet-NaN guard words at
     * both ends (handles created before keep what they have).  tiny_batch_debug_check() waits for the device and returns the number of
     * guard words any kernel has overwritten (0 = no out-of-bounds write anywhere; < 0: a HIP error, e.g. a fault); an out-of-bounds
     * READ returns NaN and surfaces in the results.  The kernels are the shipped ones.  tiny_batch_debug_poke writes one word just
     * outside a work array of a guarded handle (which = 0 in front, 1 behind): the self-test of the checker. */
    int tiny_batch_debug_guards(int on);
    /* How often tiny_batch_mpc_run_* captured a hipGraph for this handle so far (replays of a captured graph do not count): a test hook. */
    int tiny_batch_debug_graph_captures(TinyBatch *tb);
    long long tiny_batch_debug_check(void);
    int tiny_batch_debug_poke(TinyBatch *tb, int which);
    /* The ARITHMETIC the next solve computes in — the contract behind the kernel name:
     *   TINY_BATCH_ARITH_EXACT (0): every product and sum a separately rounded operation in the reference's order: results bitwise
     *                               equal to the compiled reference (tinytype = float, SSE2 build);
     *   TINY_BATCH_ARITH_FMA   (1): fused multiply-add chains: within the reference's own fp64-vs-fp32 spread,
     *                               chosen explicitly (tiny_batch_select_kernel 1 or 3), never by the automatic choice;
     *   < 0: no kernel would run (TINY_BATCH_EUNSUPPORTED; tiny_batch_last_error() names the opt-in). */
    enum { TINY_BATCH_ARITH_EXACT = 0, TINY_BATCH_ARITH_FMA = 1 };
    int tiny_batch_arithmetic(TinyBatch *tb);
    /* Force a kernel variant: 0 = auto = EXACT arithmetic: the compiled kernels of the class (nx + nu <= 16: row kernels,
     * 16 < nx + nu <= 64: wave / tile kernels) or, for a class outside the compiled lists, the run-time-dimension kernel 4;
     * 1 = streaming MFMA kernel (state in HBM, fma arithmetic; the only kernel of a class whose dimensions admit no exact order),
     * 2 = exact arithmetic on the compiled kernels (bitwise equal to the reference's SSE2 build), 3 = fma arithmetic on the row
     * kernels (nx + nu <= 16) or on the state-on-chip wave / tile kernels (16 < nx + nu <= 64, N <= 50),
     * 4 = exact arithmetic with run-time dimensions (admm_generic.hip: one thread per instance, state in HBM; any nx <= 36, nu <= 32
     * with nx, nu each <= 4 or a multiple of 4 — the classes the reference's summation orders are defined and pinned for; bitwise
     * equal to the reference compiled for that class; the any-class fallback, slow).  Bounds may be batch-shared or per instance in
     * every variant (per-instance bounds stay on the register-resident 16-lane kernels for N <= 64; the quad kernel and longer
     * horizons hand over to the kernels that stream their state). */
    int tiny_batch_select_kernel(TinyBatch *tb, int variant);
    /* Which row kernel variants 2/3 (and auto) launch: 0 = auto (4 where it exists, else 1 where (nx,nu,N) has an unrolled
     * instantiation, else 2 for N <= 64, else 3), 1 = rowlane (16 lanes per instance, unrolled, state in registers/LDS),
     * 2 = rowloop (rolled loops, state in registers/LDS, any N <= 64), 3 = rowstream (any N, state in HBM),
     * 4 = quadlane (4 lanes per instance, nx = 4 and nu = 1 only), 5 = tile16 (16 instances per wavefront as the columns of
     * a 16x16 MFMA tile, gain x state products on the matrix cores in both arithmetic modes, state in registers/LDS; nx = 12,
     * nu = 4 and an instantiated horizon; the auto choice for launches dispatched longest first (tiny_batch_set_dispatch) from 160 instances per compute unit on.  Per-instance bounds and a per-instance reference
     * array are served by its "pi" instantiations (kernel name `tile16<...,pi>`): the rows reach LDS by LDS-DMA, one resident row per
     * instance when the table does not change along the horizon, a ring of step slots otherwise; with fp16 storage, inside a closed-loop
     * run with per-instance tables, or with per-step per-instance bounds beside a trajectory table too long for the LDS left, the handle
     * falls back to the auto choice among the 16-lane kernels).  For 16 < nx + nu <= 64 (one wavefront per instance): 6 = wavestream (state in HBM, any
     * N), 7 = waveres (state in registers/LDS, N <= 50; the auto choice up to 2 048 and for 4 097 ... 6 144 instances), 8 = tile48 (nx = 32, nu = 16,
     * N <= 50, fp32 storage: sixteen instances per workgroup as the columns of 16x16 MFMA tiles, duals in LDS;
     * the auto choice for 2 049 ... 4 096 and from 6 145 instances on (rounds of the launch)).  All of them compute identical results. */
    int tiny_batch_set_row_kernel(TinyBatch *tb, int family);
    /* Storage precision of the per-instance horizon arrays (the twelve work arrays, Xref, bounds) in HBM:
     * 32 = fp32 like the reference (default); 16 = IEEE binary16 storage with fp32 arithmetic (BASELINE.json
     * configs[4]): every assignment to a work array rounds to nearest even, products, sums and the four residual
     * reductions stay fp32.  Row kernels only (nx + nu <= 16, batch-shared bounds).  Host-side arrays stay float; values
     * are rounded when they are stored.  Changing the precision restarts the workspace from zero, like create.
     * With bits = 16 the DUALS y, g stay fp32 wherever the kernel a call resolves to keeps them ("fp16 states with fp32 residual
     * accumulation": the register-resident 16-lane and quad kernels — with 16-bit duals part of a batch stalls short of the tolerances,
     * see below).  That is a PREFERENCE (round 4): a solve or step call that resolves to another kernel — per-instance bounds under
     * fp16, the optional terms, a forced rolled / streaming row kernel, the six single-function calls — converts the duals pair to
     * binary16 (rounding to nearest even) and runs, and converts back when a later call resolves to a kernel with fp32 duals;
     * tiny_batch_kernel_name() says which (",h16d" / ",h16").  tiny_batch_set_storage_ex(tb, 16, 16) forces binary16 everywhere,
     * tiny_batch_set_storage_ex(tb, 16, 32) makes fp32 duals a requirement. */
    int tiny_batch_set_storage(TinyBatch *tb, int bits);
    /* The same with the precision of the DUALS y, g chosen separately: (32, 32) and (16, 16) are tiny_batch_set_storage;
     * (16, 32) keeps the duals — the running sums of the primal residuals, admm.cpp:69-70 — in fp32 while the other ten
     * work arrays, Xref and the bounds are binary16 ("fp16 states, fp32 residual accumulation" read literally).  With fp16
     * duals, increments below half an fp16 ulp of y, g are lost and part of a batch stalls short of the tolerances
     * (DESIGN.md, configs[4]); fp32 duals remove that at 1/12 more state traffic.  Register-resident kernels only (rowlane
     * and quadlane: an instantiated (nx, nu, N), batch-shared bounds, no optional terms); anything else returns
     * TINY_BATCH_EUNSUPPORTED from solve. */
    int tiny_batch_set_storage_ex(TinyBatch *tb, int bits, int dual_bits);

    /* Dispatch order of the register-resident 16-lane row kernels, unrolled and rolled (a launch of batch/4 workgroups is a
     * few rounds deep and iteration counts are uneven, so what starts last decides when the launch ends; results never depend on the order).
     * mode 0: index order.  mode 1: longest first by a predicted iteration count — one fma forward sweep over the first
     * four horizon steps from the current workspace gives the largest primal residual per group of four instances, a bucket sort orders the groups;
     * applied to launches of at least 4096 groups, a no-op elsewhere.  mode 2 (second session of round 4): longest first by the iteration counts the
     * PREVIOUS solve of this workspace left in iter[] (the largest of a group's / tile's instances) — the order for warm-started launches, where the
     * sweep of mode 1 sees nothing (residuals of the size of the tolerance) and consecutive MPC steps are strongly correlated; index order where no
     * such history exists (after a reset, after an upload of iter[]).  It also orders the tiles / groups of an on-chip closed-loop run
     * (tiny_batch_mpc_run_async) by the solve before the run; a run that starts from a reset workspace is ordered by mode 1's predictor of its first,
     * cold solve (0.96 -> 0.84 ms per MPC step of that first run, 65 536 instances).  65 536 tracking instances, ms per warm-started MPC step: 1.39 -> 1.22 step by step
     * on the 16-lane kernel, 0.92 -> 0.81 inside the on-chip loop of the 16-instances-per-wave kernel (tools/warm_dispatch_ab.py).
     * mode -1 (default, round 4): automatic — mode 1 for a launch that starts from a reset workspace (where the iteration counts spread widely:
     * 2.20 -> 1.76 ms per solve of 65 536 tracking instances), mode 2 for warm-started ones.  The automatic kernel choice takes the
     * 16-instances-per-wave kernel only for cold-start launches that are dispatched longest first (from 160 instances per compute unit on: it loses
     * to the 16-lane kernel in index order) and for closed-loop runs of that size.  The 16-instances-per-wave kernel (tile16) orders its tiles of
     * sixteen instances by the largest key of their four groups. */
    int tiny_batch_set_dispatch(TinyBatch *tb, int mode);
    /* The tile queue of the 16-instances-per-wave kernel (tile16) under longest-first dispatch.  A launch of q tiles per wave slot ends in a partial
     * round: the slots return from their q-th tile together and the few tiles left occupy a fraction of the chip for one more tile's length.  With
     * stride k every k-th wave takes its tiles from the SHORT end of the predicted order instead of the long one, fits one tile more into the same
     * time, and the partial round disappears (65 536 tracking instances: makespan 132.5 -> 123 iterations simulated on the true counts,
     * tests/fuzz/sim_tile_deque.py).  -1 (default): automatic — for cold-start launches in predicted order of at least three tiles per slot and at most 32 768 tiles stride 4
     * (four to eight tiles per slot: the headline's 65 536 instances) or 8, otherwise one counter; 0: one counter; 1 .. 255: that stride.  Results never depend on it. */
    int tiny_batch_set_tile_queue(TinyBatch *tb, int stride);
    /* What the most recent solve launch actually did: 0 = index order (also when mode 1 / 2 did not apply: small launch, a kernel
     * without dispatch order, no history), 1 = longest first by the predicted iteration count, 2 = the caller's order, 3 = longest first by the
     * previous solve's iteration counts. */
    int tiny_batch_dispatch_applied(TinyBatch *tb);
    /* The caller's own order (e.g. from the iteration counts of the previous MPC step): d_order is a device array holding a
     * permutation of the ceil(batch/4) group indices, workgroup b solves instances 4*d_order[b] .. +3; it must stay valid
     * until the solves that use it have finished.  NULL returns to tiny_batch_set_dispatch's mode.  The order lists groups of FOUR
     * instances: it applies to the kernels that solve four instances per wave (unrolled and rolled 16-lane kernels); the
     * 16-instances-per-wave kernel (tile16) ignores it when forced, and the automatic choice stays off tile16 while an order is set
     * (tiny_batch_dispatch_applied() says what a launch did). */
    int tiny_batch_set_dispatch_order_device(TinyBatch *tb, const int *d_order);

    /* ---- the two terms the reference ships commented out, off by default ----
     * en_coeff_d2p: backward_pass_grad adds "+ coeff_d2p * d.col(i)" to p.col(i) (the trailing comment of
     * src/tinympc/admm.cpp:20; TinyCache::coeff_d2p, types.hpp:33).  en_uref: update_linear_cost computes
     * r = -(Uref o R) - rho*(znew - y) with TinyWorkspace::Uref and ::R (types.hpp:93,83) — the input-side twin of
     * admm.cpp:81-82, which is what the line commented out at admm.cpp:79 stands for.  A handle with a term switched on
     * runs the row kernel that streams its state (and the step kernels); nx + nu <= 16, variants 0/2/3.  With both off
     * (the default) nothing changes anywhere.  In exact arithmetic the results are bit-identical to Eigen evaluating
     * those expressions over the reference's types (pinned by the test suite). */
    int tiny_batch_set_optional_terms(TinyBatch *tb, int en_uref, int en_coeff_d2p);
    /* R: nu input-cost weights as stored in TinyWorkspace::R (types.hpp:83; the reference's codegen stores R + rho there) */
    int tiny_batch_set_input_cost(TinyBatch *tb, const float *R);
    /* coeff_d2p: nx x nu, column-major (types.hpp:33; tiny_riccati() computes it) */
    int tiny_batch_set_coeff_d2p(TinyBatch *tb, const float *coeff_d2p);
    /* Uref: [N-1][nu] when shared != 0, else [batch][N-1][nu] (types.hpp:93) */
    int tiny_batch_set_uref(TinyBatch *tb, const float *uref, int shared);

    /* ---- offline setup: Riccati cache precompute (src/tinympc/codegen.cpp:254-292), fp64, host ---- */
    /* A (nx x nx), B (nx x nu) column-major; Q (nx), R (nu) diagonals WITHOUT rho (the routine adds it,
     * codegen.cpp:255-256).  Outputs column-major.  *iters = Riccati iterations run (1000 = not converged,
     * the reference then keeps the last iterate too).  coeff_d2p may be NULL. */
    int tiny_riccati(int nx, int nu, const double *A, const double *B, const double *Q, const double *R, double rho,
                     double *Kinf, double *Pinf, double *Quu_inv, double *AmBKt, double *coeff_d2p, int *iters);

#ifdef __cplusplus
}
#endif
#endif /* TINYMPC_BATCH_H */
