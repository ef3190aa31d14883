"""Randomised differential test of the DISPATCH ORDERS at sizes where they act (test infrastructure; run on an MI355X):

    python tests/fuzz/fuzz_dispatch.py [seconds] [seed]

The other drivers use small batches, where tiny_batch_set_dispatch is a no-op (it acts on launches of >= 4 096 groups).  Every round here draws a batch
of 16 384 ... 70 000 quadrotor instances (N = 30), the 16-lane or the 16-instances-per-wave kernel, settings (iteration limits, termination stride,
tolerances, bound switches), a dispatch mode (-1 automatic, 0 index, 1 predictor, 2 history), a tile-queue stride (-1 ... 9) and a chain of one cold and
up to three warm-started solves (duals reset or kept, x0 moved), in half the rounds followed by an on-chip closed-loop run of 2 ... 4 MPC steps, and requires all twelve work arrays, the
residuals, status and iter after every solve to equal the ORACLE's bit for bit — whatever the order in which waves took their groups / tiles — and
tiny_batch_dispatch_applied() to report the order the mode and the history imply; the run's u0 trajectory, final iter / status / x0 must equal the oracle's loop."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import accelerated_tinympc_amd as T  # noqa: E402
from oracle import oracle as O  # noqa: E402

pr = T.problems
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
N = 30
prob = pr.quadrotor(20, N)
t_end, rounds, solves, t_note = time.time() + budget, 0, 0, time.time()
seen = {}
while time.time() < t_end:
    if time.time() - t_note > 30:
        print(f"... {rounds} rounds, {solves} solves so far", flush=True); t_note = time.time()
    B = int(rng.integers(16384, 70001))
    fam = int(rng.choice([1, 5]))
    settings = dict(abs_pri_tol=float(rng.choice([1e-3, 1e-3, 1e-2, 0.0])), abs_dua_tol=float(rng.choice([1e-3, 1e-3, 1e-1])),
                    max_iter=int(rng.choice([1, 2, 9, 30, 100])), check_termination=int(rng.choice([1, 1, 1, 2, 5])),
                    en_state_bound=int(rng.integers(2)), en_input_bound=1)
    settings = dict(O.DEFAULT_SETTINGS, **settings)
    mode, stride = int(rng.choice([-1, 0, 1, 2])), int(rng.integers(-1, 10))
    x0, table, start = pr.tracking_batch(B, N, seed=int(rng.integers(1 << 30)), spread=float(rng.choice([0.05, 0.3])))
    bnds = pr.bounds_arrays(prob)
    sol = T.TinyBatchSolver(prob, B, settings=settings)
    sol.select_kernel(2); sol.set_row_kernel(fam); sol.set_dispatch(mode); sol.set_tile_queue(stride)
    sol.set_bounds(*bnds); sol.set_xref_window(table, start); sol.set_x0(x0)
    orc = O.Oracle(prob, np.float32, settings)
    st = O.new_state(B, 12, 4, N); st["x"][:, 0] = x0
    xr = pr.expand_windows(table, start, N)
    history = False
    from_reset = rng.random() < 0.25   # no chain: the on-chip run below starts from the reset workspace (its tiles by the predictor of its first, cold solve)
    for k in range(0 if from_reset else int(rng.integers(1, 5))):
        if k > 0:
            if rng.random() < 0.7:
                st["y"][:] = 0; st["g"][:] = 0; sol.reset_dual_variables()
            x0 = (x0 * np.float32(rng.choice([1.0, 0.99, 1.02]))).astype(np.float32); st["x"][:, 0] = x0; sol.set_x0(x0)
        orc.solve(st, *bnds, xr, nthreads=8); sol.solve(); solves += 1
        big = (B + 3) // 4 >= 4096 and settings["max_iter"] > 1
        want = 0
        if big:
            cold = k == 0
            want = {0: 0, 1: 1, 2: 0 if (cold or not history) else 3, -1: 1 if cold else (3 if history else 0)}[mode]
        got_d = sol.dispatch_applied()
        assert got_d == want, f"dispatch applied {got_d}, expected {want}: mode {mode}, solve {k}, B {B}, family {fam}, settings {settings}"
        history = settings["max_iter"] > 0
        seen[(sol.kernel_name(), got_d)] = seen.get((sol.kernel_name(), got_d), 0) + 1
        got = sol.get_state()
        for name in O.STATE_ORDER + ("residuals", "status", "iter"):
            g_, r_ = got[name], st[name]
            same = np.array_equal(g_, r_) and (g_.dtype.kind != "f" or np.array_equal(np.signbit(g_), np.signbit(r_)))
            if not same:
                print(f"MISMATCH round {rounds} solve {k}: {name}; B {B} family {fam} mode {mode} stride {stride} settings {settings} kernel {sol.kernel_name()} applied {got_d}")
                sys.exit(1)
    if (from_reset or rng.random() < 0.5) and settings["max_iter"] > 0:   # ... and an on-chip closed-loop run from the state the chain left (tiny_batch_mpc_run_traj: K MPC steps in
        K, adv = int(rng.integers(2, 5)), int(rng.integers(0, 2))   # one launch, its tiles / groups in history order) against the oracle's loop (quadrotor_tracking.cpp:93-118)
        traj = sol.mpc_run_traj(K, adv)
        d_run = sol.dispatch_applied()
        if from_reset and (B + 3) // 4 >= 4096 and settings["max_iter"] > 1:
            assert d_run == {0: 0, 1: 1, 2: 0, -1: 1}[mode], f"run from a reset workspace: dispatch applied {d_run}, mode {mode}"
        x = st["x"][:, 0].copy(); ws = start.copy()
        for k in range(K):
            st["x"][:, 0] = x; st["y"][:] = 0; st["g"][:] = 0
            idx = np.minimum(ws[:, None] + np.arange(N)[None], table.shape[0] - 1)
            orc.solve(st, *bnds, np.ascontiguousarray(table[idx]), nthreads=8); solves += 1
            if not np.array_equal(traj[k], st["u"][:, 0]):
                print(f"MISMATCH round {rounds} on-chip run step {k}: u0; B {B} family {fam} mode {mode} stride {stride} settings {settings} kernel {sol.closed_loop_kernel_name()} applied {d_run}")
                sys.exit(1)
            x = orc.plant_step(x, st["u"][:, 0]); ws = ws + adv
        it_g, st_g, _ = sol.get_status()
        if not (np.array_equal(it_g, st["iter"]) and np.array_equal(st_g, st["status"]) and np.array_equal(sol.get_x0(), x)):
            print(f"MISMATCH round {rounds} on-chip run: iter / status / x0 after {K} steps; B {B} family {fam} mode {mode} settings {settings}")
            sys.exit(1)
        seen[("run:" + sol.closed_loop_kernel_name(), d_run)] = seen.get(("run:" + sol.closed_loop_kernel_name(), d_run), 0) + 1
    sol.close()
    rounds += 1
print(f"fuzz ok: {rounds} rounds, {solves} solves of 16 384 ... 70 000 instances, every bit equal to the oracle under every dispatch order; (kernel, order applied): "
      + ", ".join(f"{k[0]} / {k[1]}: {v}" for k, v in sorted(seen.items())))
