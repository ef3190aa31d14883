"""Randomised check of the native-names compatibility API (include/tinympc_admm.h; run on an MI355X): random sequences of
tiny_solve / forward_pass / update_slack / update_dual / update_linear_cost / termination_condition / backward_pass_grad on a
caller-owned TinySolver whose bounds, reference, settings and gains change between calls, mirrored with the CPU oracle.
Half of the rounds run the double build of the same names (libtinympc_wrapper64.so, TINYMPC_TINYTYPE_DOUBLE) against the fp64 oracle.
      python tests/fuzz/fuzz_native.py [seconds] [seed]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import accelerated_tinympc_amd as T  # noqa: E402
from accelerated_tinympc_amd import native  # noqa: E402
from oracle import oracle as O  # noqa: E402

pr = T.problems
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end, rounds, calls, t_note = time.time() + budget, 0, 0, time.time()
while time.time() < t_end:
    if time.time() - t_note > 30:
        print(f"... {rounds} rounds, {calls} calls", flush=True); t_note = time.time()
    double = rng.random() < 0.5
    dt = np.float64 if double else np.float32
    kind, N = ([("quad", 30), ("quad", 10), ("quad", 17), ("cartpole", 10), ("r8_4", 9)] if double else
               [("quad", 30), ("quad", 17), ("cartpole", 10), ("odd", 7)])[rng.integers(5 if double else 4)]
    prob = {"quad": lambda: pr.quadrotor(20, N), "cartpole": lambda: pr.cartpole(N, riccati=O.riccati), "odd": lambda: pr.random_system(8, 3, N, seed=99),
            "r8_4": lambda: pr.random_system(8, 4, N, seed=804, riccati=O.riccati)}[kind]()
    nx, nu = prob["nx"], prob["nu"]
    settings = dict(O.DEFAULT_SETTINGS, max_iter=int(rng.choice([1, 5, 30])), check_termination=int(rng.choice([1, 2])))
    ns = native.NativeSolver(prob, settings, dtype=dt)
    bnds = [np.asarray(a, dt).copy() for a in pr.bounds_arrays(prob)]
    xref = (rng.standard_normal((N, nx)) * 0.2).astype(dt)
    st = O.new_state(1, nx, nu, N, dt)
    for k in O.STATE_ORDER:
        st[k][:] = (rng.standard_normal(st[k].shape) * 0.2).astype(dt)
    st["iter"][:] = 2; st["status"][:] = 11
    hist = []
    for _ in range(int(rng.integers(3, 10))):
        # mutate the caller's structs
        if rng.random() < 0.3:
            sc = rng.uniform(0.2, 1.0); bnds = [(a * sc).astype(dt) for a in pr.bounds_arrays(prob)]
        if rng.random() < 0.3:
            xref = (rng.standard_normal((N, nx)) * 0.2).astype(dt)
        if rng.random() < 0.2:
            settings = dict(settings, max_iter=int(rng.choice([1, 5, 30])), en_input_bound=int(rng.integers(2)), abs_pri_tol=float(rng.choice([1e-3, 0.3])))
        if rng.random() < 0.1:
            st["d"][:] = (rng.standard_normal(st["d"].shape) * 0.1).astype(dt)
        for name, arr in zip(("x_min", "x_max", "u_min", "u_max"), bnds):
            ns.a[name][:] = arr
        ns.a["Xref"][:] = xref
        for f in ("abs_pri_tol", "abs_dua_tol", "max_iter", "check_termination", "en_state_bound", "en_input_bound"):
            setattr(ns.settings, f, settings[f])
        for k in O.STATE_ORDER:
            ns.a[k][:] = st[k][0]
        w = ns.work
        (w.primal_residual_state, w.primal_residual_input, w.dual_residual_state, w.dual_residual_input) = map(float, st["residuals"][0])
        w.iter, w.status = int(st["iter"][0]), int(st["status"][0])
        orc = O.Oracle(prob, dt, settings)
        fn = ["tiny_solve", *O.Oracle.STEP_FUNCTIONS][rng.integers(7)]
        hist.append(fn); calls += 1
        if fn == "tiny_solve":
            orc.solve(st, *bnds, xref[None]); rc = ns.tiny_solve()
            ok = rc == (1 if st["status"][0] == 11 else 0)
        else:
            ref_rv = orc.step(fn, st, *bnds, xref[None]); rv = ns.call(fn)
            ok = fn != "termination_condition" or bool(rv) == bool(ref_rv[0])
        if not all(np.all(np.isfinite(st[n_])) for n_ in O.STATE_ORDER):
            break
        for k in O.STATE_ORDER:
            ok = ok and np.array_equal(ns.a[k], st[k][0]) and np.array_equal(np.signbit(ns.a[k]), np.signbit(st[k][0]))
        ok = ok and np.array_equal(ns.residuals, st["residuals"][0]) and ns.work.iter == st["iter"][0] and ns.work.status == st["status"][0]
        if not ok:
            print(f"MISMATCH {'double' if double else 'float'} {kind} N={N} after {hist} settings {settings}"); sys.exit(1)
    rounds += 1
print(f"fuzz ok: {rounds} rounds, {calls} calls of the native-names API equal to the oracle bit for bit")
