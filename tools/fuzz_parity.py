"""Randomised differential test of the exact kernels against the CPU oracle (test infrastructure; run on an MI355X):

    python tools/fuzz_parity.py [seconds] [seed]

Every round draws a problem class with an exact kernel, a batch size, settings (iteration limits, termination stride,
tolerances, bound switches), bounds (per step, some infeasible or infinite), a reference (shared / per instance / sliding
window), a random warm workspace (with zeros and negative zeros) and a row-kernel family, runs a chain of solves and
requires all twelve work arrays, the residuals, status and iter to equal the oracle's bit for bit."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import accelerated_tinympc_amd as T  # noqa: E402
from oracle import oracle as O  # noqa: E402

pr = T.problems
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
CLASSES = [("quad", 30), ("quad", 25), ("quad", 20), ("quad", 10), ("quad", 17), ("quad", 7), ("quad", 36), ("cartpole", 10),
           ("cartpole", 23), ("odd", 7), ("odd", 13), ("rand32", 50)]
t_end, rounds, solves, t_note = time.time() + budget, 0, 0, time.time()
while time.time() < t_end:
    if time.time() - t_note > 30:
        print(f"... {rounds} rounds, {solves} solves so far", flush=True)
        t_note = time.time()
    kind, N = CLASSES[rng.integers(len(CLASSES))]
    prob = {"quad": lambda: pr.quadrotor(20, N), "cartpole": lambda: pr.cartpole(N), "odd": lambda: pr.random_system(8, 3, N, seed=99),
            "rand32": lambda: pr.random_system(32, 16, N)}[kind]()
    nx, nu = prob["nx"], prob["nu"]
    B = int(rng.choice([1, 2, 3, 4, 5, 15, 16, 17, 63, 64, 65, 200])) if kind != "rand32" else int(rng.choice([1, 3, 9]))
    settings = dict(abs_pri_tol=float(rng.choice([0.0, 1e-3, 1e-2, 0.5])), abs_dua_tol=float(rng.choice([0.0, 1e-3, 1e-1, 5.0])),
                    max_iter=int(rng.choice([0, 1, 2, 3, 7, 20, 45])), check_termination=int(rng.choice([1, 1, 2, 3, 7])),
                    en_state_bound=int(rng.integers(2)), en_input_bound=int(rng.integers(2)))
    xmn, xmx, umn, umx = [a.copy() for a in pr.bounds_arrays(prob)]
    scale = rng.uniform(0.05, 1.0)
    xmn *= scale; xmx *= scale * rng.uniform(0.5, 1.5); umn *= rng.uniform(0.1, 1.0, size=umn.shape).astype(np.float32); umx *= scale
    if rng.random() < 0.3:   # infeasible and infinite entries (examples/codegen_random.cpp:28-31 has min > max)
        umn[rng.integers(N - 1), rng.integers(nu)] = 3.0
        xmx[rng.integers(N), rng.integers(nx)] = np.inf
        xmn[rng.integers(N), rng.integers(nx)] = -np.inf
    bnds = (xmn, xmx, umn, umx)
    sol = T.TinyBatchSolver(prob, B, settings=settings)
    fams = [0] + ([f for f in (1, 2, 3, 4) if kind != "rand32"])
    fam = int(rng.choice(fams))
    try:
        sol.set_row_kernel(fam)
    except T.TinyBatchError:
        sol.set_row_kernel(0)
    sol.set_bounds(*bnds)
    mode = rng.integers(3) if nx <= 16 else rng.integers(2)
    if mode == 0:
        xref = (rng.standard_normal((N, nx)) * 0.3).astype(np.float32); sol.set_xref(xref)
    elif mode == 1:
        xref = (rng.standard_normal((B, N, nx)) * 0.3).astype(np.float32); sol.set_xref(xref)
    else:
        table = (rng.standard_normal((N + 40, nx)) * 0.3).astype(np.float32)
        start = rng.integers(0, 40, size=B).astype(np.int32)
        sol.set_xref_window(table, start); xref = pr.expand_windows(table, start, N)
    st = O.new_state(B, nx, nu, N)
    if rng.random() < 0.7:   # warm workspace, sprinkled with zeros and negative zeros
        for k in O.STATE_ORDER:
            v = (rng.standard_normal(st[k].shape) * 0.3).astype(np.float32)
            v[rng.random(v.shape) < 0.1] = 0.0
            v[rng.random(v.shape) < 0.05] = -0.0
            st[k][:] = v
        st["residuals"][:] = rng.uniform(0, 1, size=(B, 4)).astype(np.float32)
        st["iter"][:] = rng.integers(1, 9, size=B); st["status"][:] = 11
        sol.set_state(st)
    else:
        x0 = rng.uniform(-0.5, 0.5, size=(B, nx)).astype(np.float32)
        st["x"][:, 0] = x0; sol.set_x0(x0)
    orc = O.Oracle(prob, np.float32, settings)
    for k in range(int(rng.integers(1, 4))):
        if rng.random() < 0.6:
            st["y"][:] = 0; st["g"][:] = 0; sol.reset_dual_variables()
        orc.solve(st, *bnds, xref, nthreads=8); sol.solve(); solves += 1
        got = sol.get_state()
        for name in O.STATE_ORDER + ("residuals", "status", "iter"):
            if not np.array_equal(got[name], st[name]) or (got[name].dtype == np.float32 and not np.array_equal(np.signbit(got[name]), np.signbit(st[name]))):
                print(f"MISMATCH round {rounds} {kind} N={N} B={B} kernel {sol.kernel_name()} settings {settings} xref mode {mode} solve {k}: {name}")
                bad = np.argwhere(~((got[name] == st[name]) & (np.signbit(got[name]) == np.signbit(st[name]))))
                for idx in bad[:6]:
                    idx = tuple(idx)
                    print("   ", idx, "gpu", got[name][idx], "oracle", st[name][idx], "iter gpu/oracle", got["iter"][idx[0]], st["iter"][idx[0]])
                sys.exit(1)
    sol.close(); rounds += 1
print(f"fuzz ok: {rounds} rounds, {solves} solves, all bitwise equal to the oracle (signs of zeros included)")
