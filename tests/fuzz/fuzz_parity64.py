"""Randomised differential test of the fp64 library (include/tinympc_batch64.h) against the CPU oracle's fp64
instantiation (test infrastructure; run on an MI355X):

    python tests/fuzz/fuzz_parity64.py [seconds] [seed]

Every round draws one of the three instantiated classes with a random horizon and batch size, settings, bounds (shared or
per instance, some infeasible or infinite), a reference (shared or per instance), a cold or random warm workspace (with
zeros and negative zeros), runs a chain of solves and requires all twelve work arrays, the residuals, status and iter to
equal the oracle's bit for bit."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import accelerated_tinympc_amd as T  # noqa: E402
from oracle import oracle as O  # noqa: E402

pr = T.problems
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
CLASSES = [("quad", 10), ("quad", 30), ("quad", 20), ("quad", 7), ("quad", 41), ("cartpole", 10), ("cartpole", 3), ("cartpole", 33), ("r8_4", 9), ("r8_4", 26), ("r12_2", 11), ("r4_2", 8), ("r4_4", 6), ("r16_4", 10)]
t_end, rounds, solves, t_note, overflowed = time.time() + budget, 0, 0, time.time(), 0
while time.time() < t_end:
    if time.time() - t_note > 45:
        print(f"... {rounds} rounds, {solves} solves so far", flush=True)
        t_note = time.time()
    kind, N = CLASSES[rng.integers(len(CLASSES))]
    prob = {"quad": lambda: pr.quadrotor(20, N), "cartpole": lambda: pr.cartpole(N), }.get(kind, lambda: pr.random_system(int(kind[1:].split("_")[0]), int(kind.split("_")[1]), N, seed=7))()
    nx, nu = prob["nx"], prob["nu"]
    B = int(rng.choice([1, 2, 3, 63, 64, 65, 200, 257]))
    settings = dict(abs_pri_tol=float(rng.choice([0.0, 1e-3, 1e-2, 0.5])), abs_dua_tol=float(rng.choice([0.0, 1e-3, 1e-1, 5.0])),
                    max_iter=int(rng.choice([0, 1, 2, 3, 7, 20, 45])), check_termination=int(rng.choice([1, 1, 2, 3, 7])),
                    en_state_bound=int(rng.integers(2)), en_input_bound=int(rng.integers(2)))
    xmn, xmx, umn, umx = [np.asarray(a, np.float64).copy() for a in pr.bounds_arrays(prob)]
    scale = rng.uniform(0.05, 1.0)
    xmn *= scale; xmx *= scale * rng.uniform(0.5, 1.5); umn *= rng.uniform(0.1, 1.0, size=umn.shape); umx *= scale
    if rng.random() < 0.3:
        umn[rng.integers(N - 1), rng.integers(nu)] = 3.0
        xmx[rng.integers(N), rng.integers(nx)] = np.inf
        xmn[rng.integers(N), rng.integers(nx)] = -np.inf
    bnds = (xmn, xmx, umn, umx)
    if rng.random() < 0.3:
        bnds = tuple(a[None] * rng.uniform(0.3, 1.0, size=(B,) + a.shape) for a in bnds)
    sol = T.TinyBatchSolver64(prob, B, settings=settings)
    if rng.random() < 0.4:
        sol.select_kernel(1)   # one thread per instance; otherwise automatic (sixteen lanes per instance where instantiated)
    sol.set_bounds(*bnds)
    xref = rng.standard_normal((N, nx) if rng.random() < 0.5 else (B, N, nx)) * 0.3
    sol.set_xref(xref)
    st = O.new_state(B, nx, nu, N, np.float64)
    if rng.random() < 0.7:
        for k in O.STATE_ORDER:
            v = rng.standard_normal(st[k].shape) * 0.3
            v[rng.random(v.shape) < 0.1] = 0.0
            v[rng.random(v.shape) < 0.05] = -0.0
            st[k][:] = v
        st["residuals"][:] = rng.uniform(0, 1, size=(B, 4))
        st["iter"][:] = rng.integers(1, 9, size=B); st["status"][:] = 11
        sol.set_state(st)
    else:
        x0 = rng.uniform(-0.5, 0.5, size=(B, nx))
        st["x"][:, 0] = x0; sol.set_x0(x0)
    orc = O.Oracle(prob, np.float64, settings)
    for k in range(int(rng.integers(1, 4))):
        if rng.random() < 0.6:
            st["y"][:] = 0; st["g"][:] = 0; sol.reset_dual_variables()
        rc_ref = orc.solve(st, *bnds, xref, nthreads=8); rc = sol.solve(); solves += 1
        if not all(np.all(np.isfinite(st[n_])) for n_ in O.STATE_ORDER):
            overflowed += 1
            break
        got = sol.get_state()
        if rc != (1 if rc_ref else 0):
            print(f"MISMATCH round {rounds}: return code {rc} vs {rc_ref}"); sys.exit(1)
        for name in O.STATE_ORDER + ("residuals", "status", "iter"):
            g_, r_ = got[name], st[name]
            if g_.dtype.kind == "f":
                nn = np.isnan(g_) & np.isnan(r_)
                same = np.all((g_ == r_) | nn) and np.array_equal(np.signbit(g_)[~nn], np.signbit(r_)[~nn])
            else:
                same = np.array_equal(g_, r_)
            if not same:
                print(f"MISMATCH round {rounds} {kind} N={N} B={B} kernel {sol.kernel_name()} settings {settings} per-instance bounds {bnds[0].ndim == 3} xref {xref.shape} solve {k}: {name}")
                bad = np.argwhere(~((g_ == r_) & (np.signbit(g_) == np.signbit(r_)))) if g_.dtype.kind == "f" else np.argwhere(g_ != r_)
                for idx in bad[:6]:
                    idx = tuple(idx)
                    print("   ", idx, "gpu", g_[idx], "oracle", r_[idx], "iter gpu/oracle", got["iter"][idx[0]], st["iter"][idx[0]])
                sys.exit(1)
    sol.close(); rounds += 1
print(f"fuzz64 ok: {rounds} rounds, {solves} solves, all bitwise equal to the fp64 oracle (signs of zeros included); "
      f"{overflowed} rounds left the finite range and were not compared")
