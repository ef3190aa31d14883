"""Randomised consistency check (run on an MI355X): in fma arithmetic the three 16-lane row kernels (unrolled, rolled, state
in HBM) share their step arithmetic (rowlane_math.h), so they must agree with each other bit for bit even though none of them
is bit-comparable with the reference.      python tests/fuzz/fuzz_fast_families.py [seconds] [seed]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import accelerated_tinympc_amd as T  # noqa: E402

pr = T.problems
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
NAMES = ("x", "u", "q", "r", "p", "d", "v", "vnew", "z", "znew", "g", "y", "residuals", "status", "iter")
t_end, rounds, t_note = time.time() + budget, 0, time.time()
while time.time() < t_end:
    if time.time() - t_note > 30:
        print(f"... {rounds} rounds", flush=True); t_note = time.time()
    kind, N = [("quad", 30), ("quad", 20), ("quad", 10), ("odd", 7)][rng.integers(4)]
    prob = {"quad": lambda: pr.quadrotor(20, N), "odd": lambda: pr.random_system(8, 3, N, seed=99)}[kind]()
    nx, nu = prob["nx"], prob["nu"]
    B = int(rng.choice([1, 4, 17, 64]))
    settings = dict(abs_pri_tol=float(rng.choice([0.0, 1e-3])), abs_dua_tol=float(rng.choice([0.0, 1e-3])), max_iter=int(rng.choice([1, 4, 30])),
                    check_termination=int(rng.choice([1, 3])), en_state_bound=int(rng.integers(2)), en_input_bound=1)
    x0 = rng.uniform(-0.4, 0.4, size=(B, nx)).astype(np.float32)
    xref = (rng.standard_normal((B, N, nx)) * 0.2).astype(np.float32)
    warm = {k: None for k in ("d", "v", "z", "g", "y")}
    states = []
    for fam in (1, 2, 3):
        sol = T.TinyBatchSolver(prob, B, settings=settings)
        sol.select_kernel(3); sol.set_row_kernel(fam)
        sol.set_bounds(*pr.bounds_arrays(prob)); sol.set_xref(xref); sol.set_x0(x0)
        for k in warm:
            if warm[k] is None:
                warm[k] = (rng.standard_normal(sol.get_array(k).shape) * 0.05).astype(np.float32)
            sol.set_array(k, warm[k])
        for _ in range(2):
            sol.reset_dual_variables(); sol.solve()
        states.append((sol.kernel_name(), sol.get_state()))
        sol.close()
    for name in NAMES:
        for kn, stt in states[1:]:
            if not np.array_equal(stt[name], states[0][1][name], equal_nan=stt[name].dtype.kind == "f"):
                print(f"MISMATCH {kn} vs {states[0][0]}: {name}  ({kind} N={N} B={B} {settings})"); sys.exit(1)
    rounds += 1
print(f"fuzz ok: {rounds} rounds, rowlane == rowloop == rowstream in fma arithmetic, bit for bit")
