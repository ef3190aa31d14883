"""Randomised check (run on an MI355X): tiny_batch_mpc_run_async(K) — on-chip closed loop or graph replay — must leave exactly
the state of K calls of tiny_batch_mpc_step_async, for random classes, kernels, arithmetic modes, batches, settings,
window advances and step counts.      python tests/fuzz/fuzz_mpc.py [seconds] [seed]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import accelerated_tinympc_amd as T  # noqa: E402

pr = T.problems
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
CLASSES = [("quad", 30), ("quad", 20), ("quad", 10), ("quad", 17), ("quad", 36), ("cartpole", 10), ("cartpole", 23), ("odd", 7)]
NAMES = ("x", "u", "q", "r", "p", "d", "v", "vnew", "z", "znew", "g", "y", "residuals", "status", "iter")
t_end, rounds, t_note, onchip = time.time() + budget, 0, time.time(), 0
while time.time() < t_end:
    if time.time() - t_note > 30:
        print(f"... {rounds} rounds ({onchip} on chip)", flush=True); t_note = time.time()
    kind, N = CLASSES[rng.integers(len(CLASSES))]
    prob = {"quad": lambda: pr.quadrotor(20, N), "cartpole": lambda: pr.cartpole(N), "odd": lambda: pr.random_system(8, 3, N, seed=99)}[kind]()
    nx, nu = prob["nx"], prob["nu"]
    B = int(rng.choice([1, 3, 4, 16, 17, 64, 130]))
    settings = dict(abs_pri_tol=float(rng.choice([1e-3, 1e-2])), abs_dua_tol=float(rng.choice([1e-3, 1e-1])), max_iter=int(rng.choice([1, 3, 20, 60])),
                    check_termination=int(rng.choice([1, 1, 2, 5])), en_state_bound=int(rng.integers(2)), en_input_bound=int(rng.integers(2)))
    variant = int(rng.choice([2, 3, 1])) if kind != "odd" or True else 2
    fam = int(rng.choice([0, 0, 1, 2, 3, 4, 5, 5]))   # 5 = tile16 (round 4: its MPC loop stays on chip too)
    K, adv = int(rng.integers(1, 9)), int(rng.choice([0, 1, 2]))
    table = (rng.standard_normal((N + 60, nx)) * 0.1).astype(np.float32)
    start = rng.integers(0, 40, size=B).astype(np.int32)
    x0 = rng.uniform(-0.2, 0.2, size=(B, nx)).astype(np.float32)
    windowed = rng.random() < 0.6
    shared_ref = fam == 5 and rng.random() < 0.5   # one shared reference: the other mode the 16-instances-per-wave kernel serves
    sols = []
    for _ in range(2):
        s = T.TinyBatchSolver(prob, B, settings=settings)
        try:
            s.select_kernel(variant)
            if variant != 1:
                s.set_row_kernel(fam)
        except T.TinyBatchError:
            s.select_kernel(0); s.set_row_kernel(0)
        s.set_bounds(*pr.bounds_arrays(prob))
        if windowed:
            s.set_xref_window(table, start)
        elif shared_ref:
            s.set_xref(table[:N])
        else:
            s.set_xref(pr.expand_windows(table, start, N))
        s.set_x0(x0)
        sols.append(s)
    a, b = sols
    onchip += a.kernel_name().startswith(("rowlane", "quadlane", "tile16")) and K > 1
    for rnd in range(2):
        a.mpc_run_async(K, adv)
        for _ in range(K):
            b.mpc_step_async(adv)
        sa, sb = a.get_state(), b.get_state()
        xa, xb = a.get_x0(), b.get_x0()
        for name in NAMES:
            ga, gb = sa[name], sb[name]
            same = np.array_equal(ga, gb, equal_nan=True) if ga.dtype.kind == "f" else np.array_equal(ga, gb)
            if not same or not np.array_equal(xa, xb, equal_nan=True):
                print(f"MISMATCH round {rounds}.{rnd} {a.kernel_name()} B={B} K={K} adv={adv} windowed={windowed} settings {settings}: {name}")
                sys.exit(1)
    a.close(); b.close(); rounds += 1
print(f"fuzz ok: {rounds} rounds ({onchip} through the on-chip closed loop), mpc_run == step by step, bit for bit")
