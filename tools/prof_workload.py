#!/usr/bin/env python3
"""Run ONE kernel of the library on its BASELINE.json workload for a few launches — the program that
tools/collect_kernel_counters.sh puts behind `rocprofv3 ... --` (one kernel per process, so that per-kernel averages of
the profiler are averages over launches of that one workload).

    python3 tools/prof_workload.py <workload> [--launches K] [--batch B]

Prints one JSON line: kernel name, hipEvent kernel time, iteration statistics, algorithmic flops of the launch.
"""
from __future__ import annotations

import argparse
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

# name -> (problem, batch, xref setup, select_kernel variant, row family, settings, solver class)
WORKLOADS = {
    "rowlane_exact": dict(prob="q30", batch=65536, ref="track", variant=2, family=1),
    "rowlane_fast": dict(prob="q30", batch=65536, ref="track", variant=3, family=1),
    "tile16_exact": dict(prob="q30", batch=65536, ref="track", variant=2, family=5),
    "tile16_fast": dict(prob="q30", batch=65536, ref="track", variant=3, family=5),
    # the headline workload with per-instance tables (admm_tile16_pi.hip): every instance its own bounds (constant along the horizon: one resident row),
    # its own reference trajectory (per step: a ring fed from the tile image), both per step; and the 16-lane kernel on the same inputs
    "tile16_pi_bounds_const": dict(prob="q30", batch=65536, ref="track", variant=2, family=5, pi_bounds="const"),
    "tile16_pi_xref_steps": dict(prob="q30", batch=65536, ref="inst", variant=2, family=5),
    "tile16_pi_both_steps": dict(prob="q30", batch=65536, ref="inst", variant=2, family=5, pi_bounds="steps"),
    "rowlane_pi_both_steps": dict(prob="q30", batch=65536, ref="inst", variant=2, family=1, pi_bounds="steps"),
    "rowloop_exact": dict(prob="q17", batch=65536, ref="track", variant=2, family=2),
    "rowstream_exact": dict(prob="q17", batch=65536, ref="track", variant=2, family=3),
    "stream_3_1": dict(prob="q30", batch=65536, ref="track", variant=1, family=0),
    "quadlane_exact": dict(prob="cp", batch=32768, ref="zero", variant=2, family=4, settings=dict(max_iter=150)),
    "waveres_exact": dict(prob="r32", batch=16384, ref="zero", variant=2, family=7),
    "waveres_exact_2048": dict(prob="r32", batch=2048, ref="zero", variant=2, family=7),
    "waveres_fast": dict(prob="r32", batch=16384, ref="zero", variant=3, family=7),
    "tile48_exact": dict(prob="r32", batch=16384, ref="zero", variant=2, family=8),
    "tile48_exact_2048": dict(prob="r32", batch=2048, ref="zero", variant=2, family=8),
    "wavestream_exact": dict(prob="r32", batch=16384, ref="zero", variant=2, family=6),
    "stream_8_4": dict(prob="r32", batch=16384, ref="zero", variant=1, family=0),
    "rows64": dict(prob="q30", batch=65536, ref="track", f64=True, which=2),
    "thread64": dict(prob="q30", batch=65536, ref="track", f64=True, which=1),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", choices=list(WORKLOADS))
    ap.add_argument("--launches", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--dispatch", type=int, default=0)
    args = ap.parse_args()
    import accelerated_tinympc_amd as T
    from bench import Cost

    pr = T.problems
    w = WORKLOADS[args.workload]
    B = args.batch or w["batch"]
    if w["prob"] == "q30":
        prob = pr.quadrotor(20, 30)
    elif w["prob"] == "q17":
        prob = pr.quadrotor(20, 17)
    elif w["prob"] == "cp":
        prob = pr.cartpole(10)
    else:
        prob = pr.random_system(32, 16, 50, seed=1234)
    nx, nu, N = prob["nx"], prob["nu"], prob["N"]
    settings = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1, en_state_bound=1, en_input_bound=1)
    settings.update(w.get("settings", {}))
    if w["ref"] in ("track", "inst"):
        x0, table, start = pr.tracking_batch(B, N)
    elif w["prob"] == "cp":
        rng = np.random.default_rng(1)
        x0 = np.tile(np.array([[0, 0, 0.1, 0]], np.float32), (B, 1)) + rng.uniform(-0.05, 0.05, size=(B, 4)).astype(np.float32)
    else:
        x0, _ = pr.random_batch(B, nx, N)
    if w.get("f64"):
        sol = T.TinyBatchSolver64(prob, B, settings=settings)
        sol.select_kernel(w["which"])
        sol.set_bounds(*[np.asarray(a, np.float64) for a in pr.bounds_arrays(prob)])
        sol.set_xref(pr.expand_windows(table, start, N).astype(np.float64))
        zero = {k: np.zeros_like(v) for k, v in sol.get_state().items() if k not in ("iter", "status", "residuals")}
        import time
        ms = []
        for _ in range(args.launches + 1):
            for k, v in zero.items():
                sol.set_array(k, v)
            sol.set_x0(x0.astype(np.float64))
            t0 = time.perf_counter()
            sol.solve()
            ms.append((time.perf_counter() - t0) * 1e3)
        ms = ms[1:]
        it, st, _ = sol.get_status()
    else:
        sol = T.TinyBatchSolver(prob, B, settings=settings)
        sol.select_kernel(w["variant"])
        if w["family"]:
            sol.set_row_kernel(w["family"])
        bnds = pr.bounds_arrays(prob)
        if w.get("pi_bounds"):   # the same values for every instance (the iterates do not change); "steps": one entry differs along the horizon
            bnds = [np.broadcast_to(a, (B,) + a.shape).copy() for a in bnds]
            if w["pi_bounds"] == "steps":
                bnds[1][:, 5, 0] *= 1.0000001
        sol.set_bounds(*bnds)
        if w["ref"] == "inst":
            sol.set_xref(pr.expand_windows(table, start, N))
        elif w["ref"] == "track":
            sol.set_xref_window(table, start)
        else:
            sol.set_xref(np.zeros((N, nx), np.float32))
        sol.set_dispatch(args.dispatch)
        sol.enable_timing(True)
        ms = []
        for r in range(args.launches + 1):
            sol.reset_workspace()
            sol.set_x0(x0)
            sol.solve_async()
            sol.synchronize()
            if r:
                ms.append(sol.last_solve_ms())
        it, st, _ = sol.get_status()
    cost = Cost(nx, nu, N)
    fl = cost.flops_of(it, st)
    k_ms = float(np.mean(ms))
    peak = 78.6 if w.get("f64") else 157.3
    print(json.dumps(dict(workload=args.workload, kernel=sol.kernel_name(), batch=B, nx=nx, nu=nu, N=N, launches=args.launches,
                          kernel_ms=k_ms, kernel_ms_all=ms, mean_iters=float(it.mean()), max_iters=int(it.max()),
                          frac_converged=float(np.mean(st == 1)), alg_flops_per_launch=fl, alg_bytes_per_launch=B * cost.b_solve * (2 if w.get("f64") else 1),
                          tflops=fl / (k_ms * 1e-3) / 1e12, roof_frac=fl / (k_ms * 1e-3) / 1e12 / peak, roof_tflops=peak)), flush=True)
    sol.close()


if __name__ == "__main__":
    main()
