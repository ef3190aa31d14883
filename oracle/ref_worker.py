#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (oracle/): one worker process of bench.py's `cpu_baseline.all_cores` leg.

The compiled reference (oracle/_ref: the reference's own src/tinympc/admm.cpp with the vendored Eigen, built by oracle/Makefile)
keeps its solver in process-global objects, exactly like the reference's generated wrapper (tiny_wrapper.cpp: one global
`tiny_data_solver`), so it cannot be driven from several threads of one process.  The all-core figure therefore runs ONE PROCESS
PER CORE, each timing cold-start tiny_solve() calls over its own contiguous slice of the same workload bench.py gives the GPU.
bench.py starts these workers before it touches the GPU and sums their rates; nothing here is on the product path.

    python oracle/ref_worker.py --config tracking --total 65536 --start 0 --count 2048 --seconds 6 --settings '{...}'

Prints one JSON line: {"solves": n, "seconds": t, "mean_iters": m, "kind": "reference" | "port"}.
"""
from __future__ import annotations

import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True, choices=["tracking", "random32"])
    ap.add_argument("--total", type=int, required=True)
    ap.add_argument("--start", type=int, required=True)
    ap.add_argument("--count", type=int, required=True)
    ap.add_argument("--seconds", type=float, default=6.0)
    ap.add_argument("--settings", required=True)
    ap.add_argument("--go-at", type=float, default=0.0, help="time.time() at which every worker starts its timed loop")
    a = ap.parse_args()
    import accelerated_tinympc_amd as T  # problem data and the workload generators only: no library is loaded, no GPU is touched
    from oracle import oracle as O
    pr = T.problems
    settings = json.loads(a.settings)
    if a.config == "tracking":
        N = 30
        prob = pr.quadrotor(20, N)
        gx0, table, gstart = pr.tracking_batch(a.total, N)
        lo, hi = a.start, min(a.total, a.start + a.count)
        x0, xr = gx0[lo:hi], pr.expand_windows(table, gstart[lo:hi], N)
    else:
        N = 50
        prob = pr.random_system(32, 16, N, seed=1234)
        gx0, xref0 = pr.random_batch(a.total, 32, N)
        lo, hi = a.start, min(a.total, a.start + a.count)
        x0, xr = gx0[lo:hi], xref0
    nx, nu = prob["nx"], prob["nu"]
    kind = "reference" if O.have_ref(np.float32, nx, nu, N) else "port"
    solver = (O.Reference if kind == "reference" else O.Oracle)(prob, np.float32, settings)
    bnds = pr.bounds_arrays(prob)
    nb = len(x0)

    def one_pass():
        st = O.new_state(nb, nx, nu, N)
        st["x"][:, 0] = x0
        t0 = time.perf_counter()
        solver.solve(st, *bnds, xr, nthreads=1)
        return time.perf_counter() - t0, float(st["iter"].mean())

    one_pass()  # page everything in
    while time.time() < a.go_at:
        time.sleep(0.005)
    n, t, its = 0, 0.0, []
    while t < a.seconds:
        dt, mi = one_pass()
        n += nb; t += dt; its.append(mi)
    print(json.dumps({"solves": n, "seconds": t, "mean_iters": float(np.mean(its)), "kind": kind}), flush=True)


if __name__ == "__main__":
    main()
