"""Test-infrastructure study (drives the oracle): what bounds the 16-instances-per-wave kernel beyond its per-iteration cost.
From the TRUE iteration counts of the bench workload (the oracle's) it simulates the launch — tiles of 16 instances in lock step, 1 024
one-wave-per-SIMD slots, list scheduling in dispatch order — for several predictors and groupings: lock-step factor, makespan,
workgroup-granular against wave-granular release of a CU, instances regrouped by window start / predictor / true count.
DESIGN.md section 5.4 quotes its output.   python tests/fuzz/sim_tile_dispatch.py"""
import sys, heapq, numpy as np
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[2]))
import accelerated_tinympc_amd as T
from oracle import oracle as O
pr = T.problems
prob = pr.quadrotor(20, 30); B = 65536; N = 30
x0, table, start = pr.tracking_batch(B, N)
xmn, xmx, umn, umx = pr.bounds_arrays(prob)
xr = pr.expand_windows(table, start, N)
def run(max_iter):
    st = O.new_state(B, 12, 4, N); st["x"][:, 0] = x0
    O.Oracle(prob, np.float32, dict(O.DEFAULT_SETTINGS, max_iter=max_iter)).solve(st, xmn, xmx, umn, umx, xr, nthreads=8)
    return st
full = run(100)
it = full["iter"].astype(np.int64)
def makespan(tile_iters, slots=1024, fixed=3.5):
    h = [0.0] * slots; heapq.heapify(h)
    for t in tile_iters:
        s = heapq.heappop(h); heapq.heappush(h, s + fixed + t)
    return max(h)
tile_true = it.reshape(-1, 16).max(axis=1)
ideal = makespan(tile_true[np.argsort(-tile_true)])
print("tiles sorted by TRUE tile count:", ideal, " index order:", makespan(tile_true))
def evalkey(name, key):
    k = key.reshape(-1, 16).max(axis=1)
    print(f"{name:60s} makespan {makespan(tile_true[np.argsort(-k)]):7.1f}  corr(inst) {np.corrcoef(key, it)[0,1]:.3f} corr(tile) {np.corrcoef(k, tile_true)[0,1]:.3f}")
# current predictor
A, Bm, K = prob["Adyn"].astype(np.float64), prob["Bdyn"].astype(np.float64), prob["Kinf"].astype(np.float64)
x = x0.astype(np.float64); key = np.zeros(B)
for i in range(8):
    u = -(x @ K.T)
    key = np.maximum(key, np.max(np.abs(x - np.clip(x, -5, 5)), axis=1)); key = np.maximum(key, np.max(np.abs(u - np.clip(u, -0.5, 0.5)), axis=1))
    x = x @ A.T + u @ Bm.T
evalkey("current: max primal residual, 8 steps, 1 sweep", key)
for k in (1, 2, 3, 4, 6, 8):
    st = run(k)
    r = st["residuals"].astype(np.float64)  # pri_x, pri_u, dua_x, dua_u ? order (ps, pi, ds, di)
    evalkey(f"after {k} iterations: max of 4 residuals", r.max(axis=1))
    evalkey(f"after {k} iterations: primal u residual", r[:, 1])
    evalkey(f"after {k} iterations: dual residuals max", np.maximum(r[:, 2], r[:, 3]))
    if k >= 2:
        prev = run(k - 1)["residuals"].astype(np.float64)
        rate = (r.max(axis=1) + 1e-9) / (prev.max(axis=1) + 1e-9)
        # predicted remaining iterations for geometric decay to 1e-3
        pred = k + np.log(np.maximum(r.max(axis=1), 1e-3) / 1e-3) / np.maximum(-np.log(np.clip(rate, 1e-3, 0.999)), 1e-3)
        evalkey(f"after {k} iterations: geometric extrapolation to tol", np.minimum(pred, 100))
# window start as key (instances tracking the same part of the trajectory)
evalkey("window start index", start.astype(np.float64))

print("---- workgroup granularity ----")
k = key.reshape(-1, 16).max(axis=1)
order = np.argsort(-k)
tt = tile_true[order]
# (1) 4 tiles per workgroup, a CU (4 slots) is released when all four are done: 256 CU-slots, job time = max of 4
wg = tt.reshape(-1, 4).max(axis=1)
print("workgroups of 4 tiles on 256 CUs (CU freed when its slowest tile ends):", makespan(wg, slots=256))
print("independent tiles on 1024 SIMD slots:", makespan(tt, slots=1024))
# lower bound
print("work / slots:", (tile_true.sum() + 3.5 * len(tile_true)) / 1024)

print("---- instance-level regrouping ----")
def run2(name, perm, sort_tiles_by=None):
    t = it[perm].reshape(-1, 16).max(axis=1)
    if sort_tiles_by is None:
        order = np.arange(len(t))
    else:
        order = np.argsort(-sort_tiles_by[perm].reshape(-1, 16).max(axis=1))
    print(f"{name:70s} lockstep {t.mean() / it.mean():.3f}  makespan {makespan(t[order]):7.1f}")
idx = np.arange(B)
run2("index order tiles, sorted by predictor", idx, key)
run2("instances sorted by window start (stable), tiles by predictor", np.argsort(start, kind="stable"), key)
run2("instances sorted by predictor", np.argsort(-key), key)
# two-level: bucket by window start, within bucket by predictor
perm = np.lexsort((-key, start))
run2("instances sorted by (window start, predictor)", perm, key)
perm = np.lexsort((start, -np.round(key, 2)))
run2("instances sorted by (rounded predictor, window start)", perm, key)
run2("instances sorted by TRUE count", np.argsort(-it), it.astype(float))
# how well does the mean iteration count of the window predict?
m = np.zeros(271); 
for s_ in range(271): m[s_] = it[start == s_].mean()
print("std of iteration count within a window", np.mean([it[start == s_].std() for s_ in range(271)]), "overall std", it.std())
