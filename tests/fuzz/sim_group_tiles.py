"""Test-infrastructure study (drives the oracle), round 4: would tiles formed from GROUPS adjacent in predicted order (instead of sixteen consecutive
instances) shorten the headline launch?  Replays the true iteration counts like sim_tile_dispatch.py.  Result: no — 134.5 against
132.5 iterations of makespan, lock step 1.125 either way; only the TRUE counts would (125.5 for groups of four, 116.5 for instances).
    python tests/fuzz/sim_group_tiles.py"""
import sys, heapq, numpy as np
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[2]))
import accelerated_tinympc_amd as T
from oracle import oracle as O
pr = T.problems
prob = pr.quadrotor(20, 30); B = 65536; N = 30
x0, table, start = pr.tracking_batch(B, N)
xmn, xmx, umn, umx = pr.bounds_arrays(prob)
xr = pr.expand_windows(table, start, N)
st = O.new_state(B, 12, 4, N); st["x"][:, 0] = x0
O.Oracle(prob, np.float32, dict(O.DEFAULT_SETTINGS, max_iter=100)).solve(st, xmn, xmx, umn, umx, xr, nthreads=8)
it = st["iter"].astype(np.int64)
def makespan(tile_iters, slots=1024, fixed=3.5):
    h = [0.0] * slots; heapq.heapify(h)
    for t in tile_iters:
        s = heapq.heappop(h); heapq.heappush(h, s + fixed + t)
    return max(h)
A, Bm, K = prob["Adyn"].astype(np.float64), prob["Bdyn"].astype(np.float64), prob["Kinf"].astype(np.float64)
def predictor(steps):
    x = x0.astype(np.float64); key = np.zeros(B)
    for i in range(steps):
        u = -(x @ K.T)
        key = np.maximum(key, np.max(np.abs(x - np.clip(x, -5, 5)), axis=1)); key = np.maximum(key, np.max(np.abs(u - np.clip(u, -0.5, 0.5)), axis=1))
        x = x @ A.T + u @ Bm.T
    return key
key = predictor(8)
def report(name, perm_inst, tile_key=None):
    t = it[perm_inst].reshape(-1,16).max(axis=1)
    if tile_key is None: order = np.arange(len(t))
    else: order = np.argsort(-tile_key)
    print(f"{name:72s} lockstep {t.mean()/it.mean():.3f} makespan {makespan(t[order]):7.1f}")
idx=np.arange(B)
tk = key.reshape(-1,16).max(axis=1)
report("current: 16 consecutive instances per tile, tiles sorted by predictor", idx, tk)
# groups of g consecutive instances sorted by predicted key; tiles = 16/g adjacent groups in that order
for g in (4, 2, 1, 8):
    gk = key.reshape(-1, g).max(axis=1)
    go = np.argsort(-gk, kind="stable")
    perm = (go[:,None]*g + np.arange(g)[None,:]).reshape(-1)
    report(f"tiles = {16//g} groups of {g} adjacent in predicted order", perm)
    # with the bucket sort's resolution (exponent + 3 mantissa bits)
    b = (np.float32(gk).view(np.uint32) & 0x7fffffff) >> 20
    go = np.argsort(-b.astype(np.int64), kind="stable")
    perm = (go[:,None]*g + np.arange(g)[None,:]).reshape(-1)
    report(f"   same with the 2048-bucket sort (order inside a bucket = index)", perm)
# true count groups
gk = it.reshape(-1,4).max(axis=1); go=np.argsort(-gk,kind="stable"); perm=(go[:,None]*4+np.arange(4)[None,:]).reshape(-1)
report("tiles = 4 groups of 4 adjacent in TRUE-count order", perm)
report("instances sorted by TRUE count", np.argsort(-it))
# alternative predictors: more steps
for s in (2, 30):
    k2 = predictor(s); gk = k2.reshape(-1,4).max(axis=1); go=np.argsort(-gk,kind="stable"); perm=(go[:,None]*4+np.arange(4)[None,:]).reshape(-1)
    report(f"tiles = 4 groups of 4, predictor over {s} steps", perm)
print("work/slots lower bound (no lockstep):", (it.sum()/16 + 3.5*4096)/1024, " mean iters", it.mean())
