"""Test-infrastructure study (drives the oracle): the two-ended tile queue of admm_tile16.hip (tiny_batch_set_tile_queue).

Under longest-first dispatch a launch of q tiles per wave slot ends in a PARTIAL ROUND: the tiles of a tracking batch take 17 ... 28 lock-step
iterations apart from a thin tail (up to 85), list scheduling in descending order hands every slot q tiles of nearly the same total, and the
few hundred tiles left over occupy a sixth of the chip for one more tile's length (makespan 132.5 iterations for 113.8 of work per slot at
65 536 instances).  No order of ONE queue avoids that; letting every k-th wave draw from the SHORT end of the same order does: such a wave fits
one tile more into the same time.  This script replays the oracle's TRUE iteration counts (tiles of sixteen in lock step, 1 024 one-wave-per-SIMD
slots, 3.5 iterations of fixed cost per tile) for the predictor's order and several strides and batch sizes.  DESIGN.md 5.4 quotes its output;
tools/t16_queue_ab.py is the measurement on the chip.      python tests/fuzz/sim_tile_deque.py [batch,seed ...]
With the argument `regroup` instead: what INSTANCE-level regrouping into tiles could add on the headline batch (lock step 1.125) — by the true counts, by the
predictor, and by the residuals after k real ADMM iterations of every instance (which no launch has for free)."""
import sys, heapq, numpy as np
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[2]))
import accelerated_tinympc_amd as T
from oracle import oracle as O
pr = T.problems
prob = pr.quadrotor(20, 30); N = 30
xmn, xmx, umn, umx = pr.bounds_arrays(prob)
A, Bm, K = (prob[k].astype(np.float64) for k in ("Adyn", "Bdyn", "Kinf"))
FIX, SLOTS = 3.5, 1024


def workload(B, seed):
    x0, table, start = pr.tracking_batch(B, N, seed=seed)
    st = O.new_state(B, 12, 4, N); st["x"][:, 0] = x0
    O.Oracle(prob, np.float32, dict(O.DEFAULT_SETTINGS, max_iter=100)).solve(st, xmn, xmx, umn, umx, pr.expand_windows(table, start, N), nthreads=8)
    x = x0.astype(np.float64); key = np.zeros(B)
    for i in range(4):  # dispatch_order.hip's predictor: largest primal residual of the LQR rollout over four steps
        u = -(x @ K.T)
        key = np.maximum(key, np.max(np.abs(x - np.clip(x, -5, 5)), axis=1)); key = np.maximum(key, np.max(np.abs(u - np.clip(u, -0.5, 0.5)), axis=1))
        x = x @ A.T + u @ Bm.T
    return st["iter"].astype(np.int64), key


def makespan(tiles_in_order, stride):
    """tiles_in_order: lock-step iteration counts in queue order (predicted longest first); every stride-th slot claims from the short end"""
    n, h, t, mk = len(tiles_in_order), 0, 0, 0.0
    slots = [(0.0, s) for s in range(SLOTS)]; heapq.heapify(slots)
    while h + t < n:
        e, s = heapq.heappop(slots)
        if stride and s % stride == 0: j = n - 1 - t; t += 1
        else: j = h; h += 1
        e += FIX + tiles_in_order[j]; mk = max(mk, e); heapq.heappush(slots, (e, s))
    return mk


if sys.argv[1:] == ["regroup"]:
    B = 65536
    x0, table, start = pr.tracking_batch(B, N)
    xr = pr.expand_windows(table, start, N)
    def run(max_iter):
        st = O.new_state(B, 12, 4, N); st["x"][:, 0] = x0
        O.Oracle(prob, np.float32, dict(O.DEFAULT_SETTINGS, max_iter=max_iter)).solve(st, xmn, xmx, umn, umx, xr, nthreads=8)
        return st
    it = run(100)["iter"].astype(np.int64)
    x = x0.astype(np.float64); key = np.zeros(B)
    for i in range(4):
        u = -(x @ K.T)
        key = np.maximum(key, np.max(np.abs(x - np.clip(x, -5, 5)), axis=1)); key = np.maximum(key, np.max(np.abs(u - np.clip(u, -0.5, 0.5)), axis=1))
        x = x @ A.T + u @ Bm.T
    def report(name, k):
        perm = np.argsort(-k, kind="stable")                     # tiles of instances adjacent in the key's order ...
        tiles = it[perm].reshape(-1, 16).max(1)
        q_own = tiles[np.argsort(-k[perm].reshape(-1, 16).max(1), kind="stable")]    # ... queued by that key
        q_pred = tiles[np.argsort(-key[perm].reshape(-1, 16).max(1), kind="stable")]  # ... queued by the library's predictor (what a caller's batch order gets)
        print(f"{name:60s} corr {np.corrcoef(k, it)[0, 1]:6.3f}  lock step {tiles.mean() / it.mean():.3f}  queued by the key itself: one counter {makespan(q_own, 0):6.1f} stride 4 {makespan(q_own, 4):6.1f}"
              f"   by the predictor: {makespan(q_pred, 0):6.1f} / {makespan(q_pred, 4):6.1f}", flush=True)
    report("index order (the batch as it is)", -np.arange(B, dtype=np.float64))
    report("TRUE counts", it.astype(np.float64))
    report("predictor (4 steps of the LQR rollout)", key)
    report("window start of the reference (what a caller could sort by)", -start.astype(np.float64))
    for k in (2, 4, 6, 10):
        r = run(k)["residuals"].astype(np.float64)
        report(f"largest residual after {k} iterations", r.max(axis=1))
    sys.exit(0)
cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(65536, 20241024), (65536, 7), (40960, 3), (49152, 4), (81920, 9), (98304, 5), (131072, 6)]
for B, seed in cases:
    it, key = workload(B, seed)
    tile_true, tile_key = it.reshape(-1, 16).max(1), key.reshape(-1, 16).max(1)
    for name, k in (("predictor's order", tile_key), ("order by TRUE counts", tile_true)):
        tiles = tile_true[np.argsort(-k, kind="stable")]
        print(f"B={B:7d} seed {seed}: {len(tiles) / SLOTS:4.1f} tiles per slot, work per slot {(tiles.sum() + FIX * len(tiles)) / SLOTS:6.1f}  {name:22s} "
              + "  ".join(f"stride {s}: {makespan(tiles, s):6.1f}" for s in (0, 16, 8, 4, 2)), flush=True)
