"""Launch-tail analysis of the headline workload (run on an MI355X):   python tools/launch_tail.py [batch]

Solves the bench workload once, takes the per-instance iteration counts the kernel reports, and replays them through a model
of the hardware dispatcher (2 048 wave slots = 256 CUs x 4 SIMDs x 2 waves, a free slot takes the next wave of 4 instances) in
several orders: as launched, longest first with the true counts (needs an oracle of the future), and longest first by the
largest residual after ONE iteration (a predictor that needs no history).  Unit = one ADMM iteration of one wave."""
import heapq
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import accelerated_tinympc_amd as T  # noqa: E402

pr = T.problems
N, B = 30, int(sys.argv[1]) if len(sys.argv) > 1 else 65536
SLOTS = 2048
prob = pr.quadrotor(20, N)
x0, table, start = pr.tracking_batch(B, N)
settings = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1, en_state_bound=1, en_input_bound=1)


def solve(max_iter):
    s = T.TinyBatchSolver(prob, B, settings=dict(settings, max_iter=max_iter))
    s.set_bounds(*pr.bounds_arrays(prob)); s.set_xref_window(table, start); s.set_x0(x0)
    s.solve()
    it, st, res = s.get_status()
    s.close()
    return it, res


def makespan(w):
    h = [0] * SLOTS
    heapq.heapify(h)
    for x in w:
        heapq.heappush(h, heapq.heappop(h) + int(x))
    return max(h)


it, _ = solve(100)
_, res1 = solve(1)
w = it.reshape(-1, 4).max(1)
key = res1.max(1).reshape(-1, 4).max(1)
print(f"iterations per instance: mean {it.mean():.2f}  p50 {np.percentile(it, 50):.0f}  p99 {np.percentile(it, 99):.0f}  max {it.max()}")
print(f"iterations per wave (max of 4 neighbours): mean {w.mean():.2f}  (+{(w.mean() / it.mean() - 1) * 100:.1f} % divergence)")
print(f"even slots (sum / {SLOTS}):                              {w.sum() / SLOTS:.1f}")
print(f"as launched (in order):                                {makespan(w)}")
print(f"longest first, true counts:                            {makespan(np.sort(w)[::-1])}")
print(f"longest first by the largest residual after 1 iteration: {makespan(w[np.argsort(-key)])}   (corr with the count {np.corrcoef(it, res1.max(1))[0, 1]:.2f})")
rng = np.random.default_rng(0)
print(f"random orders:                                         {[makespan(rng.permutation(w)) for _ in range(3)]}")
