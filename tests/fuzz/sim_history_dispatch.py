"""Test-infrastructure study (drives the oracle): dispatch order of WARM-STARTED launches (tiny_batch_set_dispatch(2), dispatch_order.hip).

The predictor of cold-start launches (one sweep from the workspace) is blind on a warm workspace; the instance's own past is not.  This script runs
the warm-started tracking loop of the bench (65 536 instances, 40 MPC steps) on the ORACLE, records every instance's iteration count at every step,
and replays the launches — units of 16 instances in lock step on 1 024 wave slots (admm_tile16.hip) and units of 4 on 2 048 (the 16-lane kernels),
list scheduling in dispatch order — for: index order, order by the previous step's largest count per unit (what the library does), by its sum, by
the true counts; for single-step launches and for on-chip runs of 20 steps (a unit's job is then its total over the run; key: the step before the run,
or the previous run's total, which only the kernel could record).  DESIGN.md section 5.4 quotes its output; tools/warm_dispatch_ab.py is the
measurement on the chip.       python tests/fuzz/sim_history_dispatch.py [batch]"""
import sys, heapq, numpy as np
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[2]))
import accelerated_tinympc_amd as T
from oracle import oracle as O
pr = T.problems
prob = pr.quadrotor(20, 30); N = 30
B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 65536
x0, table, start = pr.tracking_batch(B, N)
bnds = pr.bounds_arrays(prob)
orc = O.Oracle(prob, np.float32, O.DEFAULT_SETTINGS)
st = O.new_state(B, 12, 4, N); x = x0.copy(); start = start.copy()
counts = []
for k in range(40):   # quadrotor_tracking.cpp:93-118: x0, duals reset, window slid, solve, plant step
    st["x"][:, 0] = x; st["y"][:] = 0; st["g"][:] = 0
    idx = np.minimum(start[:, None] + np.arange(N)[None], table.shape[0] - 1)
    orc.solve(st, *bnds, np.ascontiguousarray(table[idx]), nthreads=8)
    counts.append(st["iter"].astype(np.int64))
    x = orc.plant_step(x, st["u"][:, 0]); start = start + 1
counts = np.array(counts, dtype=np.float64)   # [step][instance]
if "--save" in sys.argv: np.save('/tmp/cl_iters.npy', counts.astype(np.int64))   # for tests/fuzz/sim_runahead.py
print(f"mean iterations per solve, steps 20-39: {counts[20:].mean():.2f}")


def makespan(jobs, slots, fixed):
    h = [0.0] * slots; heapq.heapify(h); m = 0.0
    for j in jobs:
        e = heapq.heappop(h) + j + fixed; m = max(m, e); heapq.heappush(h, e)
    return m


for unit, slots, name in ((16, 1024, "16 instances per wave, 1 024 slots"), (4, 2048, "4 instances per wave, 2 048 slots")):
    r = counts.reshape(40, -1, unit)
    tm, ts = r.max(2), r.sum(2)
    print(f"{name}: lock step {tm[20:].sum() * unit / counts[20:].sum():.3f}")
    keys = (("index order", None), ("previous step, largest count", lambda k: tm[k - 1]), ("previous step, sum", lambda k: ts[k - 1]), ("TRUE counts", lambda k: tm[k]))
    print("  single warm-started step (makespan in iterations, mean over steps 20-39; work per slot "
          f"{np.mean([(tm[k].sum() + len(tm[k])) / slots for k in range(20, 40)]):.1f}):")
    for kn, kf in keys:
        v = [makespan(tm[k] if kf is None else tm[k][np.argsort(-kf(k), kind='stable')], slots, 1.0) for k in range(20, 40)]
        print(f"    {kn:32s} {np.mean(v):7.1f}")
    run = tm[20:].sum(0)
    print(f"  on-chip run of 20 steps (a unit's job: {run.mean():.0f} iterations on average, {run.min():.0f} ... {run.max():.0f}; work per slot {run.sum() / slots:.0f}):")
    for kn, key in (("index order", None), ("step before the run, largest count", tm[19]), ("step before the run, sum", ts[19]), ("previous run's total", tm[:20].sum(0)), ("TRUE totals", run)):
        print(f"    {kn:36s} {makespan(run if key is None else run[np.argsort(-key, kind='stable')], slots, 0.0):7.0f}")
