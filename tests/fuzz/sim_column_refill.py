"""Test-infrastructure study (drives the oracle's true iteration counts), round 4: would COLUMN REFILL — a finished column of a 16-instance tile takes the
next instance from the queue, finished columns being written out in batches of F — shorten the headline launch?  Event-driven replay on 1 024 wave slots.
Result: no.  A flush stalls all sixteen columns of its tile; at the epilogue's real cost (3.5 iterations' worth) every policy is
slower than the launch as it is (133), and even a flush three times cheaper gains 7 % at most (F = 6, instance-level order) before any of the
bookkeeping a refill needs inside the iteration loop is paid.     python tests/fuzz/sim_column_refill.py   (needs /tmp/its.npy from sim_group_tiles.py or recomputes)"""
import sys, heapq, numpy as np
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[2]))
import accelerated_tinympc_amd as T
from oracle import oracle as O
pr = T.problems
prob = pr.quadrotor(20, 30); N = 30; B = 65536
x0, table, start = pr.tracking_batch(B, N)
st = O.new_state(B, 12, 4, N); st['x'][:, 0] = x0
O.Oracle(prob, np.float32, dict(O.DEFAULT_SETTINGS, max_iter=100)).solve(st, *pr.bounds_arrays(prob), pr.expand_windows(table, start, N), nthreads=8)
it = st['iter'].astype(np.int64)
A, Bm, K = prob["Adyn"].astype(np.float64), prob["Bdyn"].astype(np.float64), prob["Kinf"].astype(np.float64)
x = x0.astype(np.float64); key = np.zeros(B)
for i in range(4):
    u = -(x @ K.T)
    key = np.maximum(key, np.max(np.abs(x - np.clip(x, -5, 5)), axis=1)); key = np.maximum(key, np.max(np.abs(u - np.clip(u, -0.5, 0.5)), axis=1))
    x = x @ A.T + u @ Bm.T
def simulate(order, F, c_flush=3.5, c_refill=0.15, slots=1024, cols=16, min_fill=1):
    # event-driven per slot; slots are independent except for the shared queue -> process slots in time order with a heap of (time, slot)
    q = list(order)  # instance ids in dispatch order
    qi = 0
    rem = np.zeros((slots, cols), dtype=np.int64)   # remaining iterations of active columns (0 = empty/done)
    done = np.zeros((slots, cols), dtype=bool)       # finished, waiting for a flush
    t = np.zeros(slots)
    heap = []
    for s in range(slots):
        n = min(cols, len(q) - qi)
        rem[s, :n] = it[q[qi:qi+n]]; qi += n
        if n: heapq.heappush(heap, (0.0, s))
    end = 0.0
    while heap:
        tt, s = heapq.heappop(heap)
        act = rem[s] > 0
        if not act.any():
            # everything done: final flush
            if done[s].any():
                tt += c_flush; done[s] = False
            # refill whole tile if queue non-empty
            n = min(cols, len(q) - qi)
            if n:
                rem[s, :n] = it[q[qi:qi+n]]; qi += n
                tt += c_refill
                heapq.heappush(heap, (tt, s))
            else:
                end = max(end, tt)
            continue
        # run until the next event: the next column finishing
        step = rem[s][act].min()
        rem[s][act] -= step
        tt += step
        newly = act & (rem[s] == 0)
        done[s] |= newly
        nd = done[s].sum()
        if nd >= F and qi < len(q):
            tt += c_flush
            idx = np.where(done[s])[0]
            n = min(len(idx), len(q) - qi)
            done[s] = False
            rem[s, idx[:n]] = it[q[qi:qi+n]]; qi += n
            tt += c_refill
        heapq.heappush(heap, (tt, s))
    return end
order_tile = np.argsort(-key.reshape(-1,16).max(axis=1), kind="stable")
order_inst_by_tile = (order_tile[:,None]*16 + np.arange(16)[None,:]).reshape(-1)
order_inst = np.argsort(-key, kind="stable")
print("baseline (no refill, F=16 => flush only when all done):", simulate(order_inst_by_tile, 17))
for F in (2, 4, 6, 8, 12):
    for name, o in (("tiles in predicted order", order_inst_by_tile), ("instances in predicted order", order_inst), ("index order", np.arange(B))):
        print(f"F={F:2d} {name:32s} makespan {simulate(o, F):7.1f}   (flush 3.5)   {simulate(o, F, c_flush=5.0):7.1f} (flush 5.0)")
print("---- cheaper flush (stores overlap other tiles' work; the launch's own fixed cost 2.5 is charged once per slot at the start) ----")
for cf in (0.75, 1.0, 1.5):
    for F in (1, 2, 4, 6, 8):
        print(f"flush {cf} F={F}: instances in predicted order {simulate(order_inst, F, c_flush=cf)+2.5:7.1f}   tiles in predicted order {simulate(order_inst_by_tile, F, c_flush=cf)+2.5:7.1f}")
