#!/usr/bin/env python3
"""Developer aid (drives the oracle: test infrastructure, CPU only): replay the iteration counts of the headline workload
(65 536 tracking instances) through an in-order wave dispatcher to price lock-step loss (a wave runs as long as the slowest
of its G instances) and the launch tail, for index order, longest-first by true counts, and orders / regroupings derived
from the residuals after 1-3 iterations (what the on-device predictor sees).   python tests/fuzz/sim_grouping.py [G] [slots]
G = 4, slots = 2048 is the 16-lane kernel at two waves per SIMD; G = 16, slots = 1024 the 16-instances-per-wave kernel."""
import sys, heapq
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import accelerated_tinympc_amd as T
from oracle import oracle as O

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
pr = T.problems; prob = pr.quadrotor(20, 30); B = 65536
x0, table, start = pr.tracking_batch(B, 30)
xr = pr.expand_windows(table, start, 30)


def solve(**kw):
    st = O.new_state(B, 12, 4, 30); st['x'][:, 0] = x0
    O.Oracle(prob, np.float32, dict(O.DEFAULT_SETTINGS, **kw)).solve(st, *pr.bounds_arrays(prob), xr, nthreads=8)
    return st


it = solve()['iter']
ideal = it.sum() / (slots * G)


def makespan(work):
    h = [0] * slots; heapq.heapify(h)
    for w in work:
        heapq.heappush(h, heapq.heappop(h) + w)
    return max(h)


def rep(name, w):
    print(f'{name:46s} lock-step {w.sum() * G / it.sum():.3f}   makespan / ideal {makespan(w) / ideal:.3f}')


w = it.reshape(-1, G).max(1)
rep('index order', w)
rep('groups longest first, true counts', np.sort(w)[::-1])
rep('instances sorted by true count', it[np.argsort(-it, kind='stable')].reshape(-1, G).max(1))
for nit in (1, 2, 3):
    r = solve(max_iter=nit, abs_pri_tol=0, abs_dua_tol=0)['residuals']
    for nm, k in (('primal', np.maximum(r[:, 0], r[:, 1])), ('all four', r.max(1))):
        gk = k.reshape(-1, G).max(1)
        rep(f'groups by max {nm} residual after {nit}', w[np.argsort(-gk, kind="stable")])
        rep(f'instances by max {nm} residual after {nit}', it[np.argsort(-k, kind="stable")].reshape(-1, G).max(1))
