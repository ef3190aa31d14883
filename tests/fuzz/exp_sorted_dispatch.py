"""Developer experiment (run on an MI355X): the headline launch with the batch permuted on the host — by the TRUE iteration counts
(the ceiling of any regrouping / refill scheme) and by what a predictor can see (residuals after 1-3 iterations).
      python tests/fuzz/exp_sorted_dispatch.py"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import accelerated_tinympc_amd as T
pr = T.problems
prob = pr.quadrotor(20, 30); B = 65536
x0, table, start = pr.tracking_batch(B, 30)
def run(x0, start, dispatch, label, kernel=2):
    s = T.TinyBatchSolver(prob, B); s.select_kernel(kernel)
    s.set_bounds(*pr.bounds_arrays(prob)); s.set_xref_window(table, start); s.set_dispatch(dispatch); s.enable_timing(True)
    ms = []
    for r in range(7):
        s.reset_workspace(); s.set_x0(x0); s.solve_async(); s.synchronize()
        if r >= 2: ms.append(s.last_solve_ms())
    it = s.get_status()[0]; s.close()
    print(f"{label:70s} {np.median(ms):.3f} ms (min {np.min(ms):.3f}), mean iters {it.mean():.2f}", flush=True)
    return it
for kernel in (2, 3):
    it = run(x0, start, 0, f"kernel {kernel}: index order", kernel)
    run(x0, start, 1, f"kernel {kernel}: longest first by the predictor (bench default)", kernel)
    o = np.argsort(-it, kind="stable")
    run(x0[o], start[o], 0, f"kernel {kernel}: instances sorted by their TRUE iteration count (no lock-step loss, longest first)", kernel)
# instance-level sort by what a predictor can see: residuals after 1, 2, 3 iterations
for nit in (1, 2, 3):
    s = T.TinyBatchSolver(prob, B, settings=dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=nit, check_termination=1, en_state_bound=1, en_input_bound=1))
    s.set_bounds(*pr.bounds_arrays(prob)); s.set_xref_window(table, start); s.set_x0(x0); s.solve()
    res = s.get_status()[2]; s.close()
    for nm, k in (("max primal", np.maximum(res[:, 0], res[:, 1])), ("max of all four", res.max(1)), ("sum", res.sum(1))):
        o = np.argsort(-k, kind="stable")
        run(x0[o], start[o], 0, f"kernel 2: instances sorted by {nm} residual after {nit} iteration(s)", 2)
