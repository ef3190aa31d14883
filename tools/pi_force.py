import sys, numpy as np
sys.path.insert(0, '.')
import accelerated_tinympc_amd as T
pr = T.problems
prob = pr.quadrotor(20, 30); B = 65536
x0, table, start = pr.tracking_batch(B, 30)
sol = T.TinyBatchSolver(prob, B); sol.set_dispatch(1)
sol.set_bounds(*pr.bounds_arrays(prob))
if len(sys.argv) > 1 and sys.argv[1] == "shared_ref": sol.set_xref(table[:30].copy())
else: sol.set_xref_window(table, start)
sol.enable_timing(True); ms = []
for r in range(7):
    sol.reset_workspace(); sol.set_x0(x0); sol.solve_async(); sol.synchronize()
    if r >= 2: ms.append(sol.last_solve_ms())
print(f"{sol.kernel_name():28s} {np.median(ms):.3f} ms", flush=True)
