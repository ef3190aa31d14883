import sys, numpy as np
sys.path.insert(0, '.')
import accelerated_tinympc_amd as T
pr = T.problems
prob = pr.random_system(32, 16, 50, seed=1234)
B = 4096; mi = 40
x0, xr = pr.random_batch(B, 32, 50)
sol = T.TinyBatchSolver(prob, B, settings=dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=mi, check_termination=1, en_state_bound=1, en_input_bound=1))
sol.select_kernel(2); sol.set_row_kernel(8); sol.set_bounds(*pr.bounds_arrays(prob)); sol.set_xref(xr); sol.enable_timing(True)
for r in range(2):
    sol.reset_workspace(); sol.set_x0(x0); sol.solve_async(); sol.synchronize()
print(sol.kernel_name(), sol.last_solve_ms(), "ms")
res = sol.get_status()[2][:2].reshape(-1)
steps = mi * 49
names = ["x fwd: S1+elementwise", "x fwd: wait B1", "x fwd: S2+put", "x fwd: wait B2", "x bwd: pn+put+store+lin", "x bwd: wait B3", "x bwd: fetch+tk", "x bwd: loads+M3 dot"]
for n, v in zip(names, res): print(f"{n:28s} {v/steps:8.0f} cycle-counter ticks per step")
print("sum per pair", sum(res[:8]) / steps)
