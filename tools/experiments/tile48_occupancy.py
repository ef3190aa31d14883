import sys, numpy as np
sys.path.insert(0, '.')
import accelerated_tinympc_amd as T
pr = T.problems
B = 16384; mi = 20
for variant in (2, 3):
  for N in (20, 23, 24, 25, 26, 30, 50):
    prob = pr.random_system(32, 16, N, seed=1234)
    x0, xr = pr.random_batch(B, 32, N)
    sol = T.TinyBatchSolver(prob, B, settings=dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=mi, check_termination=1, en_state_bound=1, en_input_bound=1))
    sol.select_kernel(variant); sol.set_row_kernel(8); sol.set_bounds(*pr.bounds_arrays(prob)); sol.set_xref(xr); sol.enable_timing(True); ms = []
    for r in range(3):
        sol.reset_workspace(); sol.set_x0(x0); sol.solve_async(); sol.synchronize()
        if r: ms.append(sol.last_solve_ms())
    t = np.mean(ms)
    print(sol.kernel_name(), "N", N, f"{t:.3f} ms", f"{t / mi / (N - 1) * 1e3:.2f} us per step pair of the batch", flush=True); sol.close()
