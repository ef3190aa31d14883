import sys, numpy as np
sys.path.insert(0, '.')
import accelerated_tinympc_amd as T
pr = T.problems
prob = pr.random_system(32, 16, 50, seed=1234)
fams = [int(a) for a in sys.argv[1:]] or [7, 8]
for variant in (2, 3):
  for B in (2048, 16384):
    x0, xr = pr.random_batch(B, 32, 50)
    for fam in fams:
      for mi in (20, 40):
        sol = T.TinyBatchSolver(prob, B, settings=dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=mi, check_termination=1, en_state_bound=1, en_input_bound=1))
        sol.select_kernel(variant); sol.set_row_kernel(fam); sol.set_bounds(*pr.bounds_arrays(prob)); sol.set_xref(xr); sol.enable_timing(True); ms = []
        for r in range(3):
            sol.reset_workspace(); sol.set_x0(x0); sol.solve_async(); sol.synchronize()
            if r: ms.append(sol.last_solve_ms())
        print(sol.kernel_name(), B, mi, f"{np.mean(ms):.3f} ms", flush=True); sol.close()
  # the workload of the bench configuration (early exit)
  for B in (2048, 16384):
    x0, xr = pr.random_batch(B, 32, 50)
    for fam in fams:
        sol = T.TinyBatchSolver(prob, B, settings=dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1, en_state_bound=1, en_input_bound=1))
        sol.select_kernel(variant); sol.set_row_kernel(fam); sol.set_bounds(*pr.bounds_arrays(prob)); sol.set_xref(xr); sol.enable_timing(True); ms = []
        for r in range(3):
            sol.reset_workspace(); sol.set_x0(x0); sol.solve_async(); sol.synchronize()
            if r: ms.append(sol.last_solve_ms())
        print(sol.kernel_name(), B, "early exit", f"{np.mean(ms):.3f} ms", "mean iters", sol.get_status()[0].mean(), flush=True); sol.close()
