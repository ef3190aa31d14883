import sys, numpy as np
sys.path.insert(0, '.')
import accelerated_tinympc_amd as T
pr = T.problems
prob = pr.random_system(32, 16, 50, seed=1234)
for variant in (2, 3):
  for B in (2048, 2304, 2560, 3072, 4096, 4352, 6144):
    x0, xr = pr.random_batch(B, 32, 50)
    row = []
    for fam in (7, 8):
        sol = T.TinyBatchSolver(prob, B, settings=dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1, en_state_bound=1, en_input_bound=1))
        sol.select_kernel(variant); sol.set_row_kernel(fam); sol.set_bounds(*pr.bounds_arrays(prob)); sol.set_xref(xr); sol.enable_timing(True); ms = []
        for r in range(3):
            sol.reset_workspace(); sol.set_x0(x0); sol.solve_async(); sol.synchronize()
            if r: ms.append(sol.last_solve_ms())
        row.append((sol.kernel_name(), float(np.mean(ms)))); sol.close()
    print(B, " | ".join(f"{n} {t:.2f} ms" for n, t in row), flush=True)
