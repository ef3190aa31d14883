"""Kernel time of the headline workload (65 536 tracking instances, longest-first dispatch) on tile16, cold start and warm (second solve, duals kept)."""
import sys, numpy as np
sys.path.insert(0, '.')
import accelerated_tinympc_amd as T
pr = T.problems
prob = pr.quadrotor(20, 30); B = 65536
x0, table, start = pr.tracking_batch(B, 30)
for exact in (True, False):
    sol = T.TinyBatchSolver(prob, B); sol.set_dispatch(1); sol.select_kernel(2 if exact else 3)
    sol.set_bounds(*pr.bounds_arrays(prob)); sol.set_xref_window(table, start); sol.enable_timing(True)
    cold, warm = [], []
    for r in range(9):
        sol.reset_workspace(); sol.set_x0(x0); sol.solve_async(); sol.synchronize()
        if r >= 2: cold.append(sol.last_solve_ms())
        sol.set_x0(x0 * 1.01); sol.solve_async(); sol.synchronize()      # warm start from the solved state (duals kept): the non-COLD instantiation
        if r >= 2: warm.append(sol.last_solve_ms())
    sol.mpc_run_async(2, 1); sol.wait()
    sol.set_row_kernel(5)
    ms = []
    import time
    for r in range(4):
        sol.reset_workspace(); sol.set_x0(x0); sol.set_xref_window(table, start); sol.synchronize()
        t0 = time.perf_counter(); sol.mpc_run_async(20, 1); sol.wait(); ms.append((time.perf_counter() - t0) * 1e3 / 20)
    print(f"{sys.argv[1]:28s} {sol.kernel_name():24s} cold {np.median(cold):.4f} ms (min {min(cold):.4f})  warm {np.median(warm):.4f} ms   on-chip closed loop ({sol.closed_loop_kernel_name()}) {np.median(ms):.4f} ms per MPC step", flush=True)
    sol.close()
