import sys, time, numpy as np
sys.path.insert(0, '.')
import accelerated_tinympc_amd as T
pr = T.problems; prob = pr.quadrotor(20, 30); B = 65536
x0, table, start = pr.tracking_batch(B, 30)
for fam in (5, 1):
    sol = T.TinyBatchSolver(prob, B); sol.select_kernel(2); sol.set_row_kernel(fam)
    sol.set_bounds(*pr.bounds_arrays(prob))
    for mode in (0, -1, 0, -1):
        sol.set_dispatch(mode)
        sol.reset_workspace(); sol.set_x0(x0); sol.set_xref_window(table, start); sol.synchronize()
        t0 = time.perf_counter(); sol.mpc_run_async(20, 1); sol.synchronize(); ms = (time.perf_counter() - t0) * 1e3 / 20
        print(f"{sol.closed_loop_kernel_name()} dispatch {mode:2d}: FIRST run of 20 steps from a reset workspace {ms:.4f} ms per MPC step (applied {sol.dispatch_applied()})", flush=True)
    sol.close()
