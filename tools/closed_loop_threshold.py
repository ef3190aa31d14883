"""On-chip closed loop (tiny_batch_mpc_run_async, 20 steps, warm-started tracking): sixteen-instances-per-wave kernel against the 16-lane kernel over the batch size.
Timed: the SECOND run of 20 steps after a settling one (the bench's closed_loop leg) — its tiles / groups are dispatched by the counts of the solve before it."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import accelerated_tinympc_amd as T
pr = T.problems
prob = pr.quadrotor(20, 30)
for B in (8192, 16384, 24576, 32768, 40960, 49152, 65536, 98304, 131072):
    x0, table, start = pr.tracking_batch(B, 30)
    out = []
    for fam in (5, 1):
        sol = T.TinyBatchSolver(prob, B); sol.set_row_kernel(fam)
        sol.set_bounds(*pr.bounds_arrays(prob)); sol.set_xref_window(table, start); sol.set_x0(x0)
        sol.mpc_run_async(20, 1); sol.wait()
        ms = []
        for r in range(4):
            sol.reset_workspace(); sol.set_x0(x0); sol.set_xref_window(table, start); sol.mpc_run_async(20, 1); sol.synchronize()
            t0 = time.perf_counter(); sol.mpc_run_async(20, 1); sol.wait(); ms.append((time.perf_counter() - t0) * 1e3 / 20)
        out.append(f"{sol.closed_loop_kernel_name().split('<')[0]} {np.median(ms):.4f}")
        sol.close()
    print(B, " | ".join(out), flush=True)
