"""Same-box A/B of the dispatch order of WARM-STARTED launches (tiny_batch_set_dispatch: 0 index order, 2 longest first by the previous solve's iteration
counts, -1 automatic) on the tracking loop: wall time per MPC step of (a) step-by-step tiny_batch_mpc_step_async (one solve launch per step) and (b) the
on-chip loop tiny_batch_mpc_run_async(20), on the 16-lane kernel and the 16-instances-per-wave kernel.   python tools/warm_dispatch_ab.py [batch ...]"""
import sys, time, numpy as np
sys.path.insert(0, '.')
import accelerated_tinympc_amd as T
pr = T.problems
prob = pr.quadrotor(20, 30)
for B in [int(a) for a in sys.argv[1:]] or [65536]:
    x0, table, start = pr.tracking_batch(B, 30)
    for fam, name in ((1, "rowlane"), (5, "tile16"), (0, "auto")):
        sol = T.TinyBatchSolver(prob, B); sol.select_kernel(2); sol.set_row_kernel(fam)
        sol.set_bounds(*pr.bounds_arrays(prob))
        out = []
        for mode in (0, 2, -1):
            sol.set_dispatch(mode)
            sol.reset_workspace(); sol.set_x0(x0); sol.set_xref_window(table, start)
            for _ in range(20): sol.mpc_step_async(1)          # settle the warm start
            sol.synchronize()
            t0 = time.perf_counter()
            for _ in range(20): sol.mpc_step_async(1)
            sol.synchronize(); step_ms = (time.perf_counter() - t0) * 1e3 / 20
            kn, da = sol.kernel_name(), sol.dispatch_applied()
            sol.reset_workspace(); sol.set_x0(x0); sol.set_xref_window(table, start)
            sol.mpc_run_async(20, 1); sol.synchronize()
            t0 = time.perf_counter(); sol.mpc_run_async(20, 1); sol.synchronize(); run_ms = (time.perf_counter() - t0) * 1e3 / 20
            out.append(f"dispatch {mode:2d}: step-by-step {step_ms:.4f} ms ({kn}, applied {da})  on-chip run {run_ms:.4f} ms ({sol.closed_loop_kernel_name()}, applied {sol.dispatch_applied()})")
        print(f"B={B} {name:8s} " + "\n                   ".join(out), flush=True)
        sol.close()
