"""Should longest-first dispatch be the default?  Wall time per call of cold solves, warm mpc_step (one launch sequence per step) and mpc_run (on chip) with
tiny_batch_set_dispatch 0 / 1, automatic kernel choice, several batch sizes."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import accelerated_tinympc_amd as T
pr = T.problems
prob = pr.quadrotor(20, 30)
for B in (8192, 16384, 32768, 65536):
    x0, table, start = pr.tracking_batch(B, 30)
    for disp in (0, 1):
        sol = T.TinyBatchSolver(prob, B); sol.set_dispatch(disp)
        sol.set_bounds(*pr.bounds_arrays(prob)); sol.set_xref_window(table, start)
        def wall(fn, n):
            fn(); sol.synchronize()
            t0 = time.perf_counter()
            for _ in range(n): fn()
            sol.synchronize()
            return (time.perf_counter() - t0) * 1e3 / n
        def cold(): sol.reset_workspace(); sol.set_x0(x0); sol.solve_async()
        c = wall(cold, 10); kc = sol.kernel_name()
        sol.reset_workspace(); sol.set_x0(x0); sol.set_xref_window(table, start); sol.solve()
        s = wall(lambda: sol.mpc_step_async(1), 20)
        sol.reset_workspace(); sol.set_x0(x0); sol.set_xref_window(table, start)
        r = wall(lambda: sol.mpc_run_async(10, 1), 3) / 10
        print(f"B={B:6d} dispatch {disp}: cold solve {c:.4f} ms ({kc}, applied {sol.dispatch_applied()})  mpc_step {s:.4f} ms  mpc_run {r:.4f} ms per step ({sol.closed_loop_kernel_name()})", flush=True)
        sol.close()
