"""Kernel time of the headline workload with per-instance tables, sixteen-instances-per-wave kernel against the 16-lane kernel.
   python tools/pi_time.py [dispatch 0|1]"""
import sys, numpy as np
sys.path.insert(0, '.')
import accelerated_tinympc_amd as T
pr = T.problems
prob = pr.quadrotor(20, 30); B = 65536
x0, table, start = pr.tracking_batch(B, 30)
shared = pr.bounds_arrays(prob)
disp = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rng = np.random.default_rng(0)
for mode in ("shared", "bounds_const", "bounds_steps", "xref_steps", "xref_const", "both_steps", "bounds_const_xref_steps"):
    for fam in (0, 1):
        sol = T.TinyBatchSolver(prob, B)
        sol.set_dispatch(disp)
        if mode in ("bounds_const", "bounds_const_xref_steps"):
            sol.set_bounds(*tuple(np.broadcast_to(a, (B,) + a.shape).copy() for a in shared))
        elif mode in ("bounds_steps", "both_steps"):
            bb = [np.broadcast_to(a, (B,) + a.shape).copy() for a in shared]
            bb[1][:, 5, 0] *= 1.0000001  # one entry differs along the horizon: the table goes through the ring (same iterates to fp32 noise)
            sol.set_bounds(*bb)
        else:
            sol.set_bounds(*shared)
        if mode in ("xref_steps", "both_steps", "bounds_const_xref_steps"):
            sol.set_xref(pr.expand_windows(table, start, 30))
        elif mode == "xref_const":
            sol.set_xref(np.repeat(table[start][:, None, :], 30, axis=1).copy())
        else:
            sol.set_xref_window(table, start)
        sol.set_row_kernel(fam)
        sol.enable_timing(True); ms = []
        for r in range(7):
            sol.reset_workspace(); sol.set_x0(x0); sol.solve_async(); sol.synchronize()
            if r >= 2: ms.append(sol.last_solve_ms())
        print(f"{mode:26s} {sol.kernel_name():28s} {np.median(ms):.3f} ms  mean iterations {sol.get_status()[0].mean():.1f}", flush=True)
        sol.close()
