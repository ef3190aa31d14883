"""Kernel time of tile16 against the 16-lane kernel over the batch size (tracking workload, longest-first dispatch where it applies): where should the automatic choice switch?"""
import sys, numpy as np
sys.path.insert(0, '.')
import accelerated_tinympc_amd as T
pr = T.problems
prob = pr.quadrotor(20, 30)
for B in (32768, 36864, 40960, 45056, 49152, 57344, 65536):
    x0, table, start = pr.tracking_batch(B, 30)
    out = []
    for fam in (5, 1):
        for disp in (0, 1):
            sol = T.TinyBatchSolver(prob, B); sol.set_dispatch(disp); sol.set_row_kernel(fam)
            sol.set_bounds(*pr.bounds_arrays(prob)); sol.set_xref_window(table, start); sol.enable_timing(True)
            ms = []
            for r in range(7):
                sol.reset_workspace(); sol.set_x0(x0); sol.solve_async(); sol.synchronize()
                if r >= 2: ms.append(sol.last_solve_ms())
            out.append(f"{sol.kernel_name().split('<')[0]}/{'sorted' if sol.dispatch_applied() == 1 else 'index'} {np.median(ms):.4f}")
            sol.close()
    print(B, " | ".join(out), flush=True)
