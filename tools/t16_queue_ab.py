"""Same-box A/B of tile16's tile queue (tiny_batch_set_tile_queue): kernel ms of a cold-start launch in predicted longest-first order for several batch
sizes and tail strides (0 = one counter), and a check that the results do not depend on it.   python tools/t16_queue_ab.py [strides] [batches]"""
import sys, numpy as np
sys.path.insert(0, '.')
import accelerated_tinympc_amd as T
pr = T.problems
prob = pr.quadrotor(20, 30)
strides = [int(s) for s in (sys.argv[1] if len(sys.argv) > 1 else "0,8,-1").split(",")]
batches = [int(s) for s in (sys.argv[2] if len(sys.argv) > 2 else "65536").split(",")]
for B in batches:
    x0, table, start = pr.tracking_batch(B, 30)
    sol = T.TinyBatchSolver(prob, B); sol.set_dispatch(1); sol.select_kernel(2); sol.set_row_kernel(5)
    sol.set_bounds(*pr.bounds_arrays(prob)); sol.set_xref_window(table, start); sol.enable_timing(True)
    ref = None; line = []
    for k in strides:
        sol.set_tile_queue(k)
        ms = []
        for r in range(9):
            sol.reset_workspace(); sol.set_x0(x0); sol.solve_async(); sol.synchronize()
            if r >= 2: ms.append(sol.last_solve_ms())
        st = sol.get_state()
        if ref is None: ref = st
        same = all(np.array_equal(st[a].view(np.uint32) if st[a].dtype == np.float32 else st[a], ref[a].view(np.uint32) if ref[a].dtype == np.float32 else ref[a]) for a in st)
        line.append(f"stride {k:3d}: {np.median(ms):.4f} ms (min {min(ms):.4f}){'' if same else '  RESULTS DIFFER'}")
    print(f"B={B:7d} {sol.kernel_name()} mean iters {ref['iter'].mean():.2f}  " + "   ".join(line), flush=True)
    sol.close()
