"""Cost of the two optional terms (Uref, coeff_d2p) on the register-resident 16-lane kernel (round 4) against the same solve without them and against
the streaming row kernel a flagged handle used to be routed to.   python tools/experiments/optional_terms_cost.py"""
import sys, numpy as np
sys.path.insert(0, '.')
import accelerated_tinympc_amd as T
pr = T.problems
prob = dict(pr.quadrotor(20, 30)); B = 65536
rng = np.random.default_rng(0)
prob["coeff_d2p"] = (rng.standard_normal((12, 4)) * 0.02).astype(np.float32)
x0, table, start = pr.tracking_batch(B, 30)
uref = rng.uniform(-0.1, 0.1, (B, 29, 4)).astype(np.float32)
def run(flags, family):
    sol = T.TinyBatchSolver(prob, B)
    sol.set_bounds(*pr.bounds_arrays(prob)); sol.set_xref_window(table, start)
    sol.set_row_kernel(family)
    if flags:
        sol.set_input_cost(prob["R"]); sol.set_coeff_d2p(prob["coeff_d2p"]); sol.set_uref(uref); sol.set_optional_terms(True, True)
    sol.enable_timing(True); ms = []
    for r in range(5):
        sol.reset_workspace(); sol.set_x0(x0); sol.solve_async(); sol.synchronize()
        if r >= 2: ms.append(sol.last_solve_ms())
    it = sol.get_status()[0].mean()
    print(f"{sol.kernel_name():32s} optional terms {'on ' if flags else 'off'}  {np.mean(ms):.3f} ms  mean iterations {it:.2f}", flush=True)
    sol.close()
run(False, 1); run(True, 1); run(True, 3)
