import sys, numpy as np
sys.path.insert(0, '.')
import accelerated_tinympc_amd as T
pr = T.problems
bad = 0
nx, nu, N = 32, 16, 50
prob = pr.random_system(nx, nu, N)
for B in (1, 3, 16, 17, 66, 259):
    rng = np.random.default_rng(B)
    x0 = rng.uniform(-1, 1, size=(B, nx)).astype(np.float32)
    xref = (rng.standard_normal((B, N, nx)) * 0.2).astype(np.float32)
    bnds = pr.bounds_arrays(prob)
    for settings in (dict(max_iter=40), dict(max_iter=25, check_termination=4), dict(max_iter=1), dict(max_iter=100, abs_pri_tol=5e-2, abs_dua_tol=5e-2),
                     dict(max_iter=12, en_state_bound=0, en_input_bound=0), dict(max_iter=0)):
        s = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1, en_state_bound=1, en_input_bound=1); s.update(settings)
        sols = []
        for fam in (7, 8):
            sol = T.TinyBatchSolver(prob, B, settings=s)
            sol.select_kernel(2); sol.set_row_kernel(fam)
            sol.set_bounds(*bnds); sol.set_xref(xref); sol.set_x0(x0)
            sols.append(sol)
        for k in range(3):
            sts = []
            for sol in sols:
                sol.reset_dual_variables(); rc = sol.solve(); st = sol.get_state(); st["rc"] = np.array([rc]); sts.append(st)
            for name in sts[0]:
                a, b = np.asarray(sts[0][name]), np.asarray(sts[1][name])
                if a.tobytes() != b.tobytes():
                    bad += 1
                    if bad < 40:
                        d = np.argwhere(a != b)
                        print("DIFF", B, settings, k, name, len(d), d[:4].tolist(), sols[1].kernel_name(), flush=True)
        for sol in sols: sol.close()
    print("B", B, "done, bad so far", bad, flush=True)
print("BAD", bad)
sys.exit(1 if bad else 0)
