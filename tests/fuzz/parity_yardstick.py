"""Developer aid: how far is the HIP result from the fp32 reference, measured against how far the reference's
own fp64 build is from its fp32 build (the algorithm's intrinsic rounding sensitivity)."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import accelerated_tinympc_amd as T
from oracle import oracle as O

pr = T.problems
prob = pr.quadrotor(20, 30)
B = 1024
for config in ("hover", "tracking"):
    for settings in (dict(O.DEFAULT_SETTINGS), dict(O.DEFAULT_SETTINGS, abs_pri_tol=0, abs_dua_tol=0, max_iter=10),
                     dict(O.DEFAULT_SETTINGS, abs_pri_tol=0, abs_dua_tol=0, max_iter=30)):
        if config == "hover":
            x0, xr = pr.hover_batch(B, 30)
        else:
            x0, table, start = pr.tracking_batch(B, 30)
            xr = pr.expand_windows(table, start, 30)
        outs = {}
        for dt in (np.float32, np.float64):
            xmn, xmx, umn, umx = pr.bounds_arrays(prob, dt)
            st = O.new_state(B, 12, 4, 30, dt); st["x"][:, 0] = x0
            O.Oracle(prob, dt, settings).solve(st, xmn, xmx, umn, umx, xr.astype(dt), nthreads=8)
            outs[dt] = st
        sol = T.TinyBatchSolver(prob, B, settings=settings)
        sol.set_bounds(*pr.bounds_arrays(prob)); sol.set_xref(xr); sol.set_x0(x0); sol.solve()
        g = sol.get_state(); sol.close()
        r32, r64 = outs[np.float32], outs[np.float64]
        print(f"== {config} max_iter={settings['max_iter']} tol={settings['abs_pri_tol']}: mean iters ref32 {r32['iter'].mean():.2f} gpu {g['iter'].mean():.2f} ref64 {r64['iter'].mean():.2f}")
        fg, f64 = g["iter"] != r32["iter"], r64["iter"] != r32["iter"]
        print(f"   iter flips vs ref32: gpu {fg.mean():.3f} (mean|d| {np.abs(g['iter']-r32['iter']).mean():.2f} max {np.abs(g['iter']-r32['iter']).max()})   ref64 {f64.mean():.3f} (mean|d| {np.abs(r64['iter']-r32['iter']).mean():.2f} max {np.abs(r64['iter']-r32['iter']).max()})")
        for k in ("u", "x", "d", "y", "g", "p"):
            fl = 0.5 if k in ("u",) else max(1.0, float(np.abs(r32[k]).max())) if k in ("p", "g", "y", "d") else 1.0
            def e(a, m):
                if not (~m).any(): return np.array([0.0])
                aa = a[k][~m].astype(np.float64).reshape((~m).sum(), -1); bb = r32[k][~m].astype(np.float64).reshape((~m).sum(), -1)
                return np.max(np.abs(aa - bb), axis=1) / fl
            eg, e6 = e(g, fg), e(r64, f64)
            print(f"   {k}: gpu-vs-ref32 max {eg.max():.2e} med {np.median(eg):.2e} | ref64-vs-ref32 max {e6.max():.2e} med {np.median(e6):.2e}")
        # first control only
        eg = np.max(np.abs(g["u"][:, 0].astype(np.float64) - r32["u"][:, 0]), axis=1) / 0.5
        e6 = np.max(np.abs(r64["u"][:, 0] - r32["u"][:, 0]), axis=1) / 0.5
        print(f"   u.col(0) ALL instances: gpu max {eg.max():.2e} med {np.median(eg):.2e} | ref64 max {e6.max():.2e} med {np.median(e6):.2e}")
