"""Developer aid: per-array error of the HIP path vs the oracle after 1..k iterations on a random warm state."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import accelerated_tinympc_amd as T
from oracle import oracle as O

pr = T.problems
which = sys.argv[1] if len(sys.argv) > 1 else "quad"
prob = {"quad": lambda: pr.quadrotor(20, 30), "cart": lambda: pr.cartpole(10), "r32": lambda: pr.random_system(32, 16, 50),
        "odd": lambda: pr.random_system(8, 3, 7, seed=99)}[which]()
nx, nu, N = prob["nx"], prob["nu"], prob["N"]
B = 20
rng = np.random.default_rng(0)
st0 = O.new_state(B, nx, nu, N)
for k in O.STATE_ORDER:
    st0[k][:] = (rng.standard_normal(st0[k].shape) * 0.1).astype(np.float32)
xref = (rng.standard_normal((B, N, nx)) * 0.2).astype(np.float32)
xmn, xmx, umn, umx = pr.bounds_arrays(prob)
for mi in (1, 2, 5):
    settings = dict(O.DEFAULT_SETTINGS, max_iter=mi, abs_pri_tol=0.0, abs_dua_tol=0.0)
    a = O.copy_state(st0)
    O.Oracle(prob, np.float32, settings).solve(a, xmn, xmx, umn, umx, xref)
    sol = T.TinyBatchSolver(prob, B, settings=settings)
    sol.set_bounds(xmn, xmx, umn, umx); sol.set_xref(xref); sol.set_state(st0)
    sol.solve()
    g = sol.get_state()
    print(f"--- max_iter={mi} kernel={sol.kernel_name()}")
    for k in O.STATE_ORDER + ("residuals", "iter", "status"):
        d = np.abs(g[k].astype(np.float64) - a[k])
        idx = np.unravel_index(np.argmax(d), d.shape)
        print(f"  {k:9s} max|diff| {d.max():.3e}  scale {np.abs(a[k]).max():.3g}  at {idx}")
    sol.close()
