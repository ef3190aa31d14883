"""Developer aid: kernel-time throughput of every BASELINE.json configuration on one GPU (not the headline bench)."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import accelerated_tinympc_amd as T

pr = T.problems


def measure(name, prob, B, setup, variants=(0,), settings=None, reps=5):
    for v in variants:
        sol = T.TinyBatchSolver(prob, B, settings=settings)
        try:
            sol.select_kernel(v)
        except T.TinyBatchError as e:
            print(f"{name:34s} variant {v}: unsupported ({e})"); sol.close(); continue
        sol.set_bounds(*pr.bounds_arrays(prob))
        x0 = setup(sol)
        sol.enable_timing(True)
        ms = []
        for r in range(reps + 1):
            sol.reset_workspace(); sol.set_x0(x0); sol.solve_async(); sol.synchronize()
            if r: ms.append(sol.last_solve_ms())
        it, st, _ = sol.get_status()
        print(f"{name:34s} {sol.kernel_name():28s} B={B:6d}  {np.mean(ms):8.3f} ms  {B / np.mean(ms) * 1e3:12.4g} solves/s  "
              f"mean iters {it.mean():6.2f}  converged {np.mean(st == 1):.3f}")
        sol.close()


def hover(B, N=30):
    x0, xref = pr.hover_batch(B, N)
    def f(sol): sol.set_xref(xref); return x0
    return f


def track(B, N=30):
    x0, table, start = pr.tracking_batch(B, N)
    def f(sol): sol.set_xref_window(table, start); return x0
    return f


def rand(B, prob):
    rng = np.random.default_rng(0)
    x0 = rng.uniform(-1, 1, size=(B, prob["nx"])).astype(np.float32)
    def f(sol): sol.set_xref(np.zeros((prob["N"], prob["nx"]), np.float32)); return x0
    return f


q30 = pr.quadrotor(20, 30)
measure("cfg1 hover 1 instance", q30, 1, hover(1), (0, 3, 1))
measure("cfg2 hover 4096", q30, 4096, hover(4096), (0, 3, 1))
measure("cfg3 tracking 65536", q30, 65536, track(65536), (0, 3, 1))
r32 = pr.random_system(32, 16, 50)
measure("cfg4 random 32/16/50 2048 (1 of 8 GPUs)", r32, 2048, rand(2048, r32), (0,))
measure("cfg4 random 32/16/50 16384", r32, 16384, rand(16384, r32), (0,))
cp = pr.cartpole(10)
rngc = np.random.default_rng(1)
def cps(sol):
    sol.set_xref(np.zeros((10, 4), np.float32)); return np.tile(np.array([[0, 0, 0.1, 0]], np.float32), (sol.B, 1)) + rngc.uniform(-0.05, 0.05, size=(sol.B, 4)).astype(np.float32)
measure("cfg5 cartpole 32768", cp, 32768, cps, (0, 3, 1), settings=dict(max_iter=150))
measure("cfg5 quadrotor hover 32768", q30, 32768, hover(32768), (0, 3))
q17 = pr.quadrotor(20, 17)
measure("any-N quadrotor N=17 65536", q17, 65536, track(65536, 17), (0, 3, 1))
fixed = dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=10)
measure("cfg3 tracking fixed 10 iters", q30, 65536, track(65536), (0, 3, 1), settings=fixed)


# ---- cfg5 as BASELINE.json states it: mixed cartpole + quadrotor batch in ONE call, fp16 storage / fp32 arithmetic ----
def group(storage, variant, reps=5, dual_bits=None):
    """32768 cartpole + 32768 quadrotor (tracking) instances solved by tiny_batch_group_solve; wall time of the group."""
    import ctypes
    sols, x0s = [], []
    for prob, setup, st in ((cp, cps, dict(max_iter=150)), (q30, track(32768), None)):
        sol = T.TinyBatchSolver(prob, 32768, settings=st)
        sol.select_kernel(variant); sol.set_storage(storage, dual_bits)
        sol.set_bounds(*pr.bounds_arrays(prob))
        x0s.append(setup(sol)); sols.append(sol)
    ts = []
    for r in range(reps + 1):
        for sol, x0 in zip(sols, x0s):
            sol.reset_workspace(); sol.set_x0(x0)
        for sol in sols: sol.synchronize()
        t0 = time.perf_counter(); T.solve_group(sols); t1 = time.perf_counter()
        if r: ts.append((t1 - t0) * 1e3)
    out = []
    for sol in sols:
        it, stt, _ = sol.get_status()
        out.append((sol.kernel_name(), it.copy(), float(np.mean(stt == 1)), sol.get_u()[:, 0].copy(), stt.copy()))
        sol.close()
    ms = float(np.median(ts))
    dnote = "" if dual_bits in (None, storage) else f" (duals fp{dual_bits})"
    print(f"cfg5 mixed 32768 cartpole + 32768 quadrotor, storage fp{storage}{dnote}, variant {variant}: {ms:7.3f} ms wall per group "
          f"-> {65536 / ms * 1e3:.4g} solves/s; " + "; ".join(f"{k}: mean iters {i.mean():.2f}, converged {c:.3f}" for k, i, c, _, _ in out))
    return out


for variant in (2, 3):
    ref = group(32, variant)
    h16 = group(16, variant, dual_bits=16)
    h16d = group(16, variant, dual_bits=32)
    for (k32, i32, c32, u32, s32), (k16, i16, c16, u16, s16), (kd, idd, cd, ud, sd) in zip(ref, h16, h16d):
        same = i32 == i16
        scale = max(np.abs(u32).max(), 1e-6)
        du = np.abs(u16[same] - u32[same]).max() / scale if same.any() else float("nan")
        print(f"   fp16-storage drift of {k16}: iteration count differs for {np.mean(~same):.3f} of the instances "
              f"(mean {i16.mean() - i32.mean():+.2f}); u0 of the others within {du:.2e} of the input scale")
        for tag, u, st_ in (("16-bit duals", u16, s16), ("fp32 duals", ud, sd)):
            bad = (st_ != 1) & (s32 == 1)
            e = np.abs(u[bad] - u32[bad]).max() / scale if bad.any() else 0.0
            print(f"   {tag}: {np.mean(bad):.3f} of the instances stop at max_iter (fp32 converges them); their u0 is off by up to {e:.2e} of the input scale")


# ---- where fp16 storage pays: the state is STREAMED through L2/HBM every iteration (horizons beyond the register-resident kernels) ----
def streamed(N, storage):
    q = pr.quadrotor(20, N)
    B = 65536
    x0, table, start = pr.tracking_batch(B, N)
    sol = T.TinyBatchSolver(q, B)
    sol.select_kernel(2)
    sol.set_storage(storage, storage)
    sol.set_bounds(*pr.bounds_arrays(q)); sol.set_xref_window(table, start)
    sol.enable_timing(True); ms = []
    for r in range(4):
        sol.reset_workspace(); sol.set_x0(x0); sol.solve_async(); sol.synchronize()
        if r: ms.append(sol.last_solve_ms())
    it, st, _ = sol.get_status()
    print(f"quadrotor N={N} x{B}, {sol.kernel_name():30s} storage fp{storage}: {np.mean(ms):8.3f} ms  {B / np.mean(ms) * 1e3:10.4g} solves/s  "
          f"mean iters {it.mean():.2f} converged {np.mean(st == 1):.3f}  workspace {6 * N * 16 * (storage // 8) * B / 1e6:.0f} MB "
          f"({6 * N * 16 * (storage // 8)} B per instance: {288e9 / (6 * N * 16 * (storage // 8)) / 1e6:.1f} M instances per 288 GB)")
    sol.close()


for N in (80, 100):
    for storage in (32, 16):
        streamed(N, storage)
