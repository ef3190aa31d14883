#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the COMPILED REFERENCE.

Runs only in the build container (needs /root/reference and `make -C oracle ref`).  Each
fixture is data only: the live-in set and the live-out set of tiny_solve() (all twelve work
arrays, the four residuals, status, iter and the return code) for a list of solves, produced
by the reference's own Eigen code through oracle/ref_shim.cpp.  Nothing of the reference's
source is stored.

Scenarios:
  quad_hover_f32_N30   examples/quadrotor_hovering.cpp closed loop, solves k = 0,1,2,24,69
  quad_hover_f64_N10   the same as shipped (fp64, N=10): per-step iter/status/u0 of all 70 solves
  quad_track_f32_N30   examples/quadrotor_tracking.cpp closed loop, solves k = 0,1,100
  quad_batch_f32_N30   16 randomised x0 (hover), cold start + 2 warm starts, early exit and fixed 10 iterations
  cartpole_f32_N10     examples/codegen_cartpole.cpp model, gains from the reference's tiny_codegen()
  random_f32_32_16_50  seeded nx=32,nu=16 system, gains from tiny_codegen(), 2 solves x 4 instances
  dims_f32_8_3_7       odd sizes (nx, nu not multiples of 4), random stable system
  codegen_random_f32_2_2_3  the problem of examples/codegen_random.cpp:19-31 (n = 2, m = 2, N = 3, rho = 0.1, x_min > x_max and
                       u_min > u_max — infeasible boxes, per-row values), gains from the reference's tiny_codegen(); 24 instances,
                       three chained solves, bounds stored as full arrays (bnd_*)
  riccati_*            reference tiny_codegen() cache for cartpole and the random system
  closed_loop_traces   whole closed loops of the compiled reference (tiny_solve + the examples' Eigen plant step) over
                       batches of 64 (20) instances: u.col(0), iter, status of EVERY step and the final x0 — hovering 70
                       steps, tracking 110 steps with sliding per-instance windows, cartpole 100 steps, (8,3,7) 25 steps
"""
import ctypes as C
import json
import re
import shutil
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import oracle as O  # noqa: E402
import accelerated_tinympc_amd as T  # noqa: E402  (host-side problem definitions only; no GPU is touched)

pr = T.problems
OUT = Path(__file__).resolve().parent
PROB_KEYS = ("Kinf", "Pinf", "Quu_inv", "AmBKt", "Adyn", "Bdyn", "Q")


def ref_riccati(nx, nu, N, A, B, Q, R, rho):
    """Reference tiny_codegen() (codegen.cpp:218-696) in a /tmp scratch dir; parse the emitted numbers."""
    lib = C.CDLL(str(ROOT / "oracle" / "_ref" / "libtinympc_ref_riccati.so"))
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    a = np.ascontiguousarray(np.asarray(A, np.float64).T).ravel()
    b = np.ascontiguousarray(np.asarray(B, np.float64).T).ravel()
    q, r = np.ascontiguousarray(Q, np.float64).ravel().copy(), np.ascontiguousarray(R, np.float64).ravel().copy()
    xmn, xmx = np.full(nx * N, -5.0), np.full(nx * N, 5.0)
    umn, umx = np.full(nu * (N - 1), -5.0), np.full(nu * (N - 1), 5.0)
    lib.ref_codegen.argtypes = ([C.c_int] * 3 + [C.POINTER(C.c_double)] * 8 + [C.c_double] * 3 +
                                [C.c_int] * 2 + [C.c_char_p] * 2)
    scratch = Path(tempfile.mkdtemp(prefix="tinympc_codegen_", dir="/tmp"))
    out = scratch / "gen"
    try:
        lib.ref_codegen(nx, nu, N, dp(a), dp(b), dp(q), dp(r), dp(xmn), dp(xmx), dp(umn), dp(umx), float(rho),
                        1e-3, 1e-3, 100, 1, b"/root/reference", str(out).encode())
        txt = (out / "src" / "tiny_data_workspace.cpp").read_text()
    finally:
        shutil.rmtree(scratch, ignore_errors=True)

    def grab(name, rows, cols):
        # "(tiny_MatrixNuNx() << v, v, ...).finished()" entries are row-major (print_matrix, codegen.cpp:118-129)
        m = re.search(r"/\*\s*" + name + r"\s*\*/(.*?)\.finished\(\)", txt, re.S) or \
            re.search(name + r"[^<]*<<(.*?)\)\.finished\(\)", txt, re.S)
        vals = [float(v) for v in re.findall(r"\(tinytype\)(-?[0-9.eE+-]+)", m.group(1))]
        assert len(vals) == rows * cols, (name, len(vals), rows, cols)
        return np.array(vals).reshape(rows, cols)
    return txt, grab


def parse_cache(txt, nx, nu):
    """Pull rho, Kinf, Pinf, Quu_inv, AmBKt, Q out of the generated workspace (codegen.cpp:322-470)."""
    nums = lambda s: [float(v) for v in re.findall(r"\(tinytype\)(-?[0-9.]+(?:[eE][-+]?[0-9]+)?)", s)]
    m = re.search(r"TinyCache cache\s*=\s*\{(.*?)\};", txt, re.S)
    body = m.group(1)
    parts = re.split(r"\(tiny_Matrix\w+\(\)\s*<<", body)
    rho = nums(parts[0])[0]
    mats = [np.array(nums(p)) for p in parts[1:]]
    Kinf, Pinf, Quu, Am = mats[0].reshape(nu, nx), mats[1].reshape(nx, nx), mats[2].reshape(nu, nu), mats[3].reshape(nx, nx)
    return rho, Kinf, Pinf, Quu, Am


def closed_loop(prob, dt, N, x0, xref_fn, steps, keep, settings=None, batch_first=False):
    """Drive the compiled reference through an MPC loop; record pre/post state of the solves in `keep`."""
    ref = O.Reference(prob, dt, settings)
    nx, nu = prob["nx"], prob["nu"]
    xmn, xmx, umn, umx = pr.bounds_arrays(prob, dt)
    B = x0.shape[0]
    st = O.new_state(B, nx, nu, N, dt)
    x0 = x0.astype(dt).copy()
    rec, trace = [], []
    for k in range(steps):
        xref = xref_fn(k).astype(dt)
        st["x"][:, 0] = x0
        st["y"][:] = 0
        st["g"][:] = 0
        pre = O.copy_state(st) if k in keep else None
        rc = ref.solve(st, xmn, xmx, umn, umx, xref)
        trace.append((int(rc), st["iter"].copy(), st["status"].copy(), st["u"][:, 0].copy()))
        if k in keep:
            rec.append(dict(k=k, pre=pre, post=O.copy_state(st), xref=xref.copy(), rc=int(rc)))
        # the examples' own Eigen expression x1 = work.Adyn*x0 + work.Bdyn*work.u.col(0) (quadrotor_hovering.cpp:110-111),
        # compiled from the reference's types (oracle/ref_shim.cpp: ref_plant_step) — not a numpy matmul
        x0 = ref.plant_step(x0, st["u"][:, 0])
    return rec, trace


def save(name, prob, dt, settings, recs, extra=None):
    d = {}
    meta = dict(name=name, nx=prob["nx"], nu=prob["nu"], N=prob["N"], rho=float(prob["rho"]),
                dtype=np.dtype(dt).name, settings=settings, nsolves=len(recs),
                bounds=[float(prob[k]) for k in ("x_min", "x_max", "u_min", "u_max")],
                ks=[r.get("k", i) for i, r in enumerate(recs)], rcs=[r["rc"] for r in recs],
                settings_per_solve=[r.get("settings") for r in recs])
    for k in PROB_KEYS:
        d["prob_" + k] = np.asarray(prob[k], np.float64)
    for s, r in enumerate(recs):
        d[f"s{s}_xref"] = r["xref"]
        for key, v in r["pre"].items():
            d[f"s{s}_pre_{key}"] = v
        for key, v in r["post"].items():
            d[f"s{s}_post_{key}"] = v
    if extra:
        d.update(extra)
    d["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(OUT / f"{name}.npz", **d)
    print(f"wrote {name}.npz  ({(OUT / (name + '.npz')).stat().st_size / 1024:.1f} KiB, {len(recs)} solves)")


def main():
    S = dict(O.DEFAULT_SETTINGS)
    # ---- quadrotor hovering, fp32 N=30 -------------------------------------------------------
    prob = pr.quadrotor(20, 30)
    hov = np.tile(pr.HOVER_XREF, (30, 1))
    recs, trace = closed_loop(prob, np.float32, 30, pr.HOVER_X0[None], lambda k: hov, 70, {0, 1, 2, 24, 69})
    extra = dict(trace_iter=np.array([t[1][0] for t in trace]), trace_status=np.array([t[2][0] for t in trace]),
                 trace_rc=np.array([t[0] for t in trace]), trace_u0=np.array([t[3][0] for t in trace]))
    save("quad_hover_f32_N30", prob, np.float32, S, recs, extra)
    print("  k=0 u0", trace[0][3][0], "iter", trace[0][1][0], " total iters", int(extra["trace_iter"].sum()))
    # ---- as shipped: fp64 N=10 ----------------------------------------------------------------
    prob10 = pr.quadrotor(20, 10)
    hov10 = np.tile(pr.HOVER_XREF, (10, 1))
    recs, trace = closed_loop(prob10, np.float64, 10, pr.HOVER_X0[None], lambda k: hov10, 70, {0, 69})
    extra = dict(trace_iter=np.array([t[1][0] for t in trace]), trace_status=np.array([t[2][0] for t in trace]),
                 trace_rc=np.array([t[0] for t in trace]), trace_u0=np.array([t[3][0] for t in trace]))
    save("quad_hover_f64_N10", prob10, np.float64, S, recs, extra)
    print("  k=0 u0", trace[0][3][0], " total iters", int(extra["trace_iter"].sum()))
    # ---- quadrotor tracking, fp32 N=30 ----------------------------------------------------------
    table = pr.y_axis_line()
    recs, trace = closed_loop(prob, np.float32, 30, table[0][None], lambda k: table[k:k + 30], 120, {0, 1, 100})
    extra = dict(trace_iter=np.array([t[1][0] for t in trace]), trace_u0=np.array([t[3][0] for t in trace]))
    save("quad_track_f32_N30", prob, np.float32, S, recs, extra)
    # ---- batch of 32 randomised hover problems: cold + 2 warm, early-exit and fixed-iteration ----------
    x0b, _ = pr.hover_batch(16, 30, seed=7)
    recs, _ = closed_loop(prob, np.float32, 30, x0b, lambda k: hov, 3, {0, 1, 2})
    S10 = dict(S, abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=10)
    recs10, _ = closed_loop(prob, np.float32, 30, x0b, lambda k: hov, 3, {0, 1, 2}, settings=S10)
    for r in recs: r["settings"] = S
    for r in recs10: r["settings"] = S10
    save("quad_batch_f32_N30", prob, np.float32, S, recs + recs10)
    # ---- tracking batch with per-instance windows (config 3 shape, 24 instances) ------------------
    x0t, tab, start = pr.tracking_batch(24, 30, seed=11)
    ref = O.Reference(prob, np.float32, S)
    xmn, xmx, umn, umx = pr.bounds_arrays(prob, np.float32)
    st = O.new_state(24, 12, 4, 30, np.float32)
    st["x"][:, 0] = x0t
    pre = O.copy_state(st)
    xr = pr.expand_windows(tab, start, 30)
    rc = ref.solve(st, xmn, xmx, umn, umx, xr)
    save("quad_trackbatch_f32_N30", prob, np.float32, S, [dict(pre=pre, post=O.copy_state(st), xref=xr, rc=int(rc))],
         dict(window_start=start, table=tab))
    # ---- cartpole with the reference's own Riccati ----------------------------------------------
    cp = pr.cartpole(10, riccati=O.riccati)
    txt, _ = ref_riccati(4, 1, 10, cp["Adyn"], cp["Bdyn"], cp["Q_raw"], cp["R"], cp["rho"])
    rho, K, P, Qi, Am = parse_cache(txt, 4, 1)
    np.savez_compressed(OUT / "riccati_cartpole.npz", Kinf=K, Pinf=P, Quu_inv=Qi, AmBKt=Am, rho=rho,
                        A=cp["Adyn"], B=cp["Bdyn"], Q=cp["Q_raw"], R=cp["R"])
    print("  cartpole ref Kinf", K.ravel(), "Quu_inv", Qi.ravel())
    cpr = dict(cp, Kinf=K, Pinf=P, Quu_inv=Qi, AmBKt=Am)
    S150 = dict(S, max_iter=150)  # codegen_cartpole.cpp:80
    x0c = np.array([[0.0, 0, 0.1, 0]])
    zr = np.zeros((10, 4))
    recs, trace = closed_loop(cpr, np.float32, 10, x0c, lambda k: zr, 30, {0, 1, 2, 29}, settings=S150)
    save("cartpole_f32_N10", cpr, np.float32, S150, recs, dict(trace_iter=np.array([t[1][0] for t in trace])))
    # ---- random nx=32 nu=16 N=50 ------------------------------------------------------------------
    rs = pr.random_system(32, 16, 50, seed=1234, riccati=O.riccati)
    txt, _ = ref_riccati(32, 16, 50, rs["Adyn"], rs["Bdyn"], rs["Q_raw"], rs["R"], rs["rho"])
    rho, K, P, Qi, Am = parse_cache(txt, 32, 16)
    np.savez_compressed(OUT / "riccati_random_32_16.npz", Kinf=K, Pinf=P, Quu_inv=Qi, AmBKt=Am, rho=rho,
                        A=rs["Adyn"], B=rs["Bdyn"], Q=rs["Q_raw"], R=rs["R"])
    rsr = dict(rs, Kinf=K, Pinf=P, Quu_inv=Qi, AmBKt=Am)
    rng = np.random.default_rng(5)
    x0r = rng.uniform(-1, 1, size=(4, 32))
    xr32 = np.zeros((50, 32))
    recs, trace = closed_loop(rsr, np.float32, 50, x0r, lambda k: xr32, 2, {0, 1})
    save("random_f32_32_16_50", rsr, np.float32, S, recs)
    print("  random32 iters", trace[0][1], trace[1][1])
    # ---- odd dims 8,3,7 -----------------------------------------------------------------------------
    od = pr.random_system(8, 3, 7, seed=99, riccati=O.riccati)
    x0o = np.random.default_rng(3).uniform(-1, 1, size=(20, 8))
    xro = np.random.default_rng(4).uniform(-0.2, 0.2, size=(7, 8))
    recs, trace = closed_loop(od, np.float32, 7, x0o, lambda k: xro, 3, {0, 1, 2})
    save("dims_f32_8_3_7", od, np.float32, S, recs)


def codegen_random_fixture():
    """examples/codegen_random.cpp:19-31: the numbers are the example's (column-major in the source), the cache is what the
    reference's own tiny_codegen() emits for them."""
    nx, nu, N = 2, 2, 3
    A = np.array([[1.0, 1.0], [5.0, 2.0]])       # Adyn_data = {1, 5, 1, 2} column-major
    Bm = np.array([[3.0, 4.0], [3.0, 1.0]])      # Bdyn_data = {3, 3, 4, 1}
    Q, R, rho = np.array([1.0, 1.0]), np.array([2.0, 2.0]), 0.1
    txt, _ = ref_riccati(nx, nu, N, A, Bm, Q, R, rho)
    rho_c, K, P, Qi, Am = parse_cache(txt, nx, nu)
    # work.Q as tiny_codegen stores it: Q + rho (codegen.cpp:255, :433)
    prob = dict(nx=nx, nu=nu, N=N, rho=rho_c, Kinf=K, Pinf=P, Quu_inv=Qi, AmBKt=Am, Adyn=A, Bdyn=Bm, Q=Q + rho,
                x_min=0.0, x_max=0.0, u_min=0.0, u_max=0.0)
    dt = np.float32
    xmn = np.tile(np.array([1.0, 2.0], dt), (N, 1)); xmx = np.tile(np.array([-1.0, -2.0], dt), (N, 1))          # x_min > x_max
    umn = np.tile(np.array([2.0, 3.0], dt), (N - 1, 1)); umx = np.tile(np.array([-2.0, -3.0], dt), (N - 1, 1))  # u_min > u_max
    B = 24
    rng = np.random.default_rng(223)
    x0 = rng.uniform(-1.5, 1.5, size=(B, nx)).astype(dt)
    xref = (rng.standard_normal((B, N, nx)) * 0.3).astype(dt)
    recs = []
    st = O.new_state(B, nx, nu, N, dt)
    st["x"][:, 0] = x0
    for k, settings in enumerate((dict(O.DEFAULT_SETTINGS), dict(O.DEFAULT_SETTINGS, max_iter=7, abs_pri_tol=0.0, abs_dua_tol=0.0),
                                  dict(O.DEFAULT_SETTINGS, check_termination=3, max_iter=40))):
        pre = O.copy_state(st)
        rc = O.Reference(prob, dt, settings).solve(st, xmn, xmx, umn, umx, xref)
        recs.append(dict(k=k, pre=pre, post=O.copy_state(st), xref=xref.copy(), rc=int(rc), settings=settings))
        st["y"][:] = 0; st["g"][:] = 0
    save("codegen_random_f32_2_2_3", prob, dt, dict(O.DEFAULT_SETTINGS), recs,
         dict(bnd_xmin=xmn, bnd_xmax=xmx, bnd_umin=umn, bnd_umax=umx))
    print("  codegen_random iters", [r["post"]["iter"][:6].tolist() for r in recs])


def closed_loop_traces():
    """Whole closed loops through the compiled reference; data only (inputs + per-step outputs)."""
    S = dict(O.DEFAULT_SETTINGS)
    d, meta = {}, {}

    def run(name, prob, N, x0, xref_fn, steps, settings, extra_meta):
        _, trace = closed_loop(prob, np.float32, N, x0, xref_fn, steps, set(), settings=settings)
        ref = O.Reference(prob, np.float32, settings)
        # final state x0 after the last plant step: replay the last step's plant from the recorded u0
        d[f"{name}_x0"] = x0.astype(np.float32)
        d[f"{name}_u0"] = np.stack([t[3] for t in trace])           # (steps, B, nu)
        d[f"{name}_iter"] = np.stack([t[1] for t in trace])         # (steps, B)
        d[f"{name}_status"] = np.stack([t[2] for t in trace])
        xs = x0.astype(np.float32).copy()
        for t in trace:
            xs = ref.plant_step(xs, t[3])
        d[f"{name}_x_final"] = xs
        meta[name] = dict(steps=steps, B=int(x0.shape[0]), N=N, settings=settings, **extra_meta)
        print(f"  {name}: {steps} steps x {x0.shape[0]} instances, mean iterations {d[name + '_iter'].mean():.2f}, "
              f"converged on the last step {np.mean(d[name + '_status'][-1] == 1):.2f}")

    prob = pr.quadrotor(20, 30)
    hov = np.tile(pr.HOVER_XREF, (30, 1))
    x0h, _ = pr.hover_batch(64, 30, seed=21)
    x0h[0] = pr.HOVER_X0                                   # instance 0 = the example itself
    run("hover", prob, 30, x0h, lambda k: hov, 70, S, dict(problem="quadrotor_20hz", xref="hover"))
    table = pr.y_axis_line().astype(np.float32)
    start = (np.arange(64) * 3 % 100).astype(np.int32)     # per-instance window phases; instance 0 = the example (start 0)
    rng = np.random.default_rng(22)
    x0t = (table[start] + rng.uniform(-0.05, 0.05, size=(64, 12))).astype(np.float32)
    x0t[0] = table[0]
    d["track_start"] = start
    run("track", prob, 30, x0t, lambda k: pr.expand_windows(table, start + k, 30), 110, S,
        dict(problem="quadrotor_20hz", xref="window of y_axis_line, start_b + k", window_advance=1))
    z = np.load(OUT / "riccati_cartpole.npz")
    cp = dict(pr.cartpole(10, riccati=O.riccati), Kinf=z["Kinf"], Pinf=z["Pinf"], Quu_inv=z["Quu_inv"], AmBKt=z["AmBKt"])
    S150 = dict(S, max_iter=150)
    x0c = np.array([[0.0, 0, 0.1, 0]]) + np.random.default_rng(23).uniform(-0.05, 0.05, size=(64, 4))
    x0c[0] = [0.0, 0, 0.1, 0]
    zr = np.zeros((10, 4))
    run("cartpole", cp, 10, x0c, lambda k: zr, 100, S150, dict(problem="cartpole (reference tiny_codegen cache, riccati_cartpole.npz)", xref="zero"))
    od = pr.random_system(8, 3, 7, seed=99, riccati=O.riccati)
    x0o = np.random.default_rng(24).uniform(-1, 1, size=(20, 8))
    xro = np.random.default_rng(4).uniform(-0.2, 0.2, size=(7, 8))
    run("dims837", od, 7, x0o, lambda k: xro, 25, S, dict(problem="random_system(8,3,7,seed=99)", xref="fixed random"))
    d["dims837_xref"] = xro
    d["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(OUT / "closed_loop_traces.npz", **d)
    print(f"wrote closed_loop_traces.npz ({(OUT / 'closed_loop_traces.npz').stat().st_size / 1024:.1f} KiB)")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "codegen_random":
    O.build(ref=True)
    codegen_random_fixture()
    sys.exit(0)

if __name__ == "__main__":
    main()
    closed_loop_traces()
