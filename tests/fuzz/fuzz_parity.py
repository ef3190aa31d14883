"""Randomised differential test of the exact kernels against the CPU oracle (test infrastructure; run on an MI355X):

    python tests/fuzz/fuzz_parity.py [seconds] [seed]

Every round draws a problem class with an exact kernel, a batch size, settings (iteration limits, termination stride,
tolerances, bound switches), bounds (per step, some infeasible or infinite), a reference (shared / per instance / sliding
window), a random warm workspace (with zeros and negative zeros) and a row-kernel family, runs a chain of solves and
requires all twelve work arrays, the residuals, status and iter to equal the oracle's bit for bit.  One round in six also
switches on the two terms the reference ships commented out (admm.cpp:20 coeff_d2p, :79 Uref), one in five runs under a
random caller-supplied dispatch order."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import accelerated_tinympc_amd as T  # noqa: E402
from oracle import oracle as O  # noqa: E402

import ctypes  # noqa: E402
hip = ctypes.CDLL("libamdhip64.so")  # the runtime the library links: device buffers for caller-supplied dispatch orders


def dev_ints(a):
    p = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(a.nbytes)) == 0
    assert hip.hipMemcpy(p, ctypes.c_void_p(a.ctypes.data), ctypes.c_size_t(a.nbytes), 1) == 0
    return p


pr = T.problems
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
CLASSES = [("quad", 30), ("quad", 25), ("quad", 20), ("quad", 10), ("quad", 17), ("quad", 7), ("quad", 36), ("quad", 40), ("quad", 50), ("quad", 61), ("quad", 67), ("cartpole", 10),
           ("cartpole", 23), ("odd", 7), ("odd", 13), ("rand32", 50), ("rand32", 50), ("rand32", 23), ("rand32", 2), ("r8_4", 9), ("r12_2", 14), ("r4_2", 8), ("r4_4", 35), ("w16_8", 10), ("w16_4", 12), ("w20_8", 11), ("w24_4", 9),
           # round 4: classes outside the compiled lists, solved by the run-time-dimension exact kernel (admm_generic.hip)
           ("g20_12", 12), ("g3_2", 6), ("g8_8", 6), ("g4_3", 9), ("g36_4", 5), ("g28_16", 6)]
if len(sys.argv) > 3:   # optional third argument: only the classes whose name contains it (e.g. "rand32": the nx = 32 kernels incl. tile48)
    CLASSES = [c for c in CLASSES if sys.argv[3] in c[0]]
t_end, rounds, solves, t_note, overflowed, refused = time.time() + budget, 0, 0, time.time(), 0, 0
kernels_seen = {}
while time.time() < t_end:
    if time.time() - t_note > 30:
        print(f"... {rounds} rounds, {solves} solves so far", flush=True)
        t_note = time.time()
    kind, N = CLASSES[rng.integers(len(CLASSES))]
    if "_" in kind:
        prob = pr.random_system(int(kind[1:].split("_")[0]), int(kind.split("_")[1]), N, seed=7)
    else:
        prob = {"quad": lambda: pr.quadrotor(20, N), "cartpole": lambda: pr.cartpole(N), "odd": lambda: pr.random_system(8, 3, N, seed=99),
                "rand32": lambda: pr.random_system(32, 16, N)}[kind]()
    nx, nu = prob["nx"], prob["nu"]
    gen = kind.startswith("g")   # no compiled kernel: generic<nx,nu,exact>
    wave = kind == "rand32" or kind.startswith("w") or gen
    B = int(rng.choice([1, 2, 3, 4, 5, 15, 16, 17, 63, 64, 65, 200])) if not wave else int(rng.choice([1, 3, 9, 17, 40]))
    settings = dict(abs_pri_tol=float(rng.choice([0.0, 1e-3, 1e-2, 0.5])), abs_dua_tol=float(rng.choice([0.0, 1e-3, 1e-1, 5.0])),
                    max_iter=int(rng.choice([0, 1, 2, 3, 7, 20, 45])), check_termination=int(rng.choice([1, 1, 2, 3, 7])),
                    en_state_bound=int(rng.integers(2)), en_input_bound=int(rng.integers(2)))
    xmn, xmx, umn, umx = [a.copy() for a in pr.bounds_arrays(prob)]
    scale = rng.uniform(0.05, 1.0)
    xmn *= scale; xmx *= scale * rng.uniform(0.5, 1.5); umn *= rng.uniform(0.1, 1.0, size=umn.shape).astype(np.float32); umx *= scale
    if rng.random() < 0.3:   # infeasible and infinite entries (examples/codegen_random.cpp:28-31 has min > max)
        umn[rng.integers(N - 1), rng.integers(nu)] = 3.0
        xmx[rng.integers(N), rng.integers(nx)] = np.inf
        xmn[rng.integers(N), rng.integers(nx)] = -np.inf
    bnds = (xmn, xmx, umn, umx)
    if rng.random() < 0.25:  # per-instance bounds (served by the kernels that stream their state)
        bnds = tuple((a[None] * rng.uniform(0.3, 1.0, size=(B,) + a.shape)).astype(np.float32) for a in bnds)
        if rng.random() < 0.4:   # ... that do not change along the horizon (admm_tile16_pi.hip keeps one resident row per instance then)
            bnds = tuple(np.repeat(a[:, :1], a.shape[1], axis=1).copy() for a in bnds)
    bnds_raw = bnds
    sol = T.TinyBatchSolver(prob, B, settings=settings)
    h16 = not wave and rng.random() < 0.3   # fp16 storage / fp32 arithmetic against the oracle's _h16 restatement
    h16d = False
    if h16:
        sol.set_storage(16, 16)
        if rng.random() < 0.4:   # ... with the duals kept in fp32 (register-resident kernels only; refused elsewhere)
            try:
                sol.set_storage(16, 32); h16d = True
            except T.TinyBatchError:
                pass
    R = O.round_h16 if h16 else (lambda a: a)
    if not wave and not h16 and rng.random() < 0.06:   # the same kernel forced onto a compiled class (variant 4)
        sol.select_kernel(4); gen = True
    fams = [0] + ([1, 2, 3, 4, 5] if not wave else [6, 7, 8, 8])   # 5 = tile16 (MFMA products), 6 / 7 = streaming / on-chip wave kernel, 8 = tile48 (nx = 32)
    fam = int(rng.choice(fams))
    try:
        sol.set_row_kernel(fam)
    except T.TinyBatchError:
        sol.set_row_kernel(0)
    sol.set_bounds(*bnds)
    bnds = tuple(R(b) for b in bnds)
    mode = rng.integers(3) if nx <= 16 else rng.integers(2)
    if mode == 0:
        xref = (rng.standard_normal((N, nx)) * 0.3).astype(np.float32); sol.set_xref(xref)
    elif mode == 1:
        xref = (rng.standard_normal((B, N, nx)) * 0.3).astype(np.float32)
        if rng.random() < 0.3:   # every instance its own set point
            xref = np.repeat(xref[:, :1], N, axis=1).copy()
        sol.set_xref(xref)
    else:
        table = (rng.standard_normal((N + 40, nx)) * 0.3).astype(np.float32)
        start = rng.integers(0, 40, size=B).astype(np.int32)
        sol.set_xref_window(table, start); xref = pr.expand_windows(table, start, N)
    xref = R(xref)
    st = O.new_state(B, nx, nu, N)
    if rng.random() < 0.7:   # warm workspace, sprinkled with zeros and negative zeros
        for k in O.STATE_ORDER:
            v = (rng.standard_normal(st[k].shape) * 0.3).astype(np.float32)
            v[rng.random(v.shape) < 0.1] = 0.0
            v[rng.random(v.shape) < 0.05] = -0.0
            st[k][:] = R(v)
        st["residuals"][:] = rng.uniform(0, 1, size=(B, 4)).astype(np.float32)
        st["iter"][:] = rng.integers(1, 9, size=B); st["status"][:] = 11
        sol.set_state(st)
    else:
        x0 = rng.uniform(-0.5, 0.5, size=(B, nx)).astype(np.float32)
        st["x"][:, 0] = R(x0); sol.set_x0(x0)
    opt = (rng.integers(2), rng.integers(2)) if (nx + nu <= 16 and not gen and rng.random() < 0.17) else (0, 0)
    if opt[0] or opt[1]:   # the commented-out terms: any combination, shared or per-instance Uref, zeros and negative zeros in it
        prob = dict(prob, coeff_d2p=(rng.standard_normal((nx, nu)) * 0.05).astype(np.float32), R=rng.uniform(0.2, 3.0, nu).astype(np.float32))
        uref = (rng.standard_normal((N - 1, nu) if rng.random() < 0.5 else (B, N - 1, nu)) * 0.2).astype(np.float32)
        uref[rng.random(uref.shape) < 0.1] = 0.0
        uref[rng.random(uref.shape) < 0.05] = -0.0
        sol.set_input_cost(prob["R"]); sol.set_coeff_d2p(prob["coeff_d2p"]); sol.set_uref(uref)
        sol.set_optional_terms(opt[0], opt[1])
        settings = dict(settings, en_uref=int(opt[0]), en_coeff_d2p=int(opt[1]))
    d_order = None
    if rng.random() < 0.2:   # a caller-supplied dispatch order (random permutation of the groups of four) or the predicted one
        d_order = dev_ints(rng.permutation((B + 3) // 4).astype(np.int32))
        sol.set_dispatch_order_device(d_order.value)
    elif rng.random() < 0.2:
        sol.set_dispatch(int(rng.choice([1, 2])))   # predictor / history order (they act on launches of >= 4096 groups; smaller ones check the plumbing)
    if rng.random() < 0.3:   # tile16's two-ended tile queue at a forced stride: applies to cold-start launches of any size
        sol.set_tile_queue(int(rng.integers(0, 10)))
    orc = O.Oracle(prob, ("h16d" if h16d else "h16") if h16 else np.float32, settings)
    if opt[0]:
        orc.set_uref(R(uref))
    for k in range(int(rng.integers(1, 4))):
        if rng.random() < 0.6:
            st["y"][:] = 0; st["g"][:] = 0; sol.reset_dual_variables()
        if h16d:
            try:
                sol.solve()
            except T.TinyBatchError:   # per-instance bounds, optional terms or a forced streaming kernel: refused by design
                refused += 1
                break
            orc.solve(st, *bnds, xref, nthreads=8); solves += 1
        else:
            orc.solve(st, *bnds, xref, nthreads=8); sol.solve(); solves += 1
            kernels_seen[sol.kernel_name()] = kernels_seen.get(sol.kernel_name(), 0) + 1
        if not all(np.all(np.isfinite(st[n_])) for n_ in O.STATE_ORDER):
            overflowed += 1   # NaN / inf regime (fp16 storage overflow, unstable iteration): behaviour undefined
            break
        got = sol.get_state()
        for name in O.STATE_ORDER + ("residuals", "status", "iter"):
            g_, r_ = got[name], st[name]
            if g_.dtype == np.float32:   # NaN (overflowing fp16 storage, inf - inf) only has to be NaN on both sides
                nn = np.isnan(g_) & np.isnan(r_)
                same = np.all((g_ == r_) | nn) and np.array_equal(np.signbit(g_)[~nn], np.signbit(r_)[~nn])
            else:
                same = np.array_equal(g_, r_)
            if not same:
                print(f"MISMATCH round {rounds} h16={h16} h16d={h16d} {kind} N={N} B={B} kernel {sol.kernel_name()} settings {settings} xref mode {mode} solve {k}: {name}"
                      f" | per-instance bounds {bnds_raw[0].ndim == 3}, row kernel family {fam}, caller order {d_order is not None}, optional terms {opt}")
                bad = np.argwhere(~((got[name] == st[name]) & (np.signbit(got[name]) == np.signbit(st[name]))))
                for idx in bad[:6]:
                    idx = tuple(idx)
                    print("   ", idx, "gpu", got[name][idx], "oracle", st[name][idx], "iter gpu/oracle", got["iter"][idx[0]], st["iter"][idx[0]])
                sys.exit(1)
    finite = all(np.all(np.isfinite(st[n_])) for n_ in O.STATE_ORDER)
    if finite and nx + nu <= 16 and not h16d and not gen and rng.random() < 0.3:   # one of the six step functions on the state the chain left
        fn = O.Oracle.STEP_FUNCTIONS[rng.integers(6)]
        ref_rv = orc.step(fn, st, *bnds, xref)
        rv = getattr(sol, fn)()
        got = sol.get_state()
        ok = fn != "termination_condition" or np.array_equal(rv, ref_rv)
        for name in O.STATE_ORDER + ("residuals",):
            nn = np.isnan(got[name]) & np.isnan(st[name])
            ok = ok and np.all((got[name] == st[name]) | nn) and np.array_equal(np.signbit(got[name])[~nn], np.signbit(st[name])[~nn])
        if not ok:
            print(f"MISMATCH round {rounds} step function {fn} {kind} N={N} B={B} h16={h16} settings {settings}")
            sys.exit(1)
    sol.close(); rounds += 1
    if d_order is not None:
        hip.hipFree(d_order)
print(f"fuzz ok: {rounds} rounds, {solves} solves, all bitwise equal to the oracle (signs of zeros included); "
      f"{overflowed} rounds left the finite range and were not compared; {refused} fp32-dual rounds asked for a kernel that does not implement them and were refused")
fam = {}
for k, v in kernels_seen.items():   # solves per kernel family (tile16 split into its shared-table and per-instance-table instantiations)
    key = k.split("<")[0] + (",pi" if k.endswith(",pi>") else "")
    fam[key] = fam.get(key, 0) + v
print("solves per kernel:", ", ".join(f"{k} {v}" for k, v in sorted(fam.items())))
