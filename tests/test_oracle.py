"""CPU tests: the oracle (our C restatement) against the compiled reference's golden vectors,
and, where oracle/_ref is present, bit-for-bit against the compiled reference itself."""
import numpy as np
import pytest

from helpers import GOLDEN as GOLDEN_DIR, STATE_ORDER, bounds_of, closed_loop_case, load_fixture

FIXTURES = ["quad_hover_f32_N30", "quad_hover_f64_N10", "quad_track_f32_N30", "quad_batch_f32_N30",
            "quad_trackbatch_f32_N30", "cartpole_f32_N10", "random_f32_32_16_50", "dims_f32_8_3_7"]


@pytest.mark.parametrize("name", FIXTURES)
def test_oracle_reproduces_golden_bit_exact(oracle_mod, name):
    """Every golden solve (live-in -> live-out of the reference's tiny_solve) is reproduced exactly."""
    O = oracle_mod
    meta, prob, solves, _ = load_fixture(name)
    dt = np.dtype(meta["dtype"])
    xmn, xmx, umn, umx = bounds_of(prob, dt)
    for s in solves:
        orc = O.Oracle(prob, dt, s["settings"])
        st = {k: v.copy() for k, v in s["pre"].items()}
        rc = orc.solve(st, xmn, xmx, umn, umx, s["xref"])
        assert (rc > 0) == (s["rc"] > 0)
        for k in STATE_ORDER + ("residuals", "status", "iter"):
            assert np.array_equal(st[k], s["post"][k]), (name, s["k"], k, np.max(np.abs(st[k] - s["post"][k])))


def test_oracle_reproduces_the_codegen_random_example(oracle_mod):
    """examples/codegen_random.cpp:19-31 (n = 2, m = 2, N = 3, per-row bounds with min > max): golden vectors of the compiled
    reference, bounds stored as full arrays."""
    O = oracle_mod
    meta, prob, solves, z = load_fixture("codegen_random_f32_2_2_3")
    bnds = [z[k] for k in ("bnd_xmin", "bnd_xmax", "bnd_umin", "bnd_umax")]
    assert (bnds[0] > bnds[1]).all() and (bnds[2] > bnds[3]).all()  # the example's infeasible boxes
    for s in solves:
        st = {k: v.copy() for k, v in s["pre"].items()}
        rc = O.Oracle(prob, np.float32, s["settings"]).solve(st, *bnds, s["xref"])
        assert (rc > 0) == (s["rc"] > 0)
        for k in STATE_ORDER + ("residuals", "status", "iter"):
            assert np.array_equal(st[k], s["post"][k]), (s["k"], k)


def test_known_answers_of_the_survey(oracle_mod):
    """SURVEY.md §4 KATs of the unchanged reference, via the golden traces."""
    meta, prob, solves, z = load_fixture("quad_hover_f64_N10")
    np.testing.assert_allclose(z["trace_u0"][0], [0.488778532, 0.478938215, 0.542743696, 0.551056731], rtol=0, atol=5e-10)
    assert int(z["trace_iter"].sum()) == 1269 and z["trace_iter"][0] == 100 and z["trace_iter"][69] == 2
    assert z["trace_status"][0] == 11 and z["trace_status"][69] == 1 and z["trace_rc"][0] == 1 and z["trace_rc"][69] == 0
    meta, prob, solves, z = load_fixture("quad_hover_f32_N30")
    np.testing.assert_allclose(z["trace_u0"][0], [0.48478967, 0.476141721, 0.531687975, 0.539531589], rtol=0, atol=2e-7)
    np.testing.assert_allclose(solves[0]["post"]["residuals"][0], [0, 3.953e-2, 2.403e-2, 7.488e-3], rtol=2e-3, atol=0)
    assert z["trace_iter"][0] == 100 and z["trace_iter"][69] == 2


def test_max_iter_zero_touches_only_status_and_iter(oracle_mod):
    """admm.cpp:114-117,151: with max_iter=0 tiny_solve returns 1, status=11, iter=1, nothing else changes."""
    O = oracle_mod
    meta, prob, solves, _ = load_fixture("quad_hover_f32_N30")
    s = solves[2]
    xmn, xmx, umn, umx = bounds_of(prob, np.float32)
    st = {k: v.copy() for k, v in s["pre"].items()}
    rc = O.Oracle(prob, np.float32, dict(s["settings"], max_iter=0)).solve(st, xmn, xmx, umn, umx, s["xref"])
    assert rc == 1 and st["status"][0] == 11 and st["iter"][0] == 1
    for k in STATE_ORDER + ("residuals",):
        assert np.array_equal(st[k], s["pre"][k])


CFGS = [(np.float32, 12, 4, 30), (np.float64, 12, 4, 30), (np.float64, 12, 4, 10), (np.float32, 12, 4, 10),
        (np.float32, 4, 1, 10), (np.float64, 4, 1, 10), (np.float32, 8, 3, 7), (np.float32, 32, 16, 50),
        (np.float32, 8, 4, 9), (np.float32, 12, 2, 11), (np.float32, 4, 2, 8), (np.float32, 4, 4, 6), (np.float64, 8, 4, 9),
        (np.float32, 16, 8, 10), (np.float32, 16, 4, 10), (np.float32, 20, 8, 10), (np.float32, 24, 4, 10),
        (np.float64, 12, 2, 11), (np.float64, 4, 2, 8), (np.float64, 4, 4, 6), (np.float64, 16, 4, 10),
        (np.float32, 2, 2, 3), (np.float64, 2, 2, 3),  # examples/codegen_random.cpp:19-21
        # round 4: classes OUTSIDE the compiled kernel lists whose orders the rules still define (nx, nu each <= 4 or a multiple of 4): what the
        # run-time-dimension exact kernel (admm_generic.hip) is held to
        (np.float32, 20, 12, 12), (np.float32, 3, 2, 6), (np.float32, 8, 8, 6), (np.float32, 4, 3, 9), (np.float32, 36, 4, 5), (np.float32, 28, 16, 6)]


@pytest.mark.parametrize("dt,nx,nu,N", CFGS)
def test_oracle_bit_exact_vs_compiled_reference(oracle_mod, tinympc, dt, nx, nu, N):
    """Random warm-start states and references through the compiled reference (oracle/_ref) and the oracle."""
    O = oracle_mod
    if not O.have_ref(dt, nx, nu, N):
        pytest.skip("oracle/_ref not built here (needs /root/reference)")
    pr = tinympc.problems
    if (nx, nu) == (12, 4):
        prob = pr.quadrotor(20, N)
    elif (nx, nu) == (4, 1):
        prob = pr.cartpole(N, riccati=O.riccati)
    else:
        prob = pr.random_system(nx, nu, N, seed=nx * 100 + nu, riccati=O.riccati)
    rng = np.random.default_rng(nx + nu + N)
    B = 5
    st0 = O.new_state(B, nx, nu, N, dt)
    for k in STATE_ORDER:
        st0[k][:] = (rng.standard_normal(st0[k].shape) * 0.3).astype(dt)
    # zeros and negative zeros in the live-in state: their signs propagate into products, sums and residuals
    for k in ("x", "d", "v", "z", "g", "y"):
        st0[k][rng.random(st0[k].shape) < 0.1] = 0.0
        st0[k][rng.random(st0[k].shape) < 0.1] = -0.0
    st0["x"][0, 0] = -0.0; st0["g"][0] = 0.0; st0["y"][0] = 0.0; st0["d"][0] = 0.0   # an instance that sits at the origin
    xref = (rng.standard_normal((B, N, nx)) * 0.2).astype(dt)
    xref[0] = 0.0
    xmn, xmx, umn, umx = pr.bounds_arrays(prob, dt)
    for settings in (dict(max_iter=1, abs_pri_tol=0, abs_dua_tol=0), dict(max_iter=12, abs_pri_tol=0, abs_dua_tol=0),
                     dict(max_iter=60, check_termination=3), dict(max_iter=5, en_state_bound=0, en_input_bound=0)):
        a, b = O.copy_state(st0), O.copy_state(st0)
        ra = O.Oracle(prob, dt, settings).solve(a, xmn, xmx, umn, umx, xref)
        rb = O.Reference(prob, dt, settings).solve(b, xmn, xmx, umn, umx, xref)
        assert ra == rb
        for k in STATE_ORDER + ("residuals", "status", "iter"):
            assert np.array_equal(a[k], b[k]), (settings, k)
            if a[k].dtype.kind == "f":
                assert np.array_equal(np.signbit(a[k]), np.signbit(b[k])), (settings, k, "sign of a zero")


TERMS_CFGS = [(np.float32, 12, 4, 30), (np.float32, 4, 1, 10), (np.float32, 8, 4, 9), (np.float32, 12, 2, 11), (np.float32, 4, 2, 8),
              (np.float32, 4, 4, 6), (np.float64, 12, 4, 10), (np.float32, 16, 8, 10), (np.float32, 32, 16, 50)]


@pytest.mark.parametrize("dt,nx,nu,N", TERMS_CFGS)
def test_optional_terms_bit_exact_vs_eigen(oracle_mod, tinympc, dt, nx, nu, N):
    """The two terms the reference comments out (admm.cpp:20 coeff_d2p, :79 Uref), enabled in the oracle, against Eigen
    evaluating the same expressions over the reference's types (oracle/ref_terms_shim.cpp -> oracle/_ref)."""
    import ctypes as C
    O = oracle_mod
    suf = "f32" if dt == np.float32 else "f64"
    path = O.HERE / "_ref" / f"libtinympc_terms_{suf}_{nx}_{nu}_{N}.so"
    if not path.exists():
        pytest.skip("oracle/_ref not built here (needs /root/reference)")
    lib = C.CDLL(str(path))
    ct = C.c_float if dt == np.float32 else C.c_double
    P = C.POINTER(ct)
    lib.terms_backward_pass_grad.argtypes, lib.terms_backward_pass_grad.restype = [P] * 9, None
    lib.terms_input_cost.argtypes, lib.terms_input_cost.restype = [ct] + [P] * 5, None
    rng = np.random.default_rng(7 * nx + nu)
    pr = tinympc.problems
    prob = dict(pr.quadrotor(20, N) if (nx, nu) == (12, 4) else pr.random_system(nx, nu, N, seed=nx * 100 + nu, riccati=O.riccati))
    prob["coeff_d2p"] = rng.standard_normal((nx, nu)) * 0.3
    prob["R"] = rng.uniform(0.5, 3.0, nu)
    B = 4
    st = O.new_state(B, nx, nu, N, dt)
    for k in STATE_ORDER:
        st[k][:] = (rng.standard_normal(st[k].shape) * 0.4).astype(dt)
        st[k][rng.random(st[k].shape) < 0.08] = 0.0
        st[k][rng.random(st[k].shape) < 0.08] = -0.0
    uref = (rng.standard_normal((B, N - 1, nu)) * 0.2).astype(dt)
    uref[rng.random(uref.shape) < 0.1] = 0.0
    uref[0] = -0.0
    xref = np.zeros((N, nx), dt)
    xmn, xmx, umn, umx = pr.bounds_arrays(prob, dt)
    orc = O.Oracle(prob, dt, dict(en_uref=1, en_coeff_d2p=1))
    orc.set_uref(uref)
    a = O.copy_state(st)
    orc.step("update_linear_cost", a, xmn, xmx, umn, umx, xref)
    cm = lambda m: np.ascontiguousarray(np.asarray(m, dt).T).ravel()
    ptr = lambda v: v.ctypes.data_as(P)
    Rv = np.ascontiguousarray(prob["R"], dt)
    for b in range(B):
        r = np.zeros((N - 1, nu), dt)
        lib.terms_input_cost(ct(prob["rho"]), ptr(uref[b]), ptr(Rv), ptr(st["znew"][b]), ptr(st["y"][b]), ptr(r))
        assert np.array_equal(r, a["r"][b]) and np.array_equal(np.signbit(r), np.signbit(a["r"][b]))
    a = O.copy_state(st)
    orc.step("backward_pass_grad", a, xmn, xmx, umn, umx, xref)
    mats = [cm(prob[k]) for k in ("Kinf", "Quu_inv", "AmBKt", "Bdyn", "coeff_d2p")]
    for b in range(B):
        p, d = st["p"][b].copy(), st["d"][b].copy()
        lib.terms_backward_pass_grad(*[ptr(m) for m in mats], ptr(st["q"][b]), ptr(st["r"][b]), ptr(p), ptr(d))
        for k, v in (("p", p), ("d", d)):
            assert np.array_equal(v, a[k][b]) and np.array_equal(np.signbit(v), np.signbit(a[k][b])), k
    # switched off (the default) the oracle is the reference: identical to an oracle that never heard of the terms
    off, base = O.copy_state(st), O.copy_state(st)
    o2 = O.Oracle(prob, dt, dict(max_iter=4, abs_pri_tol=0, abs_dua_tol=0))
    o2.set_uref(uref)
    o2.solve(off, xmn, xmx, umn, umx, xref)
    plain = {k: v for k, v in prob.items() if k not in ("coeff_d2p",)}
    O.Oracle(plain, dt, dict(max_iter=4, abs_pri_tol=0, abs_dua_tol=0)).solve(base, xmn, xmx, umn, umx, xref)
    assert all(np.array_equal(off[k], base[k]) for k in STATE_ORDER)
    # and a solve with both terms on differs from it (the terms are live) while a zero Uref / zero coeff_d2p changes no value
    on = O.copy_state(st)
    o3 = O.Oracle(prob, dt, dict(max_iter=4, abs_pri_tol=0, abs_dua_tol=0, en_uref=1, en_coeff_d2p=1))
    o3.set_uref(uref)
    o3.solve(on, xmn, xmx, umn, umx, xref)
    assert not np.array_equal(on["u"], base["u"])
    zero = dict(prob, coeff_d2p=np.zeros((nx, nu)))
    z = O.copy_state(st)
    o4 = O.Oracle(zero, dt, dict(max_iter=4, abs_pri_tol=0, abs_dua_tol=0, en_uref=1, en_coeff_d2p=1))
    o4.set_uref(np.zeros((N - 1, nu), dt))
    o4.solve(z, xmn, xmx, umn, umx, xref)
    assert all(np.array_equal(z[k], base[k]) for k in STATE_ORDER)  # == : +0 and -0 compare equal


def test_riccati_oracle_vs_reference_codegen(oracle_mod):
    """oracle_riccati (codegen.cpp:254-292 restated) against the cache the reference's tiny_codegen() emitted."""
    O = oracle_mod
    for name, nx, nu in (("riccati_cartpole", 4, 1), ("riccati_random_32_16", 32, 16)):
        z = np.load(__import__("helpers").GOLDEN / f"{name}.npz")
        c, it = O.riccati(nx, nu, z["A"], z["B"], z["Q"], z["R"], float(z["rho"]))
        for k in ("Kinf", "Pinf", "Quu_inv", "AmBKt"):
            np.testing.assert_allclose(c[k], z[k], rtol=1e-9, atol=1e-9 * np.max(np.abs(z[k])))
        if nx == 4:
            assert it == 476  # "Kinf converged after 476 iterations" 
            np.testing.assert_allclose(c["Kinf"].ravel(), [-2.9121762, -4.8173684, 44.3538696, 19.7167444], atol=1e-7)


def test_fp16_storage_oracle_rounding_and_fixpoint(oracle_mod):
    """The _h16 instantiation (fp16 storage, fp32 arithmetic): its rounding primitive is IEEE binary16 round-to-nearest-even
    (checked against numpy on normals, subnormals, ties and overflow), every array it leaves behind is binary16
    representable, and with storage rounding switched off by construction (inputs that never leave the binary16 grid:
    max_iter = 0) it touches nothing, like the fp32 oracle."""
    import ctypes as C
    O = oracle_mod
    lib = O._lib()
    lib.oracle_round_h16.argtypes, lib.oracle_round_h16.restype = [C.c_float], C.c_float
    h = np.arange(0, 0x7c00, dtype=np.uint16).view(np.float16).astype(np.float32)
    ties = ((h[:-1].astype(np.float64) + h[1:].astype(np.float64)) / 2).astype(np.float32)
    rng = np.random.default_rng(0)
    rand = (rng.standard_normal(20000) * 10.0 ** rng.integers(-9, 6, 20000)).astype(np.float32)
    xs = np.concatenate([h[::7], ties[::5], -ties[::11], rand, np.array([65504, 65519.9, 65520, 1e6, -1e6, 0.0], np.float32)])
    got = np.array([lib.oracle_round_h16(float(v)) for v in xs], np.float32)
    assert np.array_equal(got, O.round_h16(xs))
    import accelerated_tinympc_amd as T
    prob = T.problems.quadrotor(20, 30)
    B = 16
    x0, table, start = T.problems.tracking_batch(B, 30, seed=1)
    xref = O.round_h16(T.problems.expand_windows(table, start, 30))
    bnds = tuple(O.round_h16(b) for b in T.problems.bounds_arrays(prob))
    st = O.new_state(B, 12, 4, 30)
    st["x"][:, 0] = O.round_h16(x0)
    rc = O.Oracle(prob, "h16").solve(st, *bnds, xref)
    assert 0 <= rc <= B and st["iter"].min() >= 2
    for k in O.STATE_ORDER:
        assert np.array_equal(st[k], O.round_h16(st[k])), k
    st2 = O.copy_state(st)
    O.Oracle(prob, "h16", dict(O.DEFAULT_SETTINGS, max_iter=0)).solve(st2, *bnds, xref)
    for k in O.STATE_ORDER:
        assert np.array_equal(st[k], st2[k])
    assert np.all(st2["status"] == 11) and np.all(st2["iter"] == 1)


def test_oracle_refuses_dims_whose_reference_result_depends_on_alignment(oracle_mod, tinympc):
    """For a result with rows >= packet size and rows % packet size != 0 (e.g. nu = 7 in fp32) Eigen's LinearVectorized
    assignment packet-evaluates a window of rows that starts at the first 16-byte aligned element of the destination
    COLUMN, so the summation order of u.col(i) changes with i mod 4 (measured against the compiled reference for
    nx = 20, nu = 7: rows [0,4) packet/sequential at i = 0, [1,5) at i = 1, ...).  The restatement does not model
    that and must say so instead of returning almost-right numbers."""
    O = oracle_mod
    prob = tinympc.problems.random_system(20, 7, 9, seed=1)
    with pytest.raises(ValueError):
        O.Oracle(prob, np.float32)
    O.Oracle(tinympc.problems.random_system(8, 3, 7, seed=1), np.float32)  # nu < packet size: fine
    # round 4: beyond Eigen's complete-unrolling limit (3n - 1 <= 110) the restatement's sequential fallback is NOT what the compiled reference
    # does (measured for nx = 40 and nx = 64 against oracle/_ref): refused too, and the product's exact kernels stop at nx = 36 for the same reason
    for nx, nu, N in ((40, 12, 6), (64, 32, 4)):
        with pytest.raises(ValueError):
            O.Oracle(tinympc.problems.random_system(nx, nu, N, seed=nx * 31 + nu), np.float32)
        if O.have_ref(np.float32, nx, nu, N):
            prob = tinympc.problems.random_system(nx, nu, N, seed=nx * 31 + nu)
            rng = np.random.default_rng(nx)
            st0 = O.new_state(3, nx, nu, N)
            for k in O.STATE_ORDER:
                st0[k][:] = (rng.standard_normal(st0[k].shape) * 0.3).astype(np.float32)
            a, b = O.copy_state(st0), O.copy_state(st0)
            bn = tinympc.problems.bounds_arrays(prob)
            xr = (rng.standard_normal((3, N, nx)) * 0.2).astype(np.float32)
            s = dict(max_iter=1, abs_pri_tol=0, abs_dua_tol=0)
            O.Oracle(prob, np.float32, s, allow_unpinned_dims=True).solve(a, *bn, xr)
            O.Reference(prob, np.float32, s).solve(b, *bn, xr)
            assert not np.array_equal(a["p"], b["p"]), "the sequential fallback now equals the reference: the refusal can go"


def test_oracle_and_host_riccati_under_address_and_ub_sanitizers(tmp_path):
    """sanitizers run on the CPU builds.  (a) the restatement (fp32, fp64, fp16-storage instantiations and
    the Riccati recursion) through `make -C oracle sanitize`; (b) the product's host-only Riccati routine
    (csrc/riccati.cpp) compiled stand-alone with the same flags and driven on the cartpole model."""
    import shutil
    import subprocess
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    r = subprocess.run(["make", "-s", "-C", str(root / "oracle"), "sanitize"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "_h16" in r.stdout and "ERROR" not in r.stderr
    z = np.load(__import__("helpers").GOLDEN / "riccati_cartpole.npz")  # the model of examples/codegen_cartpole.cpp:22-28
    arr = lambda a: ", ".join(repr(float(v)) for v in np.asarray(a, np.float64).T.ravel())  # column-major
    main = tmp_path / "riccati_main.cpp"
    main.write_text(f'''
#include "tinympc_batch.h"
#include <cstdio>
int main() {{
    double A[16] = {{{arr(z["A"])}}}, B[4] = {{{arr(z["B"])}}}, Q[4] = {{{arr(z["Q"])}}}, R[1] = {{{arr(z["R"])}}};
    double K[4], P[16], Qi[1], Am[16], cd[4]; int it = 0;
    int rc = tiny_riccati(4, 1, A, B, Q, R, {float(z["rho"])!r}, K, P, Qi, Am, cd, &it);
    std::printf("rc %d iters %d K %.7f %.7f %.7f %.7f\\n", rc, it, K[0], K[1], K[2], K[3]);
    return rc != 0 || it <= 0 || it >= 1000;
}}
''')
    exe = tmp_path / "riccati_san"
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                        f"-I{root / 'include'}", str(main), str(root / "accelerated-tinympc_amd" / "csrc" / "riccati.cpp"),
                        "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "iters 476" in r.stdout, r.stdout   # "Kinf converged after 476 iterations" 


@pytest.mark.parametrize("dt,nx,nu,N", CFGS)
def test_plant_step_bit_exact_vs_compiled_reference(oracle_mod, tinympc, dt, nx, nu, N):
    """x1 = work.Adyn*x0 + work.Bdyn*work.u.col(0) (examples/quadrotor_hovering.cpp:110-111): the oracle's restatement of
    Eigen's evaluation order against that very expression compiled from the reference's types (ref_shim.cpp:
    ref_plant_step), zeros and negative zeros included."""
    O = oracle_mod
    if not O.have_ref(dt, nx, nu, N):
        pytest.skip("oracle/_ref not built here (needs /root/reference)")
    pr = tinympc.problems
    if (nx, nu) == (12, 4):
        prob = pr.quadrotor(20, N)
    elif (nx, nu) == (4, 1):
        prob = pr.cartpole(N, riccati=O.riccati)
    else:
        prob = pr.random_system(nx, nu, N, seed=nx * 100 + nu, riccati=O.riccati)
    rng = np.random.default_rng(nx * 7 + nu)
    B = 3000
    x0 = (rng.standard_normal((B, nx)) * 0.5).astype(dt)
    u0 = (rng.standard_normal((B, nu)) * 0.3).astype(dt)
    for a in (x0, u0):
        a[rng.random(a.shape) < 0.15] = 0.0
        a[rng.random(a.shape) < 0.15] = -0.0
    x0[:10] = 0.0; u0[:10] = -0.0; x0[10:20] = -0.0; u0[10:20] = 0.0
    r = O.Reference(prob, dt).plant_step(x0, u0)
    o = O.Oracle(prob, dt).plant_step(x0, u0)
    assert np.array_equal(r, o) and np.array_equal(np.signbit(r), np.signbit(o))


@pytest.mark.parametrize("name", ["hover", "track", "cartpole", "dims837"])
def test_oracle_closed_loop_reproduces_reference_traces(oracle_mod, tinympc, name):
    """Whole closed loops (tiny_solve + the examples' plant step, duals reset every step, warm start, sliding windows) of
    the COMPILED REFERENCE, recorded in tests/golden/closed_loop_traces.npz, reproduced by the oracle bit for bit:
    u.col(0), iter and status of every step and the final state."""
    O, pr = oracle_mod, tinympc.problems
    z = np.load(GOLDEN_DIR / "closed_loop_traces.npz")
    prob, x0, xref_fn, steps, settings, _, _ = closed_loop_case(pr, O, z, name)
    orc = O.Oracle(prob, np.float32, settings)
    B = x0.shape[0]
    bnds = pr.bounds_arrays(prob)
    st = O.new_state(B, prob["nx"], prob["nu"], prob["N"])
    x = x0.copy()
    for k in range(steps):
        st["x"][:, 0] = x
        st["y"][:] = 0; st["g"][:] = 0
        orc.solve(st, *bnds, xref_fn(k), nthreads=4)
        assert np.array_equal(st["u"][:, 0], z[f"{name}_u0"][k]), (name, k)
        assert np.array_equal(st["iter"], z[f"{name}_iter"][k]) and np.array_equal(st["status"], z[f"{name}_status"][k]), (name, k)
        x = orc.plant_step(x, st["u"][:, 0])
    assert np.array_equal(x, z[f"{name}_x_final"])
