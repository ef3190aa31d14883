"""GPU-box twins of the pin tests (round 3): the driver's `-m gpu` run deselects tests/test_oracle.py, so the evidence that the
checker itself equals the compiled reference never appeared in its record.  oracle/_ref/*.so (the reference compiled from
/root/reference by oracle/Makefile) travels with the tree, so on the GPU box

  * the oracle-vs-compiled-reference pin of tests/test_oracle.py runs again, unchanged, and
  * the HIP kernels are compared with the COMPILED REFERENCE directly (not through the oracle) on the same random warm
    states: every work array, the residuals, status and iteration counts, bit for bit, zero signs included.

Both skip (with the reason) where oracle/_ref is absent."""
import numpy as np
import pytest

from helpers import STATE_ORDER
import test_oracle as TO

pytestmark = pytest.mark.gpu

SETTINGS = (dict(max_iter=1, abs_pri_tol=0, abs_dua_tol=0), dict(max_iter=12, abs_pri_tol=0, abs_dua_tol=0),
            dict(max_iter=60, check_termination=3), dict(max_iter=5, en_state_bound=0, en_input_bound=0))


@pytest.mark.parametrize("dt,nx,nu,N", TO.CFGS)
def test_oracle_pin_on_the_gpu_box(oracle_mod, tinympc, dt, nx, nu, N):
    TO.test_oracle_bit_exact_vs_compiled_reference(oracle_mod, tinympc, dt, nx, nu, N)


def _problem(pr, O, nx, nu, N):
    if (nx, nu) == (12, 4):
        return pr.quadrotor(20, N)
    if (nx, nu) == (4, 1):
        return pr.cartpole(N, riccati=O.riccati)
    return pr.random_system(nx, nu, N, seed=nx * 100 + nu, riccati=O.riccati)


def _warm_state(O, rng, B, nx, nu, N, dt):
    st0 = O.new_state(B, nx, nu, N, dt)
    for k in STATE_ORDER:
        st0[k][:] = (rng.standard_normal(st0[k].shape) * 0.3).astype(dt)
    for k in ("x", "d", "v", "z", "g", "y"):
        st0[k][rng.random(st0[k].shape) < 0.1] = 0.0
        st0[k][rng.random(st0[k].shape) < 0.1] = -0.0
    st0["x"][0, 0] = -0.0; st0["g"][0] = 0.0; st0["y"][0] = 0.0; st0["d"][0] = 0.0
    return st0


@pytest.mark.parametrize("dt,nx,nu,N", TO.CFGS)
def test_hip_kernels_bit_exact_vs_compiled_reference(oracle_mod, tinympc, dt, nx, nu, N):
    """tiny_batch_solve (exact arithmetic, the automatic kernel of the class) against oracle/_ref on random warm states."""
    O, T = oracle_mod, tinympc
    if not O.have_ref(dt, nx, nu, N):
        pytest.skip("oracle/_ref not built here (needs /root/reference at build time)")
    pr = T.problems
    prob = _problem(pr, O, nx, nu, N)
    rng = np.random.default_rng(1000 + nx + nu + N)
    B = 21
    st0 = _warm_state(O, rng, B, nx, nu, N, dt)
    xref = (rng.standard_normal((B, N, nx)) * 0.2).astype(dt)
    xref[0] = 0.0
    bnds = pr.bounds_arrays(prob, dt)
    for settings in SETTINGS:
        full = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1, en_state_bound=1, en_input_bound=1)
        full.update(settings)
        ref = O.copy_state(st0)
        rr = O.Reference(prob, dt, settings).solve(ref, *bnds, xref)
        try:
            sol = (T.TinyBatchSolver if dt == np.float32 else T.TinyBatchSolver64)(prob, B, settings=full)
        except T.TinyBatchError as e:
            pytest.skip(f"no kernel for this class in {np.dtype(dt).name}: {e}")
        try:
            if dt == np.float32:
                # exact arithmetic or an error, never fma: round 4 made that the contract of the automatic choice (a class outside the compiled
                # lists gets the run-time-dimension exact kernel), and tiny_batch_arithmetic() states it
                assert sol.arithmetic() == "exact", sol.kernel_name()
            sol.set_bounds(*bnds)
            sol.set_xref(xref)
            sol.set_state(st0)
            rc = sol.solve()
        except T.TinyBatchError as e:
            sol.close()
            pytest.skip(f"no exact kernel for this class in {np.dtype(dt).name}: {e}")
        got = sol.get_state()
        name = sol.kernel_name()
        sol.close()
        assert (rc > 0) == (rr > 0), (name, settings)
        for k in STATE_ORDER + ("residuals", "status", "iter"):
            assert np.array_equal(got[k], ref[k]), (name, settings, k)
            if got[k].dtype.kind == "f":
                assert np.array_equal(np.signbit(got[k]), np.signbit(ref[k])), (name, settings, k, "sign of a zero")
