"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

ctypes bindings for (a) our CPU restatement of the TinyMPC ADMM hot path
(oracle/tinympc_oracle.c, restating /root/reference/src/tinympc/admm.cpp:15-152 and
codegen.cpp:254-292) and (b) the compiled reference itself (oracle/_ref/*.so, built by
oracle/Makefile from the reference sources where they lie; present only where it was
built, i.e. in the build container and — as a prebuilt, git-ignored binary — on the GPU box).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Array convention (host-visible layout of the batched API): state-type arrays are
(B, N, nx) float arrays, input-type arrays are (B, N-1, nu): instance-major, then horizon
step, then state index — i.e. each instance is the reference's column-major nx x N matrix.
Matrices (Kinf, Adyn, ...) are passed as 2-D numpy arrays in logical (row, col) indexing and
flattened column-major internally (Eigen's storage order, types.hpp:13-21).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
STATE_X = ("x", "q", "p", "v", "vnew", "g")       # (B, N, nx)
STATE_U = ("u", "r", "d", "z", "znew", "y")       # (B, N-1, nu)
STATE_ORDER = ("x", "u", "q", "r", "p", "d", "v", "vnew", "z", "znew", "g", "y")


def build(ref: bool = True) -> None:
    """Compile the oracle (and, when /root/reference exists, oracle/_ref)."""
    target = "all" if ref else "liboracle"
    subprocess.run(["make", "-s", "-C", str(HERE), target], check=True)


def _lib():
    p = HERE / "libtinympc_oracle.so"
    if not p.exists():
        build(ref=False)
    return C.CDLL(str(p))


def _colmajor(a, dt):
    return np.ascontiguousarray(np.asarray(a, dtype=dt).T).ravel()  # column-major flat copy


def _ptr(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


class _Dtype:
    """np.float32 / np.float64 = the reference's arithmetic; the string "h16" = fp16 storage with fp32 arithmetic
    (float32 arrays holding fp16-representable values; see the header of tinympc_oracle_impl.h)."""

    def __init__(self, dt):
        self.h16 = isinstance(dt, str) and dt in ("h16", "h16d")   # "h16d": fp16 primal arrays, fp32 duals y, g
        self.np = np.dtype(np.float32 if self.h16 else dt)
        self.ct = C.c_float if self.np == np.float32 else C.c_double
        self.suf = dt if self.h16 else ("f32" if self.np == np.float32 else "f64")


def round_h16(a):
    """What an fp16 store followed by a load leaves of float32 values (round to nearest even, subnormals kept)."""
    with np.errstate(over="ignore"):
        return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


def _problem_struct(T):
    class OracleProblem(C.Structure):
        _fields_ = [("nx", C.c_int), ("nu", C.c_int), ("N", C.c_int), ("rho", T.ct),
                    ("Kinf", C.POINTER(T.ct)), ("Pinf", C.POINTER(T.ct)), ("Quu_inv", C.POINTER(T.ct)),
                    ("AmBKt", C.POINTER(T.ct)), ("Adyn", C.POINTER(T.ct)), ("Bdyn", C.POINTER(T.ct)),
                    ("Q", C.POINTER(T.ct)), ("abs_pri_tol", T.ct), ("abs_dua_tol", T.ct),
                    ("max_iter", C.c_int), ("check_termination", C.c_int),
                    ("en_state_bound", C.c_int), ("en_input_bound", C.c_int),
                    ("en_uref", C.c_int), ("en_coeff_d2p", C.c_int), ("R", C.POINTER(T.ct)), ("coeff_d2p", C.POINTER(T.ct))]
    return OracleProblem


def _work_struct(T):
    P = C.POINTER(T.ct)

    class OracleWork(C.Structure):
        _fields_ = ([(n, P) for n in STATE_ORDER] + [(n, P) for n in ("u_min", "u_max", "x_min", "x_max", "Xref")] +
                    [("primal_residual_state", T.ct), ("primal_residual_input", T.ct),
                     ("dual_residual_state", T.ct), ("dual_residual_input", T.ct), ("status", C.c_int), ("iter", C.c_int),
                     ("Uref", P)])
    return OracleWork


def _batch_struct(T):
    P = C.POINTER(T.ct)

    class OracleBatch(C.Structure):
        _fields_ = ([("batch", C.c_int)] + [(n, P) for n in STATE_ORDER] +
                    [(n, P) for n in ("u_min", "u_max", "x_min", "x_max", "Xref")] +
                    [("bound_stride_x", C.c_longlong), ("bound_stride_u", C.c_longlong),
                     ("xref_stride", C.c_longlong), ("residuals", P),
                     ("status", C.POINTER(C.c_int)), ("iter", C.POINTER(C.c_int)),
                     ("Uref", P), ("uref_stride", C.c_longlong)])
    return OracleBatch


DEFAULT_SETTINGS = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1,
                        en_state_bound=1, en_input_bound=1)  # examples/quadrotor_hovering.cpp:73-78
# Optional terms (commented out in the reference, admm.cpp:20 and :79; off unless asked for): settings["en_uref"] with
# prob["R"] and set_uref(), settings["en_coeff_d2p"] with prob["coeff_d2p"].


def new_state(B, nx, nu, N, dtype=np.float32):
    """All-zero workspace (what the reference examples start from, quadrotor_hovering.cpp:52-71)."""
    st = {k: np.zeros((B, N, nx), dtype) for k in STATE_X}
    st.update({k: np.zeros((B, N - 1, nu), dtype) for k in STATE_U})
    st["residuals"] = np.zeros((B, 4), dtype)
    st["status"] = np.zeros(B, np.int32)
    st["iter"] = np.zeros(B, np.int32)
    return st


def copy_state(st):
    return {k: v.copy() for k, v in st.items()}


def _bcast(a, B, shape, dt):
    """Return (contiguous array, per-instance stride in elements; 0 = shared)."""
    a = np.ascontiguousarray(np.asarray(a, dtype=dt))
    n = int(np.prod(shape))
    if a.size == n:
        return a.reshape(shape), 0
    assert a.size == B * n, (a.shape, B, shape)
    return a.reshape((B,) + tuple(shape)), n


class _Solver:
    """Common driver for the oracle and for a compiled-reference library."""

    def __init__(self, prob, dtype=np.float32, settings=None):
        self.T = _Dtype(dtype)
        self.prob = prob
        self.nx, self.nu, self.N = prob["nx"], prob["nu"], prob["N"]
        self.settings = dict(DEFAULT_SETTINGS)
        if settings:
            self.settings.update(settings)
        dt = self.T.np
        self._m = {k: _colmajor(prob[k], dt) for k in ("Kinf", "Pinf", "Quu_inv", "AmBKt", "Adyn", "Bdyn")}
        self._m["Q"] = np.ascontiguousarray(np.asarray(prob["Q"], dt).ravel())
        self._m["R"] = np.ascontiguousarray(np.asarray(prob.get("R", np.zeros(self.nu)), dt).ravel())
        self._m["coeff_d2p"] = _colmajor(prob.get("coeff_d2p", np.zeros((self.nx, self.nu))), dt)
        assert self._m["R"].size == self.nu and self._m["coeff_d2p"].size == self.nx * self.nu
        self.rho = dt.type(prob["rho"])
        self._uref = None

    def set_uref(self, Uref):
        """Input reference of the optional Uref term: (N-1, nu) shared or (B, N-1, nu); used only when settings["en_uref"]."""
        self._uref = None if Uref is None else np.ascontiguousarray(np.asarray(Uref, self.T.np))

    def _uref_arg(self, B):
        if not self.settings.get("en_uref"):
            return None, 0
        assert self._uref is not None, "en_uref needs set_uref()"
        return _bcast(self._uref, B, (self.N - 1, self.nu), self.T.np)

    def _prep(self, st, x_min, x_max, u_min, u_max, Xref):
        dt = self.T.np
        B = st["x"].shape[0]
        for k in STATE_ORDER + ("residuals",):
            assert st[k].dtype == dt and st[k].flags.c_contiguous, k
        xmn, sx1 = _bcast(x_min, B, (self.N, self.nx), dt)
        xmx, sx2 = _bcast(x_max, B, (self.N, self.nx), dt)
        umn, su1 = _bcast(u_min, B, (self.N - 1, self.nu), dt)
        umx, su2 = _bcast(u_max, B, (self.N - 1, self.nu), dt)
        xr, sr = _bcast(Xref, B, (self.N, self.nx), dt)
        assert sx1 == sx2 and su1 == su2
        return B, (umn, umx, xmn, xmx, xr), (sx1, su1, sr)


class Oracle(_Solver):
    """Our CPU restatement (oracle/tinympc_oracle.c)."""

    kind = "port"

    def __init__(self, prob, dtype=np.float32, settings=None, allow_unpinned_dims=False):
        """allow_unpinned_dims: accept dimensions for which this restatement is NOT bit-exact with the reference (see
        below) — only for uses that need a precision yardstick, never for a parity claim."""
        super().__init__(prob, dtype, settings)
        ps = 2 if self.T.suf == "f64" else 4
        for name, rows in (("nx", self.nx), ("nu", self.nu)):
            if rows > ps and rows % ps and not allow_unpinned_dims:
                raise ValueError(f"{name}={rows}: the reference's summation order for results with rows >= {ps} and rows % {ps} != 0 "
                                 "depends on the 16-byte alignment of each destination column (Eigen LinearVectorized "
                                 "assignment) and is not restated; see the header of tinympc_oracle_impl.h")
            if 3 * rows - 1 > 110 and not allow_unpinned_dims:
                # round 4: measured against the compiled reference for nx = 40 and nx = 64 — beyond Eigen's complete-unrolling limit
                # (3n - 1 <= EIGEN_UNROLLING_LIMIT = 110, n <= 37) the coefficient-evaluated products are NOT the plain sequential loop
                # this restatement falls back to; nothing above 37 is pinned, so nothing above 37 is claimed
                raise ValueError(f"{name}={rows}: reduction orders beyond Eigen's unrolling limit (n > 37) are not restated")
        self.lib = _lib()
        self.PS = _problem_struct(self.T)
        self.BS = _batch_struct(self.T)
        self.fn = getattr(self.lib, f"oracle_solve_batch_{self.T.suf}")
        self.fn.restype = C.c_int
        self.fn.argtypes = [C.POINTER(self.PS), C.POINTER(self.BS), C.c_int]

    def _pstruct(self):
        s = self.settings
        ct = self.T.ct
        return self.PS(self.nx, self.nu, self.N, self.rho, *[_ptr(self._m[k], ct) for k in
                       ("Kinf", "Pinf", "Quu_inv", "AmBKt", "Adyn", "Bdyn", "Q")],
                       s["abs_pri_tol"], s["abs_dua_tol"], s["max_iter"], s["check_termination"],
                       s["en_state_bound"], s["en_input_bound"], int(bool(s.get("en_uref"))), int(bool(s.get("en_coeff_d2p"))),
                       _ptr(self._m["R"], ct), _ptr(self._m["coeff_d2p"], ct))

    STEP_FUNCTIONS = ("forward_pass", "update_slack", "update_dual", "update_linear_cost", "termination_condition",
                      "backward_pass_grad")

    def step(self, fn, st, x_min, x_max, u_min, u_max, Xref):
        """Apply ONE of the six step functions of admm.hpp:12-18 to every instance, in place on `st`.
        Returns the per-instance return value (only termination_condition returns something: a bool array)."""
        assert fn in self.STEP_FUNCTIONS
        B, ins, (sx, su, sr) = self._prep(st, x_min, x_max, u_min, u_max, Xref)
        ct, WS = self.T.ct, _work_struct(self.T)
        f = getattr(self.lib, f"oracle_{fn}_{self.T.suf}")
        f.argtypes, f.restype = [C.POINTER(self.PS), C.POINTER(WS)], C.c_int
        ps = self._pstruct()
        nxt, nut = self.nx * self.N, self.nu * (self.N - 1)
        out = np.zeros(B, bool)
        ur, sur = self._uref_arg(B)
        for b in range(B):
            def sl(a, stride, n):
                flat = a.reshape(-1)
                return _ptr(flat[b * stride:b * stride + n] if stride else flat[:n], ct)
            ptrs = [sl(st[k], nxt if k in STATE_X else nut, nxt if k in STATE_X else nut) for k in STATE_ORDER]
            inp = [sl(ins[0], su, nut), sl(ins[1], su, nut), sl(ins[2], sx, nxt), sl(ins[3], sx, nxt), sl(ins[4], sr, nxt)]
            r = st["residuals"][b]
            w = WS(*ptrs, *inp, r[0], r[1], r[2], r[3], int(st["status"][b]), int(st["iter"][b]),
                   sl(ur, sur, nut) if ur is not None else None)
            rv = f(C.byref(ps), C.byref(w))
            if fn == "termination_condition":
                out[b] = bool(rv)
                st["residuals"][b] = (w.primal_residual_state, w.primal_residual_input, w.dual_residual_state,
                                      w.dual_residual_input)
        return out

    def plant_step(self, x0, u0):
        """x1 = Adyn*x0 + Bdyn*u0 per instance, in the order of the reference's examples (quadrotor_hovering.cpp:110-111).
        Always the scalar type's own arithmetic (the simulated plant is not a work array: no fp16 rounding)."""
        dt, ct = self.T.np, self.T.ct
        x0 = np.ascontiguousarray(x0, dt).reshape(-1, self.nx)
        u0 = np.ascontiguousarray(u0, dt).reshape(-1, self.nu)
        x1 = np.empty_like(x0)
        suf = "f32" if self.T.h16 else self.T.suf
        f = getattr(self.lib, f"oracle_plant_step_batch_{suf}")
        f.restype = None
        f.argtypes = [C.POINTER(self.PS), C.c_int, C.POINTER(ct), C.POINTER(ct), C.POINTER(ct)]  # the _h16 struct has the f32 layout
        ps = self._pstruct()
        f(C.byref(ps), x0.shape[0], _ptr(x0, ct), _ptr(u0, ct), _ptr(x1, ct))
        return x1

    def solve(self, st, x_min, x_max, u_min, u_max, Xref, nthreads=1, ftz=False):
        """One tiny_solve() per instance, in place on `st`.  Returns #instances that hit max_iter."""
        B, ins, (sx, su, sr) = self._prep(st, x_min, x_max, u_min, u_max, Xref)
        ct = self.T.ct
        self.lib.oracle_set_ftz_daz(1 if ftz else 0)
        ur, sur = self._uref_arg(B)
        bs = self.BS(B, *[_ptr(st[k], ct) for k in STATE_ORDER], *[_ptr(a, ct) for a in ins],
                     sx, su, sr, _ptr(st["residuals"], ct), _ptr(st["status"], C.c_int), _ptr(st["iter"], C.c_int),
                     _ptr(ur, ct) if ur is not None else None, sur)
        ps = self._pstruct()
        rc = self.fn(C.byref(ps), C.byref(bs), int(nthreads))
        self.lib.oracle_set_ftz_daz(0)
        return rc


def ref_lib_path(dtype, nx, nu, N):
    suf = "f32" if np.dtype(dtype) == np.float32 else "f64"
    return HERE / "_ref" / f"libtinympc_ref_{suf}_{nx}_{nu}_{N}.so"


def have_ref(dtype, nx, nu, N):
    return ref_lib_path(dtype, nx, nu, N).exists()


class Reference(_Solver):
    """The compiled reference (Eigen, src/tinympc/admm.cpp) behind oracle/ref_shim.cpp."""

    kind = "reference"

    def __init__(self, prob, dtype=np.float32, settings=None):
        super().__init__(prob, dtype, settings)
        path = ref_lib_path(dtype, self.nx, self.nu, self.N)
        if not path.exists():
            raise FileNotFoundError(f"{path} (build it with `make -C oracle ref` where /root/reference exists)")
        self.lib = C.CDLL(str(path))
        d = (C.c_int * 4)()
        self.lib.ref_dims(C.byref(d, 0), C.byref(d, 4), C.byref(d, 8), C.byref(d, 12))
        assert (d[0], d[1], d[2]) == (self.nx, self.nu, self.N) and bool(d[3]) == (self.T.suf == "f64")
        ct, P = self.T.ct, C.POINTER(self.T.ct)
        self.lib.ref_set_problem.argtypes = [ct] + [P] * 7
        self.lib.ref_set_problem.restype = None
        self.lib.ref_set_settings.argtypes = [ct, ct, C.c_int, C.c_int, C.c_int, C.c_int]
        self.lib.ref_set_settings.restype = None
        self.lib.ref_solve_batch.argtypes = ([C.c_int] + [P] * 17 + [C.c_longlong] * 3 +
                                             [P, C.POINTER(C.c_int), C.POINTER(C.c_int)])
        self.lib.ref_solve_batch.restype = C.c_int

    def plant_step(self, x0, u0):
        """The examples' own Eigen expression x1 = work.Adyn*x0 + work.Bdyn*work.u.col(0) (ref_shim.cpp: ref_plant_step)."""
        dt, ct = self.T.np, self.T.ct
        x0 = np.ascontiguousarray(x0, dt).reshape(-1, self.nx)
        u0 = np.ascontiguousarray(u0, dt).reshape(-1, self.nu)
        x1 = np.empty_like(x0)
        self.lib.ref_set_problem(self.rho, *[_ptr(self._m[k], ct) for k in
                                 ("Kinf", "Pinf", "Quu_inv", "AmBKt", "Adyn", "Bdyn", "Q")])
        self.lib.ref_plant_step_batch.restype = None
        self.lib.ref_plant_step_batch.argtypes = [C.c_int] + [C.POINTER(ct)] * 3
        self.lib.ref_plant_step_batch(x0.shape[0], _ptr(x0, ct), _ptr(u0, ct), _ptr(x1, ct))
        return x1

    def solve(self, st, x_min, x_max, u_min, u_max, Xref, nthreads=1, ftz=False):
        assert nthreads == 1, "the reference is single-threaded (one global solver, tiny_wrapper.cpp)"
        assert not self.settings.get("en_uref") and not self.settings.get("en_coeff_d2p"), "commented out in the reference"
        B, ins, (sx, su, sr) = self._prep(st, x_min, x_max, u_min, u_max, Xref)
        ct = self.T.ct
        s = self.settings
        self.lib.ref_set_problem(self.rho, *[_ptr(self._m[k], ct) for k in
                                 ("Kinf", "Pinf", "Quu_inv", "AmBKt", "Adyn", "Bdyn", "Q")])
        self.lib.ref_set_settings(s["abs_pri_tol"], s["abs_dua_tol"], s["max_iter"], s["check_termination"],
                                  s["en_state_bound"], s["en_input_bound"])
        return self.lib.ref_solve_batch(B, *[_ptr(st[k], ct) for k in STATE_ORDER], *[_ptr(a, ct) for a in ins],
                                        sx, su, sr, _ptr(st["residuals"], ct), _ptr(st["status"], C.c_int),
                                        _ptr(st["iter"], C.c_int))


def riccati(nx, nu, A, B, Q, R, rho):
    """fp64 Riccati cache precompute, restating codegen.cpp:254-292.  Returns dict + iteration count."""
    lib = _lib()
    P = C.POINTER(C.c_double)
    lib.oracle_riccati_f64.argtypes = [C.c_int, C.c_int, P, P, P, P, C.c_double, P, P, P, P, P]
    lib.oracle_riccati_f64.restype = C.c_int
    a, b = _colmajor(A, np.float64), _colmajor(B, np.float64)
    q, r = np.ascontiguousarray(Q, np.float64).ravel(), np.ascontiguousarray(R, np.float64).ravel()
    K, Pm, Qi, Am, cd = (np.zeros(nu * nx), np.zeros(nx * nx), np.zeros(nu * nu), np.zeros(nx * nx),
                         np.zeros(nx * nu))
    it = lib.oracle_riccati_f64(nx, nu, _ptr(a, C.c_double), _ptr(b, C.c_double), _ptr(q, C.c_double),
                                _ptr(r, C.c_double), float(rho), _ptr(K, C.c_double), _ptr(Pm, C.c_double),
                                _ptr(Qi, C.c_double), _ptr(Am, C.c_double), _ptr(cd, C.c_double))
    if it < 0:
        raise RuntimeError(f"oracle_riccati failed rc={it}")
    return dict(Kinf=K.reshape(nx, nu).T.copy(), Pinf=Pm.reshape(nx, nx).T.copy(),
                Quu_inv=Qi.reshape(nu, nu).T.copy(), AmBKt=Am.reshape(nx, nx).T.copy(),
                coeff_d2p=cd.reshape(nu, nx).T.copy()), it
