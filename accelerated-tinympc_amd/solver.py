"""ctypes host mirror of the C-ABI (include/tinympc_batch.h).

`TinyBatchSolver` keeps the names and call sequence of the reference's foreign-language
wrapper (src/tinympc/tiny_wrapper.hpp:14-23: set_x0, set_xref, set_umin/umax, set_xmin/xmax,
reset_dual_variables, call_tiny_solve, get_x, get_u), batched: every array gains a leading
instance axis.  All compute happens in the HIP library; there is no CPU fallback — a missing
library or a missing GPU raises.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

import os

from . import build as _build

PKG = Path(__file__).resolve().parent
LIB_PATH = PKG / "lib" / "libtinympc_hip.so"

ARRAY_IDS = {"x": 0, "u": 1, "q": 2, "r": 3, "p": 4, "d": 5, "v": 6, "vnew": 7, "z": 8, "znew": 9, "g": 10, "y": 11}
X_FAMILY = ("x", "q", "p", "v", "vnew", "g")
TINY_SOLVED, TINY_UNSOLVED = 1, 11

_lib = None


class TinyBatchError(RuntimeError):
    pass


def load_library(build_if_missing: bool = False) -> C.CDLL:
    """dlopen the in-tree HIP library and declare its prototypes.  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        if build_if_missing:
            _build.build()
        else:
            raise TinyBatchError(f"{LIB_PATH} is missing: run `python accelerated-tinympc_amd/build.py` "
                                 "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    # developer aid: TINYMPC_HIP_LIB names another build of the SAME library (tools/ab_kernels.py times kernel variants)
    override = os.environ.get("TINYMPC_HIP_LIB")
    if override:
        import warnings
        warnings.warn(f"TINYMPC_HIP_LIB is set: loading {override} instead of {LIB_PATH}", RuntimeWarning)
    lib = C.CDLL(override or str(LIB_PATH))
    F, I, P = C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_void_p
    D = C.POINTER(C.c_double)
    sig = {
        "tiny_batch_create": [C.POINTER(P), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int],
        "tiny_batch_set_stream": [P, P],
        "tiny_batch_synchronize": [P],
        "tiny_batch_set_cache": [P, C.c_float, F, F, F, F],
        "tiny_batch_set_dynamics": [P, F, F, F],
        "tiny_batch_set_settings": [P, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int],
        "tiny_batch_set_x0": [P, F],
        "tiny_batch_set_xref": [P, F, C.c_int],
        "tiny_batch_set_xref_window": [P, F, C.c_int, I],
        "tiny_batch_set_umin": [P, F, C.c_int], "tiny_batch_set_umax": [P, F, C.c_int],
        "tiny_batch_set_xmin": [P, F, C.c_int], "tiny_batch_set_xmax": [P, F, C.c_int],
        "tiny_batch_reset_dual_variables": [P],
        "tiny_batch_solve": [P], "tiny_batch_solve_async": [P], "tiny_batch_wait": [P, I],
        "tiny_batch_get_x": [P, F], "tiny_batch_get_u": [P, F],
        "tiny_batch_forward_pass": [P], "tiny_batch_update_slack": [P], "tiny_batch_update_dual": [P],
        "tiny_batch_update_linear_cost": [P], "tiny_batch_backward_pass_grad": [P], "tiny_batch_termination_condition": [P, I],
        "tiny_batch_get_status": [P, I, I, F], "tiny_batch_set_status": [P, I, I, F],
        "tiny_batch_set_array": [P, C.c_int, F], "tiny_batch_get_array": [P, C.c_int, F],
        "tiny_batch_reset_workspace": [P],
        "tiny_batch_set_x0_device": [P, P], "tiny_batch_get_u0_device": [P, P],
        "tiny_batch_mpc_step_async": [P, C.c_int], "tiny_batch_get_x0": [P, F],
        "tiny_batch_mpc_run_async": [P, C.c_int, C.c_int], "tiny_batch_mpc_run_traj_async": [P, C.c_int, C.c_int, P], "tiny_batch_mpc_run_traj": [P, C.c_int, C.c_int, F],
        "tiny_batch_enable_timing": [P, C.c_int], "tiny_batch_last_solve_ms": [P, F],
        "tiny_batch_select_kernel": [P, C.c_int], "tiny_batch_arithmetic": [P], "tiny_batch_debug_graph_captures": [P], "tiny_batch_set_storage": [P, C.c_int], "tiny_batch_set_storage_ex": [P, C.c_int, C.c_int],
        "tiny_batch_set_row_kernel": [P, C.c_int],
        "tiny_batch_set_dispatch": [P, C.c_int], "tiny_batch_set_tile_queue": [P, C.c_int], "tiny_batch_set_dispatch_order_device": [P, P], "tiny_batch_dispatch_applied": [P],
        "tiny_batch_set_optional_terms": [P, C.c_int, C.c_int], "tiny_batch_set_input_cost": [P, F],
        "tiny_batch_set_coeff_d2p": [P, F], "tiny_batch_set_uref": [P, F, C.c_int],
        "tiny_batch_group_solve": [C.POINTER(P), C.c_int, I],
        "tiny_batch_group_gather_u0": [C.POINTER(P), C.c_int, C.c_int, P], "tiny_batch_group_get_u0": [C.POINTER(P), C.c_int, F],
        "tiny_batch_set_array_device": [P, C.c_int, P], "tiny_batch_get_array_device": [P, C.c_int, P],
        "tiny_batch_set_xref_device": [P, P, C.c_int],
        "tiny_riccati": [C.c_int, C.c_int, D, D, D, D, C.c_double, D, D, D, D, D, I],
    }
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = args, C.c_int
    # fp64 library (include/tinympc_batch64.h)
    sig64 = {
        "tiny_batch64_create": [C.POINTER(P), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int],
        "tiny_batch64_set_cache": [P, C.c_double, D, D, D, D], "tiny_batch64_set_dynamics": [P, D, D, D],
        "tiny_batch64_set_settings": [P, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int],
        "tiny_batch64_set_x0": [P, D], "tiny_batch64_set_xref": [P, D, C.c_int],
        "tiny_batch64_set_xmin": [P, D, C.c_int], "tiny_batch64_set_xmax": [P, D, C.c_int],
        "tiny_batch64_set_umin": [P, D, C.c_int], "tiny_batch64_set_umax": [P, D, C.c_int],
        "tiny_batch64_reset_dual_variables": [P], "tiny_batch64_solve": [P],
        "tiny_batch64_set_array": [P, C.c_int, D], "tiny_batch64_get_array": [P, C.c_int, D],
        "tiny_batch64_get_status": [P, I, I, D], "tiny_batch64_set_status": [P, I, I, D],
        "tiny_batch64_select_kernel": [P, C.c_int], "tiny_batch64_mpc_step": [P], "tiny_batch64_get_first_columns": [P, D, D],
        "tiny_batch64_forward_pass": [P], "tiny_batch64_update_slack": [P], "tiny_batch64_update_dual": [P],
        "tiny_batch64_update_linear_cost": [P], "tiny_batch64_backward_pass_grad": [P], "tiny_batch64_termination_condition": [P, I],
    }
    for name, args in sig64.items():
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = args, C.c_int
    lib.tiny_batch64_destroy.argtypes, lib.tiny_batch64_destroy.restype = [P], None
    lib.tiny_batch64_last_error.argtypes, lib.tiny_batch64_last_error.restype = [], C.c_char_p
    lib.tiny_batch64_kernel_name.argtypes, lib.tiny_batch64_kernel_name.restype = [P], C.c_char_p
    lib.tiny_batch_debug_guards.argtypes, lib.tiny_batch_debug_guards.restype = [C.c_int], C.c_int
    lib.tiny_batch_debug_check.argtypes, lib.tiny_batch_debug_check.restype = [], C.c_longlong
    lib.tiny_batch_debug_poke.argtypes, lib.tiny_batch_debug_poke.restype = [P, C.c_int], C.c_int
    lib.tiny_batch_destroy.argtypes, lib.tiny_batch_destroy.restype = [P], None
    lib.tiny_batch_last_error.argtypes, lib.tiny_batch_last_error.restype = [], C.c_char_p
    lib.tiny_batch_kernel_name.argtypes, lib.tiny_batch_kernel_name.restype = [P], C.c_char_p
    lib.tiny_batch_closed_loop_kernel_name.argtypes, lib.tiny_batch_closed_loop_kernel_name.restype = [P], C.c_char_p
    _lib = lib
    return lib


def debug_guards(on: bool):
    """tiny_batch_debug_guards: guard zones around every device allocation made from now on (the debug mode of SURVEY.md section 5)."""
    load_library().tiny_batch_debug_guards(1 if on else 0)


def debug_check() -> int:
    """tiny_batch_debug_check: guard words overwritten so far by any kernel (0 = no out-of-bounds write); raises on a HIP error."""
    n = load_library().tiny_batch_debug_check()
    if n < 0:
        raise TinyBatchError(f"tiny_batch_debug_check: {load_library().tiny_batch_last_error().decode()} ({n})")
    return int(n)


def exported_symbols():
    """Names declared in include/tinympc_batch.h and include/tinympc_batch64.h (used by the CPU-side ABI test)."""
    import re
    txt = (PKG.parent / "include" / "tinympc_batch.h").read_text() + (PKG.parent / "include" / "tinympc_batch64.h").read_text()
    return sorted(set(re.findall(r"\b(tiny_(?:batch(?:64)?_[a-z0-9_]+|riccati))\s*\(", txt)))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _colmajor(m):
    return np.ascontiguousarray(np.asarray(m, dtype=np.float32).T).ravel()


def riccati(nx, nu, A, B, Q, R, rho):
    """Host fp64 cache precompute (csrc/riccati.cpp; restates codegen.cpp:254-292)."""
    lib = load_library()
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    a = np.ascontiguousarray(np.asarray(A, np.float64).T).ravel()
    b = np.ascontiguousarray(np.asarray(B, np.float64).T).ravel()
    q = np.ascontiguousarray(Q, np.float64).ravel()
    r = np.ascontiguousarray(R, np.float64).ravel()
    K, Pm, Qi, Am, cd = (np.zeros(nu * nx), np.zeros(nx * nx), np.zeros(nu * nu), np.zeros(nx * nx), np.zeros(nx * nu))
    it = C.c_int(0)
    rc = lib.tiny_riccati(nx, nu, dp(a), dp(b), dp(q), dp(r), float(rho), dp(K), dp(Pm), dp(Qi), dp(Am), dp(cd), C.byref(it))
    if rc != 0:
        raise TinyBatchError(f"tiny_riccati failed rc={rc}")
    return dict(Kinf=K.reshape(nx, nu).T.copy(), Pinf=Pm.reshape(nx, nx).T.copy(), Quu_inv=Qi.reshape(nu, nu).T.copy(),
                AmBKt=Am.reshape(nx, nx).T.copy(), coeff_d2p=cd.reshape(nu, nx).T.copy(), iters=it.value)


def solve_group(solvers) -> int:
    """tiny_batch_group_solve: several problem classes (one TinyBatchSolver each) solved as one overlapping group.
    Returns the number of instances, over all classes, that hit max_iter."""
    lib = load_library()
    hs = (C.c_void_p * len(solvers))(*[s._h for s in solvers])
    n = C.c_int(0)
    rc = lib.tiny_batch_group_solve(hs, len(solvers), C.byref(n))
    if rc < 0:
        raise TinyBatchError(f"rc={rc}: {lib.tiny_batch_last_error().decode()}")
    return n.value


class TinyBatchSolver:
    """Device-resident batch of TinyMPC problem instances of one class (nx, nu, N)."""

    def __init__(self, prob: dict, batch: int, device: int = 0, settings: dict | None = None):
        self.lib = load_library()
        self.nx, self.nu, self.N, self.B = int(prob["nx"]), int(prob["nu"]), int(prob["N"]), int(batch)
        self._h = C.c_void_p()
        self._check(self.lib.tiny_batch_create(C.byref(self._h), self.nx, self.nu, self.N, self.B, int(device)))
        self.set_cache(prob)
        a, b, q = _colmajor(prob["Adyn"]), _colmajor(prob["Bdyn"]), _f32(np.asarray(prob["Q"]).ravel())
        self._check(self.lib.tiny_batch_set_dynamics(self._h, _fp(a), _fp(b), _fp(q)))
        s = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1, en_state_bound=1,
                 en_input_bound=1)  # examples/quadrotor_hovering.cpp:73-78
        if settings:
            s.update(settings)
        self.set_settings(**s)

    def set_cache(self, prob: dict):
        """TinyCache (types.hpp:26-34): rho, Kinf, Pinf, Quu_inv, AmBKt of `prob` (logical row/col numpy matrices)."""
        k, p, qi, am = (_colmajor(prob[n]) for n in ("Kinf", "Pinf", "Quu_inv", "AmBKt"))
        self._check(self.lib.tiny_batch_set_cache(self._h, float(prob["rho"]), _fp(k), _fp(p), _fp(qi), _fp(am)))

    # -- plumbing ---------------------------------------------------------------------------
    def _check(self, rc):
        if rc < 0:
            raise TinyBatchError(f"rc={rc}: {self.lib.tiny_batch_last_error().decode()}")
        return rc

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.tiny_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _xshape(self, name):
        return (self.B, self.N, self.nx) if name in X_FAMILY else (self.B, self.N - 1, self.nu)

    # -- TinySettings -----------------------------------------------------------------------
    def set_settings(self, abs_pri_tol, abs_dua_tol, max_iter, check_termination, en_state_bound, en_input_bound):
        self.settings = dict(abs_pri_tol=abs_pri_tol, abs_dua_tol=abs_dua_tol, max_iter=max_iter,
                             check_termination=check_termination, en_state_bound=en_state_bound,
                             en_input_bound=en_input_bound)
        self._check(self.lib.tiny_batch_set_settings(self._h, abs_pri_tol, abs_dua_tol, max_iter, check_termination,
                                                     en_state_bound, en_input_bound))

    # -- wrapper twins (tiny_wrapper.hpp:14-23) ------------------------------------------------
    def set_x0(self, x0):
        a = _f32(x0); assert a.shape == (self.B, self.nx), a.shape
        self._check(self.lib.tiny_batch_set_x0(self._h, _fp(a)))

    def _set_steps(self, fn, arr, steps, dim):
        a = _f32(arr)
        if a.shape == (steps, dim):
            shared = 1
        else:
            assert a.shape == (self.B, steps, dim), (a.shape, (self.B, steps, dim))
            shared = 0
        self._check(fn(self._h, _fp(a), shared))

    def set_xref(self, xref):
        self._set_steps(self.lib.tiny_batch_set_xref, xref, self.N, self.nx)

    # -- the two terms the reference ships commented out (admm.cpp:20, :79); off by default ------------------
    def set_optional_terms(self, en_uref=False, en_coeff_d2p=False):
        self._check(self.lib.tiny_batch_set_optional_terms(self._h, int(bool(en_uref)), int(bool(en_coeff_d2p))))

    def set_input_cost(self, R):
        a = _f32(np.asarray(R).ravel()); assert a.shape == (self.nu,), a.shape
        self._check(self.lib.tiny_batch_set_input_cost(self._h, _fp(a)))

    def set_coeff_d2p(self, coeff_d2p):
        assert np.asarray(coeff_d2p).shape == (self.nx, self.nu)
        a = _colmajor(coeff_d2p)
        self._check(self.lib.tiny_batch_set_coeff_d2p(self._h, _fp(a)))

    def set_uref(self, uref):
        self._set_steps(self.lib.tiny_batch_set_uref, uref, self.N - 1, self.nu)

    def set_xref_window(self, table, start):
        t = _f32(table); s = np.ascontiguousarray(start, dtype=np.int32)
        assert t.ndim == 2 and t.shape[1] == self.nx and s.shape == (self.B,)
        self._check(self.lib.tiny_batch_set_xref_window(self._h, _fp(t), t.shape[0], s.ctypes.data_as(C.POINTER(C.c_int))))

    def set_umin(self, a): self._set_steps(self.lib.tiny_batch_set_umin, a, self.N - 1, self.nu)
    def set_umax(self, a): self._set_steps(self.lib.tiny_batch_set_umax, a, self.N - 1, self.nu)
    def set_xmin(self, a): self._set_steps(self.lib.tiny_batch_set_xmin, a, self.N, self.nx)
    def set_xmax(self, a): self._set_steps(self.lib.tiny_batch_set_xmax, a, self.N, self.nx)

    def set_bounds(self, x_min, x_max, u_min, u_max):
        self.set_xmin(x_min); self.set_xmax(x_max); self.set_umin(u_min); self.set_umax(u_max)

    def reset_dual_variables(self):
        self._check(self.lib.tiny_batch_reset_dual_variables(self._h))

    def solve(self) -> int:
        """call_tiny_solve for every instance; 0 = all converged, 1 = some hit max_iter."""
        return self._check(self.lib.tiny_batch_solve(self._h))

    call_tiny_solve = solve

    def solve_async(self):
        self._check(self.lib.tiny_batch_solve_async(self._h))

    def wait(self) -> int:
        n = C.c_int(0)
        self._check(self.lib.tiny_batch_wait(self._h, C.byref(n)))
        return n.value

    # -- the six step functions of admm.hpp:12-18 ----------------------------------------------------
    def forward_pass(self): self._check(self.lib.tiny_batch_forward_pass(self._h))
    def update_slack(self): self._check(self.lib.tiny_batch_update_slack(self._h))
    def update_dual(self): self._check(self.lib.tiny_batch_update_dual(self._h))
    def update_linear_cost(self): self._check(self.lib.tiny_batch_update_linear_cost(self._h))
    def backward_pass_grad(self): self._check(self.lib.tiny_batch_backward_pass_grad(self._h))

    def termination_condition(self):
        """Returns the per-instance boolean the reference function returns (residual fields are updated on the device)."""
        conv = np.zeros(self.B, np.int32)
        self._check(self.lib.tiny_batch_termination_condition(self._h, conv.ctypes.data_as(C.POINTER(C.c_int))))
        return conv.astype(bool)

    def get_x(self): return self.get_array("x")
    def get_u(self): return self.get_array("u")

    def get_status(self):
        it = np.zeros(self.B, np.int32); st = np.zeros(self.B, np.int32); res = np.zeros((self.B, 4), np.float32)
        ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
        self._check(self.lib.tiny_batch_get_status(self._h, ip(it), ip(st), _fp(res)))
        return it, st, res

    def set_status(self, iter=None, status=None, residuals=None):
        ip = lambda a: None if a is None else np.ascontiguousarray(a, np.int32).ctypes.data_as(C.POINTER(C.c_int))
        keep = [None if a is None else np.ascontiguousarray(a, np.int32) for a in (iter, status)]
        r = None if residuals is None else _f32(residuals)
        self._check(self.lib.tiny_batch_set_status(
            self._h, None if keep[0] is None else keep[0].ctypes.data_as(C.POINTER(C.c_int)),
            None if keep[1] is None else keep[1].ctypes.data_as(C.POINTER(C.c_int)), None if r is None else _fp(r)))

    # -- whole workspace ------------------------------------------------------------------------
    def get_array(self, name):
        out = np.empty(self._xshape(name), np.float32)
        self._check(self.lib.tiny_batch_get_array(self._h, ARRAY_IDS[name], _fp(out)))
        return out

    def set_array(self, name, arr):
        a = _f32(arr); assert a.shape == self._xshape(name), (name, a.shape)
        self._check(self.lib.tiny_batch_set_array(self._h, ARRAY_IDS[name], _fp(a)))

    def get_state(self) -> dict:
        st = {k: self.get_array(k) for k in ARRAY_IDS}
        st["iter"], st["status"], st["residuals"] = self.get_status()
        return st

    def set_state(self, st: dict):
        for k in ARRAY_IDS:
            if k in st:
                self.set_array(k, st[k])
        self.set_status(st.get("iter"), st.get("status"), st.get("residuals"))

    def reset_workspace(self):
        self._check(self.lib.tiny_batch_reset_workspace(self._h))

    # -- closed loop / measurement ------------------------------------------------------------
    def mpc_step_async(self, window_advance: int = 0):
        self._check(self.lib.tiny_batch_mpc_step_async(self._h, window_advance))

    def mpc_run_async(self, steps: int, window_advance: int = 0):
        """`steps` closed-loop MPC steps replayed from one captured hipGraph."""
        self._check(self.lib.tiny_batch_mpc_run_async(self._h, steps, window_advance))

    def mpc_run_traj(self, steps: int, window_advance: int = 0) -> np.ndarray:
        """tiny_batch_mpc_run_traj: runs `steps` closed-loop MPC steps and returns u.col(0) of every step, shape (steps, B, nu).
        The library allocates the device-side trajectory buffer on this handle's own device."""
        out = np.zeros((steps, self.B, self.nu), np.float32)
        self._check(self.lib.tiny_batch_mpc_run_traj(self._h, steps, window_advance, _fp(out)))
        return out

    def get_x0(self):
        out = np.empty((self.B, self.nx), np.float32)
        self._check(self.lib.tiny_batch_get_x0(self._h, _fp(out)))
        return out

    def synchronize(self):
        self._check(self.lib.tiny_batch_synchronize(self._h))

    def enable_timing(self, on=True):
        self._check(self.lib.tiny_batch_enable_timing(self._h, 1 if on else 0))

    def last_solve_ms(self) -> float:
        ms = C.c_float(0)
        self._check(self.lib.tiny_batch_last_solve_ms(self._h, C.byref(ms)))
        return ms.value

    def kernel_name(self) -> str:
        return self.lib.tiny_batch_kernel_name(self._h).decode()

    def closed_loop_kernel_name(self) -> str:
        return self.lib.tiny_batch_closed_loop_kernel_name(self._h).decode()

    def arithmetic(self) -> str:
        """tiny_batch_arithmetic: "exact" (bitwise the reference) or "fma" for what the next solve computes in; raises where no kernel would
        run (a class without an exact kernel before the caller has opted into fma arithmetic with select_kernel(1))."""
        rc = self.lib.tiny_batch_arithmetic(self._h)
        if rc < 0:
            self._check(rc)
        return "exact" if rc == 0 else "fma"

    def select_kernel(self, variant: int):
        self._check(self.lib.tiny_batch_select_kernel(self._h, variant))

    def set_row_kernel(self, family: int):
        """0 auto, 1 rowlane (unrolled), 2 rowloop (rolled, N <= 64), 3 rowstream (state in HBM), 4 quadlane (nx=4, nu=1),
        5 tile16 (16 instances per wave, products on the matrix cores; nx=12, nu=4, instantiated N; the auto choice for launches
        of >= 32768 instances), 6 wavestream / 7 waveres (one wavefront per instance, 16 < nx + nu <= 64), 8 tile48 (nx = 32, nu = 16, N <= 50: sixteen
        instances per workgroup on the matrix cores; automatic for 2049 ... 4096 and from 6145 instances on)."""
        self._check(self.lib.tiny_batch_set_row_kernel(self._h, family))

    def set_dispatch(self, mode: int):
        """0 = workgroups in index order, 1 = longest first by a predicted iteration count (register-resident row kernels), 2 = longest first by the
        iteration counts of the previous solve of this workspace (warm-started launches), -1 (default) = automatic: 1 for a launch that starts from a
        reset workspace, 2 for a warm-started one."""
        self._check(self.lib.tiny_batch_set_dispatch(self._h, mode))

    def set_tile_queue(self, stride: int):
        """tile16's tile queue under longest-first dispatch: -1 automatic, 0 one counter, k = every k-th wave takes tiles from the short end of the order."""
        self._check(self.lib.tiny_batch_set_tile_queue(self._h, stride))

    def dispatch_applied(self) -> int:
        """0 index order, 1 predicted longest first, 2 the caller's order, 3 longest first by the previous solve's iteration counts — what the most recent
        solve launch did."""
        return self._check(self.lib.tiny_batch_dispatch_applied(self._h))

    def set_dispatch_order_device(self, d_order_ptr):
        """Device pointer to a permutation of the ceil(batch/4) group indices (int32), or None."""
        self._check(self.lib.tiny_batch_set_dispatch_order_device(self._h, C.c_void_p(d_order_ptr)))

    def set_storage(self, bits: int, dual_bits: int | None = None):
        """tiny_batch_set_storage(bits) when dual_bits is None — for bits = 16 that is binary16 storage with the duals y, g kept in
        fp32 wherever the class has a register-resident kernel (the variant that converges), 16-bit duals
        elsewhere; tiny_batch_set_storage_ex(bits, dual_bits) otherwise ((16, 16): everything binary16)."""
        if dual_bits is None:
            self._check(self.lib.tiny_batch_set_storage(self._h, bits))
        else:
            self._check(self.lib.tiny_batch_set_storage_ex(self._h, bits, dual_bits))

    def set_stream(self, stream_ptr: int):
        self._check(self.lib.tiny_batch_set_stream(self._h, C.c_void_p(stream_ptr)))


class TinyBatchSolver64:
    """The same host mirror for `typedef double tinytype` (include/tinympc_batch64.h; the reference as shipped,
    glob_opts.hpp:3): wrapper-style calls over a device-resident fp64 workspace, bitwise equal to the compiled reference."""

    def __init__(self, prob: dict, batch: int, device: int = 0, settings: dict | None = None):
        self.lib = load_library()
        self.nx, self.nu, self.N, self.B = int(prob["nx"]), int(prob["nu"]), int(prob["N"]), int(batch)
        self._h = C.c_void_p()
        self._check(self.lib.tiny_batch64_create(C.byref(self._h), self.nx, self.nu, self.N, self.B, int(device)))
        cm = lambda m: np.ascontiguousarray(np.asarray(m, np.float64).T).ravel()
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        k, p, qi, am = (cm(prob[n]) for n in ("Kinf", "Pinf", "Quu_inv", "AmBKt"))
        self._check(self.lib.tiny_batch64_set_cache(self._h, float(prob["rho"]), dp(k), dp(p), dp(qi), dp(am)))
        a, b, q = cm(prob["Adyn"]), cm(prob["Bdyn"]), np.ascontiguousarray(np.asarray(prob["Q"], np.float64).ravel())
        self._check(self.lib.tiny_batch64_set_dynamics(self._h, dp(a), dp(b), dp(q)))
        s = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1, en_state_bound=1, en_input_bound=1)
        if settings:
            s.update(settings)
        self.set_settings(**s)

    def _check(self, rc):
        if rc < 0:
            raise TinyBatchError(f"rc={rc}: {self.lib.tiny_batch64_last_error().decode()}")
        return rc

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.tiny_batch64_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _dp(a):
        return a.ctypes.data_as(C.POINTER(C.c_double))

    def _xshape(self, name):
        return (self.B, self.N, self.nx) if name in X_FAMILY else (self.B, self.N - 1, self.nu)

    def set_settings(self, abs_pri_tol, abs_dua_tol, max_iter, check_termination, en_state_bound, en_input_bound):
        self.settings = dict(abs_pri_tol=abs_pri_tol, abs_dua_tol=abs_dua_tol, max_iter=max_iter, check_termination=check_termination,
                             en_state_bound=en_state_bound, en_input_bound=en_input_bound)
        self._check(self.lib.tiny_batch64_set_settings(self._h, abs_pri_tol, abs_dua_tol, max_iter, check_termination, en_state_bound, en_input_bound))

    def set_x0(self, x0):
        a = np.ascontiguousarray(x0, np.float64); assert a.shape == (self.B, self.nx), a.shape
        self._check(self.lib.tiny_batch64_set_x0(self._h, self._dp(a)))

    def _set_steps(self, fn, arr, steps, dim):
        a = np.ascontiguousarray(arr, np.float64)
        shared = 1 if a.shape == (steps, dim) else 0
        assert shared or a.shape == (self.B, steps, dim), a.shape
        self._check(fn(self._h, self._dp(a), shared))

    def set_xref(self, xref): self._set_steps(self.lib.tiny_batch64_set_xref, xref, self.N, self.nx)

    def set_bounds(self, x_min, x_max, u_min, u_max):
        self._set_steps(self.lib.tiny_batch64_set_xmin, x_min, self.N, self.nx)
        self._set_steps(self.lib.tiny_batch64_set_xmax, x_max, self.N, self.nx)
        self._set_steps(self.lib.tiny_batch64_set_umin, u_min, self.N - 1, self.nu)
        self._set_steps(self.lib.tiny_batch64_set_umax, u_max, self.N - 1, self.nu)

    def reset_dual_variables(self): self._check(self.lib.tiny_batch64_reset_dual_variables(self._h))

    def select_kernel(self, which: int):
        """0 = automatic, 1 = one thread per instance (state in HBM, any N), 2 = sixteen lanes per instance (state in registers)."""
        self._check(self.lib.tiny_batch64_select_kernel(self._h, which))

    def kernel_name(self) -> str: return self.lib.tiny_batch64_kernel_name(self._h).decode()

    # -- the six step functions of admm.hpp:12-18 ----------------------------------------------------
    def forward_pass(self): self._check(self.lib.tiny_batch64_forward_pass(self._h))
    def update_slack(self): self._check(self.lib.tiny_batch64_update_slack(self._h))
    def update_dual(self): self._check(self.lib.tiny_batch64_update_dual(self._h))
    def update_linear_cost(self): self._check(self.lib.tiny_batch64_update_linear_cost(self._h))
    def backward_pass_grad(self): self._check(self.lib.tiny_batch64_backward_pass_grad(self._h))

    def termination_condition(self):
        conv = np.zeros(self.B, np.int32)
        self._check(self.lib.tiny_batch64_termination_condition(self._h, conv.ctypes.data_as(C.POINTER(C.c_int))))
        return conv.astype(bool)

    def mpc_step(self) -> int:
        """y = g = 0, tiny_solve, x.col(0) <- Adyn x.col(0) + Bdyn u.col(0) (quadrotor_hovering.cpp:95-111), on the device."""
        return self._check(self.lib.tiny_batch64_mpc_step(self._h))

    def first_columns(self):
        """(x.col(0), u.col(0)) as [B][nx], [B][nu]."""
        x0, u0 = np.empty((self.B, self.nx)), np.empty((self.B, self.nu))
        self._check(self.lib.tiny_batch64_get_first_columns(self._h, self._dp(x0), self._dp(u0)))
        return x0, u0

    def solve(self) -> int: return self._check(self.lib.tiny_batch64_solve(self._h))

    def get_array(self, name):
        out = np.empty(self._xshape(name), np.float64)
        self._check(self.lib.tiny_batch64_get_array(self._h, ARRAY_IDS[name], self._dp(out)))
        return out

    def set_array(self, name, arr):
        a = np.ascontiguousarray(arr, np.float64); assert a.shape == self._xshape(name), (name, a.shape)
        self._check(self.lib.tiny_batch64_set_array(self._h, ARRAY_IDS[name], self._dp(a)))

    def get_u(self): return self.get_array("u")

    def get_status(self):
        it = np.zeros(self.B, np.int32); st = np.zeros(self.B, np.int32); res = np.zeros((self.B, 4), np.float64)
        ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
        self._check(self.lib.tiny_batch64_get_status(self._h, ip(it), ip(st), self._dp(res)))
        return it, st, res

    def get_state(self) -> dict:
        st = {k: self.get_array(k) for k in ARRAY_IDS}
        st["iter"], st["status"], st["residuals"] = self.get_status()
        return st

    def set_state(self, st: dict):
        for k in ARRAY_IDS:
            if k in st:
                self.set_array(k, st[k])
        ip = lambda a: np.ascontiguousarray(a, np.int32)
        it, sta, res = ip(st["iter"]), ip(st["status"]), np.ascontiguousarray(st["residuals"], np.float64)
        self._check(self.lib.tiny_batch64_set_status(self._h, it.ctypes.data_as(C.POINTER(C.c_int)), sta.ctypes.data_as(C.POINTER(C.c_int)), self._dp(res)))
