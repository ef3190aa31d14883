"""ctypes binding of the reference's native API names (include/tinympc_admm.h): TinyCache / TinySettings /
TinyWorkspace / TinySolver with plain float arrays, `tiny_solve` and the step functions of src/tinympc/admm.hpp:10-18,
exported by lib/libtinympc_wrapper.so for ONE instance — and, with dtype=np.float64, the same structs with double members
over lib/libtinympc_wrapper64.so (the reference as checked in: typedef double tinytype, glob_opts.hpp:3).

`NativeSolver` owns the numpy arrays the structs point to (members are column-major like the reference's Eigen
matrices, i.e. the numpy arrays here are [N][nx] / [N-1][nu], C order) and is what the parity tests drive.  All compute
happens in the HIP library; nothing here has a CPU implementation.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

from . import build as _build

WRAPPER_LIB_PATH = Path(__file__).resolve().parent / "lib" / "libtinympc_wrapper.so"
WRAPPER64_LIB_PATH = Path(__file__).resolve().parent / "lib" / "libtinympc_wrapper64.so"
F = C.POINTER(C.c_float)

STATE_MEMBERS = ("x", "u", "q", "r", "p", "d", "v", "vnew", "z", "znew", "g", "y")
X_FAMILY = ("x", "q", "p", "v", "vnew", "g", "x_min", "x_max", "Xref")


def _struct_types(ct):
    """The four structs of include/tinympc_admm.h for tinytype = ct (c_float, or c_double under TINYMPC_TINYTYPE_DOUBLE)."""
    P = C.POINTER(ct)

    class TinyCache(C.Structure):  # types.hpp:26-34
        _fields_ = [("rho", ct), ("Kinf", P), ("Pinf", P), ("Quu_inv", P), ("AmBKt", P), ("coeff_d2p", P)]

    class TinySettings(C.Structure):  # types.hpp:39-47
        _fields_ = [("abs_pri_tol", ct), ("abs_dua_tol", ct), ("max_iter", C.c_int),
                    ("check_termination", C.c_int), ("en_state_bound", C.c_int), ("en_input_bound", C.c_int)]

    class TinyWorkspace(C.Structure):  # types.hpp:52-97
        _fields_ = ([("nx", C.c_int), ("nu", C.c_int), ("N", C.c_int)] + [(m, P) for m in STATE_MEMBERS] +
                    [("primal_residual_state", ct), ("primal_residual_input", ct),
                     ("dual_residual_state", ct), ("dual_residual_input", ct), ("status", C.c_int), ("iter", C.c_int),
                     ("Q", P), ("R", P), ("Adyn", P), ("Bdyn", P),
                     ("u_min", P), ("u_max", P), ("x_min", P), ("x_max", P), ("Xref", P), ("Uref", P), ("Qu", P)])

    class TinySolver(C.Structure):  # types.hpp:102-107
        _fields_ = [("settings", C.POINTER(TinySettings)), ("cache", C.POINTER(TinyCache)), ("work", C.POINTER(TinyWorkspace))]

    return TinyCache, TinySettings, TinyWorkspace, TinySolver


_struct_types_f32 = _struct_types(C.c_float)
TinyCache, TinySettings, TinyWorkspace, TinySolver = _struct_types_f32
TYPES64 = _struct_types(C.c_double)

_lib = None
_lib64 = None


def load_wrapper_library(build_if_missing: bool = False, double: bool = False) -> C.CDLL:
    global _lib, _lib64
    if double:
        if _lib64 is None:
            if not WRAPPER64_LIB_PATH.exists():
                if not build_if_missing:
                    raise RuntimeError(f"{WRAPPER64_LIB_PATH} is missing: run `python accelerated-tinympc_amd/build.py`")
                _build.build()
            lib = C.CDLL(str(WRAPPER64_LIB_PATH))
            S = C.POINTER(TYPES64[3])
            lib.tiny_solve.argtypes, lib.tiny_solve.restype = [S], C.c_int
            for fn in ("forward_pass", "update_slack", "update_dual", "update_linear_cost", "backward_pass_grad", "update_primal"):
                getattr(lib, fn).argtypes, getattr(lib, fn).restype = [S], None
            lib.termination_condition.argtypes, lib.termination_condition.restype = [S], C.c_bool
            lib.tiny_admm_set_device.argtypes = [C.c_int]
            lib.tiny_admm_set_optional_terms.argtypes = [C.c_int, C.c_int]
            _lib64 = lib
        return _lib64
    if _lib is None:
        if not WRAPPER_LIB_PATH.exists():
            if not build_if_missing:
                raise RuntimeError(f"{WRAPPER_LIB_PATH} is missing: run `python accelerated-tinympc_amd/build.py`")
            _build.build()
        lib = C.CDLL(str(WRAPPER_LIB_PATH))
        S = C.POINTER(TinySolver)
        lib.tiny_solve.argtypes, lib.tiny_solve.restype = [S], C.c_int
        for fn in ("forward_pass", "update_slack", "update_dual", "update_linear_cost", "backward_pass_grad", "update_primal"):
            getattr(lib, fn).argtypes, getattr(lib, fn).restype = [S], None
        lib.termination_condition.argtypes, lib.termination_condition.restype = [S], C.c_bool
        lib.tiny_admm_set_device.argtypes = [C.c_int]
        lib.tiny_admm_set_optional_terms.argtypes = [C.c_int, C.c_int]
        _lib = lib
    return _lib


class NativeSolver:
    """A TinySolver with numpy-backed members: `ns.a["x"]` etc. are the arrays, `ns.work` the TinyWorkspace struct."""

    def __init__(self, prob: dict, settings: dict, device: int = 0, dtype=np.float32):
        double = np.dtype(dtype) == np.float64
        self.dtype = dt = np.float64 if double else np.float32
        self.lib = load_wrapper_library(double=double)
        self.lib.tiny_admm_set_device(device)
        TinyCache, TinySettings, TinyWorkspace, TinySolver = TYPES64 if double else _struct_types_f32
        FP = C.POINTER(C.c_double if double else C.c_float)
        nx, nu, N = prob["nx"], prob["nu"], prob["N"]
        self.nx, self.nu, self.N = nx, nu, N
        cm = lambda m: np.ascontiguousarray(np.asarray(m, dt).T).ravel()  # column-major flat
        self.m = {k: cm(prob[k]) for k in ("Kinf", "Pinf", "Quu_inv", "AmBKt", "Adyn", "Bdyn")}
        self.m["Q"] = np.ascontiguousarray(prob["Q"], dt)
        self.a = {k: np.zeros((N, nx) if k in X_FAMILY else (N - 1, nu), dt)
                  for k in STATE_MEMBERS + ("u_min", "u_max", "x_min", "x_max", "Xref")}
        p = lambda arr: arr.ctypes.data_as(FP)
        self.cache = TinyCache(float(prob["rho"]), p(self.m["Kinf"]), p(self.m["Pinf"]), p(self.m["Quu_inv"]), p(self.m["AmBKt"]), None)
        self.settings = TinySettings(settings["abs_pri_tol"], settings["abs_dua_tol"], settings["max_iter"],
                                     settings["check_termination"], settings["en_state_bound"], settings["en_input_bound"])
        self.work = TinyWorkspace()
        self.work.nx, self.work.nu, self.work.N = nx, nu, N
        for k, arr in self.a.items():
            setattr(self.work, k, p(arr))
        self.work.Q, self.work.Adyn, self.work.Bdyn = p(self.m["Q"]), p(self.m["Adyn"]), p(self.m["Bdyn"])
        self.solver = TinySolver(C.pointer(self.settings), C.pointer(self.cache), C.pointer(self.work))
        # members the reference never reads unless the optional terms are switched on (admm.cpp:20, :79)
        self.m["R"] = np.ascontiguousarray(np.asarray(prob.get("R", np.zeros(nu)), dt).ravel())
        self.m["coeff_d2p"] = cm(prob.get("coeff_d2p", np.zeros((nx, nu))))
        self.a["Uref"] = np.zeros((N - 1, nu), dt)
        self.work.R, self.work.Uref, self.cache.coeff_d2p = p(self.m["R"]), p(self.a["Uref"]), p(self.m["coeff_d2p"])
        self.lib.tiny_admm_set_optional_terms(0, 0)

    def set_optional_terms(self, en_uref=False, en_coeff_d2p=False):
        """Process-wide switch of the wrapper library (tiny_admm_set_optional_terms)."""
        self.lib.tiny_admm_set_optional_terms(int(bool(en_uref)), int(bool(en_coeff_d2p)))

    def _check(self):
        code = self.lib.tiny_admm_last_error_code()
        if code < 0:
            raise RuntimeError(f"libtinympc_wrapper.so reported error {code}")

    def tiny_solve(self) -> int:
        rc = self.lib.tiny_solve(C.byref(self.solver))
        self._check()
        return rc

    def call(self, fn: str):
        rv = getattr(self.lib, fn)(C.byref(self.solver))
        self._check()
        return rv

    @property
    def residuals(self):
        w = self.work
        return np.array([w.primal_residual_state, w.primal_residual_input, w.dual_residual_state, w.dual_residual_input], self.dtype)
