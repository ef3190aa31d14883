"""MI355X-native batched TinyMPC ADMM solver (gfx950 HIP kernels behind a C-ABI).

The package directory is named `accelerated-tinympc_amd` (not a valid Python identifier);
import it as `accelerated_tinympc_amd` through the shim module at the repository root.
"""
from . import build, problems, sharding  # noqa: F401
from .solver import (ARRAY_IDS, TINY_SOLVED, TINY_UNSOLVED, TinyBatchError, TinyBatchSolver, TinyBatchSolver64,  # noqa: F401
                     debug_check, debug_guards, exported_symbols, load_library, riccati, solve_group)

__all__ = ["TinyBatchSolver", "TinyBatchSolver64", "TinyBatchError", "load_library", "riccati", "problems", "build", "sharding", "ARRAY_IDS",
           "exported_symbols", "solve_group", "TINY_SOLVED", "TINY_UNSOLVED", "debug_guards", "debug_check"]
