"""Import shim: makes the package directory `accelerated-tinympc_amd/` importable as
`accelerated_tinympc_amd` (a hyphen cannot appear in a Python module name)."""
import importlib.util
import sys
from pathlib import Path

_pkg_dir = Path(__file__).resolve().parent / "accelerated-tinympc_amd"
_spec = importlib.util.spec_from_file_location("accelerated_tinympc_amd", _pkg_dir / "__init__.py",
                                               submodule_search_locations=[str(_pkg_dir)])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["accelerated_tinympc_amd"] = _mod
_spec.loader.exec_module(_mod)
