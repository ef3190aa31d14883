#!/usr/bin/env python3
"""Write accelerated-tinympc_amd/data/quadrotor_<rate>hz.bin: rho, Adyn, Bdyn, Kinf, Pinf, Quu_inv, AmBKt, Q as flat
row-major float64 — the form examples/quadrotor_tracking_batched.cpp reads (so the C++ example needs no JSON parser)."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import accelerated_tinympc_amd as T

for rate in (20, 50, 100):
    p = T.problems.quadrotor(rate, 30)
    flat = np.concatenate([[p["rho"]]] + [np.asarray(p[k], np.float64).ravel() for k in ("Adyn", "Bdyn", "Kinf", "Pinf", "Quu_inv", "AmBKt", "Q")])
    out = Path(T.problems.DATA) / f"quadrotor_{rate}hz.bin"
    flat.astype("<f8").tofile(out)
    print("wrote", out, flat.size, "doubles")
