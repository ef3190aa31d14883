"""Shared helpers for the parity tests (test infrastructure)."""
import json
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden"
STATE_ORDER = ("x", "u", "q", "r", "p", "d", "v", "vnew", "z", "znew", "g", "y")
PROB_KEYS = ("Kinf", "Pinf", "Quu_inv", "AmBKt", "Adyn", "Bdyn", "Q")


def load_fixture(name):
    z = np.load(GOLDEN / f"{name}.npz")
    meta = json.loads(bytes(z["meta"]).decode())
    prob = dict(nx=meta["nx"], nu=meta["nu"], N=meta["N"], rho=meta["rho"])
    for k in PROB_KEYS:
        prob[k] = z["prob_" + k]
    prob["x_min"], prob["x_max"], prob["u_min"], prob["u_max"] = meta["bounds"]
    solves = []
    for s in range(meta["nsolves"]):
        pre = {k[len(f"s{s}_pre_"):]: z[k] for k in z.files if k.startswith(f"s{s}_pre_")}
        post = {k[len(f"s{s}_post_"):]: z[k] for k in z.files if k.startswith(f"s{s}_post_")}
        st = meta["settings_per_solve"][s] or meta["settings"]
        solves.append(dict(pre=pre, post=post, xref=z[f"s{s}_xref"], rc=meta["rcs"][s], settings=st, k=meta["ks"][s]))
    return meta, prob, solves, z


def bounds_of(prob, dt):
    N, nx, nu = prob["N"], prob["nx"], prob["nu"]
    return (np.full((N, nx), prob["x_min"], dt), np.full((N, nx), prob["x_max"], dt),
            np.full((N - 1, nu), prob["u_min"], dt), np.full((N - 1, nu), prob["u_max"], dt))


def rel_inf(a, b, floor):
    """per-instance relative infinity-norm error of a vs b, normalised by max(|b|_inf, floor)."""
    a = np.asarray(a, np.float64).reshape(a.shape[0], -1)
    b = np.asarray(b, np.float64).reshape(b.shape[0], -1)
    return np.max(np.abs(a - b), axis=1) / np.maximum(np.max(np.abs(b), axis=1), floor)


def scale_of(name, prob):
    """Natural magnitude used as the floor of the relative error of each work array."""
    if name in ("u", "z", "znew"):
        return max(abs(prob["u_max"]), abs(prob["u_min"]))
    if name in ("x", "v", "vnew"):
        return 1.0
    return 1.0


def closed_loop_case(pr, O, z, name):
    """(prob, x0, xref_fn(k), steps, settings, window table/start or None) of one scenario of closed_loop_traces.npz"""
    meta = json.loads(bytes(z["meta"]).decode())[name]
    if name in ("hover", "track"):
        prob = pr.quadrotor(20, 30)
    elif name == "cartpole":
        rc = np.load(GOLDEN / "riccati_cartpole.npz")
        prob = dict(pr.cartpole(10, riccati=O.riccati), Kinf=rc["Kinf"], Pinf=rc["Pinf"], Quu_inv=rc["Quu_inv"], AmBKt=rc["AmBKt"])
    else:
        prob = pr.random_system(8, 3, 7, seed=99, riccati=O.riccati)
    N = meta["N"]
    table = start = None
    if name == "hover":
        xr = np.tile(pr.HOVER_XREF, (N, 1)).astype(np.float32)
        fn = lambda k: xr
    elif name == "track":
        table, start = pr.y_axis_line().astype(np.float32), z["track_start"]
        fn = lambda k: pr.expand_windows(table, start + k, N)
    elif name == "cartpole":
        xr = np.zeros((N, 4), np.float32)
        fn = lambda k: xr
    else:
        xr = z["dims837_xref"].astype(np.float32)
        fn = lambda k: xr
    return prob, z[f"{name}_x0"], fn, meta["steps"], meta["settings"], table, start
