"""Randomised check of the host-side state machine (run on an MI355X): random sequences of set_x0 / reset_dual_variables /
reset_workspace (both folded lazily into the next solve) / set_array / set_status / kernel and row-family switches (device
layout conversion) / solve / get_*, mirrored on a numpy model that is advanced with the CPU oracle.  Every read must equal the
model bit for bit.      python tests/fuzz/fuzz_api.py [seconds] [seed]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import accelerated_tinympc_amd as T  # noqa: E402
from oracle import oracle as O  # noqa: E402

pr = T.problems
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end, rounds, ops_done, t_note = time.time() + budget, 0, 0, time.time()


def check(sol, st, what):
    got = sol.get_state()
    for name in O.STATE_ORDER + ("residuals", "status", "iter"):
        g_, r_ = got[name], st[name]
        ok = np.array_equal(g_, r_) and (g_.dtype.kind != "f" or np.array_equal(np.signbit(g_), np.signbit(r_)))
        if not ok:
            idx = tuple(np.argwhere(~((g_ == r_) & (np.signbit(g_) == np.signbit(r_)) if g_.dtype.kind == "f" else g_ != r_))[0])
            print(f"MISMATCH after {what}: {name}{idx} gpu {g_[idx]} model {r_[idx]} (kernel {sol.kernel_name()}, {kind} N={N} B={B} {settings})\n   history: {history}")
            sys.exit(1)


while time.time() < t_end:
    if time.time() - t_note > 30:
        print(f"... {rounds} rounds, {ops_done} operations", flush=True); t_note = time.time()
    kind, N = [("quad", 30), ("quad", 17), ("quad", 36), ("cartpole", 10), ("odd", 7)][rng.integers(5)]
    prob = {"quad": lambda: pr.quadrotor(20, N), "cartpole": lambda: pr.cartpole(N), "odd": lambda: pr.random_system(8, 3, N, seed=99)}[kind]()
    nx, nu = prob["nx"], prob["nu"]
    B = int(rng.choice([1, 5, 16, 33]))
    settings = dict(O.DEFAULT_SETTINGS, max_iter=int(rng.choice([1, 4, 25])), check_termination=int(rng.choice([1, 2])))
    sol = T.TinyBatchSolver(prob, B, settings=settings)
    bnds = pr.bounds_arrays(prob)
    xref = (rng.standard_normal((B, N, nx)) * 0.2).astype(np.float32)
    sol.set_bounds(*bnds); sol.set_xref(xref)
    h16 = False
    R = lambda a: a
    # the two optional terms (commented out in the reference): data always present, switched on and off at random
    prob = dict(prob, coeff_d2p=(rng.standard_normal((nx, nu)) * 0.03).astype(np.float32), R=rng.uniform(0.3, 2.0, nu).astype(np.float32))
    uref = (rng.standard_normal((N - 1, nu)) * 0.1).astype(np.float32)
    sol.set_input_cost(prob["R"]); sol.set_coeff_d2p(prob["coeff_d2p"]); sol.set_uref(uref)
    terms = (0, 0)

    def new_oracle():
        o = O.Oracle(prob, "h16" if h16 else np.float32, dict(settings, en_uref=terms[0], en_coeff_d2p=terms[1]))
        o.set_uref(R(uref))
        return o
    orc = new_oracle()
    st = O.new_state(B, nx, nu, N)
    history = []
    for _ in range(int(rng.integers(4, 14))):
        op = rng.choice(["set_x0", "reset_dual", "reset_ws", "set_array", "set_status", "switch", "solve", "solve", "read", "get_array",
                         "storage", "bounds", "xref", "terms", "uref"])
        history.append(str(op)); ops_done += 1
        if op == "set_x0":
            x0 = rng.uniform(-0.5, 0.5, size=(B, nx)).astype(np.float32)
            sol.set_x0(x0); st["x"][:, 0] = R(x0)
        elif op == "reset_dual":
            sol.reset_dual_variables(); st["y"][:] = 0; st["g"][:] = 0
        elif op == "reset_ws":
            sol.reset_workspace()
            for k in O.STATE_ORDER + ("residuals", "status", "iter"):
                st[k][:] = 0
        elif op == "set_array":
            name = O.STATE_ORDER[rng.integers(12)]
            v = (rng.standard_normal(st[name].shape) * 0.2).astype(np.float32)
            sol.set_array(name, v); st[name][:] = R(v)
        elif op == "set_status":
            st["iter"][:] = rng.integers(1, 9, size=B); st["status"][:] = rng.choice([1, 11], size=B)
            st["residuals"][:] = rng.uniform(0, 1, size=(B, 4)).astype(np.float32)
            sol.set_status(st["iter"], st["status"], st["residuals"])
        elif op == "switch":
            v = int(rng.choice([0, 2, 1, 2]))
            if v == 1 and (terms[0] or terms[1]):
                v = 2
            try:
                sol.select_kernel(v)
                if v != 1:
                    sol.set_row_kernel(int(rng.choice([0, 1, 2, 3, 4])))
            except T.TinyBatchError:
                pass
            history[-1] = f"switch->{sol.kernel_name()}"
        elif op == "solve":
            if sol.kernel_name().startswith("stream"):
                sol.select_kernel(2)  # only exact arithmetic can be mirrored bit for bit
            orc.solve(st, *[R(b_) for b_ in bnds], R(xref), nthreads=4); sol.solve()
            if not all(np.all(np.isfinite(st[n_])) for n_ in O.STATE_ORDER):
                break
            check(sol, st, "solve")
        elif op == "read":
            check(sol, st, "read")
        elif op == "storage":   # switching the storage precision restarts the workspace from zero (like create)
            if sol.kernel_name().startswith("stream"):
                sol.select_kernel(2)
            h16 = not h16
            sol.set_storage(16 if h16 else 32, 16 if h16 else 32)
            R = O.round_h16 if h16 else (lambda a: a)
            orc = new_oracle()
            for k in O.STATE_ORDER + ("residuals", "status", "iter"):
                st[k][:] = 0
            history[-1] = f"storage->{16 if h16 else 32}"
        elif op == "terms":
            terms = (int(rng.integers(2)), int(rng.integers(2)))
            if sol.kernel_name().startswith("stream"):
                sol.select_kernel(2)   # the MFMA variant refuses the terms
            sol.set_optional_terms(*terms)
            orc = new_oracle()
            history[-1] = f"terms->{terms} ({sol.kernel_name()})"
        elif op == "uref":
            uref = (rng.standard_normal((N - 1, nu) if rng.random() < 0.5 else (B, N - 1, nu)) * 0.1).astype(np.float32)
            sol.set_uref(uref)
            orc = new_oracle()
        elif op == "bounds":
            sc = rng.uniform(0.2, 1.0)
            bnds = tuple((a * sc).astype(np.float32) for a in pr.bounds_arrays(prob))
            if rng.random() < 0.3:
                bnds = tuple((a[None] * rng.uniform(0.5, 1.0, size=(B,) + a.shape)).astype(np.float32) for a in bnds)
            sol.set_bounds(*bnds)
        elif op == "xref":
            m = rng.integers(3)
            if m == 0:
                xref = (rng.standard_normal((N, nx)) * 0.2).astype(np.float32); sol.set_xref(xref)
            elif m == 1:
                xref = (rng.standard_normal((B, N, nx)) * 0.2).astype(np.float32); sol.set_xref(xref)
            else:
                table = (rng.standard_normal((N + 20, nx)) * 0.2).astype(np.float32)
                start = rng.integers(0, 20, size=B).astype(np.int32)
                sol.set_xref_window(table, start); xref = pr.expand_windows(table, start, N)
        else:
            name = O.STATE_ORDER[rng.integers(12)]
            g_ = sol.get_array(name)
            if not np.array_equal(g_, st[name]):
                print(f"MISMATCH get_array {name}; history {history}"); sys.exit(1)
    sol.close(); rounds += 1
print(f"fuzz ok: {rounds} rounds, {ops_done} API operations, device state == model after every read")
