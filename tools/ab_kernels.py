#!/usr/bin/env python3
"""Developer aid: build variants of the HIP library that differ in -D flags of ONE source file and time the headline
workload (BASELINE configs[2], tracking x 65536, cold start) on each, alternating between variants.

    python tools/ab_kernels.py build  name1="-DFOO=1" name2="-DFOO=2 -DBAR"      (CPU container or GPU box)
    python tools/ab_kernels.py time   name1 name2 [--rounds 3] [--kernels 2,3]   (GPU box)

`base` always names the stock library.  Variants live in accelerated-tinympc_amd/lib/ab/ (git-ignored *.so)."""
import os, subprocess, sys, json
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
PKG = ROOT / "accelerated-tinympc_amd"
sys.path.insert(0, str(ROOT))


def build(variants, src="admm_rowlane.hip"):
    import importlib.util
    spec = importlib.util.spec_from_file_location("b", PKG / "build.py"); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    b.build()
    ab = PKG / "lib" / "ab"; ab.mkdir(exist_ok=True)
    procs = []
    for name, flags in variants.items():
        obj = ab / f"{Path(src).stem}_{name}.o"
        cmd = ["/opt/rocm/bin/hipcc", *b.FLAGS, *flags.split(), "-c", str(PKG / "csrc" / src), "-o", str(obj)]
        procs.append((name, obj, subprocess.Popen(cmd)))
    for name, obj, p in procs:
        assert p.wait() == 0, name
        objs = [str(obj) if o.name == Path(src).stem + ".o" else str(o) for _, o in b._objs()]
        so = ab / f"libtinympc_hip_{name}.so"
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(so), *objs], check=True)
        print("built", so)


def time_one(kernels, reps):
    import numpy as np
    import accelerated_tinympc_amd as T
    pr = T.problems
    prob = pr.quadrotor(20, 30)
    B = 65536
    x0, table, start = pr.tracking_batch(B, 30)
    out = {}
    for k in kernels:
        sol = T.TinyBatchSolver(prob, B)
        sol.select_kernel(k)
        sol.set_bounds(*pr.bounds_arrays(prob)); sol.set_xref_window(table, start); sol.set_dispatch(1)
        sol.enable_timing(True)
        ms = []
        for r in range(reps + 2):
            sol.reset_workspace(); sol.set_x0(x0); sol.solve_async(); sol.synchronize()
            if r >= 2: ms.append(sol.last_solve_ms())
        it = sol.get_status()[0]
        out[sol.kernel_name()] = dict(median=float(np.median(ms)), min=float(np.min(ms)), mean_iters=float(it.mean()))
        sol.close()
    print(json.dumps(out))


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "build":
        src = "admm_rowlane.hip"
        args = sys.argv[2:]
        if args and args[0].startswith("--src="):
            src = args[0][6:]; args = args[1:]
        build(dict(a.split("=", 1) for a in args), src)
    elif mode == "one":
        time_one([int(k) for k in sys.argv[2].split(",")], int(sys.argv[3]))
    else:
        names = [a for a in sys.argv[2:] if not a.startswith("--")]
        rounds = int(next((a.split("=")[1] for a in sys.argv if a.startswith("--rounds=")), 3))
        kernels = next((a.split("=")[1] for a in sys.argv if a.startswith("--kernels=")), "2,3")
        res = {}
        for r in range(rounds):
            for n in names:
                env = dict(os.environ)
                if n != "base":
                    env["TINYMPC_HIP_LIB"] = str(PKG / "lib" / "ab" / f"libtinympc_hip_{n}.so")
                o = subprocess.run([sys.executable, __file__, "one", kernels, "8"], env=env, capture_output=True, text=True)
                line = [l for l in o.stdout.splitlines() if l.startswith("{")]
                if not line:
                    print(n, "FAILED", o.stderr[-500:]); continue
                for k, v in json.loads(line[-1]).items():
                    res.setdefault((n, k), []).append(v["median"])
                print(r, n, line[-1], flush=True)
        for (n, k), v in sorted(res.items(), key=lambda kv: kv[0][1]):
            print(f"{k:28s} {n:16s} median-of-medians {sorted(v)[len(v)//2]:.4f} ms   all {['%.4f' % x for x in v]}")
