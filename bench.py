#!/usr/bin/env python3
"""bench.py — MPC solves/s of the batched TinyMPC ADMM solver on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[2], the configuration the metric is quoted on):
  quadrotor_tracking, nx=12 nu=4 N=30 fp32, 65536 instances per GPU; instance b tracks the window of the
  301-point y_axis_line trajectory starting at b mod 271 (quadrotor_tracking.cpp:84-85,101), x0_b = Xref_b[0] +
  U(-0.05,0.05)^12, the reference example's settings (tol 1e-3/1e-3, max_iter 100, check_termination 1, both
  bounds on, quadrotor_tracking.cpp:75-80), bounds u in [-0.5,0.5], x in [-5,5].
One "step" = one cold-start tiny_solve() of every instance of the batch: reset_workspace (folded into the solve),
x0 from a device buffer, per-instance reference windows gathered on the device from the trajectory table, one
kernel launch running all ADMM iterations with per-instance early exit — preceded, by default (the library's automatic dispatch; --dispatch), by the
predictor sweep and bucket sort that let the launch start its longest instance groups first (tiny_batch_set_dispatch;
part of the timed step, not of `roofline.kernel_ms`).  Inputs are resident in HBM before the timed region.  The batch shards embarrassingly across ranks (weak scaling: 65536 instances per GPU); there is no
data-path collective — torch.distributed is used only for the barriers and the max-over-ranks of the time.

Prints ONE JSON line (rank 0).  `roofline` and `cpu_baseline` are described in DESIGN.md §Measurement.  At N=1 the line
also carries extras that never enter `value`: `fast_arithmetic` (the fma variant of the kernel), `closed_loop` (warm-started
MPC steps in one launch), `pipelined_batches` (consecutive batches double-buffered on two streams), `fp64_tinytype` (the
same workload through the `typedef double tinytype` library).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

# Workloads.  "tracking" is the one BASELINE.json's metric is quoted on (configs[2]) and the default; "random32" is
# configs[3] (codegen_random scaled to nx=32, nu=16, N=50: 16 384 instances in all, block-sharded over the ranks).
CONFIGS = {
    "tracking": dict(nx=12, nu=4, N=30, per_gpu=65536, scaling="weak",
                     metric="MPC solves/sec (batched quadrotor nx=12,nu=4,N=30)"),
    "random32": dict(nx=32, nu=16, N=50, total=16384, scaling="strong",
                     metric="MPC solves/sec (batched codegen_random nx=32,nu=16,N=50, 16384 instances over the GPUs)"),
}
PEAK_HBM_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s
PEAK_F32_TFLOPS = 157.3    # MI355X_MICROARCH.md: fp32 vector == fp32-input MFMA peak


class Cost:
    """ALGORITHMIC cost model of SURVEY.md §8(d) (stated in DESIGN.md §5.1/§6) for one problem class."""

    def __init__(self, nx, nu, N):
        # bytes per solve = every live-in array read once + every live-out array written once, bounds shared
        self.b_solve = (nx + 3 * nu * (N - 1) + 3 * nx * N + 6 * nx * N + 6 * nu * (N - 1)) * 4 + 24   # Q: 17 208 B, R: 85 976 B
        # flops per ADMM iteration; the converged iteration omits the backward sweep
        self.f_fwd = (N - 1) * (2 * nx * nx + 4 * nx * nu + nx + nu) + 2 * nx * nx + 13 * (nx * N + nu * (N - 1))
        self.f_bwd = (N - 1) * (2 * nx * nx + 4 * nx * nu + 2 * nu * nu + 2 * nx + nu)

    def flops_of(self, iters: np.ndarray, status: np.ndarray) -> float:
        it = iters.astype(np.float64)
        return float(np.sum(it * self.f_fwd + (it - (status == 1)) * self.f_bwd))


def kernel_source_sha() -> str:
    """sha256 over the kernel sources: binds a committed PMC figure (profiles/hbm_traffic.json) to the code it was taken on."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted((ROOT / "accelerated-tinympc_amd" / "csrc").glob("*")):
        if f.suffix in (".hip", ".h", ".cpp"):
            h.update(f.name.encode()); h.update(f.read_bytes())
    return h.hexdigest()[:16]


def kernel_isa_sha(kernel_name: str):
    """Hash of the gfx950 machine code of the translation unit behind `kernel_name` (accelerated-tinympc_amd/build.py: read from the
    object file, nothing is executed).  profiles/hbm_traffic.json binds a PMC figure to it: the figure survives a comment edit and
    goes stale exactly when the kernel's code changes."""
    import accelerated_tinympc_amd as T
    return T.build.kernel_isa_sha(kernel_name)


def cpu_all_cores(config: str, total: int, settings: dict, seconds: float = 5.0):
    """`cpu_baseline.all_cores`: the COMPILED REFERENCE on every host core this job may use — one process per core, because the
    reference keeps its solver in process-global objects (tiny_wrapper.cpp; oracle/ref_shim.cpp likewise) — each timing cold-start
    tiny_solve() calls over its own slice of the benchmarked workload (oracle/ref_worker.py).  Called BEFORE this process touches
    the GPU: a process that has initialised HIP must not start other programs on this pool."""
    import subprocess
    ncores = usable_cores()
    count = 2048 if config == "tracking" else 32
    go_at = time.time() + 8.0  # imports + workload generation of the slowest worker
    cmd = [sys.executable, str(ROOT / "oracle" / "ref_worker.py"), "--config", config, "--total", str(total), "--count", str(count),
           "--seconds", str(seconds), "--settings", json.dumps(settings), "--go-at", repr(go_at)]
    procs = [subprocess.Popen(cmd + ["--start", str((k * count) % max(1, total - count + 1))], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for k in range(ncores)]
    outs = []
    for pr_ in procs:
        so, se = pr_.communicate(timeout=120)
        if pr_.returncode != 0:
            return {"error": (se or "").strip().splitlines()[-1:] or ["worker failed"]}
        outs.append(json.loads(so.strip().splitlines()[-1]))
    n = sum(o["solves"] for o in outs)
    t = max(o["seconds"] for o in outs)
    return dict(value=n / t, unit="solves/s", cores=ncores, kind=outs[0]["kind"], processes=ncores, cpu_model=cpu_model(),
                per_core=[o["solves"] / o["seconds"] for o in outs],
                sample=f"{n} cold-start tiny_solve calls, {ncores} processes x {count}-instance passes of the same workload for {t:.1f} s each, "
                       f"mean {float(np.mean([o['mean_iters'] for o in outs])):.1f} iterations, FTZ/DAZ off",
                build="g++ -O3, SSE2 (x86-64 baseline): the parity build of the reference, one process per core (its solver is a process-global object)")


def cpu_model() -> str:
    try:
        for l in open("/proc/cpuinfo"):
            if l.startswith("model name"):
                return l.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cores() -> int:
    """Host cores this process may really use: the affinity mask, cut down to the cgroup's CPU quota where one is set (a GPU box
    hands a job a share of a large host: 256 visible cores, 16 of them usable)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(f).read().split()
            if f.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, n)


def cpu_baseline(prob, make_batch, settings, seconds_target=12.0, fixed10_n=0):
    """Time the reference's own CPU path (oracle/_ref, compiled Eigen code, 1 thread — it is single threaded) or,
    if that prebuilt library is absent, our C port (oracle/), on a bounded sample of the same workload.
    `make_batch(nb)` -> (x0, Xref) of the FIRST nb instances of the benchmarked batch.  Returns the JSON object and the
    reference's live-out of the first pass (the parity sample of the bench line)."""
    import accelerated_tinympc_amd as T
    from oracle import oracle as O
    pr = T.problems
    nx, nu, N = prob["nx"], prob["nu"], prob["N"]
    xmn, xmx, umn, umx = pr.bounds_arrays(prob)
    kind = "reference" if O.have_ref(np.float32, nx, nu, N) else "port"
    solver = (O.Reference if kind == "reference" else O.Oracle)(prob, np.float32, settings)  # the settings of the benchmarked run

    def run(nb, cls_solver, nthreads=1):
        x0, xr = make_batch(nb)
        nb = len(x0)  # the batch has no more instances than that (many host cores x the per-core chunk can exceed it)
        xr = xr[:nb] if getattr(xr, "ndim", 2) == 3 else xr
        st = O.new_state(nb, nx, nu, N)
        st["x"][:, 0] = x0
        t0 = time.perf_counter()
        cls_solver.solve(st, xmn, xmx, umn, umx, xr, nthreads=nthreads)
        return time.perf_counter() - t0, st

    def timed(solver_, nthreads, budget_s, chunk):
        """repeat cold-start passes over `chunk` instances until ~budget_s of wall time has been spent"""
        n, t, its, first = 0, 0.0, [], None
        while t < budget_s:
            dt, st = run(chunk, solver_, nthreads)
            first = st if first is None else first
            n += len(st["iter"]); t += dt; its.append(st["iter"].mean())
        return n, t, float(np.mean(its)), first

    t_probe, _ = run(64, solver)
    chunk = int(min(32768, max(64, 64 * 2.0 / max(t_probe, 1e-6))))
    nb, t, mi, first = timed(solver, 1, seconds_target, chunk)
    ncores = usable_cores()
    out = dict(value=nb / t, unit="solves/s", cores=1, kind=kind, cpu_model=cpu_model(), host_cores_usable=ncores,
               host_cores_visible=(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()),
               sample=f"{nb} cold-start tiny_solve calls on the first instances of the same workload ({chunk}-instance passes), "
                      f"{t:.1f} s, mean {mi:.1f} iterations, FTZ/DAZ off",
               build="g++ -O3, SSE2 (x86-64 baseline, no -march=native): the parity build of the reference (the one whose bits the "
                     "GPU kernels reproduce), not the fastest CPU build of it")
    # our multi-threaded C port on every host core this process may use, for scale (not what the reference ships)
    ncores = max(1, ncores)
    port = O.Oracle(prob, np.float32, settings)
    nbp, tp, _, _ = timed(port, ncores, 4.0, max(64, min(4096, chunk)) * ncores)
    out["port_all_cores"] = dict(value=nbp / tp, unit="solves/s", cores=ncores, kind="port", cpu_model=cpu_model(),
                                 sample=f"{nbp} solves, OpenMP over instances, {tp:.1f} s", build="gcc -O3 -ffp-contract=off -fopenmp, no -march=native (oracle/Makefile); the bit-exact-order C port, "
                                       "4x slower per core than the compiled reference: a labelled extra, not the CPU baseline")
    # the same first instances at a FIXED iteration count (tolerances 0, 10 iterations): the checker's side of the fma-mode
    # parity figure (SURVEY.md section 7 "Parity definition" (i)); not timed
    if fixed10_n:
        fs = dict(settings, abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=10)
        x0, xr = make_batch(fixed10_n)
        st10 = O.new_state(fixed10_n, nx, nu, N)
        st10["x"][:, 0] = x0
        (O.Reference if kind == "reference" else O.Oracle)(prob, np.float32, fs).solve(st10, xmn, xmx, umn, umx, xr, nthreads=1)
        first = dict(first, u_fixed10=st10["u"])
    return out, first, kind


def parity_of(gpu_u, gpu_iter, gpu_status, ref, kind, u_scale):
    """The bench line's own parity evidence: the first instances of the timed batch against the CPU reference's results
    for the same inputs (computed by the cpu_baseline leg).  max_rel_u = max over instances of |u - u_ref|_inf /
    max(|u_ref|_inf, input bound) (SURVEY.md §7 'Parity definition')."""
    n = min(gpu_u.shape[0], ref["u"].shape[0])
    g, r = gpu_u[:n].astype(np.float64).reshape(n, -1), ref["u"][:n].astype(np.float64).reshape(n, -1)
    rel = np.max(np.abs(g - r), axis=1) / np.maximum(np.max(np.abs(r), axis=1), u_scale)
    return dict(instances=int(n), max_rel_u=float(rel.max()), bitwise_u=bool(np.array_equal(gpu_u[:n], ref["u"][:n])),
                iter_mismatch=int(np.sum(gpu_iter[:n] != ref["iter"][:n])), status_mismatch=int(np.sum(gpu_status[:n] != ref["status"][:n])),
                against=f"cpu_baseline ({kind}) on the same inputs, whole horizon of u")


def fma_parity(fast, ref_first, kind, u_scale):
    """The fma-chain mode against the compiled reference (or the port) on the first instances of the batch; the reference's
    results come from the cpu_baseline leg.  Two figures, SURVEY.md section 7 'Parity definition': (i) at a fixed iteration
    count (tolerances 0, 10 iterations) u can be compared instance by instance; (ii) with early exit the iteration count
    itself may flip where a residual sits at the tolerance, so u is compared over the instances whose count agrees and the
    flip fraction is reported next to it."""
    n = min(fast["_u_early_exit"].shape[0], ref_first["u"].shape[0])

    def rel(g, r):
        g, r = g.astype(np.float64).reshape(len(g), -1), r.astype(np.float64).reshape(len(r), -1)
        return np.max(np.abs(g - r), axis=1) / np.maximum(np.max(np.abs(r), axis=1), u_scale)

    same = fast["_iter_early_exit"][:n] == ref_first["iter"][:n]
    out = {"instances": int(n), "against": f"cpu_baseline ({kind}) on the same inputs, whole horizon of u; normalised by max(|u_ref|_inf, input bound)",
           "early_exit": {"iter_flip_fraction": float(np.mean(~same)),
                          "max_rel_u_same_iteration_count": float(rel(fast["_u_early_exit"][:n][same], ref_first["u"][:n][same]).max()) if same.any() else None,
                          "max_rel_u_all": float(rel(fast["_u_early_exit"][:n], ref_first["u"][:n]).max())}}
    if "u_fixed10" in ref_first:
        m = min(n, ref_first["u_fixed10"].shape[0])
        r10 = rel(fast["_u_fixed10"][:m], ref_first["u_fixed10"][:m])
        out["fixed_10_iterations"] = {"instances": int(m), "max_rel_u": float(r10.max()), "p99_rel_u": float(np.percentile(r10, 99)),
                                      "north_star_tolerance": 1e-5, "within_1e-5": bool(r10.max() <= 1e-5)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=list(CONFIGS), default="tracking",
                    help="tracking = BASELINE.json configs[2], the headline (default); random32 = configs[3]: nx=32 nu=16 N=50, "
                         "16384 instances in all, block-sharded over the ranks (strong scaling)")
    ap.add_argument("--batch", type=int, default=0, help="instances per GPU (tracking; default 65536) / in all (random32; default 16384)")
    ap.add_argument("--mode", choices=["early_exit", "fixed10"], default="early_exit")
    ap.add_argument("--kernel", type=int, default=0, help="0 auto (exact row / wave kernel), 1 streaming, 2 row exact, 3 row fast")
    ap.add_argument("--dispatch", type=int, choices=[-1, 0, 1, 2], default=-1,
                    help="workgroup dispatch order of the row kernel: 0 index order, 1 longest first by the predicted iteration count "
                         "(tiny_batch_set_dispatch; the predictor sweep and the sort run inside the timed region), 2 longest first by the previous "
                         "solve's iteration counts (warm-started launches), -1 the library's default: 1 for the cold-start step timed here, 2 for the "
                         "warm-started launches of the closed_loop extra")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg (and with it the parity sample)")
    ap.add_argument("--no-closed-loop", action="store_true", help="skip the fixed10 / transfers legs and the warm-started closed-loop and pipelined-batches extras (profiling runs: "
                    "its launches of the same kernel would be averaged into the per-kernel statistics)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks here, BEFORE torch or HIP is touched in this process
        # (one process per GPU; the children are what the documented torch.distributed.run command would start), forward rank
        # 0's JSON line and the launcher's return code
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    if args.gpus != world:
        # inside an already launched group the two must agree: a lone rank asked for N GPUs would silently measure one
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch N>1 as\n  python -m torch.distributed.run --nnodes=1 "
                         f"--nproc-per-node {args.gpus} --master-addr 127.0.0.1 --master-port P bench.py --gpus {args.gpus} ...\n")
        raise SystemExit(2)

    # the all-core CPU baseline runs first: it starts one reference process per core, which this process may only do while it has
    # not yet initialised the GPU
    all_cores = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cfg0 = CONFIGS[args.config]
        try:
            all_cores = cpu_all_cores(args.config, (args.batch or cfg0.get("per_gpu") or cfg0["total"]),
                                      dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1, en_state_bound=1, en_input_bound=1)
                                      if args.mode == "early_exit" else
                                      dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=10, check_termination=1, en_state_bound=1, en_input_bound=1))
        except Exception as e:  # noqa: BLE001
            all_cores = {"error": f"{type(e).__name__}: {e}"}

    import torch
    import accelerated_tinympc_amd as T

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # rehearsal knobs (one-GPU box): TINYMPC_BENCH_DEVICE pins every rank to one device, TINYMPC_BENCH_BACKEND=gloo
    # replaces RCCL (two ranks cannot share a GPU under RCCL), TINYMPC_BENCH_FORCE_DIST=1 creates the process group even
    # for a single rank, so that the RCCL branch itself (init with device_id, barriers, the stats all-reduce and the final
    # all-gather on device tensors) runs on one GPU.  The driver's real runs use none of them.
    dev_index = int(os.environ.get("TINYMPC_BENCH_DEVICE", local_rank))
    backend = os.environ.get("TINYMPC_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    dist = None
    force_dist = os.environ.get("TINYMPC_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    world_seen = dist.get_world_size() if dist is not None else 1

    pr = T.problems
    cfg = CONFIGS[args.config]
    NX, NU, N = cfg["nx"], cfg["nu"], cfg["N"]
    cost = Cost(NX, NU, N)
    settings = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1, en_state_bound=1, en_input_bound=1)
    if args.mode == "fixed10":
        settings.update(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=10)
    # shard = contiguous block of the global instance index
    if args.config == "tracking":
        prob = pr.quadrotor(20, N)
        per_gpu = args.batch or cfg["per_gpu"]
        total = per_gpu * world                      # weak scaling: per_gpu instances on every rank
        gx0, table, gstart = pr.tracking_batch(total, N)
        make_batch = lambda nb: (gx0[:nb], pr.expand_windows(table, gstart[:nb], N))
    else:
        prob = pr.random_system(NX, NU, N, seed=1234)  # SURVEY.md §8(d) config 4; cache from the library's own tiny_riccati()
        total = args.batch or cfg["total"]             # strong scaling: the 16 384 instances are divided over the ranks
        gx0, xref0 = pr.random_batch(total, NX, N)
        make_batch = lambda nb: (gx0[:nb], xref0)
    lo, hi = T.sharding.block_partition(total, world, rank)
    B = hi - lo
    x0 = gx0[lo:hi]
    sol = T.TinyBatchSolver(prob, B, device=dev_index, settings=settings)
    if args.kernel:
        sol.select_kernel(args.kernel)
    sol.set_bounds(*pr.bounds_arrays(prob))
    if args.config == "tracking":
        sol.set_xref_window(table, gstart[lo:hi])
    else:
        sol.set_xref(xref0)
    sol.set_dispatch(args.dispatch)
    d_x0 = torch.from_numpy(np.ascontiguousarray(x0)).cuda()
    lib, h = sol.lib, sol._h

    def cold_name():
        """name of the kernel a COLD-START launch takes (what step() runs): the automatic choice differs for the warm-started launch that would follow a solve"""
        sol.reset_workspace()
        return sol.kernel_name()

    kname = cold_name()

    def step():
        sol.reset_workspace()
        sol._check(lib.tiny_batch_set_x0_device(h, C.c_void_p(d_x0.data_ptr())))
        sol.solve_async()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sol.synchronize()
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        # the library keeps ONE hipEvent pair per handle: of the pairs recorded in this loop only the last can be read afterwards, so only the last step
        # records one (a pair costs ~12 us of command-processor time around the solve launch: 6.3 us in front of and 5.9 us behind the kernel in the
        # rocprofv3 trace; the first nineteen pairs used to be recorded and overwritten)
        if k == args.steps - 1:
            sol.enable_timing(True)
        step()
    sol.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    kernel_ms = [sol.last_solve_ms()]  # events of the last timed step (recorded by the library on its launch stream)
    # per-step kernel durations: a second, untimed pass with an event read after every step
    for _ in range(min(args.steps, 10)):
        step()
        kernel_ms.append(sol.last_solve_ms())
    n_unsolved = sol.wait()
    # reported by the library for the launches just timed (tiny_batch_dispatch_applied), not re-derived here
    dispatch_used = {0: "index order", 1: "longest first by predicted iteration count (predictor sweep + sort inside the timed region)",
                     2: "caller-supplied order", 3: "longest first by the previous solve's iteration counts"}[sol.dispatch_applied()]
    iters, status, _ = sol.get_status()
    npar = min(B, 2048)
    gpu_u = sol.get_u()[:npar].copy() if (rank == 0 and world == 1 and not args.no_cpu) else None  # parity sample, before the extras touch the workspace
    agg = T.sharding.reduce_stats(dist, "cuda" if backend == "nccl" else "cpu", iters, status, cost.flops_of(iters, status), dt)
    dt = agg["wall_s"]  # max over ranks
    # optional epilogue of SURVEY.md §8(e), outside the timed region: every rank obtains u.col(0) of all instances with
    # one all-gather (nu floats per instance; RCCL over xGMI on the real runs).  Never allowed to break the bench line.
    gather = None
    if dist is not None:
        try:
            cap = max(T.sharding.block_partition(total, world, r)[1] - T.sharding.block_partition(total, world, r)[0] for r in range(world))
            d_u0 = torch.zeros((cap, NU), dtype=torch.float32, device="cuda")
            sol._check(lib.tiny_batch_get_u0_device(h, C.c_void_p(d_u0.data_ptr())))
            sol.synchronize()
            t_g = time.perf_counter()
            if backend == "nccl":
                d_all = torch.empty((cap * world, NU), dtype=torch.float32, device="cuda")
                dist.all_gather_into_tensor(d_all, d_u0)
                torch.cuda.synchronize()
                mine = d_all[rank * cap:rank * cap + B]
            else:
                outs = [torch.empty((cap, NU), dtype=torch.float32) for _ in range(world)]
                dist.all_gather(outs, d_u0.cpu())
                mine = outs[rank][:B].cuda()
            gather = {"op": "all_gather of u.col(0)", "bytes_per_rank": cap * NU * 4, "ms": (time.perf_counter() - t_g) * 1e3,
                      "own_block_intact": bool(torch.equal(mine, d_u0[:B]))}
        except Exception as e:  # noqa: BLE001
            gather = {"error": f"{type(e).__name__}: {e}"}  # reported in the line AND, with more than one rank, as exit code 3 below
    total_solves = total * args.steps
    value = total_solves / dt
    extras_ok = rank == 0 and world == 1 and not args.kernel and args.config == "tracking"
    # "report both fixed-iteration (tol = 0, max_iter = 10) and early-exit throughput" — the same kernel, exact
    # arithmetic, same inputs, ten iterations for every instance; and the H2D / D2H legs of a host-driven step (x0 up, u.col(0) down),
    # hipEvent-timed on pinned buffers.  Extras of rank 0: neither enters `value`.
    fixed10 = transfers = None
    if rank == 0 and world == 1 and args.mode == "early_exit" and not args.no_closed_loop:   # (profiling runs skip it: the same kernel on another workload)
        try:
            sol.set_settings(**dict(settings, abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=10))
            k10 = cold_name()   # (before the steps: it resets the workspace)
            for _ in range(2):
                step()
            sol.synchronize()
            torch.cuda.synchronize()
            nst = max(5, min(args.steps, 20))
            t_f = time.perf_counter()
            for _ in range(nst):
                step()
            sol.synchronize()
            dt_f = time.perf_counter() - t_f
            ms_f = []
            for _ in range(5):
                step()
                ms_f.append(sol.last_solve_ms())
            it10, st10, _ = sol.get_status()
            fixed10 = {"solves_per_s": B * nst / dt_f, "ms_per_step": dt_f / nst * 1e3, "kernel": k10, "kernel_ms": float(np.mean(ms_f)),
                       "iterations": int(it10.max()), "f32_frac": cost.flops_of(it10, st10) / (float(np.mean(ms_f)) * 1e-3) / 1e12 / PEAK_F32_TFLOPS,
                       "note": "tolerances 0, max_iter 10 (SURVEY.md section 8(d) config 2/3 'fixed-iteration'), exact arithmetic, wall time of the steps"}
        except Exception as e:  # noqa: BLE001
            fixed10 = {"error": f"{type(e).__name__}: {e}"}
        sol.set_settings(**settings)
        try:
            h_x0 = torch.from_numpy(np.ascontiguousarray(x0)).pin_memory()
            d_u0t = torch.zeros((B, NU), dtype=torch.float32, device="cuda")
            h_u0 = torch.empty((B, NU), dtype=torch.float32).pin_memory()
            sol._check(lib.tiny_batch_get_u0_device(h, C.c_void_p(d_u0t.data_ptr())))
            sol.synchronize()
            up, down = [], []
            for _ in range(12):
                e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                e0.record(); d_x0.copy_(h_x0, non_blocking=True); e1.record(); h_u0.copy_(d_u0t, non_blocking=True); e2.record()
                torch.cuda.synchronize()
                up.append(e0.elapsed_time(e1)); down.append(e1.elapsed_time(e2))
            up, down = float(np.median(up[2:])), float(np.median(down[2:]))
            transfers = {"h2d_ms": up, "h2d_bytes": int(h_x0.numel() * 4), "h2d_GBs": h_x0.numel() * 4 / (up * 1e-3) / 1e9,
                         "d2h_ms": down, "d2h_bytes": int(h_u0.numel() * 4), "d2h_GBs": h_u0.numel() * 4 / (down * 1e-3) / 1e9,
                         "pcie_inclusive_solves_per_s": B / (dt / args.steps + (up + down) * 1e-3),
                         "note": "x0 of the batch up, u.col(0) down, pinned host buffers, hipEvent pairs on the copy stream, median of 10; the timed "
                                 "region keeps its inputs resident in HBM, so neither leg is part of `value`"}
        except Exception as e:  # noqa: BLE001
            transfers = {"error": f"{type(e).__name__}: {e}"}
    # the opt-in fma-chain arithmetic of the same kernel, for reference (not the headline: see DESIGN.md §3)
    fast = None
    if extras_ok and kname.startswith(("rowlane", "tile16")):
        sol.select_kernel(3)
        kfast = cold_name()   # (before the steps: it resets the workspace)
        for _ in range(2):
            step()
        ms = []
        for _ in range(min(args.steps, 10)):
            step()
            ms.append(sol.last_solve_ms())
        itf, stf, _ = sol.get_status()
        fast = dict(kernel=kfast, kernel_ms=float(np.mean(ms)), solves_per_s=B / (float(np.mean(ms)) * 1e-3),
                    f32_frac=cost.flops_of(itf, stf) / (float(np.mean(ms)) * 1e-3) / 1e12 / PEAK_F32_TFLOPS,
                    mean_iters=float(itf.mean()), note="kernel time only; parity bar = the reference's own fp64/fp32 spread")
        if not args.no_cpu:
            fast["_u_early_exit"] = sol.get_u()[:npar].copy()
            fast["_iter_early_exit"] = itf[:npar].copy()
            # the same instances at a FIXED iteration count (tolerances 0, 10 iterations: SURVEY.md section 7 "Parity definition" (i))
            sol.set_settings(**dict(settings, abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=10))
            step(); sol.synchronize()
            fast["_u_fixed10"] = sol.get_u()[:npar].copy()
            sol.set_settings(**settings)
        sol.select_kernel(0)
    elif rank == 0 and world == 1 and not args.kernel and args.config == "random32" and kname.startswith(("tile48", "waveres")):
        sol.select_kernel(3)   # the nx = 32 class: kernel time of the fma instantiation only (its parity bar is held by the GPU tests)
        step()
        ms = []
        for _ in range(min(args.steps, 5)):
            step()
            ms.append(sol.last_solve_ms())
        itf, stf, _ = sol.get_status()
        fast = dict(kernel=cold_name(), kernel_ms=float(np.mean(ms)), solves_per_s=B / (float(np.mean(ms)) * 1e-3),
                    f32_frac=cost.flops_of(itf, stf) / (float(np.mean(ms)) * 1e-3) / 1e12 / PEAK_F32_TFLOPS,
                    mean_iters=float(itf.mean()), note="kernel time only; parity bar = the reference's own fp64/fp32 spread (tests/test_parity_gpu.py::test_wave_kernel_fma_arithmetic)")
        sol.select_kernel(0)

    # warm-started closed loop (how the reference's examples actually run the solver, quadrotor_tracking.cpp:93-118):
    # every MPC step = dual reset + solve + plant step + window slide, all `ksteps` of them inside one kernel launch with
    # the state staying on chip between solves (tiny_batch_mpc_run_async).
    # Reported as an extra; `value` stays the cold-start solve rate above.
    closed = None
    if extras_ok and not args.no_closed_loop:
        try:
            sol.reset_workspace()
            sol._check(lib.tiny_batch_set_x0_device(h, C.c_void_p(d_x0.data_ptr())))
            sol.set_xref_window(table, gstart[lo:hi])
            ksteps = 20
            sol.mpc_run_async(ksteps, 1)      # settles the warm start
            sol.synchronize()
            t_c = time.perf_counter()
            sol.mpc_run_async(ksteps, 1)
            sol.synchronize()
            dt_c = time.perf_counter() - t_c
            itc, stc, _ = sol.get_status()
            closed = {"kernel": sol.closed_loop_kernel_name(), "mpc_steps": ksteps, "ms_per_mpc_step": dt_c / ksteps * 1e3, "solves_per_s": B * ksteps / dt_c,
                      "mean_iters_last_step": float(itc.mean()), "frac_converged_last_step": float(np.mean(stc == 1)),
                      "dispatch_applied": sol.dispatch_applied(),
                      "note": "warm-started tracking loop on the device (tiny_batch_mpc_run_async: one launch, state on chip between solves; tiles dispatched longest first "
                              "by the iteration counts of the solve before the run, dispatch_applied 3), wall time"}
            # the same loop forced onto the other of the two kernels whose MPC loop stays on chip (round 4), for the record: the automatic choice is the
            # sixteen-instances-per-wave kernel from 160 instances per compute unit on (measured cross-over), the 16-lane kernel below
            other = 1 if closed["kernel"].startswith("tile16") else 5
            for key, fam_k in (("on_headline_kernel", 5), ("on_16_lane_kernel", 1)):
                if fam_k != other:
                    closed[key] = {"kernel": closed["kernel"], "ms_per_mpc_step": closed["ms_per_mpc_step"], "solves_per_s": closed["solves_per_s"], "note": "the automatic choice"}
                    continue
                try:
                    sol.set_row_kernel(fam_k)
                    sol.reset_workspace()
                    sol._check(lib.tiny_batch_set_x0_device(h, C.c_void_p(d_x0.data_ptr())))
                    sol.set_xref_window(table, gstart[lo:hi])
                    sol.mpc_run_async(ksteps, 1)
                    sol.synchronize()
                    k16 = sol.closed_loop_kernel_name()
                    t_c = time.perf_counter()
                    sol.mpc_run_async(ksteps, 1)
                    sol.synchronize()
                    dt_16 = time.perf_counter() - t_c
                    closed[key] = {"kernel": k16, "ms_per_mpc_step": dt_16 / ksteps * 1e3, "solves_per_s": B * ksteps / dt_16}
                except Exception as e:  # noqa: BLE001
                    closed[key] = {"error": f"{type(e).__name__}: {e}"}
            sol.set_row_kernel(0)
        except Exception as e:  # noqa: BLE001
            closed = {"error": f"{type(e).__name__}: {e}"}

    # Consecutive batches on two workspaces and two HIP streams (double buffering): the tail of one launch — waves of the
    # few instances that need 2-4x the mean iteration count — overlaps the head of the next instead of leaving CUs idle.
    # Reported as an extra; `value` and the roofline above stay the plain one-launch-after-another numbers.
    pipelined = None
    if extras_ok and not args.no_closed_loop:
        sols, streams = [], [torch.cuda.Stream(), torch.cuda.Stream()]
        try:
            for strm in streams:
                s2 = T.TinyBatchSolver(prob, B, device=dev_index, settings=settings)
                s2.set_bounds(*pr.bounds_arrays(prob))
                s2.set_xref_window(table, gstart[lo:hi])
                s2.set_stream(strm.cuda_stream)
                sols.append(s2)

            def step_on(s2):
                s2.reset_workspace()
                s2._check(lib.tiny_batch_set_x0_device(s2._h, C.c_void_p(d_x0.data_ptr())))
                s2.solve_async()
            for k in range(4):
                step_on(sols[k % 2])
            for s2 in sols:
                s2.synchronize()
            torch.cuda.synchronize()
            ksteps = max(2, args.steps)
            t_p = time.perf_counter()
            for k in range(ksteps):
                step_on(sols[k % 2])
            for s2 in sols:
                s2.synchronize()
            torch.cuda.synchronize()
            dt_p = time.perf_counter() - t_p
            pipelined = {"workspaces": 2, "steps": ksteps, "ms_per_step": dt_p / ksteps * 1e3, "solves_per_s": B * ksteps / dt_p,
                         "note": "same cold-start solves, consecutive batches alternate between two workspaces on two HIP streams "
                                 "so that launch tails overlap; wall time"}
        except Exception as e:  # noqa: BLE001
            pipelined = {"error": f"{type(e).__name__}: {e}"}
        for s2 in sols:
            s2.close()

    # The reference as shipped is `typedef double tinytype` (glob_opts.hpp:3): the same workload through the fp64 library
    # (include/tinympc_batch64.h; results bitwise equal to the reference's fp64 build).  An extra; never part of `value`.
    fp64 = None
    if extras_ok and not args.no_closed_loop and args.config == "tracking":
        try:
            s64 = T.TinyBatchSolver64(prob, B, device=dev_index, settings=settings)
            s64.set_bounds(*[np.asarray(a, np.float64) for a in pr.bounds_arrays(prob)])
            s64.set_xref(pr.expand_windows(table, gstart[lo:hi], N).astype(np.float64))
            zero = {k: np.zeros_like(v) for k, v in s64.get_state().items() if k not in ("iter", "status", "residuals")}
            x064 = x0.astype(np.float64)
            ts = []
            for _ in range(3):
                for k, v in zero.items():
                    s64.set_array(k, v)
                s64.set_x0(x064)
                t_6 = time.perf_counter()
                s64.solve()
                ts.append(time.perf_counter() - t_6)
            it6, st6, _ = s64.get_status()
            fp64 = {"kernel": s64.kernel_name(), "ms_per_solve_call": min(ts[1:]) * 1e3, "solves_per_s": B / min(ts[1:]),
                    "mean_iters": float(it6.mean()), "frac_converged": float(np.mean(st6 == 1)),
                    "note": "cold-start tiny_solve of the same instances in double; wall time of the blocking call"}
            s64.close()
        except Exception as e:  # noqa: BLE001
            fp64 = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        k_ms = float(np.mean(kernel_ms))
        fl = cost.flops_of(iters, status)  # this rank's launch
        t_mem = B * cost.b_solve / (PEAK_HBM_GBS * 1e9)
        t_flop = fl / (PEAK_F32_TFLOPS * 1e12)
        hbm_ach = B * cost.b_solve / (k_ms * 1e-3) / 1e9
        fl_ach = fl / (k_ms * 1e-3) / 1e12
        # HBM bytes per launch from the PMC passes (tools/collect_profiles.sh -> profiles/hbm_traffic.json).  The figure is
        # bound to the kernel sources it was measured on: after any change to csrc/ it is stale and reported as null.
        traffic, traffic_note = None, "no PMC figure committed for this kernel/workload"
        tf = ROOT / "profiles" / "hbm_traffic.json"
        if tf.exists():
            try:
                ent = json.loads(tf.read_text()).get(f"{kname}:{args.mode}:{B}")
                if isinstance(ent, dict):
                    isa = kernel_isa_sha(kname)
                    if ent.get("isa_sha") is not None and ent.get("isa_sha") == isa:
                        traffic, traffic_note = ent["bytes"], (f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, calibrated ({ent.get('profile')}); bound to the "
                                                               f"kernel's device code {isa}")
                    elif ent.get("isa_sha") is None and ent.get("csrc_sha") == kernel_source_sha():
                        traffic, traffic_note = ent["bytes"], f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, calibrated; kernel sources {ent['csrc_sha']}"
                    else:
                        traffic_note = f"stale: measured on device code {ent.get('isa_sha')} (sources {ent.get('csrc_sha')}), this build's kernel is {isa}"
                elif ent is not None:
                    traffic_note = "stale: figure predates the source binding"
            except Exception:
                traffic = None
        if t_flop >= t_mem:
            roof = dict(bound="mfma", achieved=fl_ach, peak=PEAK_F32_TFLOPS, unit="TFLOP/s", frac=fl_ach / PEAK_F32_TFLOPS,
                        traffic=traffic)
        else:
            roof = dict(bound="hbm", achieved=hbm_ach, peak=PEAK_HBM_GBS, unit="GB/s", frac=hbm_ach / PEAK_HBM_GBS,
                        traffic=traffic)
        roof["traffic_note"] = traffic_note
        mp = ROOT / "profiles" / "measured_peaks.json"
        if mp.exists():  # what the box itself reaches on plain instruction loops (tools/micro/peaks.hip), next to the datasheet peak
            try:
                m = json.loads(mp.read_text())
                roof["measured_peaks"] = {"f32_fma_TFLOPs": m["f32_fma_TFLOPs"], "f32_mul_add_TFLOPs": m["f32_mul_add_TFLOPs"],
                                          "hbm_read_GBs": m["hbm_read_GBs"], "frac_of_measured_fma": fl_ach / m["f32_fma_TFLOPs"],
                                          "frac_of_measured_mul_add": fl_ach / m["f32_mul_add_TFLOPs"], "source": "profiles/measured_peaks.json",
                                          "note": "exact arithmetic issues separately rounded v_mul + v_add: its own ceiling is the mul_add figure"}
            except Exception:
                pass
        per_step = np.asarray(kernel_ms[1:] if len(kernel_ms) > 1 else kernel_ms)
        roof.update(kernel=kname, kernel_ms=k_ms,
                    kernel_ms_per_step={"min": float(per_step.min()), "median": float(np.median(per_step)), "max": float(per_step.max()),
                                        "n": int(per_step.size), "note": "hipEvent pair around the solve launch, one untimed step each"},
                    hbm_GBs=hbm_ach, hbm_frac=hbm_ach / PEAK_HBM_GBS,
                    f32_TFLOPs=fl_ach, f32_frac=fl_ach / PEAK_F32_TFLOPS, alg_bytes_per_solve=cost.b_solve,
                    alg_flops_per_iteration={"forward_sweep_and_updates": cost.f_fwd, "backward_sweep": cost.f_bwd},
                    alg_flops_per_launch=fl, note="fp32 vector/MFMA peak 157.3 TFLOP/s; the path is not a dense "
                    "GEMM, 'mfma' here means the fp32 FMA roof (vector == f32-MFMA peak on gfx950)")
        what = "quadrotor_tracking" if args.config == "tracking" else "codegen_random (seeded marginally stable system, SURVEY.md §8(d) config 4)"
        line = {
            "metric": cfg["metric"], "value": value, "unit": "solves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": cfg["scaling"], "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{what} batched {B} instances per GPU, cold-start tiny_solve, "
                                   + (f"{args.mode} (tol 1e-3, max_iter 100)" if args.mode == "early_exit" else "fixed 10 iterations"),
                       "nx": NX, "nu": NU, "N": N, "instances_per_gpu": B, "instances_total": total, "parallelism": f"batch-shard x{world}",
                       "world_size_seen": world_seen, "backend": (backend if dist is not None else "none (single process)"),
                       "kernel": kname,
                       "dispatch": dispatch_used,
                       "mean_iters": agg["sum_iters"] / agg["n_instances"],
                       "max_iters": agg["max_iters"], "frac_converged": agg["n_converged"] / agg["n_instances"]},
            "roofline": roof,
        }
        if fixed10 is not None:
            line["fixed10"] = fixed10
        if transfers is not None:
            line["transfers"] = transfers
        if fast is not None:
            line["fast_arithmetic"] = fast
        if gather is not None:
            line["final_gather"] = gather
        if closed is not None:
            line["closed_loop"] = closed
        if pipelined is not None:
            line["pipelined_batches"] = pipelined
        if fp64 is not None:
            line["fp64_tinytype"] = fp64
        if not args.no_cpu and world == 1:  # rank 0 at N=1 only: the other ranks of a multi-GPU run would sit in the barrier
            cb, ref_first, kind = cpu_baseline(prob, make_batch, settings, fixed10_n=(npar if (fast is not None and "_u_fixed10" in fast) else 0))
            if all_cores is not None:
                cb["all_cores"] = all_cores
            line["cpu_baseline"] = cb
            u_scale = max(abs(prob["u_max"]), abs(prob["u_min"]))
            line["parity"] = parity_of(gpu_u, iters, status, ref_first, kind, u_scale)
            if fast is not None and "_u_early_exit" in fast:
                line["fast_arithmetic"]["parity"] = fma_parity(fast, ref_first, kind, u_scale)
        if fast is not None:
            for k in [k for k in fast if k.startswith("_")]:
                del fast[k]
        print(json.dumps(line), flush=True)
    sol.close()
    gather_failed = isinstance(gather, dict) and ("error" in gather or gather.get("own_block_intact") is False)
    if dist is not None:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            gather_failed = True
    if gather_failed and world > 1:
        raise SystemExit(3)  # the collective epilogue failed on this rank: the scaling run must not pass silently


if __name__ == "__main__":
    main()
