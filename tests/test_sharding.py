"""N>1 path: two gloo ranks shard a batch, solve their blocks and the merged result equals a single-process solve of
the whole batch.  On CPU (this container) the oracle stands in for the GPU solver — the test is then about the sharding /
reduction / gather plumbing used by bench.py; on a GPU box (-m gpu) the same two ranks solve their blocks with the HIP
library and must reproduce the oracle's answers bit for bit, and bench.py itself is rehearsed with two gloo ranks."""
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]

WORKER = r"""
import os, sys, numpy as np
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
import accelerated_tinympc_amd as T
from oracle import oracle as O
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
pr, sh = T.problems, T.sharding
NTOT = 37                                     # not divisible by the world size on purpose
prob = pr.quadrotor(20, 30)
x0, table, start = pr.tracking_batch(NTOT, 30, seed=5)
lo, hi = sh.block_partition(NTOT, world, rank)
if os.environ.get("TINYMPC_TEST_SOLVER") == "hip":   # the product path: this rank's block on the GPU, exact arithmetic
    sol = T.TinyBatchSolver(prob, hi - lo)
    assert sol.kernel_name().endswith("exact>"), sol.kernel_name()
    sol.set_bounds(*pr.bounds_arrays(prob)); sol.set_xref_window(table, start[lo:hi]); sol.set_x0(x0[lo:hi])
    sol.solve()
    st = sol.get_state(); sol.close()
else:
    st = O.new_state(hi - lo, 12, 4, 30); st["x"][:, 0] = x0[lo:hi]
    O.Oracle(prob, np.float32).solve(st, *pr.bounds_arrays(prob), pr.expand_windows(table, start[lo:hi], 30))
stats = sh.reduce_stats(dist, "cpu", st["iter"], st["status"], float(st["iter"].sum()) * 2.0, 0.1 * (rank + 1))
u0 = sh.gather_first_inputs(dist, "cpu", st["u"][:, 0].copy(), NTOT)
if rank == 0:
    np.savez(sys.argv[2], u0=u0, **{k: np.asarray(v) for k, v in stats.items()})
dist.barrier(); dist.destroy_process_group()
"""


def test_block_partition_covers_everything_once():
    import accelerated_tinympc_amd as T
    for n in (1, 7, 16, 37, 65536):
        for w in (1, 2, 3, 8):
            blocks = [T.sharding.block_partition(n, w, r) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1


import pytest


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
def test_two_gloo_ranks_with_the_hip_solver(tmp_path, oracle_mod, tinympc):
    """The same two-rank run with the HIP library solving each rank's block (both ranks on the one GPU of the box)."""
    _run_two_ranks(tmp_path, oracle_mod, tinympc, "hip")


@pytest.mark.gpu
def test_bench_two_rank_rehearsal(tmp_path):
    """bench.py as the driver launches it for N=2 (torch.distributed.run, one process per rank), rehearsed on one GPU with
    the gloo backend: the JSON line reports the ranks the process group saw, whole-job throughput over both ranks and an
    intact final gather.  (RCCL itself needs one GPU per rank: only the driver's multi-GPU run exercises it.)"""
    import json
    env = dict(os.environ, TINYMPC_BENCH_DEVICE="0", TINYMPC_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    for config, extra in (("tracking", ["--batch", "4096"]), ("random32", ["--batch", "512"])):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
               "--config", config, *extra]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]            # rank 0 prints ONE line
        d = json.loads(lines[0])
        assert d["n_gpus"] == 2 and d["config"]["world_size_seen"] == 2 and d["config"]["backend"] == "gloo"
        assert d["final_gather"]["own_block_intact"] is True, d["final_gather"]
        assert d["scaling"] == ("weak" if config == "tracking" else "strong")
        total = d["config"]["instances_total"]
        assert total == (8192 if config == "tracking" else 512) and d["config"]["instances_per_gpu"] == total // 2
        assert abs(d["value"] - total * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) / d["value"] < 1e-6
        assert d["config"]["frac_converged"] > 0.3 and "cpu_baseline" not in d
    # round 3 (advisor): `python bench.py --gpus 2` without a launcher starts the two ranks itself (children spawned before torch or
    # HIP is touched), forwards rank 0's line and the launcher's return code
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4096"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["world_size_seen"] == 2 and d["config"]["instances_total"] == 8192
    # inside an already launched group a mismatch is still refused instead of measuring the wrong number of GPUs
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1"], env=dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


@pytest.mark.gpu
def test_bench_rccl_branch_single_rank(tmp_path):
    """The RCCL branch of bench.py itself — process group on the `nccl` backend bound to the device, barriers, the stats
    all-reduce and the final all-gather on device tensors — rehearsed with ONE rank on one GPU (two ranks cannot share a
    device under RCCL).  What the driver's N > 1 launch adds on top is more ranks of the same code."""
    import json
    env = dict(os.environ, TINYMPC_BENCH_FORCE_DIST="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TINYMPC_BENCH_BACKEND", "TINYMPC_BENCH_DEVICE"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(ROOT / "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "8192",
           "--no-cpu", "--no-closed-loop"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["world_size_seen"] == 1 and d["config"]["backend"] == "nccl"
    assert d["final_gather"]["own_block_intact"] is True, d["final_gather"]
    assert d["config"]["frac_converged"] == 1.0 and d["value"] > 1e6


def test_two_gloo_ranks_match_single_process(tmp_path, oracle_mod, tinympc):
    _run_two_ranks(tmp_path, oracle_mod, tinympc, "oracle")


def _run_two_ranks(tmp_path, oracle_mod, tinympc, solver):
    O, pr = oracle_mod, tinympc.problems
    port = _free_port()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = tmp_path / "merged.npz"
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TINYMPC_TEST_SOLVER=solver)
        procs.append(subprocess.Popen([sys.executable, str(script), str(ROOT), str(out)], env=env))
    for p in procs:
        assert p.wait(timeout=180) == 0
    z = np.load(out)
    prob = pr.quadrotor(20, 30)
    x0, table, start = pr.tracking_batch(37, 30, seed=5)
    st = O.new_state(37, 12, 4, 30); st["x"][:, 0] = x0
    O.Oracle(prob, np.float32).solve(st, *pr.bounds_arrays(prob), pr.expand_windows(table, start, 30))
    assert np.array_equal(z["u0"], st["u"][:, 0])                      # gathered in global instance order
    assert int(z["n_instances"]) == 37 and float(z["sum_iters"]) == float(st["iter"].sum())
    assert int(z["max_iters"]) == int(st["iter"].max()) and int(z["n_converged"]) == int((st["status"] == 1).sum())
    assert abs(float(z["wall_s"]) - 0.2) < 1e-12                         # max over ranks
