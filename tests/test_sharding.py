"""N>1 path on CPU: two gloo ranks shard a batch, solve their blocks (the oracle stands in for the GPU solver —
this test is about the sharding / reduction / gather plumbing used by bench.py), and the merged result equals a
single-process solve of the whole batch."""
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


def test_two_gloo_ranks_match_single_process(tmp_path, oracle_mod, tinympc):
    O, pr = oracle_mod, tinympc.problems
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = tmp_path / "merged.npz"
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
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
