"""Batch sharding across GPUs of one node.

Problem instances are independent (src/tinympc/admm.cpp touches one workspace), so the batch is block-partitioned
over ranks with NO data-path collective.  torch.distributed (RCCL on GPUs, gloo on CPU in the tests) is used only
for (a) barriers / max-over-ranks of the wall time, (b) the sum / max of iteration statistics and (c) the optional
final gather of u.col(0) (nu floats per instance — latency-bound over xGMI, one all_gather).
"""
from __future__ import annotations

import numpy as np


def block_partition(n: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous block [lo, hi) of the global instance index owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def reduce_stats(dist, device, iters: np.ndarray, status: np.ndarray, flops: float, wall_s: float) -> dict:
    """Whole-job statistics: sums over ranks of iterations / converged / flops, max of iterations and wall time."""
    import torch
    s = torch.tensor([float(iters.sum()), float((status == 1).sum()), float(flops), float(iters.size)],
                     dtype=torch.float64, device=device)
    m = torch.tensor([float(iters.max()) if iters.size else 0.0, float(wall_s)], dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
    s, m = s.cpu().numpy(), m.cpu().numpy()
    return dict(sum_iters=s[0], n_converged=s[1], flops=s[2], n_instances=int(s[3]), max_iters=int(m[0]), wall_s=float(m[1]))


def gather_first_inputs(dist, device, u0_local: np.ndarray, n_total: int) -> np.ndarray:
    """Optional epilogue: every rank obtains u.col(0) of ALL instances, in global instance order."""
    import torch
    world = dist.get_world_size() if (dist is not None and dist.is_initialized()) else 1
    if world == 1:
        return u0_local
    nu = u0_local.shape[1]
    sizes = [block_partition(n_total, world, r) for r in range(world)]
    cap = max(hi - lo for lo, hi in sizes)
    buf = torch.zeros((cap, nu), dtype=torch.float32, device=device)
    buf[: u0_local.shape[0]] = torch.from_numpy(np.ascontiguousarray(u0_local)).to(device)
    outs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf)
    return np.concatenate([o[: hi - lo].cpu().numpy() for o, (lo, hi) in zip(outs, sizes)], axis=0)
