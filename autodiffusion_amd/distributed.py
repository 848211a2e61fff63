"""Data-parallel sharding of the sampling path: one process per GPU, contiguous batch shards.

Every batch row (scenario x sample) is independent through the whole loop, so ranks exchange nothing while
sampling; the only collective is one all-gather of the sampled trajectories at the end (RCCL over xGMI when
the process group's backend is "nccl"; "gloo" on CPU for tests).  This replaces the per-evaluation-step gather
the HF Trainer performs around ``prediction_step`` (transformer4planning/trainer.py:527-536, 665-716).
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Rows [lo, hi) owned by ``rank``: contiguous, sizes differ by at most one, earlier ranks get the extras."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_rows(local: torch.Tensor, total: int) -> torch.Tensor:
    """Concatenate every rank's rows (dim 0) in rank order on every rank.

    Shards may differ by one row; they are padded to the largest shard for the collective (the same trick the
    reference's evaluation loop uses to keep ragged last batches from hanging a gather) and trimmed after.
    """
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    sizes = [shard_range(total, r, world) for r in range(world)]
    max_rows = max(hi - lo for lo, hi in sizes)
    padded = local
    if local.shape[0] < max_rows:
        pad = torch.zeros(max_rows - local.shape[0], *local.shape[1:], dtype=local.dtype, device=local.device)
        padded = torch.cat([local, pad], dim=0)
    out = torch.empty(world * max_rows, *local.shape[1:], dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded.contiguous()) if local.is_cuda else \
        dist.all_gather(list(out.view(world, max_rows, *local.shape[1:]).unbind(0)), padded.contiguous())
    parts = [out[r * max_rows: r * max_rows + (hi - lo)] for r, (lo, hi) in enumerate(sizes)]
    return torch.cat(parts, dim=0)
