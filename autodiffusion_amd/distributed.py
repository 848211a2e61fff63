"""Data-parallel sharding of the sampling path: one process per GPU, contiguous batch shards.

Every batch row (scenario x sample) is independent through the whole loop, so ranks exchange nothing while
sampling; the only collective is one all-gather of the sampled trajectories at the end (RCCL over xGMI when
the process group's backend is "nccl"; "gloo" on CPU for tests).  This replaces the per-evaluation-step gather
the HF Trainer performs around ``prediction_step`` (transformer4planning/trainer.py:527-536, 665-716).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def free_port(host: str = "127.0.0.1") -> int:
    """A TCP port nobody listens on right now (bound to port 0, read back, released): the rendezvous port of a process group
    started from here.  Two launches on one host get different ports, which a port derived from the process id does not promise."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        s.bind((host, 0))
        return int(s.getsockname()[1])


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Rows [lo, hi) owned by ``rank``: contiguous, sizes differ by at most one, earlier ranks get the extras."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class RowGatherer:
    """All-gather of every rank's rows (dim 0) in rank order, with every buffer allocated once.

    The collective is enqueued behind the work already on the current stream of ``local`` (the sampling stream) and
    writes straight into a preallocated ``(world * max_rows, ...)`` tensor; nothing is allocated or concatenated per call.
    Shards may differ by one row (``total`` not a multiple of the world size): the short shards are padded to the largest
    for the collective -- the trick the reference's evaluation loop uses to keep ragged last batches from hanging a
    gather (trainer.py:665-692) -- and ``gather`` then compacts the rows in place, again without allocating.
    """

    def __init__(self, total: int, row_shape: Tuple[int, ...], device, dtype=torch.float32, always_collective: bool = False):
        self.total = int(total)
        self.always_collective = bool(always_collective) and dist.is_initialized()     # run the collective even in a group of one
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.spans = [shard_range(self.total, r, self.world) for r in range(self.world)]
        self.max_rows = max(hi - lo for lo, hi in self.spans)
        self.ragged = any(hi - lo != self.max_rows for lo, hi in self.spans)
        self.out = torch.empty(self.world * self.max_rows, *row_shape, device=device, dtype=dtype)
        lo, hi = self.spans[self.rank]
        self.rows = hi - lo
        self.pad: Optional[torch.Tensor] = None
        if self.rows < self.max_rows:
            self.pad = torch.zeros(self.max_rows, *row_shape, device=device, dtype=dtype)
        self.tmp = torch.empty(self.max_rows, *row_shape, device=device, dtype=dtype) if self.ragged else None

    def gather(self, local: torch.Tensor) -> torch.Tensor:
        """-> (total, ...) rows of all ranks, a view of the preallocated buffer (valid until the next call)."""
        if self.world == 1 and not self.always_collective:
            return local
        if local.shape[0] != self.rows:
            raise ValueError(f"rank {self.rank} owns {self.rows} rows, got {local.shape[0]}")
        src = local.contiguous()
        if self.pad is not None:
            self.pad[: self.rows].copy_(src)
            src = self.pad
        if src.is_cuda:
            dist.all_gather_into_tensor(self.out, src)
        else:                                   # gloo has no all_gather_into_tensor: gather into views of the same buffer
            dist.all_gather(list(self.out.view(self.world, self.max_rows, *self.out.shape[1:]).unbind(0)), src)
        if not self.ragged:
            return self.out
        # compact in place, front to back (a shard's destination never lies behind its source; the two may overlap, hence
        # the bounce through the preallocated shard-sized buffer)
        for r, (lo, hi) in enumerate(self.spans):
            s0 = r * self.max_rows
            if s0 != lo:
                self.tmp[: hi - lo].copy_(self.out[s0:s0 + (hi - lo)])
                self.out[lo:hi].copy_(self.tmp[: hi - lo])
        return self.out[: self.total]


def all_gather_rows(local: torch.Tensor, total: int) -> torch.Tensor:
    """One-shot form of ``RowGatherer`` (allocates its buffers; a loop should keep a ``RowGatherer``)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    return RowGatherer(total, tuple(local.shape[1:]), local.device, local.dtype).gather(local)
