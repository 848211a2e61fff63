"""Prior producer, last stage (oracle; test infrastructure only): backbone hidden states -> ``traj_logits``, the trajectory prior
the refiner conditions on, and the backbone's trajectory loss.

Restates, in plain torch on the CPU:

* ``TrajectoryDecoder.generate_trajs``     transformer4planning/models/decoder/base.py:253-260
* ``TrajectoryDecoder.compute_traj_loss``  transformer4planning/models/decoder/base.py:95-127 (the slice) and :238-241 (the default
                                           nuplan branch with ``mean_circular_loss``), :245 (rescale)
* ``mean_circular_error``                  transformer4planning/models/decoder/base.py:9-35
* ``DecoderResCat`` / ``MLP``              transformer4planning/libs/mlp.py:23-55 (via ``oracle.nets.res_cat``)

Pinned by ``tests/golden/g20_trajectory_decoder.npz``, recorded from the reference's own ``TrajectoryDecoder``.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import torch

from . import nets

SD = Dict[str, torch.Tensor]


def decode(sd: SD, h: torch.Tensor, kind: str) -> torch.Tensor:
    """``self.model``: one Linear ('linear') or DecoderResCat (anything else), base.py:48-56."""
    if kind == "linear":
        return nets.linear(h, sd, "model")
    return nets.res_cat(h, sd, "model")


def generate_trajs(sd: SD, hidden_output: torch.Tensor, pred_length: int, kind: str = "mlp") -> torch.Tensor:
    return decode(sd, hidden_output[:, -pred_length - 1:-1, :], kind)


def mean_circular_error(pred: torch.Tensor, true: torch.Tensor) -> torch.Tensor:
    wrap = lambda a: torch.atan2(torch.sin(a), torch.cos(a))  # noqa: E731
    return wrap(wrap(true) - wrap(pred)).square().mean()


def traj_slice(hidden_output: torch.Tensor, pred_length: int, context_length: Optional[int], selected_indices: Optional[Sequence[int]]):
    if context_length is None:
        return hidden_output[:, -pred_length - 1:-1, :]
    k = len(selected_indices) if selected_indices is not None else 0
    return hidden_output[:, context_length - 1 + k:context_length + pred_length - 1 + k, :]


def compute_traj_loss(sd: SD, hidden_output, label, pred_length: int, context_length=None, selected_indices=None, kind: str = "mlp",
                      loss_fn: str = "mse", no_yaw_loss: bool = False, rescale: float = 1.0):
    logits = decode(sd, traj_slice(hidden_output, pred_length, context_length, selected_indices), kind)
    d = logits[..., :2] - label[..., :2]
    if "mse" in loss_fn:
        loss = d.square().mean()
    else:
        loss = torch.where(d.abs() < 1.0, 0.5 * d * d, d.abs() - 0.5).mean()
    if not no_yaw_loss:
        loss = loss + mean_circular_error(logits[..., -1], label[..., -1])
    return loss * rescale, logits
