"""Displacement metrics of the evaluation entry (transformer4planning/trainer.py:184-234), on device tensors."""
from __future__ import annotations

from typing import Dict

import torch


def displacement_errors(pred: torch.Tensor, label: torch.Tensor) -> torch.Tensor:
    """Per-point L2 distance over (x, y): (B, S)."""
    return torch.sqrt((pred[..., 0] - label[..., 0]) ** 2 + (pred[..., 1] - label[..., 1]) ** 2)


def ade_fde(pred: torch.Tensor, label: torch.Tensor) -> Dict[str, torch.Tensor]:
    """ADE / FDE at the 1, 3, 5 and 8 s horizons, per scenario (80 points at 10 Hz)."""
    err = displacement_errors(pred, label)
    return {
        "ade1": err[:, :10].mean(1), "ade3": err[:, :30].mean(1), "ade5": err[:, :50].mean(1), "ade8": err[:, :80].mean(1),
        "fde1": err[:, 9], "fde3": err[:, 29], "fde5": err[:, 49], "fde8": err[:, -1],
    }
