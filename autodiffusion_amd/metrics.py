"""Evaluation metrics of the reference's ``compute_metrics`` (transformer4planning/trainer.py:184-305).

For tensors on the GPU the per-scenario quantities come from one fused HIP kernel (``adx_op_trajectory_metrics``), so only
B x 18 floats -- not every trajectory -- need to leave the device; CPU tensors take the same formulas in torch (used by
the host-side tests and by callers that already hold the trajectories on the host).
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch

FDE_THRESHOLD = 8.0
ADE_THRESHOLD = 8.0
HEADING_ERROR_THRESHOLD = 0.8
MISS_THRESHOLD = (6.0, 8.0, 16.0)

COLUMNS = ("ade1", "ade3", "ade5", "ade8", "fde1", "fde3", "fde5", "fde8", "ahe1", "ahe3", "ahe5", "ahe8", "fhe3", "fhe5", "fhe8",
           "maxd3", "maxd5", "maxd8")


def displacement_errors(pred: torch.Tensor, label: torch.Tensor) -> torch.Tensor:
    """Per-point L2 distance over (x, y): (B, S)."""
    return torch.sqrt((pred[..., 0] - label[..., 0]) ** 2 + (pred[..., 1] - label[..., 1]) ** 2)


def _wrap(a: torch.Tensor) -> torch.Tensor:
    return torch.atan2(torch.sin(a), torch.cos(a))


def per_scenario(pred: torch.Tensor, label: torch.Tensor) -> torch.Tensor:
    """(B, 18) table of the per-scenario quantities, columns as in ``COLUMNS``."""
    B, S, C = pred.shape
    if pred.is_cuda:
        from . import _lib
        lib = _lib.load()
        p, l = _lib.f32c(pred), _lib.f32c(label)
        out = torch.empty(B, 18, device=pred.device, dtype=torch.float32)
        _lib.check(lib.adx_op_trajectory_metrics(_lib.ptr(p), _lib.ptr(l), B, S, C, _lib.ptr(out), _lib.current_stream()),
                   "trajectory_metrics")
        return out
    rs = 80 // S if S < 80 else 1
    err = displacement_errors(pred.float(), label.float())
    h = [10 // rs, 30 // rs, 50 // rs, 80 // rs]
    cols = [err[:, :n].mean(1) for n in h] + [err[:, 9 // rs], err[:, 29 // rs], err[:, 49 // rs], err[:, -1]]
    if C == 4:
        he = _wrap(_wrap(pred[..., 3].float()) - _wrap(label[..., 3].float())).abs()
        cols += [he[:, 0]] + [he[:, :n].mean(1) for n in h[1:]] + [he[:, 29 // rs], he[:, 49 // rs], he[:, 79 // rs]]
    else:
        cols += [torch.zeros(B)] * 7
    cols += [err[:, :n].max(1)[0] for n in h[1:]]
    return torch.stack(cols, dim=1)


def ade_fde(pred: torch.Tensor, label: torch.Tensor) -> Dict[str, torch.Tensor]:
    """ADE / FDE at the 1, 3, 5 and 8 s horizons, per scenario (80 points at 10 Hz)."""
    t = per_scenario(pred, label)
    return {k: t[:, i] for i, k in enumerate(COLUMNS[:8])}


def compute_scores(table: torch.Tensor, scenario15s_id) -> Tuple[float, float]:
    """``compute_scores`` of the evaluation entry (transformer4planning/utils/nuplan_utils.py:229-314; called at trainer.py:363):
    the frames (rows of the (B, 18) per-frame ``table``) are grouped by ``scenario15s_id``; per group and per quantity q in
    {ade, fde, ahe, fhe}: ``q_score = max(1 - mean_over_frames(q3 + q5 + q8) / 3 / threshold, 0)`` -- the clamp acts on the GROUP
    mean, unlike the per-frame ``*_score`` keys of ``compute_metrics``; ``miss_score = 0`` if five or more frames of the group
    miss (maximum displacement above 6 / 8 / 16 m within 3 / 5 / 8 s), else 1; ``score = miss_score * (ade + fde + 2 ahe +
    2 fhe) / 6``.  Returns (mean score over groups, mean miss_score over groups).  Two-channel trajectories have no heading
    columns (zeros): their heading scores are 1, where the reference raises a KeyError that its caller turns into (0, 0).

    No reference output pins this function: its module does not import here (``shapely`` is absent); it is checked against a
    restatement of the same lines and analytic cases."""
    t = table.detach().double().cpu()
    ids = torch.as_tensor(scenario15s_id).reshape(-1).to(torch.int64).cpu()
    if ids.shape[0] != t.shape[0]:
        raise ValueError(f"{t.shape[0]} frames but {ids.shape[0]} scenario ids")
    c = {k: t[:, i] for i, k in enumerate(COLUMNS)}
    _, inv = torch.unique(ids, return_inverse=True)
    n_groups = int(inv.max()) + 1 if inv.numel() else 0
    if n_groups == 0:
        return 0.0, 0.0
    count = torch.zeros(n_groups, dtype=torch.float64).index_add_(0, inv, torch.ones_like(inv, dtype=torch.float64))

    def group_mean(v):
        return torch.zeros(n_groups, dtype=torch.float64).index_add_(0, inv, v) / count

    def score(q, threshold):
        return (1 - group_mean(c[f"{q}3"] + c[f"{q}5"] + c[f"{q}8"]) / 3 / threshold).clamp(min=0)
    miss = ((c["maxd3"] > MISS_THRESHOLD[0]) | (c["maxd5"] > MISS_THRESHOLD[1]) | (c["maxd8"] > MISS_THRESHOLD[2])).double()
    n_miss = torch.zeros(n_groups, dtype=torch.float64).index_add_(0, inv, miss)
    miss_score = (n_miss < 5).double()
    total = (score("ade", ADE_THRESHOLD) + score("fde", FDE_THRESHOLD) + 2 * score("ahe", HEADING_ERROR_THRESHOLD)
             + 2 * score("fhe", HEADING_ERROR_THRESHOLD)) / 6
    return float((miss_score * total).mean()), float(miss_score.mean())


def compute_metrics(pred: torch.Tensor, label: torch.Tensor, scenario15s_id=None) -> Dict[str, float]:
    """Batch means under the reference's result keys (``*_gen`` horizons, ``metric_*``, per-frame scores, miss rates); with
    ``scenario15s_id`` (one id per frame) also ``average_score`` and ``miss_score`` of :func:`compute_scores`."""
    table = per_scenario(pred, label)
    t = table.double()
    c = {k: t[:, i] for i, k in enumerate(COLUMNS)}
    r: Dict[str, float] = {}
    for n in (1, 3, 5, 8):
        r[f"ade_horizon{n}_gen"] = float(c[f"ade{n}"].mean())
        r[f"fde_horizon{n}_gen"] = float(c[f"fde{n}"].mean())
    avg_ade = (c["ade3"] + c["ade5"] + c["ade8"]) / 3
    avg_fde = (c["fde3"] + c["fde5"] + c["fde8"]) / 3
    r["metric_ade"], r["metric_fde"] = float(avg_ade.mean()), float(avg_fde.mean())
    r["ade_score"] = float((1 - avg_ade / ADE_THRESHOLD).clamp(min=0).mean())
    r["fde_score"] = float((1 - avg_fde / FDE_THRESHOLD).clamp(min=0).mean())
    if pred.shape[-1] == 4:
        r["ahe_1_gen"] = float(c["ahe1"].mean())
        for n in (3, 5, 8):
            r[f"ahe_horizon{n}_gen"] = float(c[f"ahe{n}"].mean())
            r[f"fhe_horizon{n}_gen"] = float(c[f"fhe{n}"].mean())
        avg_ahe = (c["ahe3"] + c["ahe5"] + c["ahe8"]) / 3
        avg_fhe = (c["fhe3"] + c["fhe5"] + c["fhe8"]) / 3
        r["metric_ahe"], r["metric_fhe"] = float(avg_ahe.mean()), float(avg_fhe.mean())
        r["ahe_score"] = float((1 - avg_ahe / HEADING_ERROR_THRESHOLD).clamp(min=0).mean())
        r["fhe_score"] = float((1 - avg_fhe / HEADING_ERROR_THRESHOLD).clamp(min=0).mean())
    miss = [(c[f"maxd{n}"] > th).double() for n, th in zip((3, 5, 8), MISS_THRESHOLD)]
    r["miss_rate3"], r["miss_rate5"], r["miss_rate8"] = (float(m.mean()) for m in miss)
    r["total_miss_rate"] = float(((miss[0] + miss[1] + miss[2]) >= 1).double().mean())
    if scenario15s_id is not None:
        r["average_score"], r["miss_score"] = compute_scores(table, scenario15s_id)
    return r
