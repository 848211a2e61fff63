"""CPU restatement of the multi-modal reduction ``select_k_from_mc`` (test infrastructure only).

Follows transformer4planning/models/utils.py:120-132 -> utils/modify_traj_utils.py: farthest-point sampling of
``k`` trajectories on their END POINTS starting from sample 0 (``farthest_point_sample_PointCloud``, :8-30), then
25 EM iterations of a diagonal Gaussian mixture over the whole trajectories (``run_EM_algorithm`` /
``_compute_EM_algorithm``, :124-219): the input samples are mixture components of equal weight 1/M and variance
0.1, the k modes start at the sampled trajectories with variance 0.1 and weight 1/k (the scores handed in are
multiplied by zero there, :183-184), and the result is sorted by mode weight, descending.
Pinned by tests/golden/g11_select_modes.npz (recorded from the reference).
"""
from __future__ import annotations

import math

import torch


def farthest_point_indices(end_points: torch.Tensor, k: int) -> torch.Tensor:
    """end_points (B, M, C) -> (B, k) int64; sample 0 first, then repeatedly the point farthest from the chosen set."""
    B, M, _ = end_points.shape
    idx = torch.zeros(B, k, dtype=torch.long)
    dist = torch.full((B, M), 1e10)
    far = torch.zeros(B, dtype=torch.long)
    rows = torch.arange(B)
    for i in range(k):
        idx[:, i] = far
        c = end_points[rows, far][:, None, :]
        d = ((end_points - c) ** 2).sum(-1)
        dist = torch.minimum(dist, d)
        far = dist.max(-1)[1]
    return idx


def select_modes(traj: torch.Tensor, k: int, em_iter: int = 25, org_sigma: float = 0.1, init_sigma: float = 0.1):
    """traj (B, M, L, D) -> (modes (B, k, L, D), weights (B, k)) sorted by weight, descending."""
    B, M, L, D = traj.shape
    idx = farthest_point_indices(traj[:, :, -1, :], k)
    x = traj.reshape(B, M, L * D).to(torch.float32)
    mean = x.mean(1, keepdim=True)
    x = x - mean
    mu = torch.gather(x, 1, idx[:, :, None].expand(B, k, L * D)).clone()
    sig = torch.full_like(mu, init_sigma)
    q = torch.full((B, k), 1.0 / k)
    oq = torch.full((B, M), 1.0 / M)
    for _ in range(em_iter):
        delta = x[:, :, None, :] - mu[:, None, :, :]                                   # (B, M, k, LD)
        s = sig[:, None, :, :] + 1e-5
        logn = (-0.5 * torch.log(2.0 * math.pi * s) - delta ** 2 / (2.0 * s)).sum(-1)  # (B, M, k)
        n = torch.exp(logn - logn.max(-1, keepdim=True)[0])
        w = q[:, None, :] * n
        w = w / w.sum(-1, keepdim=True)
        ow = oq[:, :, None] * w                                                        # (B, M, k)
        new_q = ow.sum(1)
        new_mu = torch.einsum("bik,bid->bkd", ow, x) / (new_q + 1e-10)[:, :, None]
        d2 = (x[:, :, None, :] - new_mu[:, None, :, :]) ** 2
        new_sig = torch.einsum("bik,bikd->bkd", ow, org_sigma + d2) / (new_q + 1e-10)[:, :, None]
        q, mu, sig = new_q, new_mu, new_sig
    q, order = torch.sort(q, descending=True, dim=1)
    mu = torch.gather(mu, 1, order[:, :, None].expand(B, k, L * D))
    return (mu + mean).reshape(B, k, L, D), q
