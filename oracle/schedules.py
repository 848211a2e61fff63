"""Noise schedules and derived per-timestep buffers (oracle; test infrastructure only).

Follows transformer4planning/models/utils.py:134-168 (beta schedules, float64),
transformer4planning/models/diffusion_loss/TrajRefiner.py:151-227 (buffers, cast
to float32 at registration) and
transformer4planning/models/diffusion_loss/diffusion_decoder.py:305-331 (decoder
twin: same buffers, kept in the schedule's own dtype).
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def betas(name: str, n: int) -> np.ndarray:
    """float64 betas, utils.py:134-168."""
    if name == "linear":
        scale = 1000.0 / n
        return np.linspace(scale * 1e-4, scale * 2e-2, n, dtype=np.float64)
    t = np.linspace(0.0, float(n), n + 1, dtype=np.float64) / n
    if name == "cosine":
        s = 0.008
        ac = np.cos((t + s) / (1.0 + s) * math.pi * 0.5) ** 2
    elif name == "sigmoid":
        start, end, tau = -3.0, 3.0, 1.0
        # the two end-point sigmoids are evaluated in float32 (utils.py:163-164 builds them from
        # default-dtype tensors), everything else in float64
        import torch
        v0 = float(torch.tensor(start / tau).sigmoid())
        v1 = float(torch.tensor(end / tau).sigmoid())
        ac = (-_sigmoid((t * (end - start) + start) / tau) + v1) / (v1 - v0)
    else:
        raise ValueError(f"unknown beta schedule {name}")
    ac = ac / ac[0]
    b = 1.0 - ac[1:] / ac[:-1]
    return np.clip(b, 0.0, 0.999)


def buffers(name: str, n: int, objective: str = "pred_noise", f32: bool = True,
            min_snr_gamma: float = 5.0) -> Dict[str, np.ndarray]:
    """All registered buffers of the in-repo diffusion wrapper (TrajRefiner.py:160-227)."""
    b = betas(name, n)
    return buffers_from_betas(b, objective, f32, min_snr_gamma)


def buffers_from_betas(b: np.ndarray, objective: str = "pred_noise", f32: bool = True,
                       min_snr_gamma: float = 5.0) -> Dict[str, np.ndarray]:
    a = 1.0 - b
    ac = np.cumprod(a)
    ac_prev = np.concatenate([[1.0], ac[:-1]])
    post_var = b * (1.0 - ac_prev) / (1.0 - ac)
    out = {
        "betas": b,
        "alphas_cumprod": ac,
        "alphas_cumprod_prev": ac_prev,
        "sqrt_alphas_cumprod": np.sqrt(ac),
        "sqrt_one_minus_alphas_cumprod": np.sqrt(1.0 - ac),
        "log_one_minus_alphas_cumprod": np.log(1.0 - ac),
        "sqrt_recip_alphas_cumprod": np.sqrt(1.0 / ac),
        "sqrt_recipm1_alphas_cumprod": np.sqrt(1.0 / ac - 1.0),
        "posterior_variance": post_var,
        "posterior_log_variance_clipped": np.log(np.maximum(post_var, 1e-20)),
        "posterior_mean_coef1": b * np.sqrt(ac_prev) / (1.0 - ac),
        "posterior_mean_coef2": (1.0 - ac_prev) * np.sqrt(a) / (1.0 - ac),
    }
    snr = ac / (1.0 - ac)
    clipped = np.minimum(snr, min_snr_gamma)
    if objective == "pred_noise":
        out["loss_weight"] = clipped / snr
    elif objective == "pred_x0":
        out["loss_weight"] = clipped
    elif objective == "pred_v":
        out["loss_weight"] = clipped / (snr + 1.0)
    else:
        raise ValueError(objective)
    if f32:
        out = {k: v.astype(np.float32) for k, v in out.items()}
    return out


def twin_betas(name: str, n: int, linear_start: float = 0.01, linear_end: float = 0.9) -> np.ndarray:
    """Decoder-twin schedules (diffusion_decoder.py:305-306).

    'cosine' is the shared helper.  'linear' is called there with
    beta_start/beta_end keywords the fork's helper does not accept (it raises in
    the reference); it is restated as linspace(linear_start, linear_end, n) -- the
    evident intent -- and is NOT pinned by any reference output.
    """
    if name == "cosine":
        return betas("cosine", n)
    if name == "linear":
        return np.linspace(linear_start, linear_end, n, dtype=np.float64)
    raise ValueError(name)


# ---- diffusers-style schedule (third-party `diffusers`, version unpinned, absent: PARITY UNPINNED)
def squaredcos_cap_v2(n: int, max_beta: float = 0.999) -> np.ndarray:
    def abar(u):
        return math.cos((u + 0.008) / 1.008 * math.pi / 2) ** 2
    return np.array([min(1.0 - abar((i + 1) / n) / abar(i / n), max_beta) for i in range(n)], dtype=np.float32)


def diffusers_betas(kind: str, n: int, beta_start: float = 1e-4, beta_end: float = 2e-2) -> np.ndarray:
    if kind == "squaredcos_cap_v2":
        return squaredcos_cap_v2(n)
    if kind == "linear":
        return np.linspace(beta_start, beta_end, n, dtype=np.float32)
    if kind == "scaled_linear":
        return np.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=np.float32) ** 2
    raise ValueError(kind)
