"""Host-side schedule tables of the samplers (product code; does not touch the oracle).

The per-timestep buffers are computed exactly as the reference registers them
(float64 schedule -> float32 buffers, transformer4planning/models/utils.py:134-168
and models/diffusion_loss/TrajRefiner.py:160-205); the coefficient rows handed to
``adx_engine_set_schedule`` are derived from those float32 buffers with float32
torch arithmetic, in the order the reference evaluates them.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence

import torch

BUFFER_NAMES = (
    "betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
    "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
    "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2",
)


def linear_beta_schedule(timesteps: int) -> torch.Tensor:
    scale = 1000 / timesteps
    return torch.linspace(scale * 0.0001, scale * 0.02, timesteps, dtype=torch.float64)


def _betas_from_alphas_cumprod(ac: torch.Tensor) -> torch.Tensor:
    ac = ac / ac[0]
    return torch.clip(1 - (ac[1:] / ac[:-1]), 0, 0.999)


def cosine_beta_schedule(timesteps: int, s: float = 0.008) -> torch.Tensor:
    t = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float64) / timesteps
    return _betas_from_alphas_cumprod(torch.cos((t + s) / (1 + s) * math.pi * 0.5) ** 2)


def sigmoid_beta_schedule(timesteps: int, start: float = -3, end: float = 3, tau: float = 1) -> torch.Tensor:
    t = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float64) / timesteps
    v_start = torch.tensor(start / tau).sigmoid()       # float32 end points, as in the reference
    v_end = torch.tensor(end / tau).sigmoid()
    ac = (-((t * (end - start) + start) / tau).sigmoid() + v_end) / (v_end - v_start)
    return _betas_from_alphas_cumprod(ac)


def make_betas(name: str, timesteps: int) -> torch.Tensor:
    if name == "linear":
        return linear_beta_schedule(timesteps)
    if name == "cosine":
        return cosine_beta_schedule(timesteps)
    if name == "sigmoid":
        return sigmoid_beta_schedule(timesteps)
    raise ValueError(f"unknown beta schedule {name}")


def diffusion_buffers(betas: torch.Tensor, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    betas = betas.to(torch.float64)
    alphas = 1.0 - betas
    ac = torch.cumprod(alphas, dim=0)
    ac_prev = torch.cat([torch.ones(1, dtype=torch.float64), ac[:-1]])
    post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
    out = {
        "betas": betas,
        "alphas_cumprod": ac,
        "alphas_cumprod_prev": ac_prev,
        "sqrt_alphas_cumprod": torch.sqrt(ac),
        "sqrt_one_minus_alphas_cumprod": torch.sqrt(1.0 - ac),
        "log_one_minus_alphas_cumprod": torch.log(1.0 - ac),
        "sqrt_recip_alphas_cumprod": torch.sqrt(1.0 / ac),
        "sqrt_recipm1_alphas_cumprod": torch.sqrt(1.0 / ac - 1),
        "posterior_variance": post_var,
        "posterior_log_variance_clipped": torch.log(post_var.clamp(min=1e-20)),
        "posterior_mean_coef1": betas * torch.sqrt(ac_prev) / (1.0 - ac),
        "posterior_mean_coef2": (1.0 - ac_prev) * torch.sqrt(alphas) / (1.0 - ac),
    }
    return {k: v.to(dtype) for k, v in out.items()}


def loss_weight(alphas_cumprod64: torch.Tensor, objective: str, min_snr_gamma: float = 5.0, clip: bool = True) -> torch.Tensor:
    snr = alphas_cumprod64 / (1 - alphas_cumprod64)
    c = snr.clone()
    if clip:
        c.clamp_(max=min_snr_gamma)
    if objective == "pred_noise":
        return c / snr
    if objective == "pred_x0":
        return c
    if objective == "pred_v":
        return c / (snr + 1)
    raise ValueError(objective)


def _row(t, a0=0.0, a1=0.0, clip=300.0, eps_mode=0.0, b0=0.0, b1=0.0, c0=0.0, c1=0.0, c2=0.0, c3=0.0, z_index=-1.0) -> List[float]:
    return [float(v) for v in (t, a0, a1, clip, eps_mode, b0, b1, c0, c1, c2, c3, z_index)]


def _x0_coefs(b: Dict[str, torch.Tensor], objective: str, t: int):
    if objective == "pred_noise":
        return float(b["sqrt_recip_alphas_cumprod"][t]), -float(b["sqrt_recipm1_alphas_cumprod"][t])
    if objective == "pred_x0":
        return 0.0, 1.0
    if objective == "pred_v":
        return float(b["sqrt_alphas_cumprod"][t]), -float(b["sqrt_one_minus_alphas_cumprod"][t])
    raise ValueError(objective)


def ddpm_rows(b: Dict[str, torch.Tensor], objective: str, clamp: float = 300.0, noise_at_every_step: bool = False,
              stochastic: bool = True) -> List[List[float]]:
    """Ancestral sampling rows, t = N-1 ... 0 (TrajRefiner.py:296-333; decoder twin diffusion_decoder.py:434-457).

    ``noise_at_every_step`` reproduces the twin, which draws a tensor at t = 0 too (masked out of x);
    ``stochastic=False`` is its ``determin`` mode (all noise zero).
    """
    n = b["betas"].shape[0]
    rows, zi = [], 0
    for t in reversed(range(n)):
        a0, a1 = _x0_coefs(b, objective, t)
        sigma = float(torch.exp(0.5 * b["posterior_log_variance_clipped"][t])) if (t > 0 and stochastic) else 0.0
        z = -1.0
        if stochastic and (t > 0 or noise_at_every_step):
            z = float(zi)
            zi += 1
        rows.append(_row(t, a0, a1, clamp, c0=float(b["posterior_mean_coef1"][t]), c1=float(b["posterior_mean_coef2"][t]),
                         c3=sigma, z_index=z))
    return rows


def ddim_time_pairs(total_timesteps: int, sampling_timesteps: int):
    times = torch.linspace(-1, total_timesteps - 1, steps=sampling_timesteps + 1)
    times = list(reversed(times.int().tolist()))
    return list(zip(times[:-1], times[1:]))


def ddim_rows(b: Dict[str, torch.Tensor], objective: str, sampling_timesteps: int, eta: float, clamp: float = 300.0):
    """In-repo DDIM (TrajRefiner.py:335-376): x0 clipped, eps re-derived from the clipped x0."""
    n = b["betas"].shape[0]
    ac = b["alphas_cumprod"]
    rows, zi = [], 0
    for t, t_next in ddim_time_pairs(n, sampling_timesteps):
        a0, a1 = _x0_coefs(b, objective, t)
        b0 = float(b["sqrt_recip_alphas_cumprod"][t])
        b1 = float(1.0 / b["sqrt_recipm1_alphas_cumprod"][t])
        if t_next < 0:
            rows.append(_row(t, a0, a1, clamp, 1.0, b0, b1, c0=1.0))
            continue
        alpha, alpha_next = ac[t], ac[t_next]
        sigma = eta * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
        c = (1 - alpha_next - sigma ** 2).sqrt()
        # the reference draws a tensor at every such step even when sigma == 0 (so draw zi keeps its place in the order the
        # reference consumes its generator), but a draw multiplied by zero is never read: it gets no index
        rows.append(_row(t, a0, a1, clamp, 1.0, b0, b1, c0=float(alpha_next.sqrt()), c2=float(c), c3=float(sigma),
                         z_index=float(zi) if float(sigma) != 0.0 else -1.0))
        zi += 1
    return rows


# ---- diffusers-style schedulers (third-party semantics restated from their public description; not pinned
# ---- by any reference output: the reference neither vendors nor pins the library)
def squaredcos_cap_v2(n: int, max_beta: float = 0.999) -> torch.Tensor:
    def abar(u):
        return math.cos((u + 0.008) / 1.008 * math.pi / 2) ** 2
    return torch.tensor([min(1 - abar((i + 1) / n) / abar(i / n), max_beta) for i in range(n)], dtype=torch.float32)


def diffusers_betas(kind: str, n: int, beta_start: float = 1e-4, beta_end: float = 2e-2) -> torch.Tensor:
    if kind == "squaredcos_cap_v2":
        return squaredcos_cap_v2(n)
    if kind == "linear":
        return torch.linspace(beta_start, beta_end, n, dtype=torch.float32)
    if kind == "scaled_linear":
        return torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2
    raise ValueError(kind)


def diffusers_timesteps(n_train: int, n_infer: int) -> List[int]:
    r = n_train // n_infer
    return [k * r for k in reversed(range(n_infer))]


def diffusers_rows(betas: torch.Tensor, n_infer: int, kind: str, clip_sample: bool = True, clip_range: float = 1.0,
                   prediction_type: str = "epsilon") -> List[List[float]]:
    """DDPMScheduler.step (variance fixed_small) / DDIMScheduler.step (eta 0) as coefficient rows."""
    if prediction_type != "epsilon":
        raise ValueError("only prediction_type='epsilon' is supported")
    ac = torch.cumprod(1.0 - betas.to(torch.float64), dim=0)
    n = ac.shape[0]
    clip = clip_range if clip_sample else 3.0e38
    rows, zi = [], 0
    for t in diffusers_timesteps(n, n_infer):
        t_prev = t - n // n_infer
        a_t = float(ac[t])
        a_prev = float(ac[t_prev]) if t_prev >= 0 else 1.0
        a0, a1 = 1.0 / math.sqrt(a_t), -math.sqrt(1 - a_t) / math.sqrt(a_t)
        if kind == "ddim":
            rows.append(_row(t, a0, a1, clip, 0.0, c0=math.sqrt(a_prev), c2=math.sqrt(1 - a_prev)))
        elif kind == "ddpm":
            cur_alpha = a_t / a_prev
            cur_beta = 1 - cur_alpha
            c0 = math.sqrt(a_prev) * cur_beta / (1 - a_t)
            c1 = math.sqrt(cur_alpha) * (1 - a_prev) / (1 - a_t)
            c3, z = 0.0, -1.0
            if t > 0:
                c3 = math.sqrt(max((1 - a_prev) / (1 - a_t) * cur_beta, 1e-20))
                z = float(zi)
                zi += 1
            rows.append(_row(t, a0, a1, clip, 0.0, c0=c0, c1=c1, c3=c3, z_index=z))
        else:
            raise ValueError(kind)
    return rows


def n_noise_draws(rows: Sequence[Sequence[float]]) -> int:
    """Number of noise tensors a schedule consumes after the initial one."""
    return int(max([r[11] for r in rows] + [-1.0])) + 1
