"""Sampling loops, data normalisation and metrics (oracle; test infrastructure only).

In-repo diffusion maths: transformer4planning/models/diffusion_loss/TrajRefiner.py
  :238-313 (x0 / eps conversions, posterior, p_sample), :315-333 (p_sample_loop),
  :335-376 (ddim_sample), :378-386 (sample + residual);
decoder twin: transformer4planning/models/diffusion_loss/diffusion_decoder.py:377-479;
evaluation-time loss: TrajRefiner.py:408-452, DiffusionAugmentation.py:75-135, diffusion_decoder.py:346-365;
normalisation: transformer4planning/models/utils.py:56-80;
chunk loop: transformer4planning/models/backbone/DiffusionAugmentation.py:137-195;
flattened-chunk wrapper: transformer4planning/models/backbone/ExplicitDiT.py:139-210 with
  transformer4planning/models/diffusion_loss/traj_diffusion.py:45-81;
metrics: transformer4planning/trainer.py:184-234.

Noise is always host-supplied: ``noise[0]`` initialises x_N, ``noise[1+i]`` is the
i-th tensor the loop draws (see tests/golden/make_golden.py for the order).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import numpy as np
import torch

from . import schedules


# ------------------------------------------------------------------ data normalisation
def normalize(x: torch.Tensor) -> torch.Tensor:
    """utils.py:56-70: x/25-1, y/5, remaining channels copied."""
    y = x.clone()
    y[..., 0] = x[..., 0] / 25 - 1.0
    y[..., 1] = x[..., 1] / 5
    return y


def denormalize(y: torch.Tensor) -> torch.Tensor:
    """utils.py:72-80: (x+1)*25, y*5, channel 2 forced to 0, channel 3 copied."""
    x = y.clone()
    x[..., 0] = (y[..., 0] + 1.0) * 25
    x[..., 1] = y[..., 1] * 5
    if y.shape[-1] > 2:
        x[..., 2] = 0
    return x


def _buf(b: Dict[str, np.ndarray], key: str, t: int) -> torch.Tensor:
    return torch.tensor(b[key][t], dtype=torch.float32)


# ------------------------------------------------------------------ in-repo DDPM / DDIM
def x0_from_output(b, objective: str, x, out, t: int):
    if objective == "pred_noise":
        return _buf(b, "sqrt_recip_alphas_cumprod", t) * x - _buf(b, "sqrt_recipm1_alphas_cumprod", t) * out
    if objective == "pred_x0":
        return out
    if objective == "pred_v":
        return _buf(b, "sqrt_alphas_cumprod", t) * x - _buf(b, "sqrt_one_minus_alphas_cumprod", t) * out
    raise ValueError(objective)


def ddpm_sample(denoise: Callable, b: Dict[str, np.ndarray], objective: str, noise: torch.Tensor,
                clamp: float = 300.0, trace: Optional[dict] = None) -> torch.Tensor:
    """Ancestral sampling in the normalised space.  ``denoise(x, t_int) -> model output``."""
    n = len(b["betas"])
    x = noise[0].clone()
    k = 1
    for t in reversed(range(n)):
        out = denoise(x, t)
        if trace is not None:
            trace.setdefault("x", []).append(x.clone())
            trace.setdefault("model_out", []).append(out.clone())
        x0 = x0_from_output(b, objective, x, out, t).clamp(-clamp, clamp)
        mean = _buf(b, "posterior_mean_coef1", t) * x0 + _buf(b, "posterior_mean_coef2", t) * x
        if t > 0:
            sigma = torch.exp(0.5 * _buf(b, "posterior_log_variance_clipped", t))
            x = mean + sigma * noise[k]
            k += 1
        else:
            x = mean
    return x


def ddim_pairs(n_train: int, n_sample: int) -> List[tuple]:
    """TrajRefiner.py:339-341: linspace(-1, N-1, n+1) truncated to int, reversed, paired."""
    times = torch.linspace(-1, n_train - 1, steps=n_sample + 1).int().tolist()
    times = list(reversed(times))
    return list(zip(times[:-1], times[1:]))


def ddim_sample(denoise: Callable, b: Dict[str, np.ndarray], objective: str, noise: torch.Tensor,
                n_sample: int, eta: float = 0.0, clamp: float = 300.0, trace: Optional[dict] = None) -> torch.Tensor:
    n = len(b["betas"])
    x = noise[0].clone()
    k = 1
    for t, t_next in ddim_pairs(n, n_sample):
        out = denoise(x, t)
        if trace is not None:
            trace.setdefault("x", []).append(x.clone())
            trace.setdefault("model_out", []).append(out.clone())
        x0 = x0_from_output(b, objective, x, out, t).clamp(-clamp, clamp)
        # eps re-derived from the clipped x0 (for pred_noise too: rederive_pred_noise=True)
        eps = (_buf(b, "sqrt_recip_alphas_cumprod", t) * x - x0) / _buf(b, "sqrt_recipm1_alphas_cumprod", t)
        if t_next < 0:
            x = x0
            continue
        a, a_next = _buf(b, "alphas_cumprod", t), _buf(b, "alphas_cumprod", t_next)
        sigma = eta * ((1 - a / a_next) * (1 - a_next) / (1 - a)).sqrt()
        c = (1 - a_next - sigma ** 2).sqrt()
        x = x0 * a_next.sqrt() + c * eps + sigma * noise[k]
        k += 1
    return x


def refiner_sample(denoise_cond: Callable, b, objective: str, transition_info, trajectory_prior, noise,
                   residual: bool = False, auto_normalize: bool = True,
                   n_sample: Optional[int] = None, eta: float = 0.0, trace=None) -> torch.Tensor:
    """DiffusionWrapper.sample for one chunk.  ``denoise_cond(x, t, transition_n, prior_n) -> out``."""
    norm = normalize if auto_normalize else (lambda v: v)
    unnorm = denormalize if auto_normalize else (lambda v: v)
    tra_n, pri_n = norm(transition_info), norm(trajectory_prior)
    fn = lambda x, t: denoise_cond(x, t, tra_n, pri_n)  # noqa: E731
    n = len(b["betas"])
    if n_sample is None or n_sample >= n:
        x = ddpm_sample(fn, b, objective, noise, trace=trace)
    else:
        x = ddim_sample(fn, b, objective, noise, n_sample, eta, trace=trace)
    x = unnorm(x)
    return x + trajectory_prior if residual else x


def refiner_generate(denoise_cond: Callable, b, objective: str, maps_unused, prior_full, init_x, noise,
                     frame_stack: int, residual: bool = False, auto_normalize: bool = True,
                     n_sample: Optional[int] = None, eta: float = 0.0) -> torch.Tensor:
    """Chunk loop of the generate() wrapper: (B,80,4) prior -> (B,80,4) refined trajectory.

    ``noise`` is (n_chunks, n_draws, B, fs, 4).
    """
    B, S, C = prior_full.shape
    if S % frame_stack:
        raise ValueError("Number of frames must be divisible by frame stack size")
    n_chunks = S // frame_stack
    pri = prior_full.reshape(B, n_chunks, frame_stack, C).permute(1, 0, 2, 3)
    transition = init_x[None].expand(B, frame_stack, C)
    preds = []
    for c in range(n_chunks):
        pred = refiner_sample(denoise_cond, b, objective, transition, pri[c], noise[c], residual, auto_normalize,
                              n_sample, eta)
        transition = pred
        preds.append(pred)
    return torch.stack(preds).permute(1, 0, 2, 3).reshape(B, S, C)


# ------------------------------------------------------------------ evaluation-time value of the training loss
def refiner_loss(denoise_cond: Callable, b: Dict[str, np.ndarray], objective: str, transition_info, trajectory_prior, x_start,
                 t: torch.Tensor, noise: torch.Tensor, auto_normalize: bool = True) -> torch.Tensor:
    """DiffusionWrapper.p_losses (TrajRefiner.py:417-452) with the random draws supplied: ``t`` (B,) int64, ``noise`` like
    ``x_start``.  ``denoise_cond(x, t_tensor, transition_n, prior_n) -> out`` (no dropout).  Returns the scalar loss."""
    norm = normalize if auto_normalize else (lambda v: v)
    x0, tra_n, pri_n = norm(x_start), norm(transition_info), norm(trajectory_prior)
    ext = lambda key: torch.as_tensor(b[key], dtype=torch.float32)[t].reshape(-1, *([1] * (x0.dim() - 1)))  # noqa: E731
    sa, s1 = ext("sqrt_alphas_cumprod"), ext("sqrt_one_minus_alphas_cumprod")
    x = sa * x0 + s1 * noise
    out = denoise_cond(x, t, tra_n, pri_n)
    target = {"pred_noise": noise, "pred_x0": x0, "pred_v": sa * noise - s1 * x0}[objective]
    loss = ((out - target) ** 2).reshape(x0.shape[0], -1).mean(dim=1)
    return (loss * torch.as_tensor(b["loss_weight"], dtype=torch.float32)[t]).mean()


def refiner_loss_grads(sd: Dict[str, torch.Tensor], forward: Callable, b: Dict[str, np.ndarray], objective: str, maps_info, transition_info,
                       trajectory_prior, x_start, t: torch.Tensor, noise: torch.Tensor, auto_normalize: bool = True):
    """What ``loss.backward()`` leaves in the denoiser's parameters after TrajectoryRefiner.forward (TrajRefiner.py:100-121 ->
    p_losses :417-452): reverse-mode differentiation of :func:`refiner_loss` over the restated network
    ``forward(sd, x, t, maps, transition_n, prior_n)``.  Returns (loss, {parameter name: gradient}); parameters that do not
    reach the loss (the mask buffers) are left out."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point()}
    full = dict(sd)
    full.update(leaves)
    loss = refiner_loss(lambda x, tt, tra_n, pri_n: forward(full, x, tt, maps_info, tra_n, pri_n), b, objective, transition_info,
                        trajectory_prior, x_start, t, noise, auto_normalize)
    names = list(leaves)
    grads = torch.autograd.grad(loss, [leaves[k] for k in names], allow_unused=True)
    return loss.detach(), {k: g for k, g in zip(names, grads) if g is not None}


def refiner_forward(denoise_cond: Callable, b, objective: str, prior_full, label_full, init_x, t, noise, frame_stack: int,
                    residual: bool = False, auto_normalize: bool = True, rescale: float = 1.0):
    """DiffusionAugmentation.forward (DiffusionAugmentation.py:75-135), loss part: chunk c is scored with the prior chunk,
    the label chunk (label - prior with ``residual``) and a transition that is ``init_x`` for chunk 0 and ZERO afterwards
    (TrajectoryRefiner.forward returns all-zero logits as its "prediction", TrajRefiner.py:115-121).
    ``t`` (n_chunks, B), ``noise`` (n_chunks, B, fs, 4).  Returns (mean loss, per-chunk losses)."""
    B, S, C = prior_full.shape
    n_chunks = S // frame_stack
    label = label_full - prior_full if residual else label_full
    lab = label.reshape(B, n_chunks, frame_stack, C)
    pri = prior_full.reshape(B, n_chunks, frame_stack, C)
    transition = init_x[None].expand(B, frame_stack, C)
    losses = []
    for c in range(n_chunks):
        losses.append(refiner_loss(denoise_cond, b, objective, transition, pri[:, c], lab[:, c], t[c], noise[c], auto_normalize) * rescale)
        transition = torch.zeros(B, frame_stack, C)
    losses = torch.stack(losses)
    return losses.mean(), losses


# ------------------------------------------------------------------ counter-based Gaussian noise (the library's generator)
def philox4x32_10(counter, key0: int, key1: int):
    """Four uint32 words (as uint64 arrays) per counter {c0, c1, c2, c3} (arrays or ints) under the 64-bit key {key0, key1}."""
    c = [np.asarray(v, dtype=np.uint64) for v in counter]
    k0, k1 = np.uint64(key0), np.uint64(key1)
    M0, M1, MASK = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [((p1 >> np.uint64(32)) ^ c[1] ^ k0) & MASK, p1 & MASK, ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & MASK, p0 & MASK]
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & MASK, (k1 + np.uint64(0xBB67AE85)) & MASK
    return c


def philox_randn(n_lead: int, batch: int, row_len: int, seed: int, row0: int = 0) -> np.ndarray:
    """Philox4x32-10 (Salmon et al. 2011) + Box-Muller, as the device generator defines it: element e of batch row b under
    leading index l comes from counter (row0 + b [64 bit], l, e // 4), key = seed; words (0,1) -> elements 4q, 4q+1 and
    words (2,3) -> 4q+2, 4q+3 with u1 = ((w >> 8) + 1) 2^-24, u2 = (w' >> 8) 2^-24, z = sqrt(-2 ln u1) (cos, sin)(2 pi u2).
    No counterpart in the reference (it calls torch.randn)."""
    quads = (row_len + 3) // 4
    l, bb, q = np.meshgrid(np.arange(n_lead, dtype=np.uint64), np.arange(batch, dtype=np.uint64), np.arange(quads, dtype=np.uint64),
                           indexing="ij")
    grow = bb + np.uint64(row0)
    c = philox4x32_10([grow & np.uint64(0xFFFFFFFF), grow >> np.uint64(32), l, q], seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    out = np.empty((n_lead, batch, quads * 4), dtype=np.float64)
    for j in (0, 1):
        u1 = ((c[2 * j] >> np.uint64(8)).astype(np.float64) + 1.0) * 2.0 ** -24
        u2 = (c[2 * j + 1] >> np.uint64(8)).astype(np.float64) * 2.0 ** -24
        r = np.sqrt(-2.0 * np.log(u1))
        out[..., 2 * j::4] = r * np.cos(2.0 * np.pi * u2)
        out[..., 2 * j + 1::4] = r * np.sin(2.0 * np.pi * u2)
    return out[..., :row_len].astype(np.float32)


# ------------------------------------------------------------------ decoder twin (DiT path)
def twin_loss(denoise: Callable, b: Dict[str, np.ndarray], label, t: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """DiffusionWrapper.train_forward of the decoder twin (diffusion_decoder.py:346-365) with the draws supplied: the
    element-wise squared error of the noise estimate at x_t = q_sample(normalize(label), t, noise)."""
    x0 = normalize(label)
    ext = lambda key: torch.as_tensor(b[key], dtype=torch.float32)[t].reshape(-1, 1, 1)  # noqa: E731
    x = ext("sqrt_alphas_cumprod") * x0 + ext("sqrt_one_minus_alphas_cumprod") * noise
    return (denoise(x, t) - noise) ** 2



def twin_sample(denoise: Callable, b: Dict[str, np.ndarray], shape, noise: Optional[torch.Tensor],
                determin: bool = True, max_action: float = 100.0):
    """diffusion_decoder.py:385-446.  Returns (denormalised trajectory, total_cls).

    Stochastic mode draws a tensor at EVERY step (the t = 0 draw is masked out of x but
    still enters the score), so ``noise`` holds 1 + N tensors.
    """
    n = len(b["betas"])
    x = torch.zeros(shape) if determin else noise[0].clone()
    total_cls = -(x ** 2).mean(dim=1)
    k = 1
    for t in reversed(range(n)):
        out = denoise(x, t)
        x0 = (_buf(b, "sqrt_recip_alphas_cumprod", t) * x - _buf(b, "sqrt_recipm1_alphas_cumprod", t) * out)
        x0 = x0.clamp(-max_action, max_action)
        mean = _buf(b, "posterior_mean_coef1", t) * x0 + _buf(b, "posterior_mean_coef2", t) * x
        if determin:
            z = torch.zeros(shape)
        else:
            z = noise[k]
            k += 1
        mask = 0.0 if t == 0 else 1.0
        x = mean + mask * torch.exp(0.5 * _buf(b, "posterior_log_variance_clipped", t)) * z
        total_cls = total_cls - (z ** 2).mean(dim=1)
    return denormalize(x), total_cls


# ------------------------------------------------------------------ diffusers-style steps (PARITY UNPINNED)
def diffusers_timesteps(n_train: int, n_infer: int) -> List[int]:
    """'leading' spacing: t_k = k * (N // n), k = n-1 ... 0."""
    r = n_train // n_infer
    return [k * r for k in reversed(range(n_infer))]


def diffusers_ddpm_step(ac: np.ndarray, n_infer: int, eps, t: int, x, z, clip: bool = True, clip_range: float = 1.0):
    n = len(ac)
    t_prev = t - n // n_infer
    a_t = float(ac[t])
    a_prev = float(ac[t_prev]) if t_prev >= 0 else 1.0
    cur_alpha = a_t / a_prev
    cur_beta = 1 - cur_alpha
    x0 = (x - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
    if clip:
        x0 = x0.clamp(-clip_range, clip_range)
    mean = (a_prev ** 0.5 * cur_beta / (1 - a_t)) * x0 + (cur_alpha ** 0.5 * (1 - a_prev) / (1 - a_t)) * x
    if t > 0:
        var = max((1 - a_prev) / (1 - a_t) * cur_beta, 1e-20)
        mean = mean + var ** 0.5 * z
    return mean


def diffusers_ddim_step(ac: np.ndarray, n_infer: int, eps, t: int, x, clip: bool = True, clip_range: float = 1.0):
    """eta = 0, set_alpha_to_one=True, use_clipped_model_output=False."""
    n = len(ac)
    t_prev = t - n // n_infer
    a_t = float(ac[t])
    a_prev = float(ac[t_prev]) if t_prev >= 0 else 1.0
    x0 = (x - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
    if clip:
        x0 = x0.clamp(-clip_range, clip_range)
    return a_prev ** 0.5 * x0 + (1 - a_prev) ** 0.5 * eps


def explicit_generate(denoise_flat: Callable, ac: np.ndarray, n_infer: int, prior_full, init_x_flat, noise,
                      frame_stack: int, data_mean, data_std, kind: str = "ddim") -> torch.Tensor:
    """Flattened-chunk wrapper (ExplicitDiT.generate): chunks are (B, fs*4) vectors, horizon 1.

    ``denoise_flat(x(B,1,fs*4), t, transition(B,1,fs*4), prior(B,1,fs*4)) -> (B,1,fs*4)``;
    ``noise`` is (n_chunks, 1 + n_infer, B, 1, fs*4) (DDIM uses only element 0).
    The prior enters un-normalised; the data_mean/std affine is applied to the
    re-assembled output only (ExplicitDiT.py:151,193-194).
    """
    B, S, C = prior_full.shape
    n_chunks = S // frame_stack
    pri = prior_full.reshape(B, n_chunks, frame_stack * C).permute(1, 0, 2)
    transition = init_x_flat[None].expand(B, frame_stack * C)
    preds = []
    for c in range(n_chunks):
        x = noise[c, 0].clone()
        k = 1
        for t in diffusers_timesteps(len(ac), n_infer):
            eps = denoise_flat(x, t, transition[:, None, :], pri[c][:, None, :])
            if kind == "ddim":
                x = diffusers_ddim_step(ac, n_infer, eps, t, x)
            else:
                x = diffusers_ddpm_step(ac, n_infer, eps, t, x, noise[c, k])
                k += 1
        transition = x[:, 0]
        preds.append(transition)
    out = torch.stack(preds).permute(1, 0, 2).reshape(B, n_chunks * frame_stack, C)
    mean = torch.as_tensor(data_mean, dtype=torch.float32)
    std = torch.as_tensor(data_std, dtype=torch.float32)
    return out * std + mean


def transition_generate(denoise_z: Callable, z_emb: Callable, x_from_z: Callable, ac: np.ndarray, n_infer: int, prior_full, init_z,
                        noise, frame_stack: int, data_mean, data_std) -> torch.Tensor:
    """Latent-carrying wrapper (StrDiffDiT.generate, StrDiffDiT.py:202-235, with DiffusionTransitionModel.
    conditional_sample, diffusion_transition.py:43-78).  BEST EFFORT, parity unpinned: the reference's denoiser for
    this variant is missing from its tree; here z enters through ``z_emb`` into the transition slot.

    ``denoise_z(x(B,fs,4), t, transition(B,fs,Cp), prior(B,fs,4)) -> z (B,fs,z_dim)``; ``noise`` is
    (n_chunks, draws, B, fs, 4), one draw per step with t > 0.  The sample starts from the chunk's prior
    (diffusion_transition.py:58); the prior's first frame is zeroed (StrDiffDiT.py:182).
    """
    B, S, C = prior_full.shape
    n_chunks = S // frame_stack
    cond = prior_full.clone()
    cond[:, :1] = 0.0
    cond = cond.reshape(B, n_chunks, frame_stack, C)
    z = init_z[None].expand(B, *init_z.shape)
    preds = []
    for c in range(n_chunks):
        pri = cond[:, c]
        x = pri.clone()
        tran = z_emb(z)
        k = 0
        for t in diffusers_timesteps(len(ac), n_infer):
            z_out = denoise_z(x, t, tran, pri)
            eps = x_from_z(z_out)
            zz = None
            if t > 0:
                zz = noise[c, k]
                k += 1
            x = diffusers_ddpm_step(ac, n_infer, eps, t, x, zz)
        z = z_out
        preds.append(x)
    out = torch.stack(preds, dim=1).reshape(B, S, C)
    mean = torch.as_tensor(data_mean, dtype=torch.float32)
    std = torch.as_tensor(data_std, dtype=torch.float32)
    return out * std + mean


def transition_forward(denoise_z: Callable, z_emb: Callable, x_from_z: Callable, ac: np.ndarray, prior_full, init_z, t, noise,
                       frame_stack: int, data_mean, data_std):
    """Loss of the latent-carrying wrapper's forward() (StrDiffDiT.py:96-155 with DiffusionTransitionModel.forward,
    diffusion_transition.py:95-130; ``add_noise`` per the public DDPMScheduler semantics).  Parity unpinned, like
    :func:`transition_generate`.  The noised chunks are those of the NORMALISED PRIOR (StrDiffDiT.py:163: ``xs = batch[1]``);
    ``t`` (n_chunks, B) int64 in [1, N), ``noise`` (n_chunks, B, fs, 4).  Returns (loss, un-normalised predictions (B, S, 4))."""
    B, S, C = prior_full.shape
    n_chunks = S // frame_stack
    mean = torch.as_tensor(data_mean, dtype=torch.float32)
    std = torch.as_tensor(data_std, dtype=torch.float32)
    cond = prior_full.clone()
    cond[:, :1] = 0.0
    cond = cond.reshape(B, n_chunks, frame_stack, C)
    xs = ((prior_full - mean) / std).reshape(B, n_chunks, frame_stack, C)
    sa = torch.as_tensor(np.sqrt(ac), dtype=torch.float32)
    s1 = torch.as_tensor(np.sqrt(1.0 - ac), dtype=torch.float32)
    z = init_z[None].expand(B, *init_z.shape)
    preds, losses = [], []
    for c in range(n_chunks):
        noisy = sa[t[c]][:, None, None] * xs[:, c] + s1[t[c]][:, None, None] * noise[c]
        z = denoise_z(noisy, t[c], z_emb(z), cond[:, c])
        pred = x_from_z(z)
        preds.append(pred)
        losses.append((pred - noise[c]) ** 2)
    return torch.stack(losses).mean(), torch.stack(preds, dim=1).reshape(B, S, C) * std + mean


# ------------------------------------------------------------------ metrics
def displacement_metrics(pred: np.ndarray, label: np.ndarray) -> Dict[str, np.ndarray]:
    """ADE/FDE at 1/3/5/8 s per scenario (trainer.py:184-234): L2 over (x, y) per point."""
    err = np.sqrt((pred[..., 0] - label[..., 0]) ** 2 + (pred[..., 1] - label[..., 1]) ** 2)
    return {
        "ade1": err[:, :10].mean(1), "ade3": err[:, :30].mean(1), "ade5": err[:, :50].mean(1), "ade8": err[:, :80].mean(1),
        "fde1": err[:, 9], "fde3": err[:, 29], "fde5": err[:, 49], "fde8": err[:, -1],
    }


def evaluation_metrics(pred: np.ndarray, label: np.ndarray) -> Dict[str, float]:
    """Batch means of the evaluation entry (trainer.py:184-305): ADE/FDE horizons, heading errors, scores, miss rates."""
    S = label.shape[1]
    rs = int(80 / S) if S < 80 else 1
    err = np.sqrt((pred[..., 0] - label[..., 0]) ** 2 + (pred[..., 1] - label[..., 1]) ** 2)
    ade = {n: err[:, :(10 * n if n != 1 else 10) // rs].mean(1) for n in (1, 3, 5, 8)}
    fde = {1: err[:, 9 // rs], 3: err[:, 29 // rs], 5: err[:, 49 // rs], 8: err[:, -1]}
    r = {}
    for n in (1, 3, 5, 8):
        r[f"ade_horizon{n}_gen"] = ade[n].mean()
        r[f"fde_horizon{n}_gen"] = fde[n].mean()
    avg_ade, avg_fde = (ade[3] + ade[5] + ade[8]) / 3, (fde[3] + fde[5] + fde[8]) / 3
    r["metric_ade"], r["metric_fde"] = avg_ade.mean(), avg_fde.mean()
    r["ade_score"] = np.maximum(1 - avg_ade / 8, 0).mean()
    r["fde_score"] = np.maximum(1 - avg_fde / 8, 0).mean()
    if pred.shape[-1] == 4:
        wrap = lambda a: np.arctan2(np.sin(a), np.cos(a))
        he = np.abs(wrap(wrap(pred[..., -1]) - wrap(label[..., -1])))
        ahe = {n: he[:, :(10 * n) // rs].mean(1) for n in (3, 5, 8)}
        fhe = {3: he[:, 29 // rs], 5: he[:, 49 // rs], 8: he[:, 79 // rs]}
        r["ahe_1_gen"] = he[:, 0].mean()
        for n in (3, 5, 8):
            r[f"ahe_horizon{n}_gen"] = ahe[n].mean()
            r[f"fhe_horizon{n}_gen"] = fhe[n].mean()
        avg_ahe, avg_fhe = (ahe[3] + ahe[5] + ahe[8]) / 3, (fhe[3] + fhe[5] + fhe[8]) / 3
        r["metric_ahe"], r["metric_fhe"] = avg_ahe.mean(), avg_fhe.mean()
        r["ahe_score"] = np.maximum(1 - avg_ahe / 0.8, 0).mean()
        r["fhe_score"] = np.maximum(1 - avg_fhe / 0.8, 0).mean()
    miss = [(err[:, :(10 * n) // rs].max(1) > th).astype(np.float64) for n, th in ((3, 6), (5, 8), (8, 16))]
    r["miss_rate3"], r["miss_rate5"], r["miss_rate8"] = (m.mean() for m in miss)
    r["total_miss_rate"] = ((miss[0] + miss[1] + miss[2]) >= 1).astype(np.float64).mean()
    return {k: float(v) for k, v in r.items()}


def evaluation_scores(pred: np.ndarray, label: np.ndarray, scenario15s_id) -> tuple:
    """``compute_scores`` (utils/nuplan_utils.py:229-314) on the per-frame items ``compute_metrics`` saves (trainer.py:196-301):
    frames grouped by scenario id in order of first appearance; per group the mean over frames of (h3 + h5 + h8), divided by 3 and
    by the threshold, subtracted from 1 and clamped at 0, for ADE / FDE (8 m) and AHE / FHE (0.8 rad); ``miss_score`` 0 when the
    group has 5 or more missed frames; ``score = miss_score * (ade + fde + 2 ahe + 2 fhe) / 6``.  Plain loops, as the reference.
    Returns (average score, average miss_score).  Parity unpinned: the reference module needs ``shapely`` to import."""
    S = label.shape[1]
    rs = int(80 / S) if S < 80 else 1
    err = np.sqrt((pred[..., 0] - label[..., 0]) ** 2 + (pred[..., 1] - label[..., 1]) ** 2)
    wrap = lambda a: np.arctan2(np.sin(a), np.cos(a))  # noqa: E731
    he = np.abs(wrap(wrap(pred[..., -1]) - wrap(label[..., -1])))
    groups: dict = {}
    for i, sid in enumerate(scenario15s_id):
        groups.setdefault(int(sid), []).append(i)
    scores, miss_scores = [], []
    for sid, frames in groups.items():
        sums = {"ade": [], "fde": [], "ahe": [], "fhe": []}
        n_miss = 0
        for i in frames:
            sums["ade"].append(sum(err[i, :(10 * n) // rs].mean() for n in (3, 5, 8)))
            sums["fde"].append(err[i, 29 // rs] + err[i, 49 // rs] + err[i, -1])
            sums["ahe"].append(sum(he[i, :(10 * n) // rs].mean() for n in (3, 5, 8)))
            sums["fhe"].append(he[i, 29 // rs] + he[i, 49 // rs] + he[i, 79 // rs])
            n_miss += int(any(err[i, :(10 * n) // rs].max() > th for n, th in ((3, 6), (5, 8), (8, 16))))
        sc = {k: max(1 - float(np.mean(v)) / 3 / (8.0 if k in ("ade", "fde") else 0.8), 0) for k, v in sums.items()}
        miss_score = 0 if n_miss >= 5 else 1
        scores.append(miss_score * (sc["ade"] + sc["fde"] + 2 * sc["ahe"] + 2 * sc["fhe"]) / 6)
        miss_scores.append(miss_score)
    return float(np.average(scores)), float(np.average(miss_scores))
