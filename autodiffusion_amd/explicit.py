"""Flattened-chunk refiner: ``TrajDiffusion`` + the ``ExplicitDiT`` wrapper.

Mirrors transformer4planning/models/diffusion_loss/traj_diffusion.py:13-81 (``TrajDiffusion``) and the chunk loop
of transformer4planning/models/backbone/ExplicitDiT.py:139-210: every chunk of ``frame_stack`` trajectory points is
ONE token of ``frame_stack * 4`` channels (horizon 1), denoised by the encoder/decoder Transformer with a
diffusers-style DDIM schedule (``DDPM`` is available too); the previous chunk's *normalised* prediction is the next
chunk's transition token, the prior enters un-normalised, and the ``data_mean / data_std`` affine is applied to the
re-assembled trajectory only.

The reference imports its denoiser class for this variant (``DiffusionForTraj``) from a module that does not define
it, and its scheduler from the third-party ``diffusers`` package (no pinned version); the denoiser here is the
reference's ``Transformer`` instantiated with the constructor arguments ``TrajDiffusion._build_model`` passes
(its ``# 791`` comment at transformer.py:243 is exactly this instantiation's conditioning length), and the scheduler
steps follow the public DDIMScheduler / DDPMScheduler semantics (``schedules.diffusers_rows``).
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional, Sequence

import torch
from torch import nn

from . import _lib, schedules
from .refiner import Transformer


class TrajDiffusion(nn.Module):
    """cfg needs: trajectory_dim, frame_stack, n_maps_token, maps_dim, map_cond and ``scheduler`` with
    num_train_timesteps, beta_start, beta_end, beta_schedule, clip_sample, prediction_type, num_inference_steps
    (optional: ``kind`` "ddim" | "ddpm", precision, elide)."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        width = cfg.trajectory_dim * cfg.frame_stack
        self.diffusion = Transformer(
            input_dim=width, output_dim=width, horizon=1, n_cond_steps=cfg.n_maps_token, cond_dim=cfg.maps_dim,
            n_prior_steps=1, prior_dim=width, causal_attn=True, map_cond=cfg.map_cond,
            precision=getattr(cfg, "precision", "fp32"), elide=getattr(cfg, "elide", 3))
        sc = cfg.scheduler
        if sc.prediction_type != "epsilon":
            raise ValueError(f"Unsupported prediction type {sc.prediction_type}")
        self.kind = getattr(cfg, "kind", "ddim")
        self.register_buffer("betas", schedules.diffusers_betas(sc.beta_schedule, sc.num_train_timesteps, sc.beta_start,
                                                                sc.beta_end).to(torch.float32), persistent=False)

    def schedule_rows(self):
        sc = self.cfg.scheduler
        betas = schedules.diffusers_betas(sc.beta_schedule, sc.num_train_timesteps, sc.beta_start, sc.beta_end)
        return schedules.diffusers_rows(betas, sc.num_inference_steps, self.kind, clip_sample=bool(sc.clip_sample))

    def make_noise(self, batch: int, device, n_chunks: Optional[int] = None):
        n = 1 + schedules.n_noise_draws(self.schedule_rows())
        lead = (n,) if n_chunks is None else (n_chunks, n)
        return torch.randn(*lead, batch, 1, self.cfg.trajectory_dim * self.cfg.frame_stack, device=device)

    @torch.no_grad()
    def generate(self, label, transition_info, trajectory_prior, maps_info, noise: Optional[torch.Tensor] = None,
                 same_maps: bool = False, seed: Optional[int] = None, row0: int = 0):
        """label (B, fs*4) [shape only], transition_info (B, fs*4) or (B, 1, fs*4), trajectory_prior (B, fs*4),
        maps_info (B, 788, maps_dim) -> (B, fs*4).  ``noise``: (1 + draws, B, 1, fs*4); None draws it on the device
        (``seed``, ``row0``).  ``same_maps``: maps_info is the tensor of the previous call (the next chunk of the same
        trajectories), so its embedding and attention caches are kept.  (Per-chunk tensors are fresh on every call; the
        engine serves such callers from its own staging buffers, one captured graph for all of them.)"""
        B = label.shape[0]
        W = self.cfg.trajectory_dim * self.cfg.frame_stack
        eng = self.diffusion.engine()
        eng.set_schedule(self.schedule_rows())
        if noise is None:
            eng.seed_from_torch(seed, row0)
        maps = _lib.f32c(maps_info) if self.diffusion.map_cond else None
        out = eng.sample_a(maps, _lib.f32c(transition_info).reshape(B, 1, W), _lib.f32c(trajectory_prior).reshape(B, 1, W),
                           _lib.f32c(noise).reshape(-1, B, 1, W) if noise is not None else None,
                           _lib.FLAG_KEEP_MAPS if same_maps else 0)
        return out.squeeze(1)

    def forward(self, label, transition_info, trajectory_prior, maps_info, t: Optional[torch.Tensor] = None,
                noise: Optional[torch.Tensor] = None):
        """-> (loss (B, fs*4) element-wise squared error of the noise estimate, pred (B, fs*4)): traj_diffusion.py:83-124.
        The timestep is uniform in [1, num_train_timesteps) as there; ``add_noise`` is the public
        ``sqrt(acp[t]) x + sqrt(1 - acp[t]) noise`` (third-party scheduler: parity unpinned).  With gradients enabled and
        trainable parameters the denoiser call is the differentiable training path (``train.py``), so the loss and ``pred`` --
        which the wrapper feeds to the next chunk as its transition, un-detached as in the reference -- carry gradients;
        under ``torch.no_grad()`` only the values are computed, on the sampling engine."""
        from . import train
        if train.wants_grad(self.diffusion):
            return self._forward_impl(label, transition_info, trajectory_prior, maps_info, t, noise, True)
        with torch.no_grad():
            return self._forward_impl(label, transition_info, trajectory_prior, maps_info, t, noise, False)

    def _forward_impl(self, label, transition_info, trajectory_prior, maps_info, t, noise, grad: bool):
        from . import ops, train
        B = label.shape[0]
        W = self.cfg.trajectory_dim * self.cfg.frame_stack
        x0 = _lib.f32c(label).reshape(B, 1, W)
        sc = self.cfg.scheduler
        if t is None:
            t = torch.randint(1, sc.num_train_timesteps, (B,), device=x0.device)
        t = t.to(device=x0.device, dtype=torch.int64).contiguous()
        noise = torch.randn_like(x0) if noise is None else _lib.f32c(noise).reshape(B, 1, W)
        acp = torch.cumprod(1.0 - self.betas.to(torch.float32), dim=0)
        x, _ = ops.q_sample(x0, noise, t, acp.sqrt(), (1.0 - acp).sqrt())
        if grad:       # the transition keeps its autograd history (init_x, or the previous chunk's prediction)
            cond = dict(maps_info=maps_info, transition_info=transition_info.to(torch.float32).reshape(B, 1, W),
                        trajectory_prior=_lib.f32c(trajectory_prior).reshape(B, 1, W))
            pred = train.denoise_with_grad(self.diffusion, x, t, cond, self.diffusion._n_head).reshape(B, W)
        else:
            cond = dict(maps_info=maps_info, transition_info=_lib.f32c(transition_info).reshape(B, 1, W),
                        trajectory_prior=_lib.f32c(trajectory_prior).reshape(B, 1, W))
            pred = self.diffusion(x, t, cond).reshape(B, W)
        return (pred - noise.reshape(B, W)) ** 2, pred


class ExplicitDiT(nn.Module):
    """The refiner half of the reference's ``ExplicitDiT`` wrapper (phase 1, the STR backbone, is passed in as
    ``str_model`` or bypassed with ``generate(trajectory_prior=..., maps_info=...)``)."""

    def __init__(self, config, str_model=None):
        super().__init__()
        config.trajectory_dim = 4
        config.maps_dim = getattr(config, "maps_dim", 256)
        config.n_maps_token = getattr(config, "n_maps_token", 788)
        self.cfg = self.config = config
        self.trajectory_dim = 4
        self.frame_stack = config.frame_stack
        self.map_cond = config.map_cond
        self.str_model = [str_model]
        self.diffusion = TrajDiffusion(config)
        self.init_x = nn.Parameter(torch.randn(self.frame_stack * self.trajectory_dim))
        self.register_data_mean_std(config.data_mean, config.data_std)

    def register_data_mean_std(self, mean: Sequence[float], std: Sequence[float], namespace: str = "data"):
        self.register_buffer(f"{namespace}_mean", torch.as_tensor(mean, dtype=torch.float32))
        self.register_buffer(f"{namespace}_std", torch.as_tensor(std, dtype=torch.float32))

    def _unnormalize_x(self, xs):
        return xs * self.data_std.to(xs.device) + self.data_mean.to(xs.device)

    def _normalize_x(self, xs):
        return (xs - self.data_mean.to(xs.device)) / self.data_std.to(xs.device)

    def _phase1(self, kwargs):
        if "trajectory_prior" in kwargs and "maps_info" in kwargs:
            return kwargs["trajectory_prior"], kwargs["maps_info"]
        if self.str_model[0] is None:
            raise ValueError("needs either trajectory_prior/maps_info or a str_model for phase 1")
        str_result = self.str_model[0].generate(**kwargs)
        return str_result["traj_logits"], str_result["maps_info"]

    @torch.no_grad()
    def generate(self, noise: Optional[torch.Tensor] = None, seed: Optional[int] = None, row0: int = 0, **kwargs):
        """-> {"traj_logits": (B, 80, 4) float32}.  ``noise``: (n_chunks, 1 + draws, B, 1, fs*4); None draws it on the device."""
        trajectory_prior, maps_info = self._phase1(kwargs)
        B, S, C = trajectory_prior.shape
        if S % self.frame_stack != 0:
            raise ValueError("Number of frames must be divisible by frame stack size")
        n_chunks = S // self.frame_stack
        width = self.frame_stack * C
        prior = _lib.f32c(trajectory_prior).reshape(B, n_chunks, width)
        if noise is None and seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
        transition = self.init_x[None].expand(B, width).contiguous()
        label = torch.empty(B, width, device=prior.device)          # shape only
        preds = []
        for c in range(n_chunks):
            # device noise: every chunk takes its own stream of the generator (seed + chunk index)
            transition = self.diffusion.generate(label, transition, prior[:, c].contiguous(), maps_info,
                                                 noise=None if noise is None else noise[c], same_maps=c > 0,
                                                 seed=None if noise is not None else seed + c, row0=row0)
            preds.append(transition)
        out = torch.stack(preds, dim=1).reshape(B, S, C)
        return {"traj_logits": self._unnormalize_x(out).to(torch.float32)}

    def forward(self, return_dict=None, t: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None, **kwargs):
        """The training objective (ExplicitDiT.py:80-137): per chunk the squared error of the noise estimate for the
        (data_mean/std-normalised) label chunk, the previous chunk's model output as transition.  ``t`` (n_chunks, B),
        ``noise`` (n_chunks, B, fs*4) fix the draws.  Differentiable when gradients are enabled (see TrajDiffusion.forward);
        the value only under ``torch.no_grad()``."""
        from . import train
        if train.wants_grad(self.diffusion.diffusion):
            return self._forward_impl(t, noise, kwargs)
        with torch.no_grad():
            return self._forward_impl(t, noise, kwargs)

    def _forward_impl(self, t, noise, kwargs):
        from .refiner import LTMOutput
        trajectory_prior, maps_info = self._phase1(kwargs)
        trajectory_label = kwargs.get("trajectory_label", None)
        if trajectory_label is None:
            raise ValueError("forward() needs trajectory_label")
        B, S, C = trajectory_prior.shape
        if S % self.frame_stack != 0:
            raise ValueError("Number of frames must be divisible by frame stack size")
        n_chunks = S // self.frame_stack
        width = self.frame_stack * C
        prior = _lib.f32c(trajectory_prior).reshape(B, n_chunks, width)
        label = self._normalize_x(_lib.f32c(trajectory_label.to(prior.device))).reshape(B, n_chunks, width)
        transition = self.init_x[None].expand(B, width).contiguous()
        losses, preds = [], []
        for c in range(n_chunks):
            l, transition = self.diffusion(label[:, c].contiguous(), transition, prior[:, c].contiguous(), maps_info,
                                           t=None if t is None else t[c], noise=None if noise is None else noise[c])
            losses.append(l)
            preds.append(transition)
        final_predict = self._unnormalize_x(torch.stack(preds, dim=1).reshape(B, S, C))
        loss = torch.stack(losses)
        return LTMOutput(loss=loss.mean(), loss_items=loss, logits=final_predict, pred_dict={"traj_logits": final_predict},
                         trajectory_label=trajectory_label)


def explicit_config(frame_stack: int = 10, num_train_timesteps: int = 100, num_inference_steps: int = 10,
                    beta_schedule: str = "squaredcos_cap_v2", map_cond: bool = True, kind: str = "ddim",
                    data_mean=(30.0, 10.0, 0.0, 0.0), data_std=(100.0, 20.0, 1.0, 1.0), precision: str = "fp32",
                    elide: int = 3) -> SimpleNamespace:
    """Config with the fields of config/explicitDiT*.yaml that the sampling path reads."""
    scheduler = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=0.0001, beta_end=0.02,
                                beta_schedule=beta_schedule, clip_sample=True, prediction_type="epsilon",
                                num_inference_steps=num_inference_steps)
    return SimpleNamespace(frame_stack=frame_stack, map_cond=map_cond, scheduler=scheduler, kind=kind, data_mean=list(data_mean),
                           data_std=list(data_std), precision=precision, elide=elide, maps_dim=256, n_maps_token=788,
                           trajectory_dim=4)
