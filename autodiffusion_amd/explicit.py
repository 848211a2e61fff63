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

    def _staging(self, B: int, n_noise: int, device):
        """Persistent per-(batch, device) buffers for the per-chunk tensors: the engine keys its captured hipGraph on the
        pointers it is given, and the chunk loop would otherwise hand it fresh tensors (a new capture) for every chunk."""
        key = (B, n_noise, str(device))
        if getattr(self, "_stage_key", None) != key:
            W = self.cfg.trajectory_dim * self.cfg.frame_stack
            f = dict(device=device, dtype=torch.float32)
            self._stage = (torch.empty(B, 1, W, **f), torch.empty(B, 1, W, **f), torch.empty(n_noise, B, 1, W, **f), torch.empty(B, 1, W, **f))
            self._stage_key = key
        return self._stage

    @torch.no_grad()
    def generate(self, label, transition_info, trajectory_prior, maps_info, noise: Optional[torch.Tensor] = None,
                 same_maps: bool = False):
        """label (B, fs*4) [shape only], transition_info (B, fs*4) or (B, 1, fs*4), trajectory_prior (B, fs*4),
        maps_info (B, 788, maps_dim) -> (B, fs*4).  ``noise``: (1 + draws, B, 1, fs*4), default fresh N(0, 1).
        ``same_maps``: maps_info is the tensor of the previous call (the next chunk of the same trajectories), so its
        embedding and attention caches are kept."""
        B = label.shape[0]
        if transition_info.dim() == 2:
            transition_info = transition_info.unsqueeze(1)
        prior = trajectory_prior.unsqueeze(1) if trajectory_prior.dim() == 2 else trajectory_prior
        eng = self.diffusion.engine()
        eng.set_schedule(self.schedule_rows())
        if noise is None:
            noise = self.make_noise(B, prior.device)
        maps = _lib.f32c(maps_info) if self.diffusion.map_cond else None
        tb, pb, nb, ob = self._staging(B, noise.shape[0], prior.device)
        tb.copy_(transition_info.reshape(tb.shape)); pb.copy_(prior.reshape(pb.shape)); nb.copy_(noise.reshape(nb.shape))
        eng.sample_a(maps, tb, pb, nb, _lib.FLAG_KEEP_MAPS if same_maps else 0, out=ob)
        return ob.squeeze(1).clone()

    def forward(self, label, transition_info, trajectory_prior, maps_info):
        raise NotImplementedError("training forward (diffusion loss) is outside the accelerated sampling path")


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

    @torch.no_grad()
    def generate(self, noise: Optional[torch.Tensor] = None, **kwargs):
        """-> {"traj_logits": (B, 80, 4) float32}.  ``noise``: (n_chunks, 1 + draws, B, 1, fs*4)."""
        if "trajectory_prior" in kwargs and "maps_info" in kwargs:
            trajectory_prior, maps_info = kwargs["trajectory_prior"], kwargs["maps_info"]
        else:
            if self.str_model[0] is None:
                raise ValueError("generate() needs either trajectory_prior/maps_info or a str_model for phase 1")
            str_result = self.str_model[0].generate(**kwargs)
            trajectory_prior, maps_info = str_result["traj_logits"], str_result["maps_info"]
        B, S, C = trajectory_prior.shape
        if S % self.frame_stack != 0:
            raise ValueError("Number of frames must be divisible by frame stack size")
        n_chunks = S // self.frame_stack
        width = self.frame_stack * C
        prior = _lib.f32c(trajectory_prior).reshape(B, n_chunks, width)
        if noise is None:
            noise = self.diffusion.make_noise(B, prior.device, n_chunks)
        transition = self.init_x[None].expand(B, width).contiguous()
        label = torch.empty(B, width, device=prior.device)          # shape only
        preds = []
        for c in range(n_chunks):
            transition = self.diffusion.generate(label, transition, prior[:, c].contiguous(), maps_info, noise=noise[c], same_maps=c > 0)
            preds.append(transition)
        out = torch.stack(preds, dim=1).reshape(B, S, C)
        return {"traj_logits": self._unnormalize_x(out).to(torch.float32)}

    def forward(self, return_dict=None, **kwargs):
        raise NotImplementedError("training forward (diffusion loss) is outside the accelerated sampling path; use generate()")


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
