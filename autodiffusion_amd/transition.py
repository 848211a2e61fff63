"""Latent-carrying refiner: ``DiffusionTransitionModel`` + the ``StrDiffDiT`` wrapper.  BEST EFFORT, parity unpinned.

Mirrors transformer4planning/models/diffusion_loss/diffusion_transition.py:11-93 and the chunk loop of
transformer4planning/models/backbone/StrDiffDiT.py:157-235: per chunk of ``frame_stack`` points the sample starts
from the chunk's prior (not from noise), a diffusers-style DDPM loop calls the denoiser, whose output is a latent
``z (B, fs, z_dim)`` that ``x_from_z`` projects to the 4-channel model output; the latent of the last step is carried
to the next chunk; the prior enters with its first frame zeroed and un-normalised; ``data_mean / data_std`` is applied
to the re-assembled trajectory only.

The reference imports the denoiser of this variant (``TransformerForDiffusion``, called as
``model(x, t, z, cond, prior)``) from a module that does not define it, so how ``z`` enters the network cannot be
recovered.  Here ``z`` takes the transition slot of the encoder/decoder ``Transformer`` (whose embedding of that slot
is named ``prior_z_emb``) through one extra Linear ``z_emb (z_dim -> prior_dim)``; everything else is the reference's
``Transformer`` with ``output_dim = z_dim``.  The scheduler follows the public DDPMScheduler semantics
(``schedules.diffusers_rows``).  ``StrDiffDiT.generate`` runs all chunks x steps as ONE captured graph inside the engine
(``adx_generate_transition``: fused encoder-row pipeline per step, the decoder's operators, x_from_z, the scheduler step, the
latent carried on the device); ``generate_eager`` / ``DiffusionTransitionModel.generate`` drive the same operators from the host.  ``forward()`` of both
classes returns the evaluation-time VALUE of the training loss (StrDiffDiT.py:96-155, diffusion_transition.py:95-130) without
an autograd graph.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional, Sequence

import torch
from torch import nn

from . import _lib, ops, schedules
from .refiner import Transformer, _Lin


class DiffusionTransitionModel(nn.Module):
    """cfg needs: x_shape (4,), z_shape (fs, z_dim), frame_stack, n_maps_token, cond_dim, prior_dim and ``scheduler``
    (num_train_timesteps, beta_start, beta_end, beta_schedule, clip_sample, prediction_type, num_inference_steps)."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        z_dim = cfg.z_shape[-1]
        self.diffusion = Transformer(
            input_dim=cfg.x_shape[-1], output_dim=z_dim, horizon=cfg.frame_stack, n_cond_steps=cfg.n_maps_token,
            cond_dim=cfg.cond_dim, n_prior_steps=cfg.frame_stack, prior_dim=cfg.prior_dim, causal_attn=True,
            precision=getattr(cfg, "precision", "fp32"), elide=int(getattr(cfg, "elide", 3)))
        self.z_emb = _Lin(z_dim, cfg.prior_dim)
        self.x_from_z = _Lin(cfg.z_shape[1], cfg.x_shape[0])
        if cfg.scheduler.prediction_type != "epsilon":
            raise ValueError(f"Unsupported prediction type {cfg.scheduler.prediction_type}")

    def schedule_rows(self):
        sc = self.cfg.scheduler
        betas = schedules.diffusers_betas(sc.beta_schedule, sc.num_train_timesteps, sc.beta_start, sc.beta_end)
        return schedules.diffusers_rows(betas, sc.num_inference_steps, "ddpm", clip_sample=bool(sc.clip_sample))

    def make_noise(self, batch: int, device, n_chunks: Optional[int] = None):
        n = schedules.n_noise_draws(self.schedule_rows())
        lead = (n,) if n_chunks is None else (n_chunks, n)
        return torch.randn(*lead, batch, self.cfg.frame_stack, self.cfg.x_shape[-1], device=device)

    @torch.no_grad()
    def generate(self, xs, z, cond, prior, noise: Optional[torch.Tensor] = None, seed: Optional[int] = None, row0: int = 0):
        """xs (B, fs, 4) [shape only], z (B, fs, z_dim), cond = maps_info (B, 788, cond_dim), prior (B, fs, 4)
        -> (z_next (B, fs, z_dim), x_next (B, fs, 4)).  ``noise``: (draws, B, fs, 4); None draws it on the device with the
        library's counter-based generator (``seed``, ``row0`` = global index of batch row 0)."""
        rows = self.schedule_rows()
        B = prior.shape[0]
        prior = _lib.f32c(prior)
        if noise is None:
            if seed is None:
                seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
            fs, C = self.cfg.frame_stack, self.cfg.x_shape[-1]
            noise = ops.randn(max(schedules.n_noise_draws(rows), 1), B, fs * C, seed, row0, device=prior.device).reshape(-1, B, fs, C)
        transition = ops.linear(_lib.f32c(z), self.z_emb.weight, self.z_emb.bias)
        condition = dict(maps_info=_lib.f32c(cond), transition_info=transition, trajectory_prior=prior)
        trajectory = prior.clone()
        z_output = None
        for row in rows:
            t = torch.full((B,), int(row[0]), dtype=torch.long, device=prior.device)
            z_output = self.diffusion(trajectory, t, condition)
            model_output = ops.linear(z_output, self.x_from_z.weight, self.x_from_z.bias)
            zi = int(row[11])
            trajectory = ops.sched_step(trajectory, model_output, noise[zi] if zi >= 0 else None, row)
        return z_output, trajectory

    def train_tables(self):
        """sqrt(alphas_cumprod), sqrt(1 - alphas_cumprod) of the training schedule (what DDPMScheduler.add_noise reads)."""
        sc = self.cfg.scheduler
        ac = torch.cumprod(1.0 - schedules.diffusers_betas(sc.beta_schedule, sc.num_train_timesteps, sc.beta_start, sc.beta_end), dim=0)
        return ac.sqrt(), (1.0 - ac).sqrt()

    @torch.no_grad()
    def forward(self, z, x, cond, prior, t: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None):
        """One training-style evaluation of a chunk (diffusion_transition.py:95-130), VALUE ONLY: noise the chunk ``x (B, fs, 4)``
        at one random timestep per row in [1, num_train_timesteps), one denoiser call with the latent ``z``, ``x_from_z`` ->
        (pred_z (B, fs, z_dim), pred_x (B, fs, 4), target = the noise).  ``t`` / ``noise`` supply the two draws.  Runs without
        autograd: the results carry no graph, so ``backward()`` on a loss built from them raises -- the training of this
        variant is not built (its denoiser class is missing from the reference tree)."""
        x = _lib.f32c(x)
        B = x.shape[0]
        n_train = int(self.cfg.scheduler.num_train_timesteps)
        if noise is None:
            noise = torch.randn(x.shape, device=x.device)
        if t is None:
            t = torch.randint(1, n_train, (B,), device=x.device)
        t = t.to(device=x.device, dtype=torch.int64)
        sa, s1 = self.train_tables()
        noisy_x, _ = ops.q_sample(x, _lib.f32c(noise), t, sa, s1)
        transition = ops.linear(_lib.f32c(z), self.z_emb.weight, self.z_emb.bias)
        condition = dict(maps_info=_lib.f32c(cond), transition_info=transition, trajectory_prior=_lib.f32c(prior))
        pred_z = self.diffusion(noisy_x, t, condition)
        pred_x = ops.linear(pred_z, self.x_from_z.weight, self.x_from_z.bias)
        if self.cfg.scheduler.prediction_type != "epsilon":
            raise ValueError(f"Unsupported prediction type {self.cfg.scheduler.prediction_type}")
        return pred_z, pred_x, noise


class StrDiffDiT(nn.Module):
    """The refiner half of the reference's ``StrDiffDiT`` wrapper (phase 1, the STR backbone, is passed in as
    ``str_model`` or bypassed with ``generate(trajectory_prior=..., maps_info=...)``)."""

    def __init__(self, cfg, str_model=None):
        super().__init__()
        self.cfg = cfg
        self.frame_stack = cfg.frame_stack
        self.x_shape = tuple(cfg.x_shape)
        self.z_shape = tuple(cfg.z_shape)
        self.map_cond = cfg.map_cond
        self.learnable_init_z = cfg.learnable_init_z
        self.str_model = [str_model]
        self.transition_model = DiffusionTransitionModel(cfg)
        self.register_data_mean_std(cfg.data_mean, cfg.data_std)
        if self.learnable_init_z:
            self.init_z = nn.Parameter(torch.randn(list(cfg.z_shape)))

    def register_data_mean_std(self, mean: Sequence[float], std: Sequence[float], namespace: str = "data"):
        self.register_buffer(f"{namespace}_mean", torch.as_tensor(mean, dtype=torch.float32))
        self.register_buffer(f"{namespace}_std", torch.as_tensor(std, dtype=torch.float32))

    def _unnormalize_x(self, xs):
        return xs * self.data_std.to(xs.device) + self.data_mean.to(xs.device)

    @torch.no_grad()
    def generate(self, noise: Optional[torch.Tensor] = None, seed: Optional[int] = None, row0: int = 0, **kwargs):
        """-> {"traj_logits": (B, 80, 4) float32}.  ``noise``: (n_chunks, draws, B, fs, 4); None draws it on the device."""
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
        if getattr(self.cfg, "context_frames", 0) % self.frame_stack != 0:
            raise ValueError("Number of context frames must be divisible by frame stack size")
        n_chunks = S // self.frame_stack
        tm = self.transition_model
        rows = tm.schedule_rows()
        draws = schedules.n_noise_draws(rows)
        fs = self.frame_stack
        if noise is None and draws > 0:               # the library's counter-based generator: chunk c draws with seed + c
            if seed is None:
                seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
            noise = torch.stack([ops.randn(draws, B, fs * C, seed + c, row0, device=trajectory_prior.device) for c in range(n_chunks)])
        eng = tm.diffusion.engine()                   # the denoiser's engine (level-3 elision: fused encoder rows, unfused decoder)
        eng.set_extra([("z_emb.weight", tm.z_emb.weight), ("z_emb.bias", tm.z_emb.bias), ("x_from_z.weight", tm.x_from_z.weight),
                       ("x_from_z.bias", tm.x_from_z.bias)])
        eng.set_schedule(rows)
        if self.learnable_init_z:
            init_z = _lib.f32c(self.init_z)
        else:
            init_z = torch.zeros(*self.z_shape, device=trajectory_prior.device)
        stamp = (self.data_mean.data_ptr(), self.data_mean._version, self.data_std.data_ptr(), self.data_std._version, str(trajectory_prior.device))
        if self.__dict__.get("_ms_stamp") != stamp:   # {mean[4], std[4]} on the device, rebuilt only when a buffer changed
            self.__dict__["_ms"] = torch.cat([self.data_mean.float().reshape(4), self.data_std.float().reshape(4)]).to(trajectory_prior.device).contiguous()
            self.__dict__["_ms_stamp"] = stamp
        out = eng.generate_transition(_lib.f32c(maps_info), _lib.f32c(trajectory_prior), init_z,
                                      None if noise is None else _lib.f32c(noise).reshape(n_chunks, -1, B, fs * C), self.__dict__["_ms"], fs)
        return {"traj_logits": out}

    @torch.no_grad()
    def generate_eager(self, noise: Optional[torch.Tensor] = None, seed: Optional[int] = None, row0: int = 0, **kwargs):
        """The same loop driven from the host, one library call per operator (``DiffusionTransitionModel.generate`` per chunk):
        the form the captured loop is checked against."""
        trajectory_prior, maps_info = kwargs["trajectory_prior"], kwargs["maps_info"]
        B, S, C = trajectory_prior.shape
        n_chunks = S // self.frame_stack
        conditions = _lib.f32c(trajectory_prior).clone()
        conditions[:, :1] = 0.0                                       # StrDiffDiT.py:182
        conditions = conditions.reshape(B, n_chunks, self.frame_stack, C)
        if noise is None and seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
        if self.learnable_init_z:
            z = self.init_z[None].expand(B, *self.z_shape).contiguous()
        else:
            z = torch.zeros(B, *self.z_shape, device=conditions.device)
        xs = torch.empty(B, self.frame_stack, C, device=conditions.device)      # shape only
        preds = []
        for c in range(n_chunks):
            z, x_next = self.transition_model.generate(xs, z, maps_info, conditions[:, c].contiguous(),
                                                       noise=None if noise is None else noise[c],
                                                       seed=None if noise is not None else seed + c, row0=row0)
            preds.append(x_next)
        out = torch.stack(preds, dim=1).reshape(B, S, C)
        return {"traj_logits": self._unnormalize_x(out).to(torch.float32)}

    def _normalize_x(self, xs):
        return (xs - self.data_mean.to(xs.device)) / self.data_std.to(xs.device)

    @torch.no_grad()
    def forward(self, return_dict=None, t: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None, **kwargs):
        """The loss the evaluation entry reports for this wrapper (StrDiffDiT.py:96-155), VALUE ONLY (no autograd graph; see
        DiffusionTransitionModel.forward).  As the reference does, the chunks that are noised are those of the normalised
        trajectory PRIOR (its ``_preprocess_batch`` takes ``xs = batch[1]``, the prior: the label never enters the loss), the
        conditioning prior has its first frame zeroed and stays un-normalised, the latent returned by a chunk conditions the
        next, the loss is the plain mean of the element-wise squared error, and ``logits`` is the un-normalised ``x_from_z``
        output re-assembled to (B, 80, 4).  ``t`` (n_chunks, B) and ``noise`` (n_chunks, B, fs, 4) supply the draws."""
        from .refiner import LTMOutput
        if "trajectory_prior" in kwargs and "maps_info" in kwargs:
            trajectory_prior, maps_info = kwargs["trajectory_prior"], kwargs["maps_info"]
        else:
            if self.str_model[0] is None:
                raise ValueError("forward() needs either trajectory_prior/maps_info or a str_model for phase 1")
            str_result = self.str_model[0].generate(**kwargs)
            trajectory_prior, maps_info = str_result["traj_logits"], str_result["maps_info"]
        B, S, C = trajectory_prior.shape
        if S % self.frame_stack != 0:
            raise ValueError("Number of frames must be divisible by frame stack size")
        if getattr(self.cfg, "context_frames", 0) % self.frame_stack != 0:
            raise ValueError("Number of context frames must be divisible by frame stack size")
        n_chunks, fs = S // self.frame_stack, self.frame_stack
        prior = _lib.f32c(trajectory_prior)
        conditions = prior.clone()
        conditions[:, :1] = 0.0
        conditions = conditions.reshape(B, n_chunks, fs, C)
        xs = self._normalize_x(prior).reshape(B, n_chunks, fs, C)
        z = self.init_z[None].expand(B, *self.z_shape).contiguous() if self.learnable_init_z else torch.zeros(B, *self.z_shape, device=prior.device)
        preds, losses = [], []
        for c in range(n_chunks):
            z, pred_x, target = self.transition_model(z, xs[:, c].contiguous(), maps_info, conditions[:, c].contiguous(),
                                                      t=None if t is None else t[c], noise=None if noise is None else noise[c])
            preds.append(pred_x)
            losses.append((pred_x - target) ** 2)
        loss_items = torch.stack(losses)                                          # (n_chunks, B, fs, C)
        logits = self._unnormalize_x(torch.stack(preds, dim=1).reshape(B, S, C))
        return LTMOutput(loss=loss_items.mean(), loss_items=loss_items, logits=logits, pred_dict={"traj_logits": logits},
                         trajectory_label=self._unnormalize_x(xs.permute(1, 0, 2, 3)))


def transition_config(frame_stack: int = 10, z_dim: int = 768, num_train_timesteps: int = 100, num_inference_steps: int = 10,
                      beta_schedule: str = "squaredcos_cap_v2", learnable_init_z: bool = True,
                      data_mean=(30.0, 10.0, 0.0, 0.0), data_std=(100.0, 20.0, 1.0, 1.0), precision: str = "fp32") -> SimpleNamespace:
    """Config with the fields of config/strDiffDiT.yaml that the sampling path reads."""
    scheduler = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=0.0001, beta_end=0.02,
                                beta_schedule=beta_schedule, variance_type="fixed_small", clip_sample=True,
                                prediction_type="epsilon", num_inference_steps=num_inference_steps)
    return SimpleNamespace(frame_stack=frame_stack, x_shape=(4,), z_shape=(frame_stack, z_dim), n_maps_token=788, cond_dim=256,
                           prior_dim=4, map_cond=True, learnable_init_z=learnable_init_z, context_frames=0, scheduler=scheduler,
                           data_mean=list(data_mean), data_std=list(data_std), precision=precision)
