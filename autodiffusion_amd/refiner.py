"""Drop-in classes for the reference's live refinement path (variant A), backed by libadx.so.

Interfaces mirrored (paths relative to the reference tree):

* ``Transformer``          transformer4planning/models/diffusion_loss/transformer.py:10-273
* ``DiffusionWrapper``     transformer4planning/models/diffusion_loss/TrajRefiner.py:123-458
* ``TrajectoryRefiner``    transformer4planning/models/diffusion_loss/TrajRefiner.py:50-121
* ``DiffusionAugmentation`` (refiner part) transformer4planning/models/backbone/DiffusionAugmentation.py:29-195

Parameter and buffer names are the reference's, so a reference ``state_dict`` loads
with ``load_state_dict``.  The modules only *hold* parameters; every forward runs in
the HIP engine.  Implemented: sampling (eval mode); the forward-only evaluation of the
training loss (what the HF-Trainer evaluation entry computes before it samples,
transformer4planning/trainer.py:643); and, whenever gradients are enabled and a parameter
requires them, the differentiable training path of ``train.py`` (fp32; the chunks of a
trajectory run as one batch whose map embedding is computed once per scenario), so
``loss.backward()`` and any torch optimizer work on these classes.

Deviations, all deliberate:
* ``Transformer.forward`` returns ``(B, T, C)`` always (the reference's trailing
  ``squeeze()`` drops size-1 axes, transformer.py:273).
* caller tensors are never modified (the reference normalises ``prior[...]`` in place
  inside the condition dict, TrajRefiner.py:322-323).
* every sampling entry point takes an optional ``noise`` tensor so results are
  reproducible and shard-invariant; without it, noise is drawn on the device inside
  the captured graph (counter-based generator keyed by ``seed`` and the global batch
  row ``row0 + b``; ``seed=None`` takes a fresh seed from torch's CPU generator).
* dropout: the differentiable training path applies the reference's dropout layers when the
  denoiser is in training mode (``p_drop_emb`` / ``p_drop_attn``, see ``train.py``), with
  draws from a device-side integer hash instead of torch's generator, so a training run
  matches the reference in distribution, and exactly when the multipliers are supplied
  (``train.set_dropout``).  The value-only loss under ``torch.no_grad()`` runs in the sampling
  engine and is the dropout-free value whatever the mode (the evaluation entry calls it in
  ``.eval()``, where the two agree).  ``forward`` takes
  optional ``t`` / ``noise``.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from collections import OrderedDict
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import _lib, schedules
from .engine import Engine


class LTMOutput(OrderedDict):
    """The reference's output record (DiffusionAugmentation.py:19-25): fields readable as attributes and as mapping items;
    fields that are None are left out of the mapping, as transformers' ModelOutput does."""

    def __init__(self, loss=None, logits=None, trajectory_label=None, loss_items=None, pred_dict=None):
        super().__init__()
        for k, v in (("loss", loss), ("logits", logits), ("trajectory_label", trajectory_label), ("loss_items", loss_items),
                     ("pred_dict", pred_dict)):
            object.__setattr__(self, k, v)
            if v is not None:
                self[k] = v


# ------------------------------------------------------------------ parameter holders
class _Lin(nn.Module):
    def __init__(self, n_in: int, n_out: int, bias: bool = True):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(n_out, n_in) * 0.02)
        if bias:
            self.bias = nn.Parameter(torch.zeros(n_out))


class _Norm(nn.Module):
    def __init__(self, d: int):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d))
        self.bias = nn.Parameter(torch.zeros(d))


class _MHA(nn.Module):
    def __init__(self, d: int):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.randn(3 * d, d) * 0.02)
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.out_proj = _Lin(d, d)


class _EncoderLayer(nn.Module):
    def __init__(self, d: int):
        super().__init__()
        self.self_attn = _MHA(d)
        self.linear1 = _Lin(d, 4 * d)
        self.linear2 = _Lin(4 * d, d)
        self.norm1 = _Norm(d)
        self.norm2 = _Norm(d)


class _DecoderLayer(nn.Module):
    def __init__(self, d: int):
        super().__init__()
        self.self_attn = _MHA(d)
        self.multihead_attn = _MHA(d)
        self.linear1 = _Lin(d, 4 * d)
        self.linear2 = _Lin(4 * d, d)
        self.norm1 = _Norm(d)
        self.norm2 = _Norm(d)
        self.norm3 = _Norm(d)


class _Stack(nn.Module):
    def __init__(self, layers):
        super().__init__()
        self.layers = nn.ModuleList(layers)


class SinusoidalPosEmb(nn.Module):
    """models/diffusion_loss/positional_embedding.py:5-17."""

    def __init__(self, dim: int):
        super().__init__()
        self.dim = dim

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        from . import ops
        return ops.timestep_embedding(x, self.dim)


class Transformer(nn.Module):
    """Encoder/decoder denoiser.  Constructor arguments as in the reference; ``precision`` ("fp32" | "bf16" | "bf16x3" | "f16":
    the split-bf16 engine whose two per-step kernels over all conditioning rows multiply operands rounded once to fp16)
    and ``elide`` (skip mask-unobservable work) select the engine build."""

    def __init__(self, input_dim: int, output_dim: int, horizon: int, n_cond_steps: int = None, cond_dim: int = 0,
                 n_prior_steps: int = 0, prior_dim: int = 0, n_layer: int = 4, n_head: int = 8, n_emb: int = 32,
                 p_drop_emb: float = 0.1, p_drop_attn: float = 0.1, causal_attn: bool = False, time_as_cond: bool = True,
                 obs_as_cond: bool = False, n_cond_layers: int = 2, map_cond: bool = True,
                 precision: str = "fp32", elide: int = 3) -> None:
        super().__init__()
        if not time_as_cond or not causal_attn or cond_dim <= 0 or n_cond_layers <= 0:
            raise NotImplementedError("only the configuration the refiner instantiates is implemented: "
                                      "time_as_cond=True, causal_attn=True, cond_dim>0, n_cond_layers>0")
        T = horizon
        T_cond = 1 + n_prior_steps + horizon + (n_cond_steps if map_cond else 0)
        self.out_features = output_dim
        self.input_emb = _Lin(input_dim, n_emb)
        self.pos_emb = nn.Parameter(torch.randn(1, T, n_emb) * 0.02)
        self.time_emb = SinusoidalPosEmb(n_emb)
        self.map_cond = map_cond
        self.cond_obs_emb = _Lin(cond_dim, n_emb) if map_cond else None
        self.prior_z_emb = _Lin(prior_dim, n_emb)
        self.cond_pos_emb = nn.Parameter(torch.randn(1, T_cond, n_emb) * 0.02 if map_cond else torch.zeros(1, T_cond, n_emb))
        self.encoder = _Stack([_EncoderLayer(n_emb) for _ in range(n_cond_layers)])
        self.decoder = _Stack([_DecoderLayer(n_emb) for _ in range(n_layer)])
        i = torch.arange(T)[:, None]
        mask = torch.zeros(T, T).masked_fill(torch.arange(T)[None, :] > i, float("-inf"))
        mem = torch.zeros(T, T_cond).masked_fill(torch.arange(T_cond)[None, :] > i + 1, float("-inf"))
        self.register_buffer("mask", mask)
        self.register_buffer("memory_mask", mem)
        self.ln_f = _Norm(n_emb)
        self.head = _Lin(n_emb, output_dim)
        self.T, self.T_cond, self.horizon = T, T_cond, horizon
        self.time_as_cond, self.obs_as_cond, self.encoder_only = True, True, False
        self._dims = dict(d_model=n_emb, n_head=n_head, n_enc_layers=n_cond_layers, n_layers=n_layer, horizon=horizon,
                          in_dim=input_dim, out_dim=output_dim, n_prior_tokens=n_prior_steps,
                          prior_dim=prior_dim, n_maps_token=n_cond_steps or 0, maps_dim=cond_dim, map_cond=int(map_cond))
        if n_prior_steps != horizon:
            # the conditioning sequence holds `horizon` transition rows and `n_prior_steps` prior rows; both are
            # embedded by prior_z_emb, so they must have the same token count in every reference instantiation
            raise NotImplementedError("n_prior_steps must equal horizon")
        self.precision = precision
        self.elide = int(elide)
        self.p_drop_emb, self.p_drop_attn = float(p_drop_emb), float(p_drop_attn)      # recorded; no kernel applies them
        self._n_head = int(n_head)
        self._engine: Optional[Engine] = None
        self._engine_key = None

    # engine build/sync
    def engine(self) -> Engine:
        key = (self.precision, self.elide)
        if self._engine is None or self._engine_key != key:
            prec = _lib.PRECISIONS[self.precision]
            if self._engine is not None:
                self._engine.close()
            self._engine = Engine(variant=_lib.VARIANT_A, precision=prec, elide=int(self.elide), **self._dims)
            self._engine_key = key
        self._engine.sync_from(self)
        return self._engine

    def forward(self, noisy_x: torch.Tensor, timestep, condition: Dict[str, torch.Tensor], **kwargs) -> torch.Tensor:
        B = noisy_x.shape[0]
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.long, device=noisy_x.device)
        elif t.dim() == 0:
            t = t[None].to(noisy_x.device)
        if t.shape[0] != B:
            if t.shape[0] != 1:
                raise ValueError("timestep must be a scalar or have one entry per batch row")
            t = t.expand(B)
        return self.engine().denoise_a(noisy_x, t, condition.get("maps_info") if self.map_cond else None,
                                       condition["transition_info"], condition["trajectory_prior"])


class DiffusionWrapper(nn.Module):
    """Gaussian-diffusion sampler around a denoiser; registers the reference's buffers under the same names."""

    def __init__(self, model, timesteps=1000, sampling_timesteps=None, objective="pred_noise", beta_schedule="sigmoid",
                 schedule_fn_kwargs=dict(), ddim_sampling_eta=0., min_snr_loss_weight=True, min_snr_gamma=5,
                 auto_normalize_data=True, residual=False):
        super().__init__()
        self.model = model
        self.objective = objective
        self.timesteps = timesteps
        self.residual = residual
        assert objective in {"pred_noise", "pred_x0", "pred_v"}, \
            "objective must be either pred_noise (predict noise) or pred_x0 (predict image start) or pred_v (predict v)"
        betas = schedules.make_betas(beta_schedule, timesteps)
        self.num_timesteps = int(betas.shape[0])
        self.sampling_timesteps = sampling_timesteps if sampling_timesteps is not None else timesteps
        assert self.sampling_timesteps <= timesteps
        self.is_ddim_sampling = self.sampling_timesteps < timesteps
        self.ddim_sampling_eta = ddim_sampling_eta
        for k, v in schedules.diffusion_buffers(betas).items():
            self.register_buffer(k, v)
        ac64 = torch.cumprod(1.0 - betas, dim=0)
        self.register_buffer("loss_weight", schedules.loss_weight(ac64, objective, min_snr_gamma, min_snr_loss_weight).to(torch.float32))
        self.auto_normalize_data = bool(auto_normalize_data)

    @property
    def device(self):
        return self.betas.device

    def _buffers_cpu(self):
        return {k: getattr(self, k).detach().float().cpu() for k in schedules.BUFFER_NAMES}

    def schedule_rows(self):
        """Coefficient rows of the registered schedule buffers; recomputed only when a buffer (or a sampling option)
        changed -- reading the buffers back is a device synchronisation, which must not sit in every generate()."""
        stamp = (tuple((getattr(self, k).data_ptr(), getattr(self, k)._version) for k in schedules.BUFFER_NAMES),
                 self.objective, bool(self.is_ddim_sampling), self.sampling_timesteps, float(self.ddim_sampling_eta))
        if getattr(self, "_rows_stamp", None) != stamp:
            b = self._buffers_cpu()
            if self.is_ddim_sampling:
                self._rows = schedules.ddim_rows(b, self.objective, self.sampling_timesteps, float(self.ddim_sampling_eta))
            else:
                self._rows = schedules.ddpm_rows(b, self.objective)
            self._rows_stamp = stamp
        return self._rows

    def flags(self) -> int:
        return (_lib.FLAG_NORMALIZE if self.auto_normalize_data else 0) | (_lib.FLAG_RESIDUAL if self.residual else 0)

    def make_noise(self, shape, device, n_chunks: Optional[int] = None):
        """Host-side noise in the layout the engine reads (for callers that want the tensor; the default path draws on the device)."""
        n = 1 + schedules.n_noise_draws(self.schedule_rows())
        lead = (n,) if n_chunks is None else (n_chunks, n)
        return torch.randn(*lead, *shape, device=device, dtype=torch.float32)

    @torch.no_grad()
    def sample(self, prior: Dict[str, torch.Tensor], trajectory_label: torch.Tensor, return_all_timesteps: bool = False,
               noise: Optional[torch.Tensor] = None, trace: bool = False, seed: Optional[int] = None, row0: int = 0):
        if return_all_timesteps:
            raise NotImplementedError("return_all_timesteps is not supported; pass trace=True for per-step tensors")
        eng = self.model.engine()
        eng.set_schedule(self.schedule_rows())
        if noise is None:
            eng.seed_from_torch(seed, row0)
        maps = _lib.f32c(prior["maps_info"]) if self.model.map_cond else None
        return eng.sample_a(maps, _lib.f32c(prior["transition_info"]), _lib.f32c(prior["trajectory_prior"]),
                            _lib.f32c(noise) if noise is not None else None, self.flags(), trace=trace)

    def p_losses(self, prior: Dict[str, torch.Tensor], x_start: torch.Tensor, t: Optional[torch.Tensor] = None,
                 noise: Optional[torch.Tensor] = None, return_rows: bool = False):
        """The training loss (TrajRefiner.py:417-452): normalise, q_sample at a random timestep per row, one denoiser call,
        min-SNR weighted mean squared error.  ``t`` (B,) int64 and ``noise`` like ``x_start`` may be given.  With gradients
        enabled and trainable parameters the result is differentiable (fp32 training path, ``train.py``): ``loss.backward()``
        fills the denoiser's ``.grad``s as the reference's autograd does; otherwise (``torch.no_grad()``, the evaluation
        entry) only the value is computed, on the sampling engine."""
        from . import train
        if train.wants_grad(self.model, prior.get("transition_info")):
            return self._p_losses_grad(prior, x_start, t, noise, return_rows)
        with torch.no_grad():
            return self._p_losses_value(prior, x_start, t, noise, return_rows)

    def _p_losses_grad(self, prior, x_start, t, noise, return_rows, maps_repeat: int = 1):
        from . import ops, train
        x0 = _lib.f32c(x_start)
        tran, pri = prior["transition_info"], _lib.f32c(prior["trajectory_prior"])
        if self.auto_normalize_data:
            x0, pri = ops.normalize(x0), ops.normalize(pri)
            if tran.requires_grad:           # init_x reaches the loss through the transition rows: keep this affine map in autograd
                C = tran.shape[-1]
                scale = torch.tensor([1 / 25.0, 1 / 5.0, 1.0, 1.0][:C], device=tran.device, dtype=torch.float32)
                shift = torch.tensor([-1.0, 0.0, 0.0, 0.0][:C], device=tran.device, dtype=torch.float32)
                tran = tran.to(torch.float32) * scale + shift
            else:
                tran = ops.normalize(_lib.f32c(tran))
        B = x0.shape[0]
        if t is None:
            t = torch.randint(0, self.timesteps, (B,), device=x0.device)
        t = t.to(device=x0.device, dtype=torch.int64).contiguous()
        noise = _lib.f32c(noise if noise is not None else torch.randn_like(x0))
        x, v = ops.q_sample(x0, noise, t, self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod, want_v=self.objective == "pred_v")
        cond = dict(maps_info=prior.get("maps_info"), transition_info=tran, trajectory_prior=pri)
        model_out = train.denoise_with_grad(self.model, x, t, cond, self.model._n_head, maps_repeat)
        target = {"pred_noise": noise, "pred_x0": x0, "pred_v": v}[self.objective]
        mean = train.WeightedMSE.apply(model_out, target, t, self.loss_weight)
        if return_rows:
            return mean, ops.weighted_mse(model_out.detach(), target, t, self.loss_weight)[0]
        return mean

    def _p_losses_value(self, prior, x_start, t, noise, return_rows):
        from . import ops
        x0 = _lib.f32c(x_start)
        tran, pri = _lib.f32c(prior["transition_info"]), _lib.f32c(prior["trajectory_prior"])
        if self.auto_normalize_data:
            x0, tran, pri = ops.normalize(x0), ops.normalize(tran), ops.normalize(pri)
        B = x0.shape[0]
        if t is None:
            t = torch.randint(0, self.timesteps, (B,), device=x0.device)
        t = t.to(device=x0.device, dtype=torch.int64).contiguous()
        if noise is None:
            noise = torch.randn_like(x0)
        noise = _lib.f32c(noise)
        x, v = ops.q_sample(x0, noise, t, self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod, want_v=self.objective == "pred_v")
        cond = dict(maps_info=prior.get("maps_info"), transition_info=tran, trajectory_prior=pri)
        model_out = self.model(x, t, cond)
        target = {"pred_noise": noise, "pred_x0": x0, "pred_v": v}[self.objective]
        rows, mean = ops.weighted_mse(model_out, target, t, self.loss_weight)
        return (mean, rows) if return_rows else mean

    def forward(self, prior, trajectory_label, get_inter=False, **kwargs):
        if self.training:
            return self.p_losses(prior, trajectory_label, **kwargs)
        return self.sample(prior, trajectory_label, return_all_timesteps=get_inter, **kwargs)


class TrajectoryRefiner(nn.Module):
    """cfg needs: trajectory_dim, frame_stack, n_maps_token, maps_dim, map_cond, beta_schedule, diffusion_timesteps,
    objective, ddim, normalize, residual (plus optional sampling_timesteps, precision, elide)."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        model = Transformer(
            input_dim=cfg.trajectory_dim, output_dim=cfg.trajectory_dim, horizon=cfg.frame_stack,
            n_cond_steps=cfg.n_maps_token, cond_dim=cfg.maps_dim, n_prior_steps=cfg.frame_stack, prior_dim=cfg.trajectory_dim,
            causal_attn=True, map_cond=cfg.map_cond,
            precision=getattr(cfg, "precision", "fp32"), elide=getattr(cfg, "elide", 3))
        self.diffusion = DiffusionWrapper(
            model, beta_schedule=cfg.beta_schedule, timesteps=cfg.diffusion_timesteps, objective=cfg.objective,
            sampling_timesteps=getattr(cfg, "sampling_timesteps", None),
            ddim_sampling_eta=cfg.ddim, auto_normalize_data=cfg.normalize, residual=cfg.residual)

    @torch.no_grad()
    def generate(self, label, transition_info, trajectory_prior, maps_info, noise: Optional[torch.Tensor] = None, **kwargs):
        condition = {"maps_info": maps_info, "transition_info": transition_info, "trajectory_prior": trajectory_prior}
        return self.diffusion.sample(condition, label, noise=noise, **kwargs)

    def forward(self, label, transition_info, trajectory_prior, maps_info, t: Optional[torch.Tensor] = None,
                noise: Optional[torch.Tensor] = None):
        """-> (loss, logits): the diffusion loss of one chunk times ``trajectory_loss_rescale`` and, as in the reference
        (TrajRefiner.py:100-121), all-zero ``logits`` of the label's shape."""
        condition = {"maps_info": maps_info, "transition_info": transition_info, "trajectory_prior": trajectory_prior}
        if not self.diffusion.training:
            raise RuntimeError("TrajectoryRefiner.forward computes the loss: call .train() first (generate() samples)")
        loss = self.diffusion(condition, label, t=t, noise=noise) * float(getattr(self.cfg, "trajectory_loss_rescale", 1.0))
        return loss, torch.zeros_like(label)


class DiffusionAugmentation(nn.Module):
    """The refiner half of the reference's DiffusionAugmentation wrapper: chunk loop around TrajectoryRefiner.

    Phase 1 (the frozen STR backbone that yields ``traj_logits`` (B,80,4) and ``maps_info`` (B,788,256)) is not part
    of this package.  Pass ``str_model`` (any object whose ``generate(**kwargs)`` returns that dict) to get the
    reference's ``generate(**batch)`` behaviour, or call ``generate(trajectory_prior=..., maps_info=...)`` directly.
    """

    def __init__(self, config, str_model=None):
        super().__init__()
        config.trajectory_dim = 4
        config.learnable_init_traj = True
        config.maps_dim = 256
        config.n_maps_token = 788
        self.cfg = self.config = config
        self.trajectory_dim = 4
        self.frame_stack = config.frame_stack
        self.map_cond = config.map_cond
        self.residual = config.residual
        self.str_model = [str_model]          # kept out of the module tree on purpose
        self.diffusion = TrajectoryRefiner(config)
        self.init_x = nn.Parameter(torch.randn(self.frame_stack, self.trajectory_dim))

    def _phase1(self, kwargs):
        if "trajectory_prior" in kwargs and "maps_info" in kwargs:
            return kwargs["trajectory_prior"], kwargs["maps_info"]
        if self.str_model[0] is None:
            raise ValueError("needs either trajectory_prior/maps_info or a str_model for phase 1")
        str_result = self.str_model[0].generate(**kwargs)
        return str_result["traj_logits"], str_result["maps_info"]

    @torch.no_grad()
    def generate(self, noise: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, seed: Optional[int] = None,
                 row0: int = 0, **kwargs):
        """-> {"traj_logits": (B, 80, 4) float32}.  ``noise``: (n_chunks, 1 + draws, B, fs, 4); None draws it on the device
        (``seed`` / ``row0``: generator seed and global index of batch row 0, see the module docstring)."""
        trajectory_prior, maps_info = self._phase1(kwargs)
        B, S = trajectory_prior.shape[:2]
        if S % self.frame_stack != 0:
            raise ValueError("Number of frames must be divisible by frame stack size")
        wrapper = self.diffusion.diffusion
        eng = wrapper.model.engine()
        eng.set_schedule(wrapper.schedule_rows())
        if noise is None:
            eng.seed_from_torch(seed, row0)
        maps = _lib.f32c(maps_info) if self.map_cond else None
        traj = eng.generate_a(maps, _lib.f32c(trajectory_prior), _lib.f32c(self.init_x),
                              _lib.f32c(noise) if noise is not None else None, self.frame_stack, wrapper.flags(), out=out)
        return {"traj_logits": traj}

    def forward(self, return_dict=None, t: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None, **kwargs):
        """The training objective (DiffusionAugmentation.py:75-135) -- its value under ``torch.no_grad()`` (what evaluation
        reports), a differentiable loss when gradients are enabled (one batched pass over all chunks, ``train.py``): per chunk
        one diffusion loss of the label chunk (label minus prior with ``residual``) conditioned on the prior chunk and on the
        previous chunk's ``prediction`` -- which the reference's refiner returns as zeros, so chunks after the first see a
        zero transition.  ``t`` (n_chunks, B) and ``noise`` (n_chunks, B, fs, 4) fix the random draws.
        -> LTMOutput(loss, loss_items (n_chunks,), logits (B, 80, 4) zeros, pred_dict, trajectory_label)."""
        trajectory_prior, maps_info = self._phase1(kwargs)
        trajectory_label = kwargs.get("trajectory_label", None)
        if trajectory_label is None:
            raise ValueError("forward() needs trajectory_label")
        trajectory_label = trajectory_label.to(trajectory_prior.device)
        if self.residual:
            trajectory_label = trajectory_label - trajectory_prior
        B, S = trajectory_prior.shape[:2]
        fs = self.frame_stack
        if S % fs != 0:
            raise ValueError("Number of frames must be divisible by frame stack size")
        n_chunks = S // fs
        label = _lib.f32c(trajectory_label).reshape(B, n_chunks, fs, -1)
        prior = _lib.f32c(trajectory_prior).reshape(B, n_chunks, fs, -1)
        from . import train
        if train.wants_grad(self.diffusion.diffusion.model, self.init_x):
            # Training: the chunks do not depend on each other here (every chunk after the first sees a zero transition), so
            # they run as ONE batch of n_chunks * B rows (row = chunk * B + scenario) that shares the scenarios' map tokens.
            C = label.shape[-1]
            rows = lambda v: v.permute(1, 0, 2, 3).reshape(n_chunks * B, fs, C).contiguous()  # noqa: E731
            transition = torch.cat([self.init_x[None].expand(B, fs, C).to(torch.float32),
                                    torch.zeros((n_chunks - 1) * B, fs, C, device=label.device, dtype=torch.float32)], dim=0)
            cond = {"maps_info": maps_info, "transition_info": transition, "trajectory_prior": rows(prior)}
            mean, per_row = self.diffusion.diffusion._p_losses_grad(
                cond, rows(label), None if t is None else t.reshape(-1), None if noise is None else noise.reshape(n_chunks * B, fs, C),
                True, maps_repeat=n_chunks)
            rescale = float(getattr(self.diffusion.cfg, "trajectory_loss_rescale", 1.0))
            final_predict = torch.zeros(B, S, C, device=label.device, dtype=torch.float32)
            return LTMOutput(loss=mean * rescale, loss_items=per_row.reshape(n_chunks, B).mean(dim=1) * rescale, logits=final_predict,
                             pred_dict={"traj_logits": final_predict}, trajectory_label=trajectory_label)
        transition = self.init_x[None].expand(B, *self.init_x.shape).contiguous()
        was_training = self.diffusion.training
        self.diffusion.train()                # the wrapper's forward() switches on the mode: loss when training, sample otherwise
        losses, preds = [], []
        try:
            for c in range(n_chunks):
                l, prediction = self.diffusion(label[:, c].contiguous(), transition, prior[:, c].contiguous(), maps_info,
                                               t=None if t is None else t[c], noise=None if noise is None else noise[c])
                transition = prediction
                losses.append(l.reshape(()))
                preds.append(prediction)
        finally:
            self.diffusion.train(was_training)
        final_predict = torch.stack(preds, dim=1).reshape(B, S, -1)
        loss = torch.stack(losses)
        return LTMOutput(loss=loss.mean(), loss_items=loss, logits=final_predict, pred_dict={"traj_logits": final_predict},
                         trajectory_label=trajectory_label)


def refiner_config(frame_stack: int = 10, diffusion_timesteps: int = 100, objective: str = "pred_noise",
                   beta_schedule: str = "cosine", ddim: float = 0.0, normalize: bool = True, residual: bool = False,
                   map_cond: bool = True, sampling_timesteps: Optional[int] = None, precision: str = "fp32",
                   elide: int = 3) -> SimpleNamespace:
    """The fields the CLI copies onto the model config (utils/args.py:154-183; str_base.py:1045-1053)."""
    return SimpleNamespace(trajectory_dim=4, frame_stack=frame_stack, n_maps_token=788, maps_dim=256, map_cond=map_cond,
                           beta_schedule=beta_schedule, diffusion_timesteps=diffusion_timesteps, objective=objective,
                           ddim=ddim, normalize=normalize, residual=residual, trajectory_loss_rescale=1.0,
                           sampling_timesteps=sampling_timesteps, precision=precision, elide=elide)
