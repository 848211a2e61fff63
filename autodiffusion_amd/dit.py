"""Drop-in classes for the reference's DiT / MM-DiT trajectory decoder (variant B), backed by libadx.so.

Interfaces mirrored (paths relative to the reference tree):

* ``Transformer_TrajDiffusionModel``  transformer4planning/models/diffusion_loss/diffusion_decoder.py:109-277
  (blocks: transformer4planning/libs/diffusionTransformer.py:24-199; encoders: transformer4planning/libs/mlp.py:43-55)
* ``DiffusionWrapper``                transformer4planning/models/diffusion_loss/diffusion_decoder.py:280-479
* ``DiffusionDecoder``                transformer4planning/models/diffusion_loss/diffusion_decoder.py:481-558

Parameter names are the reference's.  Buffers are float32 (the reference leaves the cosine schedule in
float64, which makes its own first Linear fail; its users call ``.float()``).  The 'linear' schedule of this
wrapper is ``linspace(linear_start, linear_end, n)`` -- the reference passes these two keywords to a helper that
does not accept them, so that branch cannot run there and no reference output pins it.
"""
from __future__ import annotations

import os
from types import SimpleNamespace
from typing import Optional

import torch
import torch.nn as nn

from . import _lib, schedules
from .engine import Engine
from .refiner import _Lin, _MHA, _Norm, SinusoidalPosEmb


def uniform_positional_embedding(n_points: int, feat_dim: int) -> torch.Tensor:
    """Initial value of the trainable position table (models/utils.py:104-117): positions 6*(n-i), base 100."""
    import math
    position = torch.tensor([[6 * (n_points - i)] for i in range(n_points)])
    div = torch.exp(torch.arange(0, feat_dim, 2).float() * (-math.log(100.0) / feat_dim))
    pe = torch.zeros(n_points, feat_dim)
    pe[:, 0::2] = torch.sin(position * div)
    pe[:, 1::2] = torch.cos(position * div)
    return pe


class _MLP(nn.Module):
    def __init__(self, n_in, hidden):
        super().__init__()
        self.linear = _Lin(n_in, hidden)
        self.layer_norm = _Norm(hidden)


class _ResCat(nn.Module):
    """fc(cat[x, relu(LN(Linear(x)))]) -- parameter holder for DecoderResCat."""

    def __init__(self, hidden_size, in_features, out_features):
        super().__init__()
        self.mlp = _MLP(in_features, hidden_size)
        self.fc = _Lin(hidden_size + in_features, out_features)


class _Seq(nn.Module):
    """Children registered under the integer names an nn.Sequential would give them."""

    def __init__(self, items: dict):
        super().__init__()
        for k, m in items.items():
            self.add_module(str(k), m)


class _PostNormEncoderLayer(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.self_attn = _MHA(d)
        self.linear1 = _Lin(d, 4 * d)
        self.linear2 = _Lin(4 * d, d)
        self.norm1 = _Norm(d)
        self.norm2 = _Norm(d)


class _DiTBlock(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.attn = _PostNormEncoderLayer(d)
        self.mlp = _Seq({0: _Lin(d, 4 * d), 2: _Lin(4 * d, d)})
        self.adaLN_modulation = _Seq({1: _Lin(d, 6 * d)})


class _RMSNormParam(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(d))   # present in the reference state_dict, never used in its forward


class _JointAttention(nn.Module):
    def __init__(self, d, h):
        super().__init__()
        for n in ("Wq_x", "Wk_x", "Wv_x", "Wq_c", "Wk_c", "Wv_c"):
            setattr(self, n, _Lin(d, d * h, bias=False))
        for n in ("ResNorm_x_q", "ResNorm_x_k", "ResNorm_c_q", "ResNorm_c_k"):
            setattr(self, n, _RMSNormParam(d))
        self.proj = _Lin(2 * d * h, d)


class _MMDiTBlock(nn.Module):
    def __init__(self, d, h):
        super().__init__()
        for s in ("x", "c"):
            setattr(self, f"adaLN_modulation_{s}", _Seq({1: _Lin(d, 6 * d)}))
            setattr(self, f"linear_{s}", _Seq({0: _Lin(d, d)}))
            setattr(self, f"mlp_{s}", _Seq({0: _Lin(d, 4 * d), 2: _Lin(4 * d, d)}))
        self.attn = _JointAttention(d, h)


class _FinalLayer(nn.Module):
    def __init__(self, d, out):
        super().__init__()
        self.linear = _Lin(d, out)
        self.adaLN_modulation = _Seq({1: _Lin(d, 2 * d)})


class Transformer_TrajDiffusionModel(nn.Module):
    """config needs n_embd, n_inner, n_head, n_layer, diffusion_backbone ('DiT' | 'MM-DiT')."""

    def __init__(self, config, out_features=4, predict_range=80, precision: Optional[str] = None):
        super().__init__()
        self.config = config
        self.out_features = out_features
        d = config.n_embd
        self.t_dim = d
        connect = {512: 800, 1024: 1600}.get(d, int(d * 1.5))
        self.connect_dim = connect
        self.predict_range = predict_range
        self.state_encoder = _Seq({0: _ResCat(config.n_inner, d, connect), 1: _ResCat(2 * connect, connect, d)})
        self.time_encoder = _Seq({1: _Lin(d, 2 * d), 3: _Lin(2 * d, 2 * d), 5: _Lin(2 * d, d)})
        self.x_encoder = _Seq({0: _ResCat(64, out_features, 128), 1: _ResCat(256, 128, 512), 2: _ResCat(1024, 512, 768),
                               3: _ResCat(2048, 768, d)})
        if config.diffusion_backbone == "DiT":
            self.blocks = nn.ModuleList([_DiTBlock(d) for _ in range(config.n_layer)])
        elif config.diffusion_backbone == "MM-DiT":
            self.blocks = nn.ModuleList([_MMDiTBlock(d, config.n_head) for _ in range(config.n_layer)])
        else:
            raise ValueError("Augment of diffusion backbone is invalid!")
        self.final_layer = _FinalLayer(d, out_features)
        self.position_embedding = nn.Parameter(uniform_positional_embedding(predict_range, d).unsqueeze(0))
        self.precision = precision or getattr(config, "precision", "fp32")
        self._engine: Optional[Engine] = None
        self._engine_key = None

    def engine(self) -> Engine:
        key = self.precision
        if self._engine is None or self._engine_key != key:
            if self._engine is not None:
                self._engine.close()
            c = self.config
            self._engine = Engine(
                variant=_lib.VARIANT_DIT if c.diffusion_backbone == "DiT" else _lib.VARIANT_MMDIT,
                precision=_lib.PRECISIONS[self.precision], elide=0,
                d_model=c.n_embd, n_head=c.n_head, n_enc_layers=0, n_layers=c.n_layer, horizon=self.predict_range,
                in_dim=self.out_features, out_dim=self.out_features, n_inner=c.n_inner, connect_dim=self.connect_dim)
            self._engine_key = key
        # RMSNorm weights of the joint attention are unused by the forward pass: not sent to the engine
        eng = self._engine
        params = [(n, p) for n, p in self.named_parameters() if ".ResNorm_" not in n]
        stamp = tuple((n, p.data_ptr(), p._version, str(p.device)) for n, p in params)
        if stamp != eng._stamp:
            eng.load(params)
            eng._stamp = stamp
        return eng

    def forward(self, x, t, state, info_dict=None):
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.long, device=x.device)
        if t.dim() == 0:
            t = t[None]
        if t.shape[0] != x.shape[0]:
            t = t.expand(x.shape[0])
        return self.engine().denoise_dit(x, t, state)


class DiffusionWrapper(nn.Module):
    def __init__(self, model, beta_schedule="linear", linear_start=0.01, linear_end=0.9, n_timesteps=10, clip_denoised=True,
                 predict_epsilon=True, max_action=100, predict_range=None, model_args=None):
        super().__init__()
        self.model = model
        self.model_args = model_args
        self.n_timesteps = int(n_timesteps)
        self.action_dim = model.out_features
        self.predict_range = predict_range
        assert clip_denoised, ""
        assert predict_epsilon, ""
        self.clip_denoised, self.max_action, self.predict_epsilon = clip_denoised, max_action, predict_epsilon
        if beta_schedule == "linear":
            betas = torch.linspace(linear_start, linear_end, self.n_timesteps, dtype=torch.float64)
        elif beta_schedule == "cosine":
            betas = schedules.cosine_beta_schedule(self.n_timesteps)
        else:
            raise ValueError(f"unknown beta schedule {beta_schedule}")
        for k, v in schedules.diffusion_buffers(betas).items():
            self.register_buffer(k, v)

    def schedule_rows(self, determin: bool):
        """Recomputed only when a schedule buffer changed: reading the buffers back synchronises the device."""
        stamp = (tuple((getattr(self, k).data_ptr(), getattr(self, k)._version) for k in schedules.BUFFER_NAMES),
                 float(self.max_action), bool(determin))
        cache = self.__dict__.setdefault("_rows_cache", {})
        if cache.get("stamp") != stamp:
            b = {k: getattr(self, k).detach().float().cpu() for k in schedules.BUFFER_NAMES}
            cache["rows"] = schedules.ddpm_rows(b, "pred_noise", clamp=float(self.max_action), noise_at_every_step=True,
                                                stochastic=not determin)
            cache["stamp"] = stamp
        return cache["rows"]

    def forward(self, hidden_state, batch_size=0, label=None, info_dict=None, get_inter=False, **kwargs):
        if self.training:
            return self.train_forward(hidden_state, label, info_dict=info_dict, **kwargs)
        assert batch_size > 0
        return self.sample_forward(hidden_state, batch_size, info_dict=info_dict, get_inter=get_inter, **kwargs)

    def train_forward(self, hidden_state, trajectory_label, info_dict=None, t: Optional[torch.Tensor] = None,
                      noise: Optional[torch.Tensor] = None):
        """The training loss (diffusion_decoder.py:346-365): normalise the label, q_sample at one random timestep per row, one
        denoiser call; returns the element-wise squared error (B, S, C) as the reference does.  With gradients enabled and trainable
        parameters the denoiser call is the differentiable training path (``train.py``, both backbones; gradients also reach
        ``hidden_state`` when it requires them); under ``torch.no_grad()`` only the value."""
        from . import train
        grad = train.wants_grad(self.model, hidden_state)
        if grad:
            return self._train_forward(hidden_state, trajectory_label, info_dict, t, noise, True)
        with torch.no_grad():
            return self._train_forward(hidden_state, trajectory_label, info_dict, t, noise, False)

    def _train_forward(self, hidden_state, trajectory_label, info_dict, t, noise, grad: bool):
        from . import ops, train
        x0 = ops.normalize(_lib.f32c(trajectory_label))
        B = x0.shape[0]
        if t is None:
            t = torch.randint(0, self.n_timesteps, (B,), device=x0.device)
        t = t.to(device=x0.device, dtype=torch.int64).contiguous()
        noise = torch.randn_like(x0) if noise is None else _lib.f32c(noise)
        x_noisy, _ = ops.q_sample(x0, noise, t, self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod)
        if grad:
            x_recon = train.denoise_dit_with_grad(self.model, x_noisy, t, hidden_state)
        else:
            x_recon = self.model(x_noisy, t, hidden_state, info_dict=info_dict)
        return (x_recon - noise) ** 2

    @torch.no_grad()
    def sample_forward(self, state, batch_size, info_dict=None, get_inter=False, mc_num=1, determin=True,
                       noise: Optional[torch.Tensor] = None, return_scores: bool = False, seed: Optional[int] = None, row0: int = 0,
                       **kwargs):
        """Returns the de-normalised trajectories: (B, S, C) for mc_num == 1, (B, mc_num, S, C) otherwise.

        ``noise`` (1 + draws, B * mc_num, S, C): None draws it on the device (``seed``, ``row0``: generator seed and global index
        of row 0 of the B * mc_num sample rows).  With ``return_scores`` also returns ``total_cls`` = -mean(x_N^2, dim=1) -
        sum_steps mean(z^2, dim=1), the sample score the reference accumulates; it depends on the noise only.
        """
        if get_inter:
            raise NotImplementedError("get_inter is not supported")
        seq_len = self.predict_range if self.predict_range is not None else state.shape[-2]
        B = batch_size * mc_num
        group = 1
        if mc_num > 1:
            determin_eff = False
            # The reference repeats the state mc_num times (diffusion_decoder.py:408-416).  The engine takes the sample group
            # instead and runs everything that depends on the conditioning alone once per scenario -- same results;
            # ADX_DIT_GROUP=0 (read per call) keeps the repeated form.
            if os.environ.get("ADX_DIT_GROUP", "1") != "0":
                group = mc_num
            else:
                state = torch.repeat_interleave(state, mc_num, dim=0)
        else:
            determin_eff = determin
        eng = self.model.engine()
        rows = self.schedule_rows(determin_eff)
        eng.set_schedule(rows)
        flags = _lib.FLAG_NORMALIZE
        if determin_eff:
            flags |= _lib.FLAG_INIT_ZERO
            noise = None
        elif noise is None:
            if return_scores:       # the scores are a function of the noise: materialise it (same generator, same values)
                from . import ops
                if seed is None:
                    seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
                noise = ops.randn(1 + schedules.n_noise_draws(rows), B, seq_len * self.action_dim, seed, row0,
                                  device=state.device).reshape(-1, B, seq_len, self.action_dim)
            else:
                eng.seed_from_torch(seed, row0)
        out = eng.sample_dit(_lib.f32c(state), _lib.f32c(noise) if noise is not None else None, flags, group=group)
        if mc_num > 1:
            out = out.reshape(batch_size, mc_num, *out.shape[1:])
        if not return_scores:
            return out
        if noise is None:
            cls = torch.zeros(B, self.action_dim, device=state.device)
        elif mc_num > 1:
            cls = -(noise ** 2).mean(dim=(2, 3)).sum(0).reshape(batch_size, mc_num)
        else:
            cls = -(noise ** 2).mean(dim=2).sum(0)
        return out, cls


class DiffusionDecoder(nn.Module):
    """config needs: predict_yaw, k, predict_range, diffusion_backbone, beta_schedule, diffusion_timesteps, mc_num and the
    fields of Transformer_TrajDiffusionModel."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.out_features = 4 if config.predict_yaw else 2
        self.k = config.k
        self.predict_range = config.predict_range
        if "DiT" not in config.diffusion_backbone:
            raise ValueError("Augment of diffusion backbone is invalid!")
        net = Transformer_TrajDiffusionModel(config, out_features=self.out_features, predict_range=self.predict_range)
        self.model = DiffusionWrapper(net, beta_schedule=config.beta_schedule, n_timesteps=config.diffusion_timesteps,
                                      predict_range=self.predict_range)

    def forward(self, label, hidden_output, info_dict=None, device=None, t: Optional[torch.Tensor] = None,
                noise: Optional[torch.Tensor] = None):
        """-> (loss, logits): summed squared error of the noise estimate times ``trajectory_loss_rescale`` and all-zero logits
        (diffusion_decoder.py:502-532).  Differentiable when gradients are enabled (see DiffusionWrapper.train_forward);
        the value only under ``torch.no_grad()``."""
        if not self.config.predict_yaw:
            label = label[..., :2].clone()
        loss = self.model.train_forward(hidden_output, label, info_dict=info_dict, t=t, noise=noise).sum()
        return loss * float(getattr(self.config, "trajectory_loss_rescale", 1.0)), torch.zeros_like(label)

    @torch.no_grad()
    def generate(self, hidden_output, info_dict=None, noise: Optional[torch.Tensor] = None, seed: Optional[int] = None, row0: int = 0):
        pred_traj_num = hidden_output.shape[0]
        assert self.config.predict_range > 0 and pred_traj_num > 0
        if self.k == 1:
            return self.model(hidden_output, info_dict=info_dict, batch_size=pred_traj_num, determin=True)
        # the reference also receives the samples' scores here, but its mode selection multiplies them by zero
        # (utils/modify_traj_utils.py:183-184), so they are not computed
        traj = self.model(hidden_output, info_dict=info_dict, batch_size=pred_traj_num, determin=False,
                          mc_num=self.config.mc_num, noise=noise, seed=seed, row0=row0)
        if self.config.mc_num > self.k:
            traj, _ = select_k_from_mc(traj, None, self.k)
        return traj


def select_k_from_mc(traj_logits: torch.Tensor, scores: torch.Tensor, k: int):
    """Reduce (B, mc_num, S, C) samples to k modes (farthest-point sampling on the end points + 25 EM iterations of a
    diagonal Gaussian mixture); returns (modes (B, k, S, C), mode weights (B, k)) ordered by weight.  As in the reference
    (transformer4planning/models/utils.py:120-132; modify_traj_utils.py:183-184 multiplies the scores by zero) the sample
    scores do not influence the result."""
    from . import ops
    return ops.select_modes(traj_logits, k, em_iter=25)


def dit_config(n_embd=256, n_layer=4, n_head=8, backbone="DiT", n_inner=None, k=1, mc_num=1, diffusion_timesteps=10,
               beta_schedule="cosine", precision="fp32") -> SimpleNamespace:
    return SimpleNamespace(n_embd=n_embd, n_inner=n_inner or 4 * n_embd, n_head=n_head, n_layer=n_layer,
                           diffusion_backbone=backbone, predict_yaw=True, k=k, predict_range=80, beta_schedule=beta_schedule,
                           diffusion_timesteps=diffusion_timesteps, mc_num=mc_num, trajectory_loss_rescale=1.0,
                           precision=precision)
