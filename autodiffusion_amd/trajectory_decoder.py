"""Drop-in class for the reference's ``TrajectoryDecoder`` (transformer4planning/models/decoder/base.py:42-260), backed by
libadx.so: the head that turns the backbone's hidden states into ``traj_logits`` -- the ``trajectory_prior`` every refiner /
diffusion wrapper conditions on (models/backbone/str_base.py:752-753, DiffusionAugmentation.py:168).

Parameter names are the reference's (``model.weight`` / ``model.bias`` for the linear head; ``model.mlp.linear.*``,
``model.mlp.layer_norm.*``, ``model.fc.*`` for ``DecoderResCat``, libs/mlp.py:23-55), so a reference ``state_dict`` loads.
Implemented, value only (no gradient: the backbone this head is trained with is not part of this library):

* ``generate_trajs(hidden_output, info_dict)``            base.py:253-260
* ``compute_traj_loss(hidden_output, label, info_dict)``  base.py:95-127 and the default nuplan branch :238-241, :245
  (``mean_circular_loss`` with ``predict_yaw``; ``trajectory_prediction_mode`` other than 'start_frame' / 'both' / 'rebalance' /
  'off_roadx100', which raise NotImplementedError, as do the waymo task and ``pred_key_points_only``).

The hidden-state slice is a strided view; the arithmetic (two products, LayerNorm + ReLU, the loss reduction) runs in the
library's kernels.  Dropout (``traj_dropout``) belongs to training and is not applied.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import check, current_stream, f32c, ptr
from .dit import _ResCat

_UNSUPPORTED_MODES = ("start_frame", "both", "rebalance", "off_roadx100")


class TrajectoryDecoder(nn.Module):
    def __init__(self, config, precision: Optional[str] = None):
        super().__init__()
        self.config = config
        self.precision = precision or getattr(config, "precision", "fp32")
        out_features = 4 if config.predict_yaw else 2
        if config.trajectory_decoder_type == "linear":
            self.model = nn.Linear(config.n_embd, out_features)
        else:
            self.model = _ResCat(config.n_inner, config.n_embd, out_features)
        if "mse" not in config.loss_fn and "l1" not in config.loss_fn:
            raise ValueError(f"loss fn not supported: {config.loss_fn}")
        self._prepared: Dict[str, torch.Tensor] = {}
        self._stamp = None

    def _weights(self) -> Dict[str, torch.Tensor]:
        params = [(n, p) for n, p in self.named_parameters() if p.dim() == 2]
        stamp = tuple((p.data_ptr(), p._version) for _, p in params)
        if stamp != self._stamp:
            self._prepared = {n: ops.prepare_weight(f32c(p), self.precision) for n, p in params}
            self._stamp = stamp
        return self._prepared

    def _lin(self, W, x, name: str, bias: torch.Tensor):
        return ops.linear(x, W[name + ".weight"], bias, precision=self.precision, prepared_n=bias.shape[0])

    @torch.no_grad()
    def decode(self, traj_hidden_state: torch.Tensor) -> torch.Tensor:
        """``self.model(traj_hidden_state)``: (B, T, n_embd) -> (B, T, 4 or 2)."""
        if not traj_hidden_state.is_cuda:
            raise RuntimeError("TrajectoryDecoder runs on the GPU only")
        W = self._weights()
        h = f32c(traj_hidden_state)
        m = self.model
        if isinstance(m, nn.Linear):
            return self._lin(W, h, "model", m.bias)
        hid = self._lin(W, h, "model.mlp.linear", m.mlp.linear.bias)
        hid = ops.layernorm(hid, m.mlp.layer_norm.weight, m.mlp.layer_norm.bias, eps=1e-5, relu=True)
        return self._lin(W, torch.cat([h, hid], dim=-1), "model.fc", m.fc.bias)

    def generate_trajs(self, hidden_output: torch.Tensor, info_dict: dict) -> torch.Tensor:
        pred_length = info_dict.get("pred_length", 0)
        assert pred_length > 0
        return self.decode(hidden_output[:, -pred_length - 1:-1, :])

    @torch.no_grad()
    def compute_traj_loss(self, hidden_output, label, info_dict, no_yaw_loss: bool = False, device=None, **kwargs):
        cfg = self.config
        if getattr(cfg, "task", "nuplan") != "nuplan" or getattr(cfg, "pred_key_points_only", False):
            raise NotImplementedError("compute_traj_loss: only the nuplan task with a full trajectory head")
        if not (getattr(cfg, "mean_circular_loss", True) and cfg.predict_yaw):
            raise NotImplementedError("compute_traj_loss: only mean_circular_loss with predict_yaw (the reference leaves the other loss unreduced)")
        if getattr(cfg, "trajectory_prediction_mode", "all_frames") in _UNSUPPORTED_MODES:
            raise NotImplementedError(f"compute_traj_loss: trajectory_prediction_mode {cfg.trajectory_prediction_mode!r}")
        pred_length = info_dict.get("pred_length", label.shape[1])
        context_length = info_dict.get("context_length", None)
        selected = info_dict.get("selected_indices", None)
        if context_length is not None:
            k = len(selected) if selected is not None else 0
            h = hidden_output[:, context_length - 1 + k:context_length + pred_length - 1 + k, :]
        else:
            h = hidden_output[:, -pred_length - 1:-1, :]
        logits = self.decode(h)
        lab = f32c(label.to(logits.device))
        B, T, C = logits.shape
        if lab.shape[0] != B or lab.shape[1] != T:
            raise ValueError("compute_traj_loss: label and trajectory hidden states disagree in shape")
        part = torch.empty(B, 2, device=logits.device, dtype=torch.float32)
        out = torch.empty(2, device=logits.device, dtype=torch.float32)
        check(_lib.load().adx_op_traj_loss(ptr(logits), ptr(lab), B, T, C, lab.shape[2], int("mse" not in cfg.loss_fn), ptr(part), ptr(out),
                                           current_stream()), "traj_loss")
        loss = out[0] if no_yaw_loss else out[0] + out[1]
        return loss * float(getattr(cfg, "trajectory_loss_rescale", 1.0)), logits


def trajectory_decoder_config(n_embd=256, n_inner=1024, trajectory_decoder_type="mlp", predict_yaw=True, loss_fn="mse", **kw):
    from types import SimpleNamespace
    base = dict(n_embd=n_embd, n_inner=n_inner, trajectory_decoder_type=trajectory_decoder_type, predict_yaw=predict_yaw, loss_fn=loss_fn,
                traj_dropout=0.0, task="nuplan", mean_circular_loss=True, trajectory_prediction_mode="all_frames", pred_key_points_only=False,
                trajectory_loss_rescale=1.0, first_frame_scele=1.0)
    base.update(kw)
    return SimpleNamespace(**base)
