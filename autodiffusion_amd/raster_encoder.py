"""Conditioning producer, first stage: the raster encoder that emits ``maps_info`` (phase 1 of ``generate()``), backed by libadx.so.

Interfaces mirrored (paths relative to the reference tree):

* ``NuplanRasterizeEncoder`` (ViT branch)  transformer4planning/models/encoder/nuplan_raster_encoder.py:69-122, 145-312
  (base class: models/encoder/base.py:6-62; ``cat_raster_seq``: models/utils.py:14-41)
* the ``transformers.ViTModel`` it instantiates (nuplan_raster_encoder.py:77-84; third-party, pinned 4.42.3 in the reference's
  requirements.txt:212): parameter names are those of that version, i.e. of a reference checkpoint
  (``image_downsample.encoder.layer.{i}.attention.attention.query.weight`` ...); state dicts written by newer ``transformers``
  releases (``layers.{i}.attention.q_proj`` ...) are renamed on load.

``forward(**kwargs) -> (input_embeds, info_dict, maps_info)`` as the reference; ``maps_info (B, 4 * 197, n_embd)`` stays on the
device and goes straight into the refiner's ``generate(trajectory_prior=..., maps_info=...)``.

Every arithmetic step is a libadx kernel: one gather turns the channels-last rasters into the ViT's patch matrix
(``adx_op_raster_patches``: the reference's permute + per-frame channel regrouping + the unfolding of the stride-16 Conv2d), the
patch embedding, q / k / v / output and MLP products are ``adx_op_linear`` (GELU in the epilogue), LayerNorm (eps 1e-12),
attention over the 197 tokens and the residual adds are the library's operators; torch only concatenates and slices.  The high-
and low-resolution rasters share the ViT, so both go through it as ONE batch of 2 * B * frames images.

Not built: the ResNet branch (``raster_encoder_type`` other than 'vit': needs torchvision's residual networks), the DINOv2
camera branch, proposals, the k-means key-point tokenizer.
"""
from __future__ import annotations

import random
import re
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import check, current_stream, f32c, ptr
from .refiner import _Lin, _Norm


def raster_patches(raster: torch.Tensor, framenum: int, traffic: bool = True) -> torch.Tensor:
    """(B, H, W, C) channels-last raster -> (B * framenum * (H/16) * (W/16), C' * 256) patch matrix, C' = C - 8 * (framenum - 1)
    (minus 4 without traffic lights); see ``adx_op_raster_patches`` in include/adx.h."""
    raster = f32c(raster)
    B, H, W, C = raster.shape
    cf = C - 8 * (framenum - 1) - (0 if traffic else 4)
    out = torch.empty(B * framenum * (H // 16) * (W // 16), cf * 256, device=raster.device, dtype=torch.float32)
    check(_lib.load().adx_op_raster_patches(ptr(raster), B, H, W, C, int(framenum), int(bool(traffic)), ptr(out), current_stream()),
          "raster_patches")
    return out


def _add(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    out = torch.empty_like(a)
    check(_lib.load().adx_op_add(ptr(a), ptr(b), ptr(out), a.numel(), current_stream()), "add")
    return out


def _tanh(x: torch.Tensor) -> torch.Tensor:
    y = torch.empty_like(x)
    check(_lib.load().adx_op_act(ptr(x), ptr(y), x.numel(), _lib.ACT_TANH, current_stream()), "act")
    return y


class _Named(nn.Module):
    """Children under given names (what nested modules of the reference would be called)."""

    def __init__(self, items: Dict[str, nn.Module]):
        super().__init__()
        for k, m in items.items():
            self.add_module(k, m)


class _ViTLayer(nn.Module):
    def __init__(self, d: int, inter: int):
        super().__init__()
        self.attention = _Named({"attention": _Named({"query": _Lin(d, d), "key": _Lin(d, d), "value": _Lin(d, d)}),
                                 "output": _Named({"dense": _Lin(d, d)})})
        self.intermediate = _Named({"dense": _Lin(d, inter)})
        self.output = _Named({"dense": _Lin(inter, d)})
        self.layernorm_before = _Norm(d)
        self.layernorm_after = _Norm(d)


_NEW_NAMES = {"attention.q_proj": "attention.attention.query", "attention.k_proj": "attention.attention.key",
              "attention.v_proj": "attention.attention.value", "attention.o_proj": "attention.output.dense",
              "mlp.fc1": "intermediate.dense", "mlp.fc2": "output.dense"}


def checkpoint_name(name: str) -> str:
    """A ViT parameter name of a newer ``transformers`` release -> its name in a transformers-4.42.3 (reference) checkpoint."""
    m = re.match(r"(.*)layers\.(\d+)\.(.*)", name)
    if not m or ".encoder.layer." in name or name.startswith("encoder.layer."):
        return name
    pre, i, rest = m.groups()
    for k, v in _NEW_NAMES.items():
        if rest.startswith(k + "."):
            rest = v + rest[len(k):]
            break
    return f"{pre}encoder.layer.{i}.{rest}"


class ViTModel(nn.Module):
    """Parameter holder + device forward of the ViT the raster encoder uses (patch 16, image 224, 12 layers by default, GELU,
    LayerNorm eps 1e-12, biased q / k / v).  ``precision``: operand type of the products ("fp32" | "bf16" | "bf16x3")."""

    EPS = 1e-12

    def __init__(self, hidden_size: int, num_channels: int, intermediate_size: int, num_attention_heads: int, num_hidden_layers: int = 12,
                 image_size: int = 224, patch_size: int = 16, precision: str = "fp32"):
        super().__init__()
        if patch_size != 16 or hidden_size % num_attention_heads:
            raise ValueError("ViTModel: patch size 16 and a hidden size divisible by the head count")
        n_tok = (image_size // patch_size) ** 2 + 1
        proj = nn.Module()
        proj.weight = nn.Parameter(torch.randn(hidden_size, num_channels, patch_size, patch_size) * 0.02)
        proj.bias = nn.Parameter(torch.zeros(hidden_size))
        emb = _Named({"patch_embeddings": _Named({"projection": proj})})
        emb.cls_token = nn.Parameter(torch.randn(1, 1, hidden_size))
        emb.position_embeddings = nn.Parameter(torch.randn(1, n_tok, hidden_size))
        self.embeddings = emb
        self.encoder = _Named({"layer": nn.ModuleList([_ViTLayer(hidden_size, intermediate_size) for _ in range(num_hidden_layers)])})
        self.layernorm = _Norm(hidden_size)
        self.pooler = _Named({"dense": _Lin(hidden_size, hidden_size)})      # in the checkpoint, unused by the encoder
        self.hidden_size, self.n_head, self.n_tok, self.precision = hidden_size, num_attention_heads, n_tok, precision
        self._prepared: Dict[str, torch.Tensor] = {}
        self._stamp = None
        self._register_load_state_dict_pre_hook(self._rename_keys)

    def _rename_keys(self, state_dict, prefix, *args):
        """load_state_dict pre-hook: accept the key layout of newer ``transformers`` releases."""
        for k in [k for k in state_dict if k.startswith(prefix) and ".layers." in "." + k[len(prefix):]]:
            state_dict[prefix + checkpoint_name(k[len(prefix):])] = state_dict.pop(k)
        own = {"weight": self.pooler.dense.weight.shape, "bias": self.pooler.dense.bias.shape}
        for leaf, shape in own.items():                 # newer releases give the (unused) pooler another width: keep ours
            k = prefix + "pooler.dense." + leaf
            if k in state_dict and state_dict[k].shape != shape:
                state_dict[k] = getattr(self.pooler.dense, leaf).detach().clone()

    def _weights(self) -> Dict[str, torch.Tensor]:
        """Weights in the form ``adx_op_linear`` takes for this precision, rebuilt when a parameter changed."""
        params = [(n, p) for n, p in self.named_parameters() if p.dim() >= 2 and "embeddings.cls" not in n and "position_embeddings" not in n
                  and not n.startswith("pooler.")]
        stamp = (self.precision,) + tuple((n, p.data_ptr(), p._version) for n, p in params)
        if stamp != self._stamp:
            self._prepared = {n: ops.prepare_weight(f32c(p).reshape(p.shape[0], -1), self.precision) for n, p in params}
            self._stamp = stamp
        return self._prepared

    def _lin(self, W, x, name: str, bias: torch.Tensor, act: int = _lib.ACT_NONE):
        return ops.linear(x, W[name + ".weight"], bias, act=act, precision=self.precision, prepared_n=bias.shape[0])

    @torch.no_grad()
    def encode_patches(self, patches: torch.Tensor, n_img: int) -> torch.Tensor:
        """patches (n_img * 196, C * 256) -> last_hidden_state (n_img, 197, hidden)."""
        W = self._weights()
        d, S = self.hidden_size, self.n_tok
        e = self.embeddings
        tok = self._lin(W, patches, "embeddings.patch_embeddings.projection", e.patch_embeddings.projection.bias).reshape(n_img, S - 1, d)
        x = torch.cat([f32c(e.cls_token).expand(n_img, 1, d), tok], dim=1).contiguous()
        x = _add(x, f32c(e.position_embeddings).expand(n_img, S, d).contiguous())
        for i, layer in enumerate(self.encoder.layer):
            p = f"encoder.layer.{i}."
            a = layer.attention
            h = ops.layernorm(x, layer.layernorm_before.weight, layer.layernorm_before.bias, eps=self.EPS)
            q = self._lin(W, h, p + "attention.attention.query", a.attention.query.bias)
            k = self._lin(W, h, p + "attention.attention.key", a.attention.key.bias)
            v = self._lin(W, h, p + "attention.attention.value", a.attention.value.bias)
            o = ops.attention(q, k, v, self.n_head, precision=self.precision)
            x = _add(x, self._lin(W, o, p + "attention.output.dense", a.output.dense.bias))
            h = ops.layernorm(x, layer.layernorm_after.weight, layer.layernorm_after.bias, eps=self.EPS)
            h = self._lin(W, h, p + "intermediate.dense", layer.intermediate.dense.bias, _lib.ACT_GELU_ERF)
            x = _add(x, self._lin(W, h, p + "output.dense", layer.output.dense.bias))
        return ops.layernorm(x, self.layernorm.weight, self.layernorm.bias, eps=self.EPS)


class NuplanRasterizeEncoder(nn.Module):
    """config needs: n_embd, n_head, raster_channels, vit_intermediate_size, raster_encoder_type ('vit'), use_speed, use_key_points,
    separate_kp_encoder, use_proposal, with_traffic_light, x_random_walk / y_random_walk (and arf_* for key points)."""

    def __init__(self, config, precision: Optional[str] = None):
        super().__init__()
        self.config = config
        self.use_proposal = config.use_proposal
        self.use_key_points = config.use_key_points
        if not (isinstance(config.raster_encoder_type, str) and "vit" in config.raster_encoder_type):
            raise NotImplementedError("only the ViT raster encoder is built (raster_encoder_type='vit'); the ResNet branch needs torchvision")
        if self.use_proposal:
            raise NotImplementedError("proposal embeddings are not built")
        if getattr(config, "camera_image_encoder", None):
            raise NotImplementedError("the DINOv2 camera branch is not built")
        kp = self.use_key_points
        if "specified" in kp:
            table = {"specified_forward": [4, 9, 19, 39, 79], "specified_backward": [79, 39, 19, 9, 4], "specified_two_backward": [79, 4],
                     "specified_first_second": [10]}
            assert kp in table, f"specified key points should be either specified_forward or specified_backward {kp}"
            self.selected_indices: List[int] = table[kp]
        elif "universal" in kp:
            self.selected_indices = [15, 31, 47, 63, 79]
        else:
            self.selected_indices = []
        precision = precision or getattr(config, "precision", "fp32")
        self.image_downsample = ViTModel(config.n_embd // 2, config.raster_channels, config.vit_intermediate_size, config.n_head,
                                         precision=precision)
        self.action_m_embed = _Named({"0": _Lin(7 if config.use_speed else 4, config.n_embd)})       # index 1 is the Tanh
        if kp != "no":
            assert config.separate_kp_encoder
            self.kps_m_embed = _Named({"0": _Lin(2, config.n_embd)})

    def _augment(self, traj: torch.Tensor, x_scale: float, y_scale: float, is_training: bool) -> torch.Tensor:
        """DataAugmentation.trajectory_linear_augmentation (models/encoder/augmentation.py:24-49): identity unless training with a
        non-zero random walk; then x / y are scaled by 1 + s * u * scale * 2 - scale with s falling linearly from 1 to 1 / n over
        the points and one host-side uniform draw u per channel (training-time data augmentation, tiny tensors: host arithmetic)."""
        if x_scale == 0 and y_scale == 0:
            return traj
        traj = traj.clone()
        n = traj.shape[1]
        s = ((torch.arange(n, device=traj.device, dtype=traj.dtype) + 1) / n).flip(0)[None, :]
        if is_training and x_scale > 0:
            traj[..., 0] *= 1 + s * random.random() * x_scale * 2 - x_scale
        if is_training and y_scale > 0:
            traj[..., 1] *= 1 + s * random.random() * y_scale * 2 - y_scale
        return traj

    @torch.no_grad()
    def maps_info(self, high_res_raster: torch.Tensor, low_res_raster: torch.Tensor, context_actions: torch.Tensor) -> torch.Tensor:
        """(B, 224, 224, C) x 2, (B, frames, 4 | 7) -> (B, frames * 197, n_embd): per frame its 196 patch tokens (high-resolution
        features, then low-resolution features) followed by its action token (nuplan_raster_encoder.py:176-203, 222)."""
        B, F = context_actions.shape[0], context_actions.shape[1]
        traffic = bool(self.config.with_traffic_light)
        c_frame = high_res_raster.shape[-1] - 8 * (F - 1) - (0 if traffic else 4)
        assert c_frame == self.config.raster_channels, "raster channel number should be {}, but got {}".format(self.config.raster_channels, c_frame)
        patches = torch.cat([raster_patches(high_res_raster, F, traffic), raster_patches(low_res_raster, F, traffic)], dim=0)
        hid = self.image_downsample.encode_patches(patches, 2 * B * F)[:, 1:, :]              # (2 B F, 196, n_embd / 2)
        hid = hid.reshape(2, B, F, hid.shape[1], hid.shape[2])
        lin = getattr(self.action_m_embed, "0")
        act = _tanh(ops.linear(f32c(context_actions), f32c(lin.weight), f32c(lin.bias)))     # (B, F, n_embd)
        state = torch.cat([hid[0], hid[1]], dim=-1)                                            # (B, F, 196, n_embd)
        return torch.cat([state, act[:, :, None, :]], dim=2).reshape(B, F * (state.shape[2] + 1), act.shape[-1])

    def select_keypoints(self, info_dict):
        label = info_dict["trajectory_label"]
        if "specified" in self.use_key_points:
            return torch.stack([label[:, i, :] for i in self.selected_indices], dim=1)      # slices, not an index tensor: nothing is uploaded (graph capture)
        if "universal" in self.use_key_points:
            return label[:, 19::20, :]
        assert False, "key points should be either specified or universal"

    @torch.no_grad()
    def forward(self, **kwargs):
        high, low = kwargs.get("high_res_raster"), kwargs.get("low_res_raster")
        context_actions, label = kwargs.get("context_actions"), kwargs.get("trajectory_label")
        is_training = kwargs.get("is_training")
        assert is_training is not None, "is_training should not be None"
        assert label is not None, "trajectory_label should not be None"
        cfg = self.config
        device = label.device
        B, pred_length = label.shape[0], label.shape[1]
        context_actions = self._augment(context_actions, cfg.x_random_walk, cfg.y_random_walk, is_training)
        maps = self.maps_info(high.to(device), low.to(device), context_actions.to(device))
        n_embd = maps.shape[-1]
        info = {"trajectory_label": label, "pred_length": pred_length, "context_length": maps.shape[1], "aug_current": kwargs.get("aug_current")}
        tail = torch.zeros(B, pred_length, n_embd, device=device, dtype=maps.dtype)
        if self.use_key_points == "no":
            embeds = torch.cat([maps, tail], dim=1)
            info["future_key_points"] = None
        else:
            kps = self.select_keypoints(info)
            assert kps.shape[1] != 0, "future points not enough to sample"
            aug = self._augment(kps.clone(), getattr(cfg, "arf_x_random_walk", 0.0), getattr(cfg, "arf_y_random_walk", 0.0), is_training)[:, :, :2]
            lin = getattr(self.kps_m_embed, "0")
            kp_emb = _tanh(ops.linear(f32c(aug), f32c(lin.weight), f32c(lin.bias)))
            embeds = torch.cat([maps, kp_emb, tail], dim=1)
            info["future_key_points"] = kps
        info["selected_indices"] = self.selected_indices
        return embeds, info, maps

