"""Conditioning producer, first stage (oracle; test infrastructure only): rasters + context actions -> ``maps_info``.

Restates, in plain torch on the CPU:

* ``cat_raster_seq``                      transformer4planning/models/utils.py:14-41
* ``NuplanRasterizeEncoder.forward``      transformer4planning/models/encoder/nuplan_raster_encoder.py:145-222 (ViT branch), :246-312
* the ``ViTModel`` it instantiates        third-party ``transformers`` (pinned 4.42.3 in the reference's requirements.txt:212;
                                          ViTConfig defaults: 12 layers, patch 16, image 224, GELU, LayerNorm eps 1e-12, qkv bias):
  Conv2d patch embedding, [CLS] + learned position table, 12 pre-LayerNorm blocks (LN -> multi-head attention -> dense + residual;
  LN -> dense -> GELU(erf) -> dense + residual), final LayerNorm; the pooler is not used by the encoder.

Pinned by ``tests/golden/g18_raster_encoder.npz``, recorded from the reference's own ``NuplanRasterizeEncoder`` (with the
``transformers`` version installed in the development container, whose ViT computes the same function under other parameter
names).  Weights are keyed by the parameter names of a reference checkpoint (transformers 4.42.3 layout).
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def cat_raster_seq(raster: torch.Tensor, framenum: int, traffic: bool = True) -> torch.Tensor:
    """(B, C, H, W) -> (B, framenum, C', H, W): every frame gets the route / road (/ traffic-light) channels plus its own 8
    agent channels, which sit at ``route + 24 + f + a * framenum`` in the input (utils.py:14-41)."""
    b, c, h, w = raster.shape
    route = c - 8 * framenum - 24
    shared = raster[:, :route + 24] if traffic else raster[:, :route + 20]
    frames = []
    for f in range(framenum):
        agents = raster[:, route + 24 + f::framenum]
        frames.append(torch.cat([shared, agents], dim=1))
    return torch.stack(frames, dim=1)


def vit_forward(sd: SD, pixel_values: torch.Tensor, n_head: int, prefix: str = "") -> torch.Tensor:
    """ViTModel(...).last_hidden_state: (N, C, 224, 224) -> (N, 197, hidden)."""
    p = prefix
    x = F.conv2d(pixel_values, sd[p + "embeddings.patch_embeddings.projection.weight"], sd[p + "embeddings.patch_embeddings.projection.bias"],
                 stride=16)
    x = x.flatten(2).transpose(1, 2)
    n, _, d = x.shape
    x = torch.cat([sd[p + "embeddings.cls_token"].expand(n, 1, d), x], dim=1) + sd[p + "embeddings.position_embeddings"]
    n_layer = len({k[len(p):].split(".")[2] for k in sd if k.startswith(p + "encoder.layer.")})
    dh = d // n_head
    for i in range(n_layer):
        q = f"{p}encoder.layer.{i}."
        h = F.layer_norm(x, (d,), sd[q + "layernorm_before.weight"], sd[q + "layernorm_before.bias"], 1e-12)
        qq = F.linear(h, sd[q + "attention.attention.query.weight"], sd[q + "attention.attention.query.bias"])
        kk = F.linear(h, sd[q + "attention.attention.key.weight"], sd[q + "attention.attention.key.bias"])
        vv = F.linear(h, sd[q + "attention.attention.value.weight"], sd[q + "attention.attention.value.bias"])
        S = x.shape[1]
        split = lambda t: t.reshape(n, S, n_head, dh).transpose(1, 2)  # noqa: E731
        att = torch.softmax(split(qq) @ split(kk).transpose(-1, -2) / math.sqrt(dh), dim=-1) @ split(vv)
        att = att.transpose(1, 2).reshape(n, S, d)
        x = x + F.linear(att, sd[q + "attention.output.dense.weight"], sd[q + "attention.output.dense.bias"])
        h = F.layer_norm(x, (d,), sd[q + "layernorm_after.weight"], sd[q + "layernorm_after.bias"], 1e-12)
        h = F.gelu(F.linear(h, sd[q + "intermediate.dense.weight"], sd[q + "intermediate.dense.bias"]))
        x = x + F.linear(h, sd[q + "output.dense.weight"], sd[q + "output.dense.bias"])
    return F.layer_norm(x, (d,), sd[p + "layernorm.weight"], sd[p + "layernorm.bias"], 1e-12)


def raster_maps_info(sd: SD, high_res_raster: torch.Tensor, low_res_raster: torch.Tensor, context_actions: torch.Tensor, n_head: int,
                     traffic: bool = True) -> torch.Tensor:
    """``maps_info`` of NuplanRasterizeEncoder.forward (ViT branch, nuplan_raster_encoder.py:176-203, 222): rasters
    (B, 224, 224, C) channels-last, context_actions (B, F, 4 or 7) -> (B, F * 197, n_embd): per context frame the 196 patch tokens
    (high-resolution features followed by low-resolution features on the feature axis) and then the frame's action token."""
    B, F_ = context_actions.shape[0], context_actions.shape[1]
    act = torch.tanh(F.linear(context_actions, sd["action_m_embed.0.weight"], sd["action_m_embed.0.bias"]))
    feats = []
    for raster in (high_res_raster, low_res_raster):
        seq = cat_raster_seq(raster.permute(0, 3, 2, 1), F_, traffic)
        b, f, c, h, w = seq.shape
        hid = vit_forward(sd, seq.reshape(b * f, c, h, w), n_head, "image_downsample.")[:, 1:, :]
        feats.append(hid.reshape(B, F_, hid.shape[1], hid.shape[2]))
    state = torch.cat(feats, dim=-1)
    return torch.cat([state, act[:, :, None, :]], dim=2).reshape(B, -1, act.shape[-1])


def encoder_forward(sd: SD, high_res_raster, low_res_raster, context_actions, trajectory_label, n_head: int, selected_indices,
                    traffic: bool = True):
    """(input_embeds, maps_info) of the encoder for ``use_key_points`` 'no' (``selected_indices`` empty) or a 'specified_*' /
    'universal' choice with a separate key-point embedding (nuplan_raster_encoder.py:262-312), evaluation mode (no augmentation)."""
    maps = raster_maps_info(sd, high_res_raster, low_res_raster, context_actions, n_head, traffic)
    B, pred_length = trajectory_label.shape[0], trajectory_label.shape[1]
    tail = torch.zeros(B, pred_length, maps.shape[-1])
    if not selected_indices:
        return torch.cat([maps, tail], dim=1), maps
    kp = trajectory_label[:, list(selected_indices), :2]
    kp_emb = torch.tanh(F.linear(kp, sd["kps_m_embed.0.weight"], sd["kps_m_embed.0.bias"]))
    return torch.cat([maps, kp_emb, tail], dim=1), maps
