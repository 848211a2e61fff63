"""Conditioning producer, middle stage (oracle; test infrastructure only): the sparse mixture-of-experts decoder between the raster
encoder and the trajectory head, and the autoregressive key-point loop of ``generate()`` that drives it.

Restates, in plain torch on the CPU:

* ``STR.generate``            transformer4planning/models/backbone/str_base.py:457-813 -- the nuplan branch without proposals,
                              ``k`` = 1, key points through a separate embedding (:542-581), the key-point loop (:583-591, :596,
                              :702-708), the final pass and the trajectory head (:712, :752-756), the returned dictionary (:773-783)
* ``STR.embedding_to_hidden`` str_base.py:379-387 -> the ``MixtralModel`` that ``STR_Mixtral`` instantiates
                              (models/backbone/mixtral.py:17-19).  The decoder itself is third-party (``transformers``, pinned 4.42.3
                              in the reference's requirements.txt:212, absent from the reference tree); restated from its published
                              algorithm: per layer  h += o_proj(causal attention(rotary(q_proj(n1)), rotary(k_proj(n1)), v_proj(n1)))
                              with n1 = RMSNorm(h), grouped key / value heads, rotate-half rotary embedding with base ``rope_theta``;
                              h += MoE(RMSNorm(h)) with router probabilities softmax(gate(x)) in fp32, the ``top_k`` largest kept and
                              renormalised to sum 1, expert e = w2_e(silu(w1_e x) * w3_e x); final RMSNorm.
* ``KeyPointMLPDeocder.generate_keypoints``  transformer4planning/models/decoder/base.py:543-560 (``k`` = 1: DecoderResCat to 2 values)

Weights are keyed by the parameter names of a reference checkpoint (transformers 4.42.3 layout: ``transformer.layers.{i}.
block_sparse_moe.gate.weight``, ``...experts.{e}.w1|w2|w3.weight``).  Pinned by ``tests/golden/g21_str_generate.npz``, recorded from
the reference's own ``STR_Mixtral.generate`` (with the ``transformers`` version installed in the development container, whose
Mixtral computes the same function under other parameter names).
"""
from __future__ import annotations

import math
from typing import Dict, Sequence

import torch
import torch.nn.functional as F

from . import nets, raster, trajectory_decoder

SD = Dict[str, torch.Tensor]


def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    return w * (x * torch.rsqrt(x.square().mean(-1, keepdim=True) + eps))


def rotary_tables(n_pos: int, dh: int, theta: float):
    """cos / sin (n_pos, dh) of the rotate-half rotary embedding: angle(p, j) = p * theta^(-2 (j mod dh/2) / dh), fp32 throughout."""
    inv = 1.0 / (theta ** (torch.arange(0, dh, 2, dtype=torch.int64).float() / dh))
    ang = torch.arange(n_pos, dtype=torch.float32)[:, None] * inv[None, :]
    ang = torch.cat([ang, ang], dim=-1)
    return ang.cos(), ang.sin()


def rotate(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """x (B, H, S, dh): x cos + rotate_half(x) sin with rotate_half(x) = [-x2, x1]."""
    h = x.shape[-1] // 2
    return x * cos + torch.cat([-x[..., h:], x[..., :h]], dim=-1) * sin


def moe_block(sd: SD, p: str, x: torch.Tensor, n_experts: int, top_k: int) -> torch.Tensor:
    """x (T, d) -> (T, d): every token through its ``top_k`` most probable experts, weighted by the renormalised probabilities."""
    prob = torch.softmax(F.linear(x, sd[p + "gate.weight"]).float(), dim=-1)
    val, idx = torch.topk(prob, top_k, dim=-1)
    val = val / val.sum(-1, keepdim=True)
    out = torch.zeros_like(x)
    for e in range(n_experts):
        tok, slot = torch.nonzero(idx == e, as_tuple=True)
        if tok.numel() == 0:
            continue
        q = f"{p}experts.{e}."
        xe = x[tok]
        he = F.silu(F.linear(xe, sd[q + "w1.weight"])) * F.linear(xe, sd[q + "w3.weight"])
        out.index_add_(0, tok, F.linear(he, sd[q + "w2.weight"]) * val[tok, slot, None])
    return out


def mixtral_forward(sd: SD, x: torch.Tensor, cfg, prefix: str = "transformer.") -> torch.Tensor:
    """last_hidden_state of the decoder on input embeddings x (B, S, d) with positions 0 .. S-1 and a causal mask.
    cfg: n_layer, n_head, n_kv_head, n_experts, top_k, rms_eps, rope_theta."""
    B, S, d = x.shape
    H, Hkv = cfg["n_head"], cfg["n_kv_head"]
    dh = d // H
    cos, sin = rotary_tables(S, dh, cfg["rope_theta"])
    causal = torch.ones(S, S, dtype=torch.bool).tril()
    for i in range(cfg["n_layer"]):
        p = f"{prefix}layers.{i}."
        n1 = rms_norm(x, sd[p + "input_layernorm.weight"], cfg["rms_eps"])
        q = F.linear(n1, sd[p + "self_attn.q_proj.weight"]).reshape(B, S, H, dh).transpose(1, 2)
        k = F.linear(n1, sd[p + "self_attn.k_proj.weight"]).reshape(B, S, Hkv, dh).transpose(1, 2)
        v = F.linear(n1, sd[p + "self_attn.v_proj.weight"]).reshape(B, S, Hkv, dh).transpose(1, 2)
        q, k = rotate(q, cos, sin), rotate(k, cos, sin)
        k = k.repeat_interleave(H // Hkv, dim=1)               # query head h reads key / value head h // (H / Hkv)
        v = v.repeat_interleave(H // Hkv, dim=1)
        s = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
        att = torch.softmax(s.masked_fill(~causal, float("-inf")), dim=-1) @ v
        x = x + F.linear(att.transpose(1, 2).reshape(B, S, d), sd[p + "self_attn.o_proj.weight"])
        n2 = rms_norm(x, sd[p + "post_attention_layernorm.weight"], cfg["rms_eps"])
        x = x + moe_block(sd, p + "block_sparse_moe.", n2.reshape(B * S, d), cfg["n_experts"], cfg["top_k"]).reshape(B, S, d)
    return rms_norm(x, sd[prefix + "norm.weight"], cfg["rms_eps"])


def sub(sd: SD, prefix: str) -> SD:
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def str_generate(sd: SD, cfg, high_res_raster, low_res_raster, context_actions, trajectory_label, selected_indices: Sequence[int],
                 traffic: bool = True):
    """``STR.generate`` -> {"traj_logits" (B, T, 4), "maps_info" (B, 788, d), "key_points_logits" (B, n_kp, 2)} (the last only with
    key points).  The loop runs the decoder on the growing prefix exactly as the reference does (one full pass per key point)."""
    enc = sub(sd, "encoder.")
    embeds, maps = raster.encoder_forward(enc, high_res_raster, low_res_raster, context_actions, trajectory_label, cfg["n_head"],
                                          selected_indices, traffic)
    B, pred_length = trajectory_label.shape[0], trajectory_label.shape[1]
    ctx = maps.shape[1]
    out = {"maps_info": maps}
    n_kp = len(selected_indices)
    if n_kp:
        # the key-point slots start as the embedding of (0, 0) and are overwritten by the embedded predictions one by one
        zero_kp = torch.tanh(F.linear(torch.zeros(B, n_kp, 2), enc["kps_m_embed.0.weight"], enc["kps_m_embed.0.bias"]))
        embeds = embeds.clone()
        embeds[:, ctx:ctx + n_kp] = zero_kp
        kp_sd = sub(sd, "key_points_decoder.")
        kps = []
        for i in range(n_kp):
            hid = mixtral_forward(sd, embeds[:, :ctx + i], cfg)[:, ctx + i - 1]
            kp = nets.res_cat(hid[:, None, :], kp_sd, "model")                                     # (B, 1, 2)
            embeds[:, ctx + i] = torch.tanh(F.linear(kp, enc["kps_m_embed.0.weight"], enc["kps_m_embed.0.bias"]))[:, 0]
            kps.append(kp[:, 0, :2])
        out["key_points_logits"] = torch.stack(kps, dim=1)
    hidden = mixtral_forward(sd, embeds, cfg)
    out["traj_logits"] = trajectory_decoder.generate_trajs(sub(sd, "traj_decoder."), hidden, pred_length, cfg.get("traj_head", "mlp"))
    out["last_hidden_state"] = hidden
    return out
