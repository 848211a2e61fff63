"""Denoiser networks, fp32 CPU restatement (oracle; test infrastructure only).

Weights are plain ``dict[str, Tensor]`` keyed by the reference's parameter names.

* Variant A -- encoder/decoder ``Transformer``:
  transformer4planning/models/diffusion_loss/transformer.py:189-273
  (embeddings :219-248, encoder :249, decoder :261-266, head :270-273, masks :106-124).
* Variant B -- ``Transformer_TrajDiffusionModel`` (DiT / MM-DiT):
  transformer4planning/models/diffusion_loss/diffusion_decoder.py:251-277 with blocks from
  transformer4planning/libs/diffusionTransformer.py:24-53 (DiTBlock), :56-117 (DiT_Attention),
  :120-179 (MM_DiTBlock), :182-199 (DiT_FinalLayer) and transformer4planning/libs/mlp.py:5-55.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# ------------------------------------------------------------------ primitives
def sinusoidal_pos_emb(t: torch.Tensor, dim: int) -> torch.Tensor:
    """positional_embedding.py:10-17 / libs/embeds.py:13-20: [sin(t*w), cos(t*w)], w_j = 1e4^(-j/(half-1))."""
    half = dim // 2
    w = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000.0) / (half - 1)))
    arg = t.to(torch.float32)[..., None] * w
    return torch.cat([arg.sin(), arg.cos()], dim=-1)


def linear(x, sd: SD, prefix: str, bias: bool = True):
    b = sd.get(prefix + ".bias") if bias else None
    return F.linear(x, sd[prefix + ".weight"], b)


def layer_norm(x, w=None, b=None, eps: float = 1e-5):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    y = (x - mu) / torch.sqrt(var + eps)
    if w is not None:
        y = y * w + b
    return y


def attention(q, k, v, n_head: int, allowed: Optional[torch.Tensor] = None, scale: Optional[float] = None):
    """softmax(q k^T * scale [masked]) v per head.  q (B,Sq,Hd), k (B,Sk,Hd), v (B,Sk,Hv)."""
    B, Sq, _ = q.shape
    Sk = k.shape[1]
    qh = q.reshape(B, Sq, n_head, -1).permute(0, 2, 1, 3)
    kh = k.reshape(B, Sk, n_head, -1).permute(0, 2, 1, 3)
    vh = v.reshape(B, Sk, n_head, -1).permute(0, 2, 1, 3)
    if scale is None:
        scale = 1.0 / math.sqrt(qh.shape[-1])
    s = torch.matmul(qh, kh.transpose(-1, -2)) * scale
    if allowed is not None:
        s = s.masked_fill(~allowed, float("-inf"))
    p = torch.softmax(s, dim=-1)
    o = torch.matmul(p, vh)
    return o.permute(0, 2, 1, 3).reshape(B, Sq, -1)


def mha(x_q, x_kv, sd: SD, prefix: str, n_head: int, allowed=None):
    """torch.nn.MultiheadAttention semantics: packed in_proj (rows q,k,v), out_proj."""
    w, b = sd[prefix + ".in_proj_weight"], sd[prefix + ".in_proj_bias"]
    d = w.shape[1]
    q = F.linear(x_q, w[:d], b[:d])
    k = F.linear(x_kv, w[d:2 * d], b[d:2 * d])
    v = F.linear(x_kv, w[2 * d:], b[2 * d:])
    o = attention(q, k, v, n_head, allowed)
    return F.linear(o, sd[prefix + ".out_proj.weight"], sd[prefix + ".out_proj.bias"])


def _ln(x, sd, prefix, eps=1e-5):
    return layer_norm(x, sd[prefix + ".weight"], sd[prefix + ".bias"], eps)


# ------------------------------------------------------------------ variant A
def variant_a_masks(T: int, S: int):
    """Boolean 'allowed' masks: causal (T,T) j<=i ; memory (T,S) j<=i+1 (transformer.py:106-124)."""
    i = torch.arange(T)[:, None]
    causal = torch.arange(T)[None, :] <= i
    memory = torch.arange(S)[None, :] <= i + 1
    return causal, memory


def variant_a_forward(sd: SD, x, t, maps_info, transition_info, trajectory_prior,
                      n_head: int = 8, map_cond: bool = True, return_inter: bool = False):
    """One denoiser call.  x (B,T,Cin), t (B,) int64, maps (B,788,256), transition/prior (B,P,Cp) -> (B,T,Cout).

    Always returns (B,T,Cout): the reference's trailing ``squeeze()`` (transformer.py:273),
    which drops size-1 axes, is deliberately not reproduced.
    """
    d = sd["input_emb.weight"].shape[0]
    inter = {}
    time = sinusoidal_pos_emb(t, d)[:, None, :]
    tok = linear(x, sd, "input_emb")
    parts = [time]
    if map_cond:
        parts.append(linear(maps_info, sd, "cond_obs_emb"))
    parts.append(linear(transition_info, sd, "prior_z_emb"))
    parts.append(linear(trajectory_prior, sd, "prior_z_emb"))
    c = torch.cat(parts, dim=1)
    S = c.shape[1]
    c = c + sd["cond_pos_emb"][:, :S]
    n_enc = len({k.split(".")[2] for k in sd if k.startswith("encoder.layers.")})
    for i in range(n_enc):
        p = f"encoder.layers.{i}"
        h = _ln(c, sd, p + ".norm1")
        c = c + mha(h, h, sd, p + ".self_attn", n_head)
        h = _ln(c, sd, p + ".norm2")
        c = c + linear(F.gelu(linear(h, sd, p + ".linear1")), sd, p + ".linear2")
        inter[f"enc{i}"] = c
    memory = c
    T = tok.shape[1]
    y = tok + sd["pos_emb"][:, :T]
    causal, mem_allowed = variant_a_masks(T, S)
    n_dec = len({k.split(".")[2] for k in sd if k.startswith("decoder.layers.")})
    for i in range(n_dec):
        p = f"decoder.layers.{i}"
        h = _ln(y, sd, p + ".norm1")
        y = y + mha(h, h, sd, p + ".self_attn", n_head, causal)
        h = _ln(y, sd, p + ".norm2")
        y = y + mha(h, memory, sd, p + ".multihead_attn", n_head, mem_allowed)
        h = _ln(y, sd, p + ".norm3")
        y = y + linear(F.gelu(linear(h, sd, p + ".linear1")), sd, p + ".linear2")
        inter[f"dec{i}"] = y
    out = linear(_ln(y, sd, "ln_f"), sd, "head")
    if return_inter:
        inter["memory"] = memory
        return out, inter
    return out


# ------------------------------------------------------------------ variant B (DiT / MM-DiT)
def res_cat(x, sd: SD, prefix: str):
    """DecoderResCat (libs/mlp.py:43-55): fc(cat[x, relu(LN(Linear(x)))]), LN eps 1e-5."""
    h = linear(x, sd, prefix + ".mlp.linear")
    h = F.relu(_ln(h, sd, prefix + ".mlp.layer_norm", 1e-5))
    return linear(torch.cat([x, h], dim=-1), sd, prefix + ".fc")


def _count(sd, prefix):
    return len({k[len(prefix):].split(".")[0] for k in sd if k.startswith(prefix)})


def time_encoder(t, sd: SD, d: int):
    h = sinusoidal_pos_emb(t, d)
    h = F.leaky_relu(linear(h, sd, "time_encoder.1"), 0.01)
    h = F.leaky_relu(linear(h, sd, "time_encoder.3"), 0.01)
    return linear(h, sd, "time_encoder.5")


def gelu_tanh(x):
    return F.gelu(x, approximate="tanh")


def post_norm_encoder_layer(z, sd: SD, prefix: str, n_head: int):
    """torch.nn.TransformerEncoderLayer defaults: post-norm, ReLU, eps 1e-5."""
    z = _ln(z + mha(z, z, sd, prefix + ".self_attn", n_head), sd, prefix + ".norm1")
    ff = linear(F.relu(linear(z, sd, prefix + ".linear1")), sd, prefix + ".linear2")
    return _ln(z + ff, sd, prefix + ".norm2")


def dit_block(x, c, sd: SD, prefix: str, n_head: int):
    mod = linear(F.silu(c), sd, prefix + ".adaLN_modulation.1")
    sh_a, sc_a, g_a, sh_m, sc_m, g_m = mod.chunk(6, dim=2)
    h = layer_norm(x, eps=1e-6) * (1 + sc_a) + sh_a
    x = x + g_a * post_norm_encoder_layer(h, sd, prefix + ".attn", n_head)
    h = layer_norm(x, eps=1e-6) * (1 + sc_m) + sh_m
    h = gelu_tanh(linear(gelu_tanh(linear(h, sd, prefix + ".mlp.0")), sd, prefix + ".mlp.2"))
    return x + g_m * h


def mm_dit_block(x, c, y, sd: SD, prefix: str, n_head: int):
    d = x.shape[-1]
    mx = linear(F.silu(y), sd, prefix + ".adaLN_modulation_x.1").chunk(6, dim=2)
    mc = linear(F.silu(y), sd, prefix + ".adaLN_modulation_c.1").chunk(6, dim=2)
    xh = layer_norm(x, eps=1e-6) * (1 + mx[1]) + mx[0]
    ch = layer_norm(c, eps=1e-6) * (1 + mc[1]) + mc[0]
    a = prefix + ".attn"
    B, S, _ = x.shape

    def heads(v):  # (B,S,H*k) -> (B,S,H,k)
        return v.reshape(B, S, n_head, -1)

    q = torch.cat([heads(F.linear(xh, sd[a + ".Wq_x.weight"])), heads(F.linear(ch, sd[a + ".Wq_c.weight"]))], dim=-1)
    k = torch.cat([heads(F.linear(xh, sd[a + ".Wk_x.weight"])), heads(F.linear(ch, sd[a + ".Wk_c.weight"]))], dim=-1)
    v = torch.cat([heads(F.linear(xh, sd[a + ".Wv_x.weight"])), heads(F.linear(ch, sd[a + ".Wv_c.weight"]))], dim=-1)
    k_dim = sd[a + ".Wq_x.weight"].shape[0] // n_head
    o = attention(q.reshape(B, S, -1), k.reshape(B, S, -1), v.reshape(B, S, -1), n_head, None, 1.0 / math.sqrt(k_dim))
    o = linear(o, sd, a + ".proj")
    x_attn = x + mx[2] * gelu_tanh(linear(o, sd, prefix + ".linear_x.0"))
    hx = layer_norm(x_attn, eps=1e-6) * (1 + mx[4]) + mx[3]
    hx = gelu_tanh(linear(gelu_tanh(linear(hx, sd, prefix + ".mlp_x.0")), sd, prefix + ".mlp_x.2"))
    x_out = x + mx[5] * hx            # residual from x, not x_attn (diffusionTransformer.py:173-174)
    c_attn = c + mc[2] * gelu_tanh(linear(o, sd, prefix + ".linear_c.0"))
    hc = layer_norm(c_attn, eps=1e-6) * (1 + mc[4]) + mc[3]
    hc = gelu_tanh(linear(gelu_tanh(linear(hc, sd, prefix + ".mlp_c.0")), sd, prefix + ".mlp_c.2"))
    c_out = c + mc[5] * hc
    return x_out, c_out


def dit_forward(sd: SD, x, t, state, n_head: int, backbone: str = "DiT", return_inter: bool = False):
    """x (B,80,4), t (B,), state (B,80,n_embd) -> (B,80,4)."""
    d = sd["position_embedding"].shape[-1]
    inter = {}
    s = state
    for i in range(_count(sd, "state_encoder.")):
        s = res_cat(s, sd, f"state_encoder.{i}")
    xe = x
    for i in range(_count(sd, "x_encoder.")):
        xe = res_cat(xe, sd, f"x_encoder.{i}")
    te = time_encoder(t, sd, d)[:, None, :]
    inter.update(state_emb=s, x_emb=xe, t_emb=te[:, 0])
    n_blk = _count(sd, "blocks.")
    h = xe + sd["position_embedding"]
    if backbone == "DiT":
        c = te + s
        for i in range(n_blk):
            h = dit_block(h, c, sd, f"blocks.{i}", n_head)
            inter[f"blk{i}_x"] = h
        y = c
    elif backbone == "MM-DiT":
        y = te + s
        c = s
        for i in range(n_blk):
            h, c = mm_dit_block(h, c, y, sd, f"blocks.{i}", n_head)
            inter[f"blk{i}_x"], inter[f"blk{i}_c"] = h, c
    else:
        raise ValueError(backbone)
    sh, sc = linear(F.silu(y), sd, "final_layer.adaLN_modulation.1").chunk(2, dim=2)
    out = linear(layer_norm(h, eps=1e-6) * (1 + sc) + sh, sd, "final_layer.linear")
    return (out, inter) if return_inter else out
