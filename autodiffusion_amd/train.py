"""Training path of the refiner: the denoiser's forward with its activations kept, and the backward that turns the loss
gradient into parameter gradients -- what ``loss.backward()`` does in the reference, where the loss of
``DiffusionWrapper.p_losses`` (models/diffusion_loss/TrajRefiner.py:417-452) is differentiated by torch autograd through
``Transformer.forward`` (models/diffusion_loss/transformer.py:189-273).

Every arithmetic step is a libadx kernel (fp32): the forward operators of ``ops`` and the backward operators declared in
``include/adx.h`` (``adx_op_linear_wgrad``, ``adx_op_layernorm_bwd``, ``adx_op_attention_bwd``, ``adx_op_gelu_bwd``,
``adx_op_colsum``, ``adx_op_weighted_mse_bwd`` ...).  torch is used for memory only (concatenation, slicing, broadcasting
copies, zero-filled gradient buffers) and for the autograd bookkeeping: ``DenoiserA`` / ``WeightedMSE`` are
``torch.autograd.Function``s, so ``loss.backward()`` and any torch optimizer work on the wrapper classes unchanged.

Dropout.  The reference trains with ``nn.Dropout`` active (``DiffusionAugmentation.py:109`` puts the refiner in train mode):
``p_drop_emb`` on the two embedded sequences (transformer.py:52,248,258) and torch's TransformerEncoder / DecoderLayer dropout
(``p_drop_attn``; torch's default 0.1 inside every DiTBlock) on attention probabilities, attention outputs, the FFN hidden layer
and the FFN output.  A module in training mode takes the same layers here (``DropoutPlan``): masks are a pure function of
(seed, layer number, position) computed on the device by an integer hash, regenerated -- not stored -- by the backward pass;
a test can instead supply the masks a reference run drew.  In ``.eval()`` nothing changes (bit for bit).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib, ops
from ._lib import check, current_stream, f32c, ptr


# ------------------------------------------------------------------------------------------------ operator wrappers
def _scratch(M: int, N: int, K: int, device) -> torch.Tensor:
    n = int(_lib.load().adx_reduce_ws_bytes(int(M), int(N), int(K)))
    return torch.empty(max(n, 16), dtype=torch.uint8, device=device)


def add(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    a, b = f32c(a), f32c(b)
    if a.shape != b.shape:
        raise ValueError("add: shapes differ")
    out = torch.empty_like(a)
    check(_lib.load().adx_op_add(ptr(a), ptr(b), ptr(out), a.numel(), current_stream()), "add")
    return out


def transpose(w: torch.Tensor) -> torch.Tensor:
    w = f32c(w)
    rows, cols = w.shape
    out = torch.empty(cols, rows, device=w.device, dtype=torch.float32)
    check(_lib.load().adx_op_transpose(ptr(w), ptr(out), rows, cols, current_stream()), "transpose")
    return out


def gelu(x: torch.Tensor) -> torch.Tensor:
    x = f32c(x)
    y = torch.empty_like(x)
    check(_lib.load().adx_op_gelu(ptr(x), ptr(y), x.numel(), current_stream()), "gelu")
    return y


def gelu_bwd(x: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    x, dy = f32c(x), f32c(dy)
    dx = torch.empty_like(x)
    check(_lib.load().adx_op_gelu_bwd(ptr(x), ptr(dy), ptr(dx), x.numel(), current_stream()), "gelu_bwd")
    return dx


def linear_wgrad(dy: torch.Tensor, x: torch.Tensor, dw: torch.Tensor, db: Optional[torch.Tensor], accumulate: bool = True) -> None:
    """dw (N, K) (+)= dy^T x, db (N,) (+)= column sums of dy; dy (..., N), x (..., K).  dw / db are written in place."""
    dy, x = f32c(dy), f32c(x)
    N, K = dy.shape[-1], x.shape[-1]
    M = dy.numel() // N
    if x.numel() // K != M or tuple(dw.shape) != (N, K) or (db is not None and db.numel() != N):
        raise ValueError("linear_wgrad: operand shapes do not match")
    ws = _scratch(M, N, K, dy.device)
    check(_lib.load().adx_op_linear_wgrad(ptr(dy), ptr(x), M, N, K, ptr(dw), ptr(db), int(accumulate), ptr(ws), ws.numel(), current_stream()),
          "linear_wgrad")


def colsum(x: torch.Tensor, out: torch.Tensor, accumulate: bool = True) -> None:
    """out (N,) (+)= column sums of x viewed as (M, N) with N = out.numel()."""
    x = f32c(x)
    N = out.numel()
    M = x.numel() // N
    if M * N != x.numel():
        raise ValueError("colsum: operand shapes do not match")
    ws = _scratch(M, N, 0, x.device)
    check(_lib.load().adx_op_colsum(ptr(x), M, N, ptr(out), int(accumulate), ptr(ws), ws.numel(), current_stream()), "colsum")


def layernorm_bwd(x: torch.Tensor, gamma: Optional[torch.Tensor], dy: torch.Tensor, eps: float = 1e-5, want_dy_xhat: bool = True):
    """-> (dx, dy * xhat or None) for y = LayerNorm(x) * gamma + beta over the last axis."""
    x, dy = f32c(x), f32c(dy)
    d = x.shape[-1]
    rows = x.numel() // d
    dx = torch.empty_like(x)
    dyx = torch.empty_like(x) if want_dy_xhat else None
    check(_lib.load().adx_op_layernorm_bwd(ptr(x), ptr(f32c(gamma)) if gamma is not None else None, ptr(dy), rows, d, float(eps), ptr(dx),
                                           ptr(dyx), current_stream()), "layernorm_bwd")
    return dx, dyx


class DropSite:
    """One dropout layer of one forward pass: rate and either (seed, site) of the device hash or the multipliers themselves."""
    __slots__ = ("p", "seed", "site", "mask")

    def __init__(self, p: float, seed: int, site: int, mask: Optional[torch.Tensor]):
        self.p, self.seed, self.site, self.mask = float(p), int(seed) & 0xFFFFFFFFFFFFFFFF, int(site), mask


class DropoutPlan:
    """The dropout layers of one forward pass, in call order.  ``masks``: a queue of multiplier tensors (0 or 1 / (1 - p)), one per
    layer in the order the reference consumes its generator -- embeddings of the conditioning sequence, per encoder layer
    [attention probabilities, attention output, FFN hidden, FFN output], embeddings of the trajectory tokens, per decoder layer
    [self-attention probabilities, its output, cross-attention probabilities, its output, FFN hidden, FFN output]; otherwise the
    device hash keyed by ``seed`` (None: a fresh one from torch's generator) and the layer number.  ``masks`` may also be a callable
    ``(layer number, p, shape) -> tensor``.  Attention layers take (B, H, Sq, Skv) multipliers, the others the shape of their input."""

    def __init__(self, p_emb: float, p_layer: float, seed: Optional[int] = None, masks=None):
        self.p_emb, self.p_layer = float(p_emb), float(p_layer)
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
        self.seed = seed
        self.masks = masks if (masks is None or callable(masks)) else list(masks)
        self.site = 0

    def next(self, p: float, shape) -> Optional[DropSite]:
        if p <= 0.0:
            return None
        mask = None
        if self.masks is not None:
            if callable(self.masks):
                mask = f32c(self.masks(self.site, float(p), tuple(shape)))
            elif not self.masks:
                raise ValueError("DropoutPlan: fewer masks than dropout layers")
            else:
                mask = f32c(self.masks.pop(0))
            if tuple(mask.shape) != tuple(shape):
                raise ValueError(f"DropoutPlan: mask of shape {tuple(mask.shape)} for a layer of shape {tuple(shape)}")
        site = DropSite(p, self.seed, self.site, mask)
        self.site += 1
        return site


def dropout(x: torch.Tensor, site: Optional[DropSite]) -> torch.Tensor:
    """x * multipliers of one dropout layer; applied to a gradient with the same ``site`` it is that layer's backward."""
    if site is None:
        return x
    x = f32c(x)
    if site.mask is not None:
        return mul(x, site.mask)
    y = torch.empty_like(x)
    check(_lib.load().adx_op_dropout(ptr(x), ptr(y), x.numel(), site.p, site.seed, site.site, current_stream()), "dropout")
    return y


def dropout_mask(shape, site: DropSite, device) -> torch.Tensor:
    """The multipliers the device hash gives for an elementwise layer of ``shape``."""
    y = torch.empty(*shape, device=device, dtype=torch.float32)
    check(_lib.load().adx_op_dropout(None, ptr(y), y.numel(), site.p, site.seed, site.site, current_stream()), "dropout")
    return y


def attention_dropout_mask(B: int, H: int, Sq: int, Skv: int, site: DropSite, device) -> torch.Tensor:
    """The multipliers the device hash gives for the attention probabilities of one layer, (B, H, Sq, Skv)."""
    y = torch.empty(B, H, Sq, Skv, device=device, dtype=torch.float32)
    check(_lib.load().adx_op_attention_dropout_mask(ptr(y), B, H, Sq, Skv, site.p, site.seed, site.site, current_stream()), "attention_dropout_mask")
    return y


def attention_stats(q, k, v, n_head: int, scale: Optional[float] = None, mask_off: int = -1, drop: Optional[DropSite] = None):
    """``ops.attention`` for a forward whose backward follows -> (o, stats); stats carries each query's log-sum-exp.  ``drop``:
    dropout on the attention probabilities (after the softmax)."""
    q, k, v = f32c(q), f32c(k), f32c(v)
    B, Sq, Skv = q.shape[0], q.shape[1], k.shape[1]
    dh = q.shape[2] // n_head
    if v.shape[2] != q.shape[2]:
        raise ValueError("attention_stats: value width must equal the key width")
    if scale is None:
        scale = dh ** -0.5
    o = torch.empty_like(q)
    stats = torch.empty(B, n_head, Sq, 2, device=q.device, dtype=torch.float32)
    if drop is None:
        check(_lib.load().adx_op_attention_stats(ptr(q), ptr(k), ptr(v), ptr(o), ptr(stats), B, n_head, Sq, Skv, dh, float(scale), int(mask_off),
                                                 current_stream()), "attention_stats")
    else:
        if drop.mask is not None and tuple(drop.mask.shape) != (B, n_head, Sq, Skv):
            raise ValueError("attention_stats: dropout mask must be (B, H, Sq, Skv)")
        check(_lib.load().adx_op_attention_stats_dropout(ptr(q), ptr(k), ptr(v), ptr(o), ptr(stats), B, n_head, Sq, Skv, dh, float(scale), int(mask_off),
                                                         drop.p, drop.seed, drop.site, ptr(drop.mask), current_stream()), "attention_stats_dropout")
    return o, stats


def attention_bwd(q, k, v, o, dout, n_head: int, scale: Optional[float] = None, mask_off: int = -1, stats: Optional[torch.Tensor] = None,
                  drop: Optional[DropSite] = None):
    """Backward of ``ops.attention`` / ``attention_stats`` (same layouts, same mask rule) -> (dq, dk, dv).  ``stats`` from
    ``attention_stats`` saves the pass that recomputes the log-sum-exps; ``drop`` must be the forward's."""
    q, k, v, o, dout = f32c(q), f32c(k), f32c(v), f32c(o), f32c(dout)
    B, Sq, Skv = q.shape[0], q.shape[1], k.shape[1]
    dh = q.shape[2] // n_head
    if v.shape[2] != q.shape[2] or o.shape != q.shape or dout.shape != q.shape:
        raise ValueError("attention_bwd: value width must equal the key width")
    if scale is None:
        scale = dh ** -0.5
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    have = stats is not None
    if have and tuple(stats.shape) != (B, n_head, Sq, 2):
        raise ValueError("attention_bwd: stats of the wrong shape")
    st = stats if have else torch.empty(B, n_head, Sq, 2, device=q.device, dtype=torch.float32)
    if drop is None:
        check(_lib.load().adx_op_attention_bwd(ptr(q), ptr(k), ptr(v), ptr(o), ptr(dout), B, n_head, Sq, Skv, dh, float(scale), int(mask_off),
                                               ptr(dq), ptr(dk), ptr(dv), ptr(st), int(have), current_stream()), "attention_bwd")
    else:
        check(_lib.load().adx_op_attention_bwd_dropout(ptr(q), ptr(k), ptr(v), ptr(o), ptr(dout), B, n_head, Sq, Skv, dh, float(scale), int(mask_off),
                                                       ptr(dq), ptr(dk), ptr(dv), ptr(st), int(have), drop.p, drop.seed, drop.site, ptr(drop.mask),
                                                       current_stream()), "attention_bwd_dropout")
    return dq, dk, dv


def weighted_mse_bwd(out, target, t, weight, upstream: Optional[torch.Tensor]) -> torch.Tensor:
    out, target = f32c(out), f32c(target)
    t = t.to(device=out.device, dtype=torch.int64).contiguous()
    w = f32c(weight).to(out.device)
    up = f32c(upstream).reshape(1) if upstream is not None else None
    n_rows = out.shape[0]
    dout = torch.empty_like(out)
    check(_lib.load().adx_op_weighted_mse_bwd(ptr(out), ptr(target), ptr(t), ptr(w), ptr(up), ptr(dout), n_rows, out.numel() // n_rows,
                                              current_stream()), "weighted_mse_bwd")
    return dout


# ------------------------------------------------------------------------------------------------ the network, forward and backward
class _Grads:
    """Gradient buffers by parameter name, zero-filled on first use; the kernels accumulate into them."""

    def __init__(self, params: Dict[str, torch.Tensor]):
        self.params = params
        self.g: Dict[str, torch.Tensor] = {}
        total = sum(p.numel() for p in params.values())
        first = next(iter(params.values()))
        self.flat = torch.zeros(total, device=first.device, dtype=torch.float32)     # one fill for every gradient
        self.at = 0

    def of(self, name: str) -> torch.Tensor:
        if name not in self.g:
            p = self.params[name]
            self.g[name] = self.flat[self.at:self.at + p.numel()].view(p.shape)
            self.at += p.numel()
        return self.g[name]


def _lin(P, x, prefix: str, rows: Optional[slice] = None):
    w, b = P[prefix + "weight"], P.get(prefix + "bias")
    if rows is not None:
        w, b = w[rows], b[rows]
    return ops.linear(x, f32c(w), f32c(b) if b is not None else None)


def _lin_bwd(P, G: _Grads, dy, x, prefix: str, rows: Optional[slice] = None, need_dx: bool = True):
    """Accumulates dW / db of ``y = x W[rows]^T + b[rows]`` and returns dx = dy W[rows] (None when not needed)."""
    gw = G.of(prefix + "weight")
    gb = G.of(prefix + "bias") if prefix + "bias" in P else None
    w = P[prefix + "weight"]
    if rows is not None:
        gw, gb, w = gw[rows], gb[rows], w[rows]
    linear_wgrad(dy, x, gw, gb, accumulate=True)
    if not need_dx:
        return None
    return ops.linear(dy, transpose(w))


def _ln_bwd(P, G: _Grads, x, dy, prefix: str):
    dx, dyx = layernorm_bwd(x, P[prefix + "weight"], dy, 1e-5)
    colsum(dyx, G.of(prefix + "weight"))
    colsum(dy, G.of(prefix + "bias"))
    return dx


def _site(dp: Optional[DropoutPlan], p: float, shape) -> Optional[DropSite]:
    return dp.next(p, shape) if dp is not None else None


def _mha(P, prefix: str, xq, xkv, n_head: int, mask_off: int, dp: Optional[DropoutPlan] = None):
    """-> (dropout(out_proj(attention)), saved).  With a plan: dropout on the probabilities and on the block's output (the
    ``dropout1`` / ``dropout2`` that follows the attention in torch's layers)."""
    d = xq.shape[-1]
    q = _lin(P, xq, prefix + "in_proj_", slice(0, d))
    k = _lin(P, xkv, prefix + "in_proj_", slice(d, 2 * d))
    v = _lin(P, xkv, prefix + "in_proj_", slice(2 * d, 3 * d))
    p_layer = dp.p_layer if dp is not None else 0.0
    sa = _site(dp, p_layer, (q.shape[0], n_head, q.shape[1], k.shape[1]))
    o, stats = attention_stats(q, k, v, n_head, mask_off=mask_off, drop=sa)
    out = _lin(P, o, prefix + "out_proj.")
    so = _site(dp, p_layer, out.shape)
    return dropout(out, so), (q, k, v, o, stats, sa, so)


def _mha_bwd(P, G, prefix: str, xq, xkv, saved, da, n_head: int, mask_off: int):
    """-> (d xq, d xkv) of ``dropout(out_proj(attention(q(xq), k(xkv), v(xkv))))``."""
    q, k, v, o, stats, sa, so = saved
    d = xq.shape[-1]
    do = _lin_bwd(P, G, dropout(da, so), o, prefix + "out_proj.")
    dq, dk, dv = attention_bwd(q, k, v, o, do, n_head, mask_off=mask_off, stats=stats, drop=sa)
    dxq = _lin_bwd(P, G, dq, xq, prefix + "in_proj_", slice(0, d))
    dxkv = add(_lin_bwd(P, G, dk, xkv, prefix + "in_proj_", slice(d, 2 * d)), _lin_bwd(P, G, dv, xkv, prefix + "in_proj_", slice(2 * d, 3 * d)))
    return dxq, dxkv


def _ffn(P, prefix: str, h, dp: Optional[DropoutPlan] = None):
    """dropout(linear2(dropout(gelu(linear1(h))))) -- torch's ``_ff_block``."""
    pre = _lin(P, h, prefix + "linear1.")
    p_layer = dp.p_layer if dp is not None else 0.0
    sh = _site(dp, p_layer, pre.shape)
    act = dropout(gelu(pre), sh)
    out = _lin(P, act, prefix + "linear2.")
    so = _site(dp, p_layer, out.shape)
    return dropout(out, so), (pre, act, sh, so)


def _ffn_bwd(P, G, prefix: str, h, saved, df):
    pre, act, sh, so = saved
    dact = dropout(_lin_bwd(P, G, dropout(df, so), act, prefix + "linear2."), sh)
    return _lin_bwd(P, G, gelu_bwd(pre, dact), h, prefix + "linear1.")


def _norm(P, prefix: str, x):
    return ops.layernorm(x, P[prefix + "weight"], P[prefix + "bias"], eps=1e-5)


def denoiser_forward(P: Dict[str, torch.Tensor], x, t, maps, transition, prior, n_head: int, n_enc: int, n_dec: int, maps_repeat: int = 1,
                     dp: Optional[DropoutPlan] = None):
    """transformer.py:189-273 with every intermediate a backward needs kept.  -> (out (B, T, C), tape).
    ``maps_repeat`` = r: ``maps`` holds B / r scenarios and batch row i uses scenario i % (B / r) -- the chunks of one trajectory
    share their map tokens, so the map embedding (the widest product of the network) is computed once and tiled."""
    d = P["input_emb.weight"].shape[0]
    B, T = x.shape[0], x.shape[1]
    tape: Dict[str, object] = {"x": x, "maps": maps, "transition": transition, "prior": prior, "maps_repeat": maps_repeat}
    parts = [ops.timestep_embedding(t, d)[:, None, :]]
    if maps is not None:
        if maps.shape[0] * maps_repeat != B:
            raise ValueError("maps_info must hold batch / maps_repeat scenarios")
        me = _lin(P, maps, "cond_obs_emb.")
        parts.append(me if maps_repeat == 1 else me.repeat(maps_repeat, 1, 1))
    parts += [_lin(P, transition, "prior_z_emb."), _lin(P, prior, "prior_z_emb.")]
    c = torch.cat(parts, dim=1)
    S = c.shape[1]
    c = add(c, P["cond_pos_emb"][:, :S].expand(B, S, d).contiguous())
    s_cond = _site(dp, dp.p_emb if dp is not None else 0.0, c.shape)          # self.drop on the conditioning sequence (transformer.py:248)
    c = dropout(c, s_cond)
    enc = []
    for i in range(n_enc):
        p = f"encoder.layers.{i}."
        h1 = _norm(P, p + "norm1.", c)
        a, sa = _mha(P, p + "self_attn.", h1, h1, n_head, -1, dp)
        c1 = add(c, a)
        h2 = _norm(P, p + "norm2.", c1)
        f, sf = _ffn(P, p, h2, dp)
        enc.append((c, h1, sa, c1, h2, sf))
        c = add(c1, f)
    memory = c
    y = add(_lin(P, x, "input_emb."), P["pos_emb"][:, :T].expand(B, T, d).contiguous())
    s_tok = _site(dp, dp.p_emb if dp is not None else 0.0, y.shape)           # self.drop on the trajectory tokens (transformer.py:258)
    y = dropout(y, s_tok)
    dec = []
    for i in range(n_dec):
        p = f"decoder.layers.{i}."
        h1 = _norm(P, p + "norm1.", y)
        a, sa = _mha(P, p + "self_attn.", h1, h1, n_head, 0, dp)
        y1 = add(y, a)
        h2 = _norm(P, p + "norm2.", y1)
        m, sm = _mha(P, p + "multihead_attn.", h2, memory, n_head, 1, dp)
        y2 = add(y1, m)
        h3 = _norm(P, p + "norm3.", y2)
        f, sf = _ffn(P, p, h3, dp)
        dec.append((y, h1, sa, y1, h2, sm, y2, h3, sf))
        y = add(y2, f)
    hn = _norm(P, "ln_f.", y)
    out = _lin(P, hn, "head.")
    tape.update(enc=enc, dec=dec, memory=memory, y=y, hn=hn, S=S, s_cond=s_cond, s_tok=s_tok)
    return out, tape


def denoiser_backward(P: Dict[str, torch.Tensor], tape, dout, n_head: int, want_transition_grad: bool = False):
    """-> ({parameter name: gradient}, d transition or None)."""
    G = _Grads(P)
    x, maps, transition, prior = tape["x"], tape["maps"], tape["transition"], tape["prior"]
    B, T = x.shape[0], x.shape[1]
    d = P["input_emb.weight"].shape[0]
    S = tape["S"]
    dy = _ln_bwd(P, G, tape["y"], _lin_bwd(P, G, f32c(dout), tape["hn"], "head."), "ln_f.")
    dmem = None
    for i in reversed(range(len(tape["dec"]))):
        p = f"decoder.layers.{i}."
        y, h1, sa, y1, h2, sm, y2, h3, sf = tape["dec"][i]
        dy2 = add(dy, _ln_bwd(P, G, y2, _ffn_bwd(P, G, p, h3, sf, dy), p + "norm3."))
        dh2, dm = _mha_bwd(P, G, p + "multihead_attn.", h2, tape["memory"], sm, dy2, n_head, 1)
        dmem = dm if dmem is None else add(dmem, dm)
        dy1 = add(dy2, _ln_bwd(P, G, y1, dh2, p + "norm2."))
        dq, dkv = _mha_bwd(P, G, p + "self_attn.", h1, h1, sa, dy1, n_head, 0)
        dy = add(dy1, _ln_bwd(P, G, y, add(dq, dkv), p + "norm1."))
    dy = dropout(dy, tape.get("s_tok"))
    colsum(dy, _slice_rows(G.of("pos_emb"), T))
    _lin_bwd(P, G, dy, x, "input_emb.", need_dx=False)
    dc = dmem
    for i in reversed(range(len(tape["enc"]))):
        p = f"encoder.layers.{i}."
        c, h1, sa, c1, h2, sf = tape["enc"][i]
        dc1 = add(dc, _ln_bwd(P, G, c1, _ffn_bwd(P, G, p, h2, sf, dc), p + "norm2."))
        dq, dkv = _mha_bwd(P, G, p + "self_attn.", h1, h1, sa, dc1, n_head, -1)
        dc = add(dc1, _ln_bwd(P, G, c, add(dq, dkv), p + "norm1."))
    dc = dropout(dc, tape.get("s_cond"))
    colsum(dc, _slice_rows(G.of("cond_pos_emb"), S))
    at = 1
    if maps is not None:
        n_maps, rep = maps.shape[1], tape["maps_repeat"]
        dme = dc[:, at:at + n_maps].contiguous()
        if rep > 1:                                                 # the tiled embedding's gradient: sum over the repeats
            folded = torch.zeros(maps.shape[0], n_maps, d, device=dme.device, dtype=torch.float32)
            colsum(dme, folded.reshape(-1))
            dme = folded
        _lin_bwd(P, G, dme, maps, "cond_obs_emb.", need_dx=False)
        at += n_maps
    n_tr = transition.shape[1]
    d_tr = _lin_bwd(P, G, dc[:, at:at + n_tr].contiguous(), transition, "prior_z_emb.", need_dx=want_transition_grad)
    at += n_tr
    _lin_bwd(P, G, dc[:, at:at + prior.shape[1]].contiguous(), prior, "prior_z_emb.", need_dx=False)
    return G.g, d_tr


def _slice_rows(g: torch.Tensor, n: int) -> torch.Tensor:
    """The first n rows of a (1, L, d) table as a flat view (contiguous: the table has one leading row)."""
    return g[0, :n].reshape(-1)


# ------------------------------------------------------------------------------------------------ autograd glue
class DenoiserA(torch.autograd.Function):
    """``Transformer.forward`` as a differentiable function of its parameters (and of ``transition_info``)."""

    @staticmethod
    def forward(ctx, meta, x, t, maps, transition, prior, *params):
        names, n_head, n_enc, n_dec, maps_repeat, dp = meta
        P = {n: p.detach() for n, p in zip(names, params)}
        out, tape = denoiser_forward(P, f32c(x), t, f32c(maps) if maps is not None else None, f32c(transition), f32c(prior), n_head, n_enc, n_dec,
                                     maps_repeat, dp)
        ctx.meta, ctx.P, ctx.tape = meta, P, tape
        for i, what in ((1, "x"), (3, "maps_info"), (5, "trajectory_prior")):
            if ctx.needs_input_grad[i]:       # the reference's autograd would propagate; this path does not produce these gradients
                raise NotImplementedError(f"the training path computes no gradient for the input `{what}` (it requires grad): detach it, "
                                          "or differentiate the producer of that tensor separately")
        ctx.want_tr = bool(ctx.needs_input_grad[4])
        return out

    @staticmethod
    def backward(ctx, dout):
        names, n_head = ctx.meta[0], ctx.meta[1]
        if ctx.tape is None:
            raise RuntimeError("the kept activations were released by the first backward: call forward again (retain_graph is not supported)")
        grads, d_tr = denoiser_backward(ctx.P, ctx.tape, dout.contiguous(), n_head, ctx.want_tr)
        ctx.tape = None
        return (None, None, None, None, d_tr, None) + tuple(grads.get(n) for n in names)


class WeightedMSE(torch.autograd.Function):
    """mean_b( weight[t_b] * mean_e (out - target)^2 ): the reduction of p_losses (TrajRefiner.py:448-452)."""

    @staticmethod
    def forward(ctx, out, target, t, weight):
        rows, mean = ops.weighted_mse(out, target, t, weight)
        ctx.save_for_backward(out, target, t, weight)
        return mean

    @staticmethod
    def backward(ctx, dmean):
        out, target, t, weight = ctx.saved_tensors
        return weighted_mse_bwd(out, target, t, weight, dmean), None, None, None


def trainable_names(module: torch.nn.Module) -> List[str]:
    return [n for n, _ in module.named_parameters()]


def set_dropout(module: torch.nn.Module, seed: Optional[int] = None, masks=None) -> None:
    """Fix the dropout draws of the NEXT differentiable call of ``module`` in training mode: the seed of the device hash, or the
    multiplier tensors themselves in layer order (see ``DropoutPlan``).  Without it every call takes a fresh seed from torch's
    generator (so ``torch.manual_seed`` makes a run repeatable)."""
    module.__dict__["_dropout_next"] = (seed, masks if (masks is None or callable(masks)) else list(masks))


def dropout_plan(module: torch.nn.Module, p_emb: float, p_layer: float) -> Optional[DropoutPlan]:
    """The plan of one forward pass: None in eval mode or with both rates zero (nothing changes then, bit for bit)."""
    fixed = module.__dict__.pop("_dropout_next", None)
    if not module.training or (p_emb <= 0.0 and p_layer <= 0.0):
        return None
    seed, masks = fixed if fixed is not None else (None, None)
    return DropoutPlan(p_emb, p_layer, seed, masks)


def denoise_with_grad(module: torch.nn.Module, x, t, condition: Dict[str, torch.Tensor], n_head: int, maps_repeat: int = 1) -> torch.Tensor:
    """Differentiable call of a ``refiner.Transformer``: the result carries a grad_fn whose backward fills the module's ``.grad``s.
    In training mode the module's dropout layers are active (``p_drop_emb``, ``p_drop_attn``)."""
    names = trainable_names(module)
    params = [p for _, p in module.named_parameters()]
    n_enc, n_dec = len(module.encoder.layers), len(module.decoder.layers)
    maps = condition.get("maps_info") if module.map_cond else None
    dp = dropout_plan(module, float(getattr(module, "p_drop_emb", 0.0)), float(getattr(module, "p_drop_attn", 0.0)))
    return DenoiserA.apply((tuple(names), n_head, n_enc, n_dec, int(maps_repeat), dp), x, t, maps, condition["transition_info"],
                           condition["trajectory_prior"], *params)


def wants_grad(module: torch.nn.Module, *inputs) -> bool:
    """Take the differentiable path?  Gradients enabled and something to train: a parameter of ``module``, or one of ``inputs``
    (tensors that enter the denoiser differentiably -- the learnable initial transition, a backbone state -- and may be the only
    trainable thing when the denoiser itself is frozen)."""
    if not torch.is_grad_enabled():
        return False
    return any(p.requires_grad for p in module.parameters()) or any(torch.is_tensor(v) and v.requires_grad for v in inputs)


# ================================================================================================ decoder twin (DiT backbone)
# Transformer_TrajDiffusionModel.forward (diffusion_decoder.py:251-277) with DiTBlock / DiT_FinalLayer
# (libs/diffusionTransformer.py:24-53, 182-199) and DecoderResCat (libs/mlp.py:43-55) under the same scheme: forward with kept
# activations, backward through the library's operators, for both backbones (MM_DiTBlock / DiT_Attention:
# libs/diffusionTransformer.py:56-179).  fp32.
def mul(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    a, b = f32c(a), f32c(b)
    if a.shape != b.shape:
        raise ValueError("mul: shapes differ")
    out = torch.empty_like(a)
    check(_lib.load().adx_op_mul(ptr(a), ptr(b), ptr(out), a.numel(), current_stream()), "mul")
    return out


def act(x: torch.Tensor, code: int) -> torch.Tensor:
    x = f32c(x)
    y = torch.empty_like(x)
    check(_lib.load().adx_op_act(ptr(x), ptr(y), x.numel(), int(code), current_stream()), "act")
    return y


def act_bwd(x: torch.Tensor, dy: torch.Tensor, code: int) -> torch.Tensor:
    x, dy = f32c(x), f32c(dy)
    dx = torch.empty_like(x)
    check(_lib.load().adx_op_act_bwd(ptr(x), ptr(dy), ptr(dx), x.numel(), int(code), current_stream()), "act_bwd")
    return dx


def group_colsum(x: torch.Tensor) -> torch.Tensor:
    """(G, R, N) -> (G, N): sum over the middle axis."""
    x = f32c(x)
    G, R, N = x.shape
    out = torch.empty(G, N, device=x.device, dtype=torch.float32)
    check(_lib.load().adx_op_group_colsum(ptr(x), G, R, N, ptr(out), current_stream()), "group_colsum")
    return out


def _res_cat(P, prefix: str, x):
    h0 = _lin(P, x, prefix + "mlp.linear.")
    h1 = ops.layernorm(h0, P[prefix + "mlp.layer_norm.weight"], P[prefix + "mlp.layer_norm.bias"], eps=1e-5)
    cat = torch.cat([x, act(h1, _lib.ACT_RELU)], dim=-1)
    return _lin(P, cat, prefix + "fc."), (x, h0, h1, cat)


def _res_cat_bwd(P, G, prefix: str, saved, dout, need_dx: bool = True):
    x, h0, h1, cat = saved
    n_in = x.shape[-1]
    dcat = _lin_bwd(P, G, dout, cat, prefix + "fc.")
    dh1 = act_bwd(h1, dcat[..., n_in:].contiguous(), _lib.ACT_RELU)
    dh0 = _ln_bwd(P, G, h0, dh1, prefix + "mlp.layer_norm.")
    dxa = _lin_bwd(P, G, dh0, x, prefix + "mlp.linear.", need_dx=need_dx)
    return add(dcat[..., :n_in].contiguous(), dxa) if need_dx else None


def _modulate(x, shift, scale):
    return ops.layernorm(x, None, None, shift, scale, eps=1e-6)


def _modulate_bwd(x, scale, dy):
    """y = LN0(x) (1 + scale) + shift -> (dx, dshift, dscale)."""
    n = ops.layernorm(x, eps=1e-6)
    dn = add(dy, mul(dy, scale))
    dx, _ = layernorm_bwd(x, None, dn, 1e-6, want_dy_xhat=False)
    return dx, dy, mul(dy, n)


def _post_norm_layer(P, prefix: str, z, n_head: int, dp: Optional[DropoutPlan] = None):
    """torch.nn.TransformerEncoderLayer with its defaults (post-norm, ReLU, dropout 0.1 on the attention probabilities, the
    attention output, the hidden layer and the feed-forward output)."""
    a, sa = _mha(P, prefix + "self_attn.", z, z, n_head, -1, dp)
    r1 = add(z, a)
    z1 = _norm(P, prefix + "norm1.", r1)
    p = _lin(P, z1, prefix + "linear1.")
    p_layer = dp.p_layer if dp is not None else 0.0
    sh = _site(dp, p_layer, p.shape)
    pa = dropout(act(p, _lib.ACT_RELU), sh)
    f = _lin(P, pa, prefix + "linear2.")
    so = _site(dp, p_layer, f.shape)
    r2 = add(z1, dropout(f, so))
    return _norm(P, prefix + "norm2.", r2), (z, sa, r1, z1, p, pa, r2, sh, so)


def _post_norm_layer_bwd(P, G, prefix: str, saved, dz2, n_head: int):
    z, sa, r1, z1, p, pa, r2, sh, so = saved
    dr2 = _ln_bwd(P, G, r2, dz2, prefix + "norm2.")
    dp = act_bwd(p, dropout(_lin_bwd(P, G, dropout(dr2, so), pa, prefix + "linear2."), sh), _lib.ACT_RELU)
    dz1 = add(dr2, _lin_bwd(P, G, dp, z1, prefix + "linear1."))
    dr1 = _ln_bwd(P, G, r1, dz1, prefix + "norm1.")
    dq, dkv = _mha_bwd(P, G, prefix + "self_attn.", z, z, sa, dr1, n_head, -1)
    return add(dr1, add(dq, dkv))


def _dit_block(P, prefix: str, h, silu_c, n_head: int, dp: Optional[DropoutPlan] = None):
    mod = _lin(P, silu_c, prefix + "adaLN_modulation.1.")
    sh_a, sc_a, g_a, sh_m, sc_m, g_m = (m.contiguous() for m in mod.chunk(6, dim=-1))
    h1 = _modulate(h, sh_a, sc_a)
    e, se = _post_norm_layer(P, prefix + "attn.", h1, n_head, dp)
    hA = add(h, mul(g_a, e))
    h2 = _modulate(hA, sh_m, sc_m)
    u = _lin(P, h2, prefix + "mlp.0.")
    ua = act(u, _lib.ACT_GELU_TANH)
    w = _lin(P, ua, prefix + "mlp.2.")
    wa = act(w, _lib.ACT_GELU_TANH)
    return add(hA, mul(g_m, wa)), (h, sc_a, g_a, se, e, hA, sc_m, g_m, h2, u, ua, w, wa)


def _dit_block_bwd(P, G, prefix: str, saved, silu_c, dout, n_head: int):
    h, sc_a, g_a, se, e, hA, sc_m, g_m, h2, u, ua, w, wa = saved
    dg_m = mul(dout, wa)
    dw = act_bwd(w, mul(dout, g_m), _lib.ACT_GELU_TANH)
    du = act_bwd(u, _lin_bwd(P, G, dw, ua, prefix + "mlp.2."), _lib.ACT_GELU_TANH)
    dhA_mod, dsh_m, dsc_m = _modulate_bwd(hA, sc_m, _lin_bwd(P, G, du, h2, prefix + "mlp.0."))
    dhA = add(dout, dhA_mod)
    dg_a = mul(dhA, e)
    dh1 = _post_norm_layer_bwd(P, G, prefix + "attn.", se, mul(dhA, g_a), n_head)
    dh_mod, dsh_a, dsc_a = _modulate_bwd(h, sc_a, dh1)
    dmod = torch.cat([dsh_a, dsc_a, dg_a, dsh_m, dsc_m, dg_m], dim=-1)
    return add(dhA, dh_mod), _lin_bwd(P, G, dmod, silu_c, prefix + "adaLN_modulation.1.")


def _heads_cat(a, b, n_head: int):
    """(B, S, H k), (B, S, H k) -> (B, S, H 2k): per head, the trajectory stream's features followed by the condition stream's."""
    B, S = a.shape[0], a.shape[1]
    return torch.cat([a.reshape(B, S, n_head, -1), b.reshape(B, S, n_head, -1)], dim=-1).reshape(B, S, -1)


def _heads_split(g, n_head: int):
    B, S = g.shape[0], g.shape[1]
    g4 = g.reshape(B, S, n_head, -1)
    k = g4.shape[-1] // 2
    return g4[..., :k].reshape(B, S, -1).contiguous(), g4[..., k:].reshape(B, S, -1).contiguous()


def _mm_stream(P, prefix: str, s: str, inp, m, op):
    """One stream of an MM-DiT block after the joint attention (libs/diffusionTransformer.py:166-179): the residual of both
    updates starts from the block input."""
    u = _lin(P, op, prefix + f"linear_{s}.0.")
    ua = act(u, _lib.ACT_GELU_TANH)
    mid = add(inp, mul(m[2], ua))
    h2 = _modulate(mid, m[3], m[4])
    w1 = _lin(P, h2, prefix + f"mlp_{s}.0.")
    w1a = act(w1, _lib.ACT_GELU_TANH)
    w2 = _lin(P, w1a, prefix + f"mlp_{s}.2.")
    w2a = act(w2, _lib.ACT_GELU_TANH)
    return add(inp, mul(m[5], w2a)), (u, ua, mid, h2, w1, w1a, w2, w2a)


def _mm_stream_bwd(P, G, prefix: str, s: str, m, op, saved, dout):
    """-> (d input through the two residuals, d op, [None, None, dg2, dsh3, dsc4, dg5])."""
    u, ua, mid, h2, w1, w1a, w2, w2a = saved
    dg5 = mul(dout, w2a)
    dw2 = act_bwd(w2, mul(dout, m[5]), _lib.ACT_GELU_TANH)
    dw1 = act_bwd(w1, _lin_bwd(P, G, dw2, w1a, prefix + f"mlp_{s}.2."), _lib.ACT_GELU_TANH)
    dmid, dsh3, dsc4 = _modulate_bwd(mid, m[4], _lin_bwd(P, G, dw1, h2, prefix + f"mlp_{s}.0."))
    dg2 = mul(dmid, ua)
    du = act_bwd(u, mul(dmid, m[2]), _lib.ACT_GELU_TANH)
    dop = _lin_bwd(P, G, du, op, prefix + f"linear_{s}.0.")
    return add(dout, dmid), dop, [None, None, dg2, dsh3, dsc4, dg5]


def _mm_block(P, prefix: str, x, c, silu_y, n_head: int):
    mx = [m.contiguous() for m in _lin(P, silu_y, prefix + "adaLN_modulation_x.1.").chunk(6, dim=-1)]
    mc = [m.contiguous() for m in _lin(P, silu_y, prefix + "adaLN_modulation_c.1.").chunk(6, dim=-1)]
    xh, ch = _modulate(x, mx[0], mx[1]), _modulate(c, mc[0], mc[1])
    a = prefix + "attn."
    proj = {n + s: _lin(P, xh if s == "x" else ch, a + f"W{n}_{s}.") for n in "qkv" for s in "xc"}
    q, k, v = (_heads_cat(proj[n + "x"], proj[n + "c"], n_head) for n in "qkv")
    scale = float(proj["qx"].shape[-1] // n_head) ** -0.5                       # 1 / sqrt(k_dim), k_dim = one stream's head width
    o, stats = attention_stats(q, k, v, n_head, scale=scale)
    op = _lin(P, o, a + "proj.")
    x_out, sx = _mm_stream(P, prefix, "x", x, mx, op)
    c_out, sc = _mm_stream(P, prefix, "c", c, mc, op)
    return x_out, c_out, (x, c, mx, mc, xh, ch, q, k, v, o, stats, scale, op, sx, sc)


def _mm_block_bwd(P, G, prefix: str, saved, silu_y, dx_out, dc_out, n_head: int):
    """``dc_out`` None: the block's condition-stream output is unused (the last block).  Its linear_c / mlp_c and the gate, shift
    and scale rows 2..5 of adaLN_modulation_c then take no part in the loss: their stream backward is skipped and they get zero
    rows (adaLN) or no gradient at all (linear_c, mlp_c: None, as torch's autograd leaves them)."""
    x, c, mx, mc, xh, ch, q, k, v, o, stats, scale, op, sx, sc = saved
    dx, dop_x, gx = _mm_stream_bwd(P, G, prefix, "x", mx, op, sx, dx_out)
    if dc_out is None:
        dc, dop, gc = None, dop_x, [None, None] + [torch.zeros_like(mc[i]) for i in range(2, 6)]
    else:
        dc, dop_c, gc = _mm_stream_bwd(P, G, prefix, "c", mc, op, sc, dc_out)
        dop = add(dop_x, dop_c)
    a = prefix + "attn."
    do = _lin_bwd(P, G, dop, o, a + "proj.")
    dq, dk, dv = attention_bwd(q, k, v, o, do, n_head, scale=scale, stats=stats)
    parts = {"q": _heads_split(dq, n_head), "k": _heads_split(dk, n_head), "v": _heads_split(dv, n_head)}
    dxh = add(add(_lin_bwd(P, G, parts["q"][0], xh, a + "Wq_x."), _lin_bwd(P, G, parts["k"][0], xh, a + "Wk_x.")),
              _lin_bwd(P, G, parts["v"][0], xh, a + "Wv_x."))
    dch = add(add(_lin_bwd(P, G, parts["q"][1], ch, a + "Wq_c."), _lin_bwd(P, G, parts["k"][1], ch, a + "Wk_c.")),
              _lin_bwd(P, G, parts["v"][1], ch, a + "Wv_c."))
    dx_mod, gx[0], gx[1] = _modulate_bwd(x, mx[1], dxh)
    dc_mod, gc[0], gc[1] = _modulate_bwd(c, mc[1], dch)
    dsilu = add(_lin_bwd(P, G, torch.cat(gx, dim=-1), silu_y, prefix + "adaLN_modulation_x.1."),
                _lin_bwd(P, G, torch.cat(gc, dim=-1), silu_y, prefix + "adaLN_modulation_c.1."))
    return add(dx, dx_mod), (dc_mod if dc is None else add(dc, dc_mod)), dsilu


def dit_forward(P: Dict[str, torch.Tensor], x, t, state, n_head: int, n_blocks: int, n_state: int, n_x: int, mm: bool = False,
                dp: Optional[DropoutPlan] = None):
    """``mm``: the MM-DiT backbone -- the condition stream starts as the state embedding and is updated by every block, the
    modulation input y = time embedding + state embedding stays fixed (diffusion_decoder.py:262-272)."""
    d = P["position_embedding"].shape[-1]
    B, S = x.shape[0], x.shape[1]
    tape: Dict[str, object] = {}
    s, st_saved = state, []
    for i in range(n_state):
        s, sv = _res_cat(P, f"state_encoder.{i}.", s)
        st_saved.append(sv)
    xe, x_saved = x, []
    for i in range(n_x):
        xe, sv = _res_cat(P, f"x_encoder.{i}.", xe)
        x_saved.append(sv)
    emb = ops.timestep_embedding(t, d)
    a1 = _lin(P, emb, "time_encoder.1.")
    b1 = act(a1, _lib.ACT_LEAKY)
    a2 = _lin(P, b1, "time_encoder.3.")
    b2 = act(a2, _lib.ACT_LEAKY)
    te = _lin(P, b2, "time_encoder.5.")
    h = add(xe, P["position_embedding"].reshape(1, -1, d)[:, :S].expand(B, S, d).contiguous())
    c = add(te[:, None, :].expand(B, S, d).contiguous(), s)
    silu_c = act(c, _lib.ACT_SILU)
    blocks = []
    cs = s
    for i in range(n_blocks):
        if mm:
            h, cs, sv = _mm_block(P, f"blocks.{i}.", h, cs, silu_c, n_head)
        else:
            h, sv = _dit_block(P, f"blocks.{i}.", h, silu_c, n_head, dp)
        blocks.append(sv)
    mod = _lin(P, silu_c, "final_layer.adaLN_modulation.1.")
    sh, sc = (m.contiguous() for m in mod.chunk(2, dim=-1))
    hf = _modulate(h, sh, sc)
    out = _lin(P, hf, "final_layer.linear.")
    tape.update(st=st_saved, xs=x_saved, time=(emb, a1, b1, a2, b2), c=c, silu_c=silu_c, blocks=blocks, h=h, sc=sc, hf=hf, S=S, mm=mm)
    return out, tape


def dit_backward(P: Dict[str, torch.Tensor], tape, dout, n_head: int, want_state_grad: bool = False):
    G = _Grads(P)
    silu_c = tape["silu_c"]
    dhf = _lin_bwd(P, G, f32c(dout), tape["hf"], "final_layer.linear.")
    dh, dsh, dsc = _modulate_bwd(tape["h"], tape["sc"], dhf)
    dsilu = _lin_bwd(P, G, torch.cat([dsh, dsc], dim=-1), silu_c, "final_layer.adaLN_modulation.1.")
    dcs = None                                                      # MM-DiT: gradient of the condition stream
    for i in reversed(range(len(tape["blocks"]))):
        if tape["mm"]:                                              # dcs None: the last block's condition output is not used
            dh, dcs, ds_i = _mm_block_bwd(P, G, f"blocks.{i}.", tape["blocks"][i], silu_c, dh, dcs, n_head)
        else:
            dh, ds_i = _dit_block_bwd(P, G, f"blocks.{i}.", tape["blocks"][i], silu_c, dh, n_head)
        dsilu = add(dsilu, ds_i)
    dc = act_bwd(tape["c"], dsilu, _lib.ACT_SILU)
    # c = time embedding (broadcast over the tokens) + state embedding
    dte = group_colsum(dc)
    emb, a1, b1, a2, b2 = tape["time"]
    da2 = act_bwd(a2, _lin_bwd(P, G, dte, b2, "time_encoder.5."), _lib.ACT_LEAKY)
    da1 = act_bwd(a1, _lin_bwd(P, G, da2, b1, "time_encoder.3."), _lib.ACT_LEAKY)
    _lin_bwd(P, G, da1, emb, "time_encoder.1.", need_dx=False)
    ds = dc if dcs is None else add(dc, dcs)                       # the state embedding feeds y and, for MM-DiT, the condition stream
    for i in reversed(range(len(tape["st"]))):
        ds = _res_cat_bwd(P, G, f"state_encoder.{i}.", tape["st"][i], ds, need_dx=(i > 0 or want_state_grad))
    # h0 = x embedding + position table
    colsum(dh, _slice_rows(G.of("position_embedding").reshape(1, -1, dh.shape[-1]), tape["S"]))
    dx = dh
    for i in reversed(range(len(tape["xs"]))):
        dx = _res_cat_bwd(P, G, f"x_encoder.{i}.", tape["xs"][i], dx, need_dx=i > 0)
    return G.g, ds


class DenoiserDiT(torch.autograd.Function):
    """``Transformer_TrajDiffusionModel.forward`` (DiT backbone) as a differentiable function of its parameters and of ``state``."""

    @staticmethod
    def forward(ctx, meta, x, t, state, *params):
        names, n_head, n_blocks, n_state, n_x, mm, dp = meta
        P = {n: p.detach() for n, p in zip(names, params)}
        out, tape = dit_forward(P, f32c(x), t, f32c(state), n_head, n_blocks, n_state, n_x, mm, dp)
        ctx.meta, ctx.P, ctx.tape = meta, P, tape
        if ctx.needs_input_grad[1]:
            raise NotImplementedError("the training path computes no gradient for the input `x` (it requires grad): detach it")
        ctx.want_state = bool(ctx.needs_input_grad[3])
        return out

    @staticmethod
    def backward(ctx, dout):
        names, n_head = ctx.meta[0], ctx.meta[1]
        if ctx.tape is None:
            raise RuntimeError("the kept activations were released by the first backward: call forward again (retain_graph is not supported)")
        grads, d_state = dit_backward(ctx.P, ctx.tape, dout.contiguous(), n_head, ctx.want_state)
        ctx.tape = None
        return (None, None, None, d_state) + tuple(grads.get(n) for n in names)


def denoise_dit_with_grad(module: torch.nn.Module, x, t, state) -> torch.Tensor:
    """Differentiable call of a ``dit.Transformer_TrajDiffusionModel`` (DiT or MM-DiT backbone).  In training mode the DiT blocks'
    encoder layers apply torch's default dropout (0.1; ``config.dropout`` overrides); MM-DiT blocks hold no dropout layer."""
    named = [(n, p) for n, p in module.named_parameters()]
    mm = module.config.diffusion_backbone == "MM-DiT"
    dp = None if mm else dropout_plan(module, 0.0, float(getattr(module.config, "dropout", 0.1)))
    meta = (tuple(n for n, _ in named), int(module.config.n_head), len(module.blocks), len(list(module.state_encoder.children())),
            len(list(module.x_encoder.children())), mm, dp)
    return DenoiserDiT.apply(meta, x, t, state, *[p for _, p in named])
