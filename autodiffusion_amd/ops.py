"""Operator-level wrappers over the C ABI (one HIP kernel launch each).  Device tensors in, device tensors out."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import StepCoef, check, current_stream, f32c, ptr


def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    lib = _lib.load()
    t = t.to(torch.int64).contiguous()
    out = torch.empty(*t.shape, dim, device=t.device, dtype=torch.float32)
    check(lib.adx_op_timestep_embedding(ptr(t), t.numel(), dim, ptr(out), current_stream()), "timestep_embedding")
    return out


def to_bf16(x: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    x = f32c(x)
    y = torch.empty(x.shape, device=x.device, dtype=torch.int16)
    check(lib.adx_op_to_bf16(ptr(x), ptr(y), x.numel(), current_stream()), "to_bf16")
    return y


def to_f16(x: torch.Tensor) -> torch.Tensor:
    """fp32 -> fp16 bit patterns (int16 tensor), saturating at +-65504."""
    x = f32c(x)
    y = torch.empty(x.shape, device=x.device, dtype=torch.int16)
    check(_lib.load().adx_op_to_f16(ptr(x), ptr(y), x.numel(), current_stream()), "to_f16")
    return y


def prepare_weight(w: torch.Tensor, precision: str = "fp32") -> torch.Tensor:
    """The device form of an (N, K) weight that ``adx_op_linear`` reads under ``precision``."""
    prec = _lib.PRECISIONS[precision]
    if prec in (_lib.PREC_F16X2, _lib.PREC_F16):        # operand-plane products only: the weight rounded once to fp16
        return to_f16(w)
    if prec == _lib.PREC_BF16:
        return to_bf16(w)
    if prec == _lib.PREC_BF16X3:
        hi = to_bf16(w)
        hi_f = (hi.to(torch.int32) << 16).view(torch.float32)
        return torch.cat([hi.reshape(-1), to_bf16(f32c(w) - hi_f).reshape(-1)])
    return f32c(w)


def linear(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, act: int = _lib.ACT_NONE,
           precision: str = "fp32", prepared_n: int = 0, workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``act(x w^T + bias)``.  With ``prepared_n`` > 0, ``w`` is already the result of ``prepare_weight`` for N = prepared_n.
    ``workspace`` (a CUDA byte tensor of at least ``linear_stage_bytes`` bytes, bf16 precisions) selects the staged entry."""
    lib = _lib.load()
    x = f32c(x)
    K = x.shape[-1]
    N = prepared_n or w.shape[0]
    M = x.numel() // K
    prec = _lib.PRECISIONS[precision]
    wd = w if prepared_n else prepare_weight(w, precision)
    b = f32c(bias) if bias is not None else None
    y = torch.empty(*x.shape[:-1], N, device=x.device, dtype=torch.float32)
    if workspace is not None:
        check(lib.adx_op_linear_staged(ptr(x), ptr(wd), ptr(b), ptr(y), M, N, K, act, prec, ptr(workspace),
                                       workspace.numel() * workspace.element_size(), current_stream()), "linear_staged")
    else:
        check(lib.adx_op_linear(ptr(x), ptr(wd), ptr(b), ptr(y), M, N, K, act, prec, current_stream()), "linear")
    return y


def stage_planes(x: torch.Tensor, precision: str) -> torch.Tensor:
    """K-blocked bf16 planes of an fp32 (M, K) activation, the X operand form of ``linear_planes``."""
    lib = _lib.load()
    x = f32c(x)
    K = x.shape[-1]
    M = x.numel() // K
    prec = _lib.PRECISIONS[precision]
    out = torch.empty(int(lib.adx_linear_planes_bytes(M, K, prec)), dtype=torch.uint8, device=x.device)
    check(lib.adx_op_stage_planes(ptr(x), M, K, prec, ptr(out), current_stream()), "stage_planes")
    return out


def block_weight(w_prepared: torch.Tensor, N: int, K: int, precision: str) -> torch.Tensor:
    """K-blocked form of a weight that ``prepare_weight`` returned, the W operand form of ``linear_planes``."""
    lib = _lib.load()
    prec = _lib.PRECISIONS[precision]
    out = torch.empty(int(lib.adx_linear_planes_bytes(N, K, prec)), dtype=torch.uint8, device=w_prepared.device)
    check(lib.adx_op_block_weight(ptr(w_prepared), N, K, prec, ptr(out), current_stream()), "block_weight")
    return out


def linear_planes(x_planes: torch.Tensor, w_blocked: torch.Tensor, bias: Optional[torch.Tensor], M: int, N: int, K: int,
                  act: int = _lib.ACT_NONE, precision: str = "bf16x3", out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``act(x w^T + bias)`` from operands already in the staged kernel's layout: only the product kernel runs."""
    lib = _lib.load()
    b = f32c(bias) if bias is not None else None
    y = out if out is not None else torch.empty(M, N, device=x_planes.device, dtype=torch.float32)
    check(lib.adx_op_linear_planes(ptr(x_planes), ptr(w_blocked), ptr(b), ptr(y), M, N, K, act, _lib.PRECISIONS[precision], current_stream()),
          "linear_planes")
    return y


def mlp256(x: torch.Tensor, w1_blocked: torch.Tensor, b1: Optional[torch.Tensor], act1: int, w2_blocked: torch.Tensor,
           b2: Optional[torch.Tensor], act2: int = _lib.ACT_NONE, gate: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None,
           precision: str = "bf16x3", out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``res + gate * act2(act1(x w1^T + b1) w2^T + b2)`` for x (M, 256) in one kernel; weights in ``block_weight`` form."""
    lib = _lib.load()
    x = f32c(x)
    M = x.shape[0]
    planes = 2 if precision == "bf16x3" else 1
    nbytes = w1_blocked.numel() * w1_blocked.element_size()
    N1 = nbytes // (256 * 2 * planes)                   # bf16 elements, one or two planes
    if N1 * 256 * 2 * planes != nbytes or w2_blocked.numel() * w2_blocked.element_size() != nbytes or x.shape[-1] != 256:
        raise ValueError("mlp256: x must be (M, 256) and both weights the block_weight form of (N1, 256) / (256, N1) matrices")
    for t, n in ((b1, N1), (b2, 256), (gate, M * 256), (res, M * 256), (out, M * 256)):
        if t is not None and t.numel() != n:
            raise ValueError("mlp256: operand of the wrong size")
    b1 = f32c(b1) if b1 is not None else None
    b2 = f32c(b2) if b2 is not None else None
    gate = f32c(gate) if gate is not None else None
    res = f32c(res) if res is not None else None
    y = out if out is not None else torch.empty(M, 256, device=x.device, dtype=torch.float32)
    check(lib.adx_op_mlp256(ptr(x), ptr(w1_blocked), ptr(b1), int(act1), ptr(w2_blocked), ptr(b2), int(act2), ptr(gate), ptr(res), ptr(y),
                            M, N1, _lib.PRECISIONS[precision], current_stream()), "adx_op_mlp256")
    return y


def linear_stage_bytes(M: int, N: int, K: int, precision: str) -> int:
    return int(_lib.load().adx_linear_stage_bytes(int(M), int(N), int(K), _lib.PRECISIONS[precision]))


def layernorm(x: torch.Tensor, gamma=None, beta=None, shift=None, scale=None, eps: float = 1e-5, relu: bool = False) -> torch.Tensor:
    lib = _lib.load()
    x = f32c(x)
    d = x.shape[-1]
    rows = x.numel() // d
    ld_mod = 0
    if shift is not None:
        shift, scale = f32c(shift), f32c(scale)
        ld_mod = d
    y = torch.empty_like(x)
    check(lib.adx_op_layernorm(ptr(x), ptr(f32c(gamma)) if gamma is not None else None, ptr(f32c(beta)) if beta is not None else None,
                               ptr(shift), ptr(scale), ld_mod, eps, int(relu), rows, d, ptr(y), current_stream()), "layernorm")
    return y


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, n_head: int, scale: Optional[float] = None,
              mask_off: int = -1, precision: str = "fp32") -> torch.Tensor:
    """q (B,Sq,H*dh), k (B,Skv,H*dh), v (B,Skv,H*dv) -> (B,Sq,H*dv)."""
    lib = _lib.load()
    q, k, v = f32c(q), f32c(k), f32c(v)
    B, Sq, Skv = q.shape[0], q.shape[1], k.shape[1]
    dh, dv = q.shape[2] // n_head, v.shape[2] // n_head
    if scale is None:
        scale = dh ** -0.5
    o = torch.empty(B, Sq, n_head * dv, device=q.device, dtype=torch.float32)
    check(lib.adx_op_attention_prec(ptr(q), ptr(k), ptr(v), ptr(o), B, n_head, Sq, Skv, dh, dv, float(scale), mask_off,
                                    _lib.PRECISIONS[precision], current_stream()), "attention")
    return o


def normalize(x: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    x = f32c(x)
    y = torch.empty_like(x)
    check(lib.adx_op_normalize(ptr(x), ptr(y), x.numel() // x.shape[-1], x.shape[-1], current_stream()), "normalize")
    return y


def denormalize(y: torch.Tensor, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = _lib.load()
    y = f32c(y)
    r = f32c(residual) if residual is not None else None
    x = torch.empty_like(y)
    check(lib.adx_op_denormalize(ptr(y), ptr(r), ptr(x), y.numel() // y.shape[-1], y.shape[-1], current_stream()), "denormalize")
    return x


def sched_step(x: torch.Tensor, model_out: torch.Tensor, z: Optional[torch.Tensor], row: Sequence[float], want_x0: bool = False):
    lib = _lib.load()
    x, model_out = f32c(x), f32c(model_out)
    z = f32c(z) if z is not None else None
    coef = torch.tensor([float(v) for v in row], dtype=torch.float32, device=x.device)
    xn = torch.empty_like(x)
    x0 = torch.empty_like(x) if want_x0 else None
    check(lib.adx_op_sched_step(ptr(x), ptr(model_out), ptr(z), ptr(xn), ptr(x0), x.numel(), ptr(coef), current_stream()), "sched_step")
    return (xn, x0) if want_x0 else xn


def select_modes(traj: torch.Tensor, k: int, em_iter: int = 25):
    """(B, M, L, D) sampled trajectories -> (modes (B, k, L, D), weights (B, k)), weights descending."""
    lib = _lib.load()
    traj = f32c(traj)
    B, M, L, D = traj.shape
    modes = torch.empty(B, k, L, D, device=traj.device, dtype=torch.float32)
    weights = torch.empty(B, k, device=traj.device, dtype=torch.float32)
    check(lib.adx_op_select_modes(ptr(traj), B, M, L, D, int(k), int(em_iter), ptr(modes), ptr(weights), current_stream()), "select_modes")
    return modes, weights


def q_sample(x0: torch.Tensor, noise: torch.Tensor, t: torch.Tensor, sqrt_acp: torch.Tensor, sqrt_1m_acp: torch.Tensor,
             want_v: bool = False):
    """x_t = sqrt_acp[t] x0 + sqrt_1m_acp[t] noise with a timestep per batch row (dim 0); also v = sqrt_acp[t] noise -
    sqrt_1m_acp[t] x0 when ``want_v``.  Returns (x_t, v or None)."""
    lib = _lib.load()
    x0, noise = f32c(x0), f32c(noise)
    t = t.to(device=x0.device, dtype=torch.int64).contiguous()
    sa, s1 = f32c(sqrt_acp).to(x0.device), f32c(sqrt_1m_acp).to(x0.device)
    n_rows = x0.shape[0]
    row_len = x0.numel() // n_rows
    xt = torch.empty_like(x0)
    v = torch.empty_like(x0) if want_v else None
    check(lib.adx_op_q_sample(ptr(x0), ptr(noise), ptr(t), ptr(sa), ptr(s1), ptr(xt), ptr(v), n_rows, row_len, current_stream()), "q_sample")
    return xt, v


def weighted_mse(out: torch.Tensor, target: torch.Tensor, t: torch.Tensor, weight: torch.Tensor):
    """(loss_rows (B,), loss_mean ()): loss_rows[b] = weight[t[b]] * mean((out[b] - target[b])**2)."""
    lib = _lib.load()
    out, target = f32c(out), f32c(target)
    t = t.to(device=out.device, dtype=torch.int64).contiguous()
    w = f32c(weight).to(out.device)
    n_rows = out.shape[0]
    row_len = out.numel() // n_rows
    rows = torch.empty(n_rows, device=out.device, dtype=torch.float32)
    mean = torch.empty(1, device=out.device, dtype=torch.float32)
    check(lib.adx_op_weighted_mse(ptr(out), ptr(target), ptr(t), ptr(w), ptr(rows), ptr(mean), n_rows, row_len, current_stream()), "weighted_mse")
    return rows, mean.reshape(())


def randn(n_lead: int, batch: int, row_len: int, seed: int, row0: int = 0, device="cuda") -> torch.Tensor:
    """(n_lead, batch, row_len) N(0,1) values from the library's counter-based generator (Philox4x32-10 + Box-Muller): the value of
    element e of batch row b under leading index l depends only on (seed, row0 + b, l, e // 4)."""
    lib = _lib.load()
    out = torch.empty(n_lead, batch, row_len, device=device, dtype=torch.float32)
    ctl = torch.empty(2, device=out.device, dtype=torch.int64)
    check(lib.adx_op_randn(ptr(out), n_lead, batch, row_len, int(seed) & 0xFFFFFFFFFFFFFFFF, int(row0), ptr(ctl), current_stream()), "randn")
    return out
