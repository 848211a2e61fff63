"""Seeded synthetic weights and nuPlan-shaped inputs.

Everything here is a pure function of integer seeds (NumPy PCG64), so the same
tensors can be regenerated on any machine: by the script that records golden
vectors, by the parity tests, by ``bench.py`` and by the multi-GPU shards (noise
is indexed by *global* batch row, which makes a shard's rows identical to the
same rows of a single-GPU run).
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import numpy as np
import torch


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.default_rng([int(seed), zlib.crc32(name.encode("utf-8"))])


def seeded_param(name: str, shape: Tuple[int, ...], seed: int) -> torch.Tensor:
    """Value of one parameter under the seeded-weight recipe.

    * 2-D (Linear / packed MHA weights): N(0,1)/sqrt(fan_in), so activations keep
      O(1) scale and attention logits are O(1) (a softmax that is far from
      uniform is what exposes mask and scale mistakes).
    * 3-D (learned positional tables): 0.5*N(0,1).
    * 1-D LayerNorm gains (``norm*.weight``, ``ln_f.weight``, ``layer_norm.weight``):
      1 + 0.1*N(0,1).
    * other 1-D (biases): 0.1*N(0,1).
    The generator is keyed by (seed, crc32(name)), so the value does not depend
    on the iteration order of the module's parameters.
    """
    g = _rng(seed, name)
    x = g.standard_normal(size=tuple(shape)).astype(np.float32)
    if len(shape) >= 3:
        x *= np.float32(0.5)
    elif len(shape) == 2:
        x *= np.float32(1.0 / np.sqrt(shape[-1]))
    else:
        leaf = name.split(".")
        is_gain = leaf[-1] == "weight" and any(
            k in leaf[-2] for k in ("norm", "ln_f", "layer_norm")
        )
        if is_gain:
            x = np.float32(1.0) + np.float32(0.1) * x
        else:
            x *= np.float32(0.1)
    return torch.from_numpy(np.ascontiguousarray(x))


def seeded_state(named_shapes: Iterable[Tuple[str, Tuple[int, ...]]], seed: int) -> Dict[str, torch.Tensor]:
    return {n: seeded_param(n, tuple(s), seed) for n, s in named_shapes}


def fill_module_(module: torch.nn.Module, seed: int) -> torch.nn.Module:
    """Overwrite every parameter of ``module`` with the seeded recipe (in place)."""
    with torch.no_grad():
        for n, p in module.named_parameters():
            p.copy_(seeded_param(n, tuple(p.shape), seed).to(p.dtype))
    return module


def reference_init_(module: torch.nn.Module, seed: int) -> torch.nn.Module:
    """Random initialisation in the style the reference applies to its refiner
    (transformer4planning/models/diffusion_loss/transformer.py:149-187): every matrix and
    positional table ~ N(0, 0.02), biases 0, LayerNorm gains 1; a 2-D ``init_x`` ~ N(0, 1)
    (DiffusionAugmentation.py:70-71).  Seeded per parameter name.
    """
    with torch.no_grad():
        for n, p in module.named_parameters():
            x = _rng(seed, n).standard_normal(size=tuple(p.shape)).astype(np.float32)
            leaf = n.split(".")
            if leaf[-1] == "init_x":
                pass
            elif p.dim() >= 2:
                x *= np.float32(0.02)
            elif leaf[-1] == "weight" and any(k in leaf[-2] for k in ("norm", "ln_f", "layer_norm")):
                x = np.ones_like(x)
            else:
                x = np.zeros_like(x)
            p.copy_(torch.from_numpy(x).to(p.dtype))
    return module


def contractive_dit_init_(module: torch.nn.Module, seed: int, readout_scale: float = 1.0, gate: float = 0.05, mod: float = 0.3,
                         side: float = 0.1) -> torch.nn.Module:
    """Seeded weights under which the DiT / MM-DiT decoder behaves like a TRAINED denoiser inside the sampler.

    An untrained decoder returns eps unrelated to x_t, the DDPM recursion then multiplies x by 1/sqrt(alpha_t) per step and the
    +-max_action clamp is what bounds the sample: final coordinates of thousands of metres, on which an error in metres says
    nothing.  Here every tensor starts from :func:`seeded_param` (all layers active, activations of order one) and four changes
    make eps = x_t * (1 + small) + r(x_t, t, state), |r| a fraction of |x_t|, which is the contraction a trained denoiser gives
    (x0 estimates inside the data range, final trajectories of tens of metres):

    * the trajectory encoder's ``fc`` matrices carry an identity from their first four inputs to their first four outputs (their
      other contributions to those outputs scaled by ``side``), and the position table is zero on those four features, so the
      residual stream holds x_t in features 0..3;
    * gate rows of every ``adaLN_modulation`` product are scaled by ``gate`` (the blocks perturb the stream instead of replacing
      it), shift / scale rows by ``mod``;
    * the final linear layer reads features 0..3 back with ``readout_scale`` (the factor that undoes the final LayerNorm's
      division by the stream's standard deviation; measured once per configuration) plus ``side`` times its seeded matrix.
    """
    with torch.no_grad():
        for n, p in module.named_parameters():
            x = seeded_param(n, tuple(p.shape), seed)
            leaf = n.split(".")
            if "adaLN_modulation" in n:
                chunks = 2 if "final_layer" in n else 6
                rows = x.shape[0] // chunks
                for c in range(chunks):
                    is_gate = chunks == 6 and c in (2, 5)
                    x[c * rows:(c + 1) * rows] *= (gate if is_gate else mod * (0.3 if chunks == 2 else 1.0))
            elif "x_encoder." in n and leaf[-2] == "fc":
                x[:4] *= side
                if leaf[-1] == "weight":
                    x[:4, :4] += torch.eye(4)
            elif leaf[-1] == "position_embedding":
                x[..., :4] = 0
            elif "final_layer.linear" in n:
                x *= side
                if leaf[-1] == "weight":
                    k = min(x.shape[0], 4)
                    x[:k, :k] += readout_scale * torch.eye(k)
            p.copy_(x.to(p.dtype))
    return module


def _rows(seed: int, tag: str, row0: int, nrows: int, tail: Tuple[int, ...]) -> np.ndarray:
    """Standard-normal block whose row r only depends on (seed, tag, global row r)."""
    out = np.empty((nrows,) + tail, dtype=np.float32)
    for i in range(nrows):
        g = np.random.default_rng([int(seed), zlib.crc32(tag.encode()), row0 + i])
        out[i] = g.standard_normal(size=tail).astype(np.float32)
    return out


def maps_info(batch: int, seed: int = 7, row0: int = 0, n_tokens: int = 788, dim: int = 256) -> torch.Tensor:
    """STR map/raster conditioning tokens, ``(B, 788, 256)`` ~ 0.5*N(0,1)."""
    return torch.from_numpy(_rows(seed, "maps", row0, batch, (n_tokens, dim)) * np.float32(0.5))


def trajectory_prior(batch: int, seed: int = 7, row0: int = 0, points: int = 80) -> torch.Tensor:
    """STR trajectory prior ``(B, 80, 4)``: forward drive with lateral wobble.

    x = cumsum(v*0.1 s), v ~ U(0,15) m/s per scenario; y = smooth lateral drift;
    channel 2 is 0; channel 3 is a small yaw.
    """
    out = np.zeros((batch, points, 4), dtype=np.float32)
    for i in range(batch):
        g = np.random.default_rng([int(seed), zlib.crc32(b"prior"), row0 + i])
        v = g.uniform(0.0, 15.0)
        out[i, :, 0] = np.cumsum(np.full(points, v * 0.1))
        out[i, :, 1] = np.cumsum(g.standard_normal(points) * 0.05) * 0.5
        out[i, :, 3] = g.standard_normal(points) * 0.05
    return torch.from_numpy(out)


def hidden_state(batch: int, n_embd: int, seed: int = 7, row0: int = 0, points: int = 80) -> torch.Tensor:
    """STR backbone hidden states for the DiT decoders, ``(B, 80, n_embd)`` ~ N(0,1)."""
    return torch.from_numpy(_rows(seed, "state", row0, batch, (points, n_embd)))


def occupancy_raster(batch: int, channels: int, seed: int = 7, size: int = 224, density: float = 0.08) -> torch.Tensor:
    """Rasterised nuPlan-style inputs ``(B, 224, 224, C)``, channels last: sparse 0 / 1 occupancy, ``density`` of the pixels set."""
    g = np.random.default_rng(seed)
    return torch.from_numpy((g.random((batch, size, size, channels)) < density).astype(np.float32))


def dropout_multipliers(shape: Tuple[int, ...], p: float, seed: int, site: int) -> torch.Tensor:
    """The multipliers of dropout layer number ``site`` of one forward pass: 0 with probability p, else 1 / (1 - p)."""
    keep = _rng(seed, f"dropout.{site}").random(tuple(shape), dtype=np.float32) >= np.float32(p)
    return torch.from_numpy(keep.astype(np.float32) / np.float32(1.0 - p))


def noise(batch: int, lead: Tuple[int, ...], tail: Tuple[int, ...], seed: int = 1234, row0: int = 0) -> torch.Tensor:
    """Gaussian noise ``lead + (B,) + tail`` where batch row r depends only on its global index.

    ``lead`` is e.g. ``(n_chunks, n_steps + 1)`` and ``tail`` is ``(T, C)``.
    """
    block = _rows(seed, "noise", row0, batch, tuple(lead) + tuple(tail))  # (B, *lead, *tail)
    nl = len(lead)
    perm = tuple(range(1, 1 + nl)) + (0,) + tuple(range(1 + nl, 1 + nl + len(tail)))
    return torch.from_numpy(np.ascontiguousarray(block.transpose(perm)))
