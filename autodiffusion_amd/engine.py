"""Python handle around an ``adx_engine`` (weights + workspace + sampling loop on the GPU)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterable, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import ModelConfig, StepCoef, check, current_stream, f32c, ptr


class Engine:
    """Owns one ``adx_engine``.  Parameters are pushed by name; the handle is freed on ``close``/GC."""

    def __init__(self, **cfg):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.AdxError("no GPU visible: the sampling engines run on an MI355X only")
        self.cfg = ModelConfig(**cfg)
        h = C.c_void_p()
        check(self.lib.adx_engine_create(C.byref(self.cfg), C.byref(h)), "adx_engine_create")
        self.h = h
        self._rows = None
        self._stamp = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.adx_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ parameters
    def load(self, named: Iterable[Tuple[str, torch.Tensor]]):
        torch.cuda.synchronize()
        for name, t in named:
            t = f32c(t)
            if not t.is_cuda:
                t = t.cuda()
            shape = (C.c_int64 * t.dim())(*t.shape)
            check(self.lib.adx_engine_set_tensor(self.h, name.encode(), ptr(t), shape, t.dim()), f"set_tensor({name})")
        check(self.lib.adx_engine_finalize(self.h), "adx_engine_finalize")   # rebuilds the weight-derived tables of a set schedule

    def sync_from(self, module: torch.nn.Module, prefix: str = ""):
        """(Re)load the parameters of ``module`` if any changed since the last call."""
        params = [(n, p) for n, p in module.named_parameters()]
        stamp = tuple((n, p.data_ptr(), p._version, str(p.device)) for n, p in params)
        if stamp != self._stamp:
            self.load((n[len(prefix):] if n.startswith(prefix) else n, p) for n, p in params)
            self._stamp = stamp

    def set_extra(self, named: Iterable[Tuple[str, torch.Tensor]]):
        """Tensors beside the denoiser's own (the latent maps of the transition wrapper), re-sent only when one changed."""
        named = list(named)
        stamp = tuple((n, p.data_ptr(), p._version, str(p.device)) for n, p in named)
        if stamp != getattr(self, "_extra_stamp", None):
            self.load(named)
            self._extra_stamp = stamp

    def generate_transition(self, maps, prior_full, init_z, noise, mean_std, frame_stack: int, flags: int = 0,
                            out: Optional[torch.Tensor] = None):
        """Latent-carrying chunk loop in one captured graph (``adx_generate_transition``)."""
        B, S = prior_full.shape[0], prior_full.shape[1]
        n_noise = 0 if noise is None else noise.shape[1]
        if out is None:
            out = torch.empty(B, S, 4, device=prior_full.device, dtype=torch.float32)
        check(self.lib.adx_generate_transition(self.h, B, S, int(frame_stack), ptr(maps), ptr(prior_full), ptr(init_z), ptr(noise), n_noise,
                                               ptr(mean_std), ptr(out), flags, current_stream()), "adx_generate_transition")
        return out

    def reserve(self, batch: int):
        check(self.lib.adx_engine_reserve(self.h, int(batch)), "adx_engine_reserve")

    def workspace_bytes(self) -> int:
        return int(self.lib.adx_engine_workspace_bytes(self.h))

    def capture_count(self) -> int:
        """hipGraph captures made by this engine so far."""
        return int(self.lib.adx_engine_capture_count(self.h))

    def set_noise_seed(self, seed: int, row0: int = 0):
        """Seed and global index of batch row 0 for sampling calls that pass ``noise=None`` (noise drawn on the device)."""
        check(self.lib.adx_engine_set_noise_seed(self.h, int(seed) & 0xFFFFFFFFFFFFFFFF, int(row0)), "adx_engine_set_noise_seed")

    def seed_from_torch(self, seed: Optional[int] = None, row0: int = 0):
        """``seed`` None: a fresh 63-bit seed from torch's CPU generator, so ``torch.manual_seed`` makes runs repeatable."""
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
        self.set_noise_seed(seed, row0)

    # ------------------------------------------------------------------ schedule
    def set_schedule(self, rows: Sequence[Sequence[float]]):
        key = tuple(tuple(r) for r in rows)
        if key == self._rows:
            return
        arr = (StepCoef * len(rows))()
        for i, r in enumerate(rows):
            arr[i] = StepCoef(*[float(v) for v in r])
        check(self.lib.adx_engine_set_schedule(self.h, arr, len(rows)), "adx_engine_set_schedule")
        self._rows = key

    # ------------------------------------------------------------------ calls
    def denoise_a(self, x, t, maps, transition, prior) -> torch.Tensor:
        B, T = x.shape[0], x.shape[1]
        x, maps, transition, prior = f32c(x), (f32c(maps) if maps is not None else None), f32c(transition), f32c(prior)
        t = t.to(torch.int64).contiguous()
        out = torch.empty(B, T, self.cfg.out_dim, device=x.device, dtype=torch.float32)
        check(self.lib.adx_denoise_a(self.h, B, ptr(x), ptr(t), ptr(maps), ptr(transition), ptr(prior), ptr(out), current_stream()),
              "adx_denoise_a")
        return out

    def denoise_dit(self, x, t, state) -> torch.Tensor:
        B = x.shape[0]
        x, state = f32c(x), f32c(state)
        t = t.to(torch.int64).contiguous()
        out = torch.empty_like(x)
        check(self.lib.adx_denoise_dit(self.h, B, ptr(x), ptr(t), ptr(state), ptr(out), current_stream()), "adx_denoise_dit")
        return out

    def sample_a(self, maps, transition, prior, noise, flags: int, out: Optional[torch.Tensor] = None, trace: bool = False):
        B = transition.shape[0]
        T, Cin = self.cfg.horizon, self.cfg.in_dim
        n_noise = 0 if noise is None else noise.shape[0]
        if out is None:
            out = torch.empty(B, T, Cin, device=transition.device, dtype=torch.float32)
        n_steps = len(self._rows)
        tx = torch.empty(n_steps, B, T, Cin, device=out.device) if trace else None
        to = torch.empty(n_steps, B, T, Cin, device=out.device) if trace else None
        check(self.lib.adx_sample_a(self.h, B, ptr(maps), ptr(transition), ptr(prior), ptr(noise), n_noise, ptr(out), ptr(tx), ptr(to),
                                    flags, current_stream()), "adx_sample_a")
        return (out, tx, to) if trace else out

    def generate_a(self, maps, prior_full, init_x, noise, frame_stack: int, flags: int, out: Optional[torch.Tensor] = None):
        B, S = prior_full.shape[0], prior_full.shape[1]
        n_noise = 0 if noise is None else noise.shape[1]
        if out is None:
            out = torch.empty(B, S, 4, device=prior_full.device, dtype=torch.float32)
        check(self.lib.adx_generate_a(self.h, B, S, frame_stack, ptr(maps), ptr(prior_full), ptr(init_x), ptr(noise), n_noise, ptr(out),
                                      flags, current_stream()), "adx_generate_a")
        return out

    def debug_stage_a(self, batch: int, stage: int, iters: int = 1):
        """Launch one kernel stage of a variant-A step on the current workspace (timing aid)."""
        check(self.lib.adx_debug_stage_a(self.h, int(batch), int(stage), int(iters), current_stream()), "adx_debug_stage_a")

    def profile(self, enable: bool):
        """Time the marked kernels of the following calls with events (calls then run without graph capture)."""
        check(self.lib.adx_engine_profile(self.h, int(bool(enable))), "adx_engine_profile")

    def profile_read(self, stage: int) -> Tuple[float, int]:
        """(summed milliseconds, launches) of one stage since the last clear; stage -1 clears."""
        ms, n = C.c_double(0.0), C.c_int(0)
        check(self.lib.adx_engine_profile_read(self.h, int(stage), C.byref(ms), C.byref(n)), "adx_engine_profile_read")
        return ms.value, n.value

    def sample_dit(self, state, noise, flags: int, out: Optional[torch.Tensor] = None, group: int = 1):
        """state (B / group, S, d_model): with group > 1 every state item conditions ``group`` consecutive sample rows."""
        group = int(group)
        B, S = state.shape[0] * group, state.shape[1]
        n_noise = 0 if noise is None else noise.shape[0]
        if noise is not None and noise.shape[1] != B:
            raise ValueError(f"noise holds {noise.shape[1]} rows, the call samples {B}")
        if out is None:
            out = torch.empty(B, S, self.cfg.in_dim, device=state.device, dtype=torch.float32)
        if group == 1:
            check(self.lib.adx_sample_dit(self.h, B, ptr(state), ptr(noise), n_noise, ptr(out), flags, current_stream()), "adx_sample_dit")
        else:
            check(self.lib.adx_sample_dit_grouped(self.h, B, group, ptr(state), ptr(noise), n_noise, ptr(out), flags, current_stream()),
                  "adx_sample_dit_grouped")
        return out
