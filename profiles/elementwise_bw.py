#!/usr/bin/env python3
"""Achieved HBM bandwidth of the elementwise / normalisation kernels against the 8 TB/s HBM3E peak of an MI355X.

Every kernel is launched through the C ABI on buffers far larger than the 256 MB of last-level cache, timed with events on the
launch stream over `iters` back-to-back launches; bytes = the ALGORITHMIC traffic of one launch (every input read once, every
output written once).  Under `rocprofv3 --kernel-trace --stats` the same command gives the per-kernel durations by name
(profiles/r03_elementwise_kernel_stats.csv); `--pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum` gives the fabric requests.

usage: elementwise_bw.py [rows_in_thousands=320]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autodiffusion_amd import _lib, ops, schedules, train  # noqa: E402

PEAK = 8000.0
dev = torch.device("cuda:0")
R = (int(sys.argv[1]) if len(sys.argv) > 1 else 320) * 1000
rows = []


def timed(name, fn, nbytes, iters=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / iters
    gbs = nbytes / us / 1e3
    rows.append((name, nbytes / 1e6, us, gbs, gbs / PEAK))
    print(f"{name:78s} {nbytes / 1e6:9.1f} MB {us:9.1f} us {gbs:8.1f} GB/s  {gbs / PEAK:5.2f}", flush=True)


print(f"{'kernel (entry point, shape)':78s} {'bytes':>12s} {'time':>12s} {'achieved':>13s}  of 8 TB/s")
lib = _lib.load()
# ---- LayerNorm family: rows of d floats, read once, written once (+ shift / scale rows read once with adaLN modulation)
for d, tag in ((1024, "layernorm_vec_kernel<4>"), (256, "layernorm_vec_kernel<1>"), (2048, "layernorm_vec_kernel<8>"), (128, "layernorm_narrow_kernel<32>"),
               (32, "layernorm_narrow_kernel<8>"), (96, "layernorm_kernel (wave per row, any width)")):
    n = R if d >= 256 else R * 8
    n = min(n, (3 << 30) // (d * 4))
    x = torch.randn(n, d, device=dev)
    g, b = torch.ones(d, device=dev), torch.zeros(d, device=dev)
    y = torch.empty_like(x)

    def ln(shift=None, scale=None):
        gg, bb = (None, None) if shift is not None else (g, b)          # adaLN: LayerNorm without affine, then x (1 + scale) + shift
        _lib.check(lib.adx_op_layernorm(_lib.ptr(x), _lib.ptr(gg), _lib.ptr(bb), _lib.ptr(shift), _lib.ptr(scale), d if shift is not None else 0, 1e-5, 0,
                                        n, d, _lib.ptr(y), _lib.current_stream()), "layernorm")
    timed(f"{tag}: adx_op_layernorm affine, {n} x {d}", ln, 2 * n * d * 4)
    if d >= 256:
        sh, sc = torch.randn(n, d, device=dev) * 0.1, torch.randn(n, d, device=dev) * 0.1
        timed(f"{tag}: adx_op_layernorm + adaLN modulate, {n} x {d}", lambda: ln(sh, sc), 4 * n * d * 4)
        del sh, sc
    del x, y
    torch.cuda.empty_cache()

# ---- scheduler step: x, model output, noise read; x_next written (16 B per element)
n = 64 * 1000 * 1000
x, eps, z = (torch.randn(n, device=dev) for _ in range(3))
row = schedules.ddpm_rows({k: v for k, v in schedules.diffusion_buffers(schedules.cosine_beta_schedule(100)).items()}, "pred_noise")[10]
coef = torch.tensor([float(v) for v in row], device=dev)
xn = torch.empty_like(x)
timed(f"sched_step_kernel: adx_op_sched_step, {n} elements",
      lambda: _lib.check(lib.adx_op_sched_step(_lib.ptr(x), _lib.ptr(eps), _lib.ptr(z), _lib.ptr(xn), None, n, _lib.ptr(coef), _lib.current_stream()), "s"),
      4 * n * 4)
# ---- forward-process sample: x0, noise read; x_t written (12 B per element) -- timestep and table gathers are per row
B, L = 200000, 320
x0, nz = torch.randn(B, L, device=dev), torch.randn(B, L, device=dev)
t = torch.randint(0, 100, (B,), device=dev)
buf = schedules.diffusion_buffers(schedules.cosine_beta_schedule(100))
sa, s1 = buf["sqrt_alphas_cumprod"].float().to(dev), buf["sqrt_one_minus_alphas_cumprod"].float().to(dev)
xt = torch.empty_like(x0)
timed(f"q_sample_kernel: adx_op_q_sample, {B} x {L}",
      lambda: _lib.check(lib.adx_op_q_sample(_lib.ptr(x0), _lib.ptr(nz), _lib.ptr(t), _lib.ptr(sa), _lib.ptr(s1), _lib.ptr(xt), None, B, L,
                                             _lib.current_stream()), "q"), 3 * B * L * 4)
# ---- data normalisation (8 B per element), residual add (12 B per element), device noise (write only: 4 B per element)
pts = x0.reshape(-1, 4)
out = torch.empty_like(pts)
timed(f"normalize_kernel: adx_op_normalize, {pts.shape[0]} points x 4", lambda: _lib.check(lib.adx_op_normalize(_lib.ptr(pts), _lib.ptr(out), pts.shape[0], 4, _lib.current_stream()), "n"),
      2 * pts.numel() * 4)
timed(f"denormalize_kernel: adx_op_denormalize, {pts.shape[0]} points x 4",
      lambda: _lib.check(lib.adx_op_denormalize(_lib.ptr(pts), None, _lib.ptr(out), pts.shape[0], 4, _lib.current_stream()), "d"), 2 * pts.numel() * 4)
a_, b_ = x0.reshape(-1), nz.reshape(-1)
o_ = torch.empty_like(a_)
timed(f"add_kernel: adx_op_add, {a_.numel()} elements", lambda: _lib.check(lib.adx_op_add(_lib.ptr(a_), _lib.ptr(b_), _lib.ptr(o_), a_.numel(), _lib.current_stream()), "a"),
      3 * a_.numel() * 4)
ctl = torch.empty(2, device=dev, dtype=torch.int64)
rn = torch.empty(51, 4096, 320, device=dev)
timed("randn_rows_kernel: adx_op_randn, 51 x 4096 x 320 (Philox4x32-10 + Box-Muller; write only)",
      lambda: _lib.check(lib.adx_op_randn(_lib.ptr(rn), 51, 4096, 320, 1234, 0, _lib.ptr(ctl), _lib.current_stream()), "r"), rn.numel() * 4)
