#!/usr/bin/env python3
"""The operand-plane product under the three split / unsplit operand forms, alternating in one warm process: split-bf16 (three matrix
instructions per product), fp16 two-term (two), single fp16 and single bf16 (one), at the four large GEMM shapes of the d = 1024 DiT decoder
(M = 256 scenarios x 80 tokens).  Events on the launch stream, minimum of 5 rounds of 20 launches each.

usage: gemm_f16x2.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autodiffusion_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M = 20480
shapes = [(4096, 1024), (1024, 4096), (3072, 1024), (6144, 1024)]
precs = ("bf16x3", "f16x2", "f16", "bf16")


def timed(fn, iters=20):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


print(f"{'shape':24s}" + "".join(f"{p:>12s}" for p in precs) + "   (us per launch; useful TFLOP/s)")
for N, K in shapes:
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.02
    bias = torch.zeros(N, device=dev)
    y = torch.empty(M, N, device=dev)
    calls = {}
    for p in precs:
        xp, wp = ops.stage_planes(x, p), ops.block_weight(ops.prepare_weight(w, p), N, K, p)
        calls[p] = (lambda xp=xp, wp=wp, p=p: ops.linear_planes(xp, wp, bias, M, N, K, precision=p, out=y))
    for _ in range(8):                      # warm the clocks with all three
        for p in precs:
            timed(calls[p])
    best = {p: 1e9 for p in precs}
    for _ in range(5):
        for p in precs:
            best[p] = min(best[p], timed(calls[p]))
    print(f"M={M} N={N} K={K}".ljust(24) + "".join(f"{best[p]:9.1f} us" for p in precs) + "   " +
          " / ".join(f"{2.0 * M * N * K / best[p] / 1e6:.0f}" for p in precs), flush=True)
