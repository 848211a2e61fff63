#!/usr/bin/env python3
"""K = 256 products of the d = 256 decoder (M = 327680 token rows): the row-stationary kernel on fp32 X (what the engine
launches) against the staged product kernel on operand planes (ADX_PLANES_MIN_K=256 lets it take K = 256), one warm process."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autodiffusion_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
os.environ["ADX_PLANES_MIN_K"] = "256"
M = int(os.environ.get("ADX_M", 327680))


def timed(fn, n=10, reps=3):
    best = float("inf")
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        for _ in range(n):
            fn()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3 / n)
    return best


for prec in sys.argv[1:] or ["bf16x3"]:
    for N, K in [(768, 256), (1024, 256), (2048, 256), (1536, 256)]:
        x = torch.randn(M, K, device=dev)
        w32 = torch.randn(N, K, device=dev) * 0.02
        w = ops.prepare_weight(w32, prec)
        b = torch.randn(N, device=dev)
        for _ in range(60):
            y0 = ops.linear(x, w, b, precision=prec, prepared_n=N)
        t_rows = timed(lambda: ops.linear(x, w, b, precision=prec, prepared_n=N))
        xp, wp, y = ops.stage_planes(x, prec), ops.block_weight(w, N, K, prec), torch.empty(M, N, device=dev)
        for _ in range(20):
            ops.linear_planes(xp, wp, b, M, N, K, precision=prec, out=y)
        t_pl = timed(lambda: ops.linear_planes(xp, wp, b, M, N, K, precision=prec, out=y))
        err = float((y - y0).abs().max() / y0.abs().max())
        tf = lambda us: 2.0 * M * N * K / us / 1e6
        print(f"{prec} M={M} N={N} K={K}: rows kernel {t_rows:7.1f} us ({tf(t_rows):6.1f} TFLOP/s)   planes kernel {t_pl:7.1f} us ({tf(t_pl):6.1f} TFLOP/s)"
              f"   max diff {err:.1e}", flush=True)
        del x, w, xp, wp, y, y0
