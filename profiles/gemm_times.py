#!/usr/bin/env python3
"""Time adx_op_linear (ADX_GEMM_STAGED=1: adx_op_linear_staged, 2: both) on the DiT GEMM shapes (events on the launch stream)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autodiffusion_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
shapes = [(5120, 4096, 1024), (5120, 1024, 4096), (5120, 6144, 1024), (20480, 4096, 1024), (327680, 1024, 256), (327680, 256, 1024),
          (327680, 1536, 256)]
if os.environ.get("ADX_GEMM_SHAPE"):      # e.g. 20480,4096,1024
    shapes = [tuple(int(v) for v in os.environ["ADX_GEMM_SHAPE"].split(","))]
for prec in sys.argv[1:] or ["bf16", "bf16x3"]:
    for M, N, K in shapes:
        x = torch.randn(M, K, device=dev)
        w = ops.prepare_weight(torch.randn(N, K, device=dev) * 0.02, prec)
        b = torch.zeros(N, device=dev)
        modes = {"0": [False], "1": [True], "2": [False, True]}[os.environ.get("ADX_GEMM_STAGED", "0")]
        wss = {st: (torch.empty(ops.linear_stage_bytes(M, N, K, prec), dtype=torch.uint8, device=dev) if st else None) for st in modes}

        def run(staged, n):
            for _ in range(n):
                ops.linear(x, w, b, precision=prec, prepared_n=N, workspace=wss[staged])
        for staged in modes:                # warm-up long enough for the clocks to settle (a fresh process starts cold)
            run(staged, 20)
        best = {st: float("inf") for st in modes}
        for _ in range(3):                  # the modes alternate, the best of three timings each is reported
            for staged in modes:
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                s.record()
                run(staged, 10)
                e.record()
                torch.cuda.synchronize()
                best[staged] = min(best[staged], s.elapsed_time(e) * 100)
        for staged in modes:
            us = best[staged]
            print(f"{prec:7s} {'staged ' if staged else 'direct '} M={M:6d} N={N:5d} K={K:5d}  {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s  "
                  f"{(M * K * 4 + M * N * 4) / us / 1e3:7.1f} GB/s (X read once + Y written)")
        del x, w, wss
