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
        for staged in modes:
            ws = torch.empty(ops.linear_stage_bytes(M, N, K, prec), dtype=torch.uint8, device=dev) if staged else None
            for _ in range(3):
                ops.linear(x, w, b, precision=prec, prepared_n=N, workspace=ws)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            s.record()
            for _ in range(10):
                ops.linear(x, w, b, precision=prec, prepared_n=N, workspace=ws)
            e.record()
            torch.cuda.synchronize()
            us = s.elapsed_time(e) * 100
            print(f"{prec:7s} {'staged ' if staged else 'direct '} M={M:6d} N={N:5d} K={K:5d}  {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s  "
                  f"{(M * K * 4 + M * N * 4) / us / 1e3:7.1f} GB/s (X read once + Y written)")
        del x, w
