#!/usr/bin/env python3
"""Dispatch table evidence: every GEMM shape of the DiT workloads through the default dispatch without a workspace ("direct":
tiled / row-stationary kernels) and through the staged LDS-DMA kernel forced for the shape (ADX_GEMM_DMA_ANY=1), alternating
in one warm process.  Staged times include the re-layout kernels of X and W."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autodiffusion_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
shapes = []
for B in (64, 256):                     # DiT d=1024: token rows B * 80
    M = B * 80
    shapes += [(M, 6144, 1024), (M, 3072, 1024), (M, 1024, 1024), (M, 4096, 1024), (M, 1024, 4096), (M, 2048, 1024), (M, 1024, 2048)]
shapes += [(327680, 1536, 256), (327680, 768, 256), (327680, 256, 256), (327680, 1024, 256), (327680, 256, 1024), (163840, 2048, 256), (163840, 256, 4096)]
for prec in sys.argv[1:] or ["bf16x3"]:
    for M, N, K in shapes:
        x = torch.randn(M, K, device=dev)
        w = ops.prepare_weight(torch.randn(N, K, device=dev) * 0.02, prec)
        b = torch.zeros(N, device=dev)
        ws = torch.empty(ops.linear_stage_bytes(M, N, K, prec), dtype=torch.uint8, device=dev)

        def run(staged, n):
            if staged:
                os.environ["ADX_GEMM_DMA_ANY"] = "1"
            else:
                os.environ.pop("ADX_GEMM_DMA_ANY", None)
            for _ in range(n):
                ops.linear(x, w, b, precision=prec, prepared_n=N, workspace=ws if staged else None)
        best = {}
        for st in (False, True):
            run(st, 10)
            best[st] = float("inf")
        for _ in range(3):
            for st in (False, True):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                s.record()
                run(st, 10)
                e.record()
                torch.cuda.synchronize()
                best[st] = min(best[st], s.elapsed_time(e) * 100)
        tiles = ((M + 255) // 256) * ((N + 127) // 128)
        print(f"{prec:7s} M={M:6d} N={N:5d} K={K:5d} tiles(256x128)={tiles:5d}  direct {best[False]:8.1f} us  staged {best[True]:8.1f} us  ratio {best[True] / best[False]:.2f}",
              flush=True)
        del x, w, ws
