#!/usr/bin/env python3
"""Time variants of the staged GEMM against each other, alternating in one warm process (events on the launch stream).
A variant is a set of the per-call tuning variables, written NAME=VALUE,NAME=VALUE; e.g.
    python profiles/gemm_sweep.py bf16x3 ADX_GLDS_FORM=1 ADX_GLDS_FORM=2,ADX_GLDS_PANEL=0 ADX_GLDS_FORM=2,ADX_GLDS_PANEL=8
Shape from ADX_GEMM_SHAPE=M,N,K (default 20480,4096,1024).  Reported times include the re-layout kernels of the staged entry."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autodiffusion_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M, N, K = [int(v) for v in os.environ.get("ADX_GEMM_SHAPE", "20480,4096,1024").split(",")]
prec = sys.argv[1]
variants = [dict(kv.split("=") for kv in v.split(",")) for v in sys.argv[2:]]
names = set(k for v in variants for k in v)
os.environ["ADX_GEMM_DMA_ANY"] = "1"
x = torch.randn(M, K, device=dev)
w = ops.prepare_weight(torch.randn(N, K, device=dev) * 0.02, prec)
b = torch.randn(N, device=dev)
ws = torch.empty(ops.linear_stage_bytes(M, N, K, prec), dtype=torch.uint8, device=dev)


def run(v, n):
    for k in names:
        os.environ.pop(k, None)
    os.environ.update(v)
    y = None
    for _ in range(n):
        y = ops.linear(x, w, b, precision=prec, prepared_n=N, workspace=ws)
    return y


ys = [run(v, 15) for v in variants]
best = [float("inf")] * len(variants)
for _ in range(4):
    for i, v in enumerate(variants):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        run(v, 10)
        e.record()
        torch.cuda.synchronize()
        best[i] = min(best[i], s.elapsed_time(e) * 100)
mult = 3 if prec == "bf16x3" else 1
for i, v in enumerate(variants):
    tf = 2.0 * M * N * K / best[i] / 1e6
    print(f"{prec} M={M} N={N} K={K}  {best[i]:8.1f} us  {tf:7.1f} TFLOP/s (executed {tf * mult:7.1f})  same bits as variant 0: {torch.equal(ys[0], ys[i])}  {sys.argv[2 + i]}",
          flush=True)
