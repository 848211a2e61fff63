#!/usr/bin/env python3
"""What the epilogue of the operand-plane product costs: the same product with its result stored as fp32 (the documented form),
as the next product's operand planes (what the DiT engine's 4d-wide layers do), or not at all.  Measurement library only
(python -m autodiffusion_amd.build --measure); same process, alternating.

usage: gemm_epilogue.py [precision ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("ADX_LIB_PATH", os.path.join(ROOT, "autodiffusion_amd", "libadx_measure.so"))
from autodiffusion_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M = 20480
shapes = [(4096, 1024), (1024, 4096)]
outs = ["fp32", "planes", "planes2", "none"]


def timed(fn, iters=20):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


for prec in sys.argv[1:] or ["f16", "bf16", "bf16x3"]:
    for N, K in shapes:
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) * 0.02
        bias = torch.zeros(N, device=dev)
        y = torch.empty(M, N, device=dev)
        xp, wp = ops.stage_planes(x, prec), ops.block_weight(ops.prepare_weight(w, prec), N, K, prec)
        call = lambda: ops.linear_planes(xp, wp, bias, M, N, K, precision=prec, out=y)  # noqa: E731
        for _ in range(10):
            timed(call)
        best = {o: 1e9 for o in outs}
        for _ in range(4):
            for o in outs:
                if o == "fp32":
                    os.environ.pop("ADX_PLANES_OUT", None)
                else:
                    os.environ["ADX_PLANES_OUT"] = o
                best[o] = min(best[o], timed(call))
        os.environ.pop("ADX_PLANES_OUT", None)
        print(f"{prec:7s} M={M} N={N} K={K}: " + "  ".join(f"{o}: {best[o]:.1f}" for o in outs) + "  us", flush=True)
