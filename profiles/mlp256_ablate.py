#!/usr/bin/env python3
"""Where the time of the fused MLP kernel goes: template ablations (ADX_MLP_ABLATE, results are wrong, timing only), split
operands, ReLU, M = 327680, hidden 1024."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# the ablation instances live in the measurement library only (python -m autodiffusion_amd.build --measure)
os.environ.setdefault("ADX_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "autodiffusion_amd", "libadx_measure.so"))
from autodiffusion_amd import ops, _lib  # noqa: E402

dev = torch.device("cuda:0")
M, N1, prec = 327680, 1024, "bf16x3"
x = torch.randn(M, 256, device=dev)
w1b = ops.block_weight(ops.prepare_weight(torch.randn(N1, 256, device=dev) * 0.06, prec), N1, 256, prec)
w2b = ops.block_weight(ops.prepare_weight(torch.randn(256, N1, device=dev) * 0.03, prec), 256, N1, prec)
b1, b2 = torch.randn(N1, device=dev) * 0.1, torch.randn(256, device=dev) * 0.1
res, y = torch.randn(M, 256, device=dev), torch.empty(M, 256, device=dev)
names = {0: "everything", 1: "no weight copies after the first", 2: "no products in phase 2", 4: "no products in phase 1", 6: "no products at all",
         7: "no products, no copies", 15: "no products, no copies, no epilogue memory traffic", 23: "no products, no copies, no X loads",
         31: "no products, no copies, no global memory traffic at all", 24: "everything but the X loads and the epilogue traffic"}
run = lambda: ops.mlp256(x, w1b, b1, _lib.ACT_RELU, w2b, b2, res=res, precision=prec, out=y)
for _ in range(40):
    run()
for abl in (int(v) for v in (sys.argv[1:] or names)):
    os.environ["ADX_MLP_ABLATE"] = str(abl)
    for _ in range(5):
        run()
    best = float("inf")
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        for _ in range(10):
            run()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 100)
    print(f"ablate {abl} ({names.get(abl, '?')}): {best:7.1f} us", flush=True)
