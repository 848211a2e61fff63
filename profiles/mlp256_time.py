#!/usr/bin/env python3
"""The fused two-layer MLP (d = 256, hidden 1024) at M = 327680 token rows against the two products it replaces (first product by
the row-stationary kernel, second by the staged kernel on operand planes -- here with the re-layout pass the engine does not pay)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autodiffusion_amd import ops, _lib  # noqa: E402

dev = torch.device("cuda:0")
M, N1 = int(os.environ.get("ADX_M", 327680)), 1024


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
    x = torch.randn(M, 256, device=dev)
    w1, w2 = torch.randn(N1, 256, device=dev) * 0.06, torch.randn(256, N1, device=dev) * 0.03
    b1, b2 = torch.randn(N1, device=dev) * 0.1, torch.randn(256, device=dev) * 0.1
    res = torch.randn(M, 256, device=dev)
    w1p, w2p = ops.prepare_weight(w1, prec), ops.prepare_weight(w2, prec)
    w1b, w2b = ops.block_weight(w1p, N1, 256, prec), ops.block_weight(w2p, 256, N1, prec)
    y = torch.empty(M, 256, device=dev)
    for act, name in ((_lib.ACT_RELU, "relu"), (_lib.ACT_GELU_TANH, "gelu_tanh")):
        for _ in range(30):
            ops.mlp256(x, w1b, b1, act, w2b, b2, act2=act if act == _lib.ACT_GELU_TANH else 0, res=res, precision=prec, out=y)
        t = timed(lambda: ops.mlp256(x, w1b, b1, act, w2b, b2, act2=act if act == _lib.ACT_GELU_TANH else 0, res=res, precision=prec, out=y))
        t1 = timed(lambda: ops.linear(x, w1p, b1, act=act, precision=prec, prepared_n=N1))
        hid = ops.linear(x, w1p, b1, act=act, precision=prec, prepared_n=N1)
        t2 = timed(lambda: ops.linear(hid, w2p, b2, precision=prec, prepared_n=256))
        fl = 4.0 * M * N1 * 256
        print(f"{prec} {name}: fused {t:7.1f} us ({fl / t / 1e6:6.1f} TFLOP/s)   two products {t1:7.1f} + {t2:7.1f} us", flush=True)
        del hid
