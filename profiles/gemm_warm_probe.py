#!/usr/bin/env python3
"""How many launches an idle GPU needs before the timing of the scaled DiT decoder's dominant GEMM settles: twelve bursts of 50
launches of the product kernel (operands in planes), events on the launch stream."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autodiffusion_amd import ops
dev = torch.device("cuda:0")
M, N, K = 20480, 4096, 1024
prec = "bf16x3"
x = torch.randn(M, K, device=dev); w32 = torch.randn(N, K, device=dev) * 0.02; b = torch.zeros(N, device=dev)
w = ops.prepare_weight(w32, prec); xp = ops.stage_planes(x, prec); wp = ops.block_weight(w, N, K, prec); y = torch.empty(M, N, device=dev)
def t(n):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n): ops.linear_planes(xp, wp, b, M, N, K, precision=prec, out=y)
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
for i in range(12):
    print(f"burst {i}: {t(50):.1f} us per launch (50 launches)", flush=True)
