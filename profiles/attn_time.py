#!/usr/bin/env python3
"""DiT / MM-DiT attention launches alone: the fp32-matrix-core kernel against the bf16-matrix-core kernel with 1, 2, 4, 8 heads per
workgroup (ADX_ATTN_HPW), events on the launch stream, one warm process."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autodiffusion_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


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


for name, B, H, dh in (("DiT d=256", 4096, 8, 32), ("DiT d=1024", 256, 16, 64), ("MM-DiT d=256", 1024, 8, 512)):
    q, k, v = (torch.randn(B, 80, H * dh, device=dev) for _ in range(3))
    gb = 4 * q.numel() * 4 / 1e9
    for _ in range(20):
        ops.attention(q, k, v, H)
    t = timed(lambda: ops.attention(q, k, v, H))
    print(f"{name} B={B} H={H} dh={dh}: fp32 kernel {t:7.1f} us ({gb / t * 1e3:5.2f} TB/s)", flush=True)
    for prec in sys.argv[1:] or ["bf16x3", "bf16"]:
        for hpw in ("auto",):
            if hpw == "auto":
                os.environ.pop("ADX_ATTN_HPW", None)
            elif H % int(hpw):
                continue
            else:
                os.environ["ADX_ATTN_HPW"] = hpw
            for _ in range(5):
                ops.attention(q, k, v, H, precision=prec)
            t = timed(lambda: ops.attention(q, k, v, H, precision=prec))
            print(f"    {prec} heads per workgroup {hpw:>4}: {t:7.1f} us ({gb / t * 1e3:5.2f} TB/s)", flush=True)
    del q, k, v
