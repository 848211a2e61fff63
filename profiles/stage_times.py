#!/usr/bin/env python3
"""Time each kernel stage of one variant-A sampling step in isolation (events on the launch stream)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda:0")
da = bench.build_model(dev, prec)
maps, prior, noise = bench.make_inputs(B, 0, dev, 1)
out = torch.empty(B, 80, 4, device=dev)
da.generate(trajectory_prior=prior, maps_info=maps, noise=noise, out=out)
torch.cuda.synchronize()
eng = da.diffusion.diffusion.model.engine()
names = {1: "fused encoder rows", 2: "last-layer attention", 3: "fused tail", 4: "chunk cache build", 5: "map cache build"}
for stage, name in names.items():
    iters = 3 if stage >= 4 else 20
    eng.debug_stage_a(B, stage, 2)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    eng.debug_stage_a(B, stage, iters)
    e.record()
    torch.cuda.synchronize()
    print(f"{prec} B={B} stage {stage} {name:22s} {s.elapsed_time(e) / iters * 1e3:9.1f} us")

# the tail right after the streaming kernels (its weight image no longer in L2), as inside a sampling step
def pair(stages, iters=20):
    for st in stages:
        eng.debug_stage_a(B, st, 1)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        for st in stages:
            eng.debug_stage_a(B, st, 1)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

print(f"{prec} B={B} stages 1+2       {pair([1, 2]):9.1f} us")
print(f"{prec} B={B} stages 1+2+3     {pair([1, 2, 3]):9.1f} us")
