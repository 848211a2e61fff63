#!/usr/bin/env python3
"""Where the time of the last-layer attention kernel goes: the kernel with parts removed (ADX_ATTN_G_ABLATE, results are wrong),
timed in isolation at the bench batch; events on the launch stream."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# the ablation instances live in the measurement library only (python -m autodiffusion_amd.build --measure)
os.environ.setdefault("ADX_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "autodiffusion_amd", "libadx_measure.so"))
import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
da = bench.build_model(dev, "bf16x3")
maps, prior, noise = bench.make_inputs(B, 0, dev, 1)
out = torch.empty(B, 80, 4, device=dev)
da.generate(trajectory_prior=prior, maps_info=maps, noise=noise, out=out)
torch.cuda.synchronize()
eng = da.diffusion.diffusion.model.engine()
names = {0: "everything", 16: "hand-placed order instead of interleaved chains (same results)", 1: "no exponentials", 2: "no weighted-sum products", 4: "no score products", 6: "no score and weighted-sum products",
         8: "no split of the weights", 9: "no exponentials, no split of the weights", 15: "loads, V projection and bookkeeping only"}
for abl, name in names.items():
    os.environ["ADX_ATTN_G_ABLATE"] = str(abl)
    eng.debug_stage_a(B, 2, 5)
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        eng.debug_stage_a(B, 2, 20)
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 20 * 1e3)
    print(f"attn_g B={B} {name:45s} {best:7.1f} us", flush=True)
os.environ.pop("ADX_ATTN_G_ABLATE", None)
