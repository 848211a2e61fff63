#!/usr/bin/env python3
"""What a caller that hands in NEW tensors on every generate() pays (the evaluation loop and the simulator do): the first call
captures a graph on the caller's pointers, the second -- pointers changed -- switches the shape to the engine's staging buffers
and captures once more, every later call copies in, replays and copies out.  Compared with a caller that reuses its buffers,
and with noise drawn on the device (noise=None)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
da = bench.build_model(dev, "bf16x3")
eng = da.diffusion.diffusion.model.engine
maps, prior, noise = bench.make_inputs(B, 0, dev, 1)
keep = []                                   # old tensors stay alive, so every call really sees new addresses


def timed(label, **kw):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    da.generate(**kw)
    torch.cuda.synchronize()
    print(f"B={B} {label:44s} {1e3 * (time.perf_counter() - t0):8.1f} ms   graph captures so far: {eng().capture_count()}", flush=True)


for i in range(5):
    m2, p2, n2 = maps.clone(), prior.clone(), noise.clone()
    keep += [m2, p2, n2]
    timed(f"fresh tensors, call {i}", trajectory_prior=p2, maps_info=m2, noise=n2)
for i in range(3):
    m2, p2 = maps.clone(), prior.clone()
    keep += [m2, p2]
    timed(f"fresh tensors, noise drawn on device, call {i}", trajectory_prior=p2, maps_info=m2)
out = torch.empty(B, 80, 4, device=dev)
da2 = bench.build_model(dev, "bf16x3")
eng = da2.diffusion.diffusion.model.engine
da = da2
for i in range(3):
    timed(f"same tensors (a second model), call {i}", trajectory_prior=prior, maps_info=maps, noise=noise, out=out)
