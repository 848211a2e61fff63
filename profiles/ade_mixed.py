#!/usr/bin/env python3
"""ADE of the refiner's split-bf16 engine and of its fp16 forms (ADX_A_MIX bits: 1 | 16 | 32 = 49 the three products of the row kernel with both
operands rounded once to fp16, 2 | 4 = 6 the weighted sums, scores and V of the last encoder attention; 55 = all of them; 63 = also fp16 o / l, gates and LayerNorm rows in memory = precision "f16")
against the fp32 engine over many scenarios, for the bench configuration (frame stack 10, DDIM 50 of 100, reference initialisation); and
the time of one generate() for each.  (The product library has the row kernel's three products as one instance; r03_ade_mixed*.txt were
taken while each product had its own.)

usage: ade_mixed.py [scenarios]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from autodiffusion_amd import metrics, synth  # noqa: E402
import test_gpu_parity as T  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
fs, N, ns = 10, 100, 50
maps = synth.maps_info(B, T.ISEED).to(dev)
prior = synth.trajectory_prior(B, T.ISEED).to(dev)
noise = synth.noise(B, (8, ns), (fs, 4), T.NSEED).to(dev)


def run(prec, mix):
    os.environ["ADX_A_MIX"] = str(mix)
    da = T._make_da(fs, N, prec, "reference", dev, ns)
    out = da.generate(trajectory_prior=prior, maps_info=maps, noise=noise)["traj_logits"].clone()
    for _ in range(2):
        da.generate(trajectory_prior=prior, maps_info=maps, noise=noise)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        da.generate(trajectory_prior=prior, maps_info=maps, noise=noise)
    torch.cuda.synchronize()
    return out.cpu(), (time.perf_counter() - t0) / 5


ref, _ = run("fp32", 0)
for mix in (0, 6, 49, 55, 63, 0, 63):
    out, dt = run("bf16x3", mix)
    ade = metrics.ade_fde(out, ref)["ade8"]
    q = torch.quantile(ade, torch.tensor([0.5, 0.9, 0.99]))
    print(f"bf16x3 mix={mix} B={B}: {dt * 1e3:7.2f} ms per generate ({B / dt:7.0f} traj/s)   ADE vs fp32 engine  median {q[0]:.2e}  p90 {q[1]:.2e}  "
          f"p99 {q[2]:.2e}  max {ade.max():.2e} m  (> 1e-2 m: {(ade > 1e-2).sum().item()} of {B})", flush=True)
