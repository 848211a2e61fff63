#!/usr/bin/env python3
"""ADE of the split-bf16 engine against the fp32 engine (itself within 1e-6 of the CPU oracle) over many scenarios:
the distribution behind the 1e-2 m gate, for the bench configuration (DDIM 50 of 100, reference initialisation)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from autodiffusion_amd import metrics, synth  # noqa: E402
import test_gpu_parity as T  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0")
fs, N, ns = 10, 100, 50
maps = synth.maps_info(B, T.ISEED).to(dev)
prior = synth.trajectory_prior(B, T.ISEED).to(dev)
noise = synth.noise(B, (8, ns), (fs, 4), T.NSEED).to(dev)
out = {}
for prec in ("fp32", "bf16x3", "bf16"):
    da = T._make_da(fs, N, prec, "reference", dev, ns)
    out[prec] = da.generate(trajectory_prior=prior, maps_info=maps, noise=noise)["traj_logits"].cpu()
for prec in ("bf16x3", "bf16"):
    ade = metrics.ade_fde(out[prec], out["fp32"])["ade8"]
    q = torch.quantile(ade, torch.tensor([0.5, 0.9, 0.99]))
    print(f"{prec:7s} tail_f32={os.environ.get('ADX_TAIL_F32', '0')} B={B}: ADE vs fp32 engine  median {q[0]:.2e}  p90 {q[1]:.2e}  p99 {q[2]:.2e}  max {ade.max():.2e} m"
          f"  (> 1e-2 m: {(ade > 1e-2).sum().item()} of {B})")
