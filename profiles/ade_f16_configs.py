#!/usr/bin/env python3
"""ADE of the refiner's f16 and bf16x3 engines against the fp32 engine on the other BASELINE configurations of the refiner: C1 (frame stack 10, 100-step
DDPM, batch 16), C3's shape (whole-trajectory chunks, DDIM 10 of 100, batch 512) and C1's sampler at batch 256.

usage: ade_f16_configs.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from autodiffusion_amd import metrics, synth  # noqa: E402
import test_gpu_parity as T  # noqa: E402

dev = torch.device("cuda:0")
for tag, fs, N, ns, B in (("C1 fs=10 DDPM 100", 10, 100, None, 16), ("fs=10 DDPM 100", 10, 100, None, 256), ("C3 fs=80 DDIM 10 of 100", 80, 100, 10, 512)):
    maps = synth.maps_info(B, T.ISEED).to(dev)
    prior = synth.trajectory_prior(B, T.ISEED).to(dev)
    noise = synth.noise(B, (80 // fs, ns or N), (fs, 4), T.NSEED).to(dev)
    out = {}
    for prec in ("fp32", "bf16x3", "f16"):
        da = T._make_da(fs, N, prec, "reference", dev, ns)
        out[prec] = da.generate(trajectory_prior=prior, maps_info=maps, noise=noise)["traj_logits"].clone().cpu()
    for prec in ("bf16x3", "f16"):
        ade = metrics.ade_fde(out[prec], out["fp32"])["ade8"]
        print(f"{tag:26s} B={B:4d} {prec:7s}: ADE vs fp32 engine median {float(ade.median()):.2e}  max {float(ade.max()):.2e} m", flush=True)
