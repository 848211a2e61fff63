#!/usr/bin/env python3
"""Does the scaled DiT decoder need all three terms of the split-bf16 product?  d=1024, h=16, L layers, B scenarios, N stochastic
DDPM steps on the same noise: final-trajectory ADE of the bf16x3 engine, of the same engine with the low halves of W dropped in
the staged GEMMs (two MFMAs per product, ADX_GLDS_ABLATE=144) and of the single-pass bf16 engine, each against the fp32 engine.
Weights N(0, 0.02) as bench.py initialises them."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autodiffusion_amd import dit, metrics, synth  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 6
B = int(sys.argv[2]) if len(sys.argv) > 2 else 192
N = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda:0")
state = synth.hidden_state(B, 1024, 7).to(dev)
noise = synth.noise(B, (N + 1,), (80, 4), 1234).to(dev)


def run(prec, abl=None):
    dec = dit.DiffusionDecoder(dit.dit_config(1024, L, 16, "DiT", diffusion_timesteps=N, precision=prec))
    torch.manual_seed(11)
    with torch.no_grad():
        for p in dec.parameters():
            p.normal_(0.0, 0.02)
    dec = dec.to(dev).eval()
    if abl:
        os.environ["ADX_GLDS_ABLATE"] = abl
    try:
        out = dec.model(state, batch_size=B, determin=False, noise=noise).clone()
        torch.cuda.synchronize()
    finally:
        os.environ.pop("ADX_GLDS_ABLATE", None)
    return out


ref = run("fp32")
print(f"DiT d=1024 L={L} B={B} N={N}: |trajectory| max {float(ref.abs().max()):.1f} m")
for name, prec, abl in (("bf16x3 (three terms)", "bf16x3", None), ("two terms (W rounded to bf16 in the staged GEMMs)", "bf16x3", "144"), ("single-pass bf16", "bf16", None)):
    out = run(prec, abl)
    m = metrics.ade_fde(out, ref)
    print(f"{name:52s} ADE vs fp32 engine: mean {float(m['ade8'].mean()):.3e} m  worst {float(m['ade8'].max()):.3e} m", flush=True)
