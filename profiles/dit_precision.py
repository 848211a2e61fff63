#!/usr/bin/env python3
"""How many bf16 terms does the scaled DiT decoder need?  BASELINE configs[4] at its own size -- hidden 1024, 24 blocks, 16 heads,
200-step ancestral DDPM (cosine), stochastic sampling, `mc` samples per scenario -- on the CONTRACTIVE fixture
(synth.contractive_dit_init_: the sampler behaves as with a trained denoiser, final trajectories are tens of metres), same noise
in every run.  Final-trajectory ADE of each engine precision against the fp32 engine:

    bf16x3   x_hi w_hi + x_lo w_hi + x_hi w_lo      (three MFMAs per product)
    bf16x2   x_hi w_hi + x_lo w_hi                  (weights rounded to bf16 once, activations split: two MFMAs)
    bf16     x_hi w_hi                              (one MFMA)
    f16x2    x_hi w + x_lo w on fp16 halves         (weights rounded to fp16 once -- 11 significant bits against bf16's 8 --,
                                                     activations split: two MFMAs)
    f16      x w, both rounded once to fp16         (one MFMA; every kernel other than the operand-plane products stays split-bf16)

usage: dit_precision.py [L=24] [scenarios=16] [mc=16] [steps=200] [backbone=DiT]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("ADX_TWO_TERM_ABLATION"):     # the two-term line needs the measurement library (python -m autodiffusion_amd.build --measure)
    os.environ.setdefault("ADX_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "autodiffusion_amd", "libadx_measure.so"))
from autodiffusion_amd import _lib, dit, metrics, synth  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 24
SC = int(sys.argv[2]) if len(sys.argv) > 2 else 16
MC = int(sys.argv[3]) if len(sys.argv) > 3 else 16
N = int(sys.argv[4]) if len(sys.argv) > 4 else 200
BB = sys.argv[5] if len(sys.argv) > 5 else "DiT"
D, H = 1024, 16
READOUT, GATE = 0.93, 0.1
dev = torch.device("cuda:0")
B = SC * MC
state = synth.hidden_state(SC, D, 7).to(dev)
noise = synth.noise(B, (N + 1,), (80, 4), 1234).to(dev)

host = dit.DiffusionDecoder(dit.dit_config(D, L, H, BB, k=MC, mc_num=MC, diffusion_timesteps=N))
synth.contractive_dit_init_(host.model.model, 11, readout_scale=READOUT, gate=GATE)
weights = {k: v.clone() for k, v in host.state_dict().items()}
del host


def run(prec, abl=None):
    if abl:                                     # measurement library only: drops the w_lo term inside the staged GEMM
        os.environ["ADX_GLDS_ABLATE"] = abl
    try:
        return _run(prec)
    finally:
        os.environ.pop("ADX_GLDS_ABLATE", None)


def _run(prec):
    dec = dit.DiffusionDecoder(dit.dit_config(D, L, H, BB, k=MC, mc_num=MC, diffusion_timesteps=N, precision=prec))
    dec.load_state_dict(weights)
    dec = dec.to(dev).eval()
    t0 = time.time()
    out = dec.model(state, batch_size=SC, determin=False, mc_num=MC, noise=noise).reshape(B, 80, 4).clone()
    torch.cuda.synchronize()
    dt = time.time() - t0
    del dec
    torch.cuda.empty_cache()
    return out, dt


ref, dt = run("fp32")
xy = ref[..., :2]
print(f"{BB} d={D} L={L} h={H}, {SC} scenarios x {MC} samples, {N} DDPM steps; fp32 engine {dt:.1f} s; |x,y| mean {float(xy.abs().mean()):.1f} m, "
      f"max {float(xy.abs().max()):.1f} m, on the clamp: {float((ref[..., 0].abs() >= 2474.9).float().mean()) * 100:.2f} %", flush=True)
cases = [(p, p, None) for p in ("bf16x3", "f16x2", "f16", "bf16x2", "bf16") if p in _lib.PRECISIONS]
if "bf16x2" not in _lib.PRECISIONS and os.environ.get("ADX_TWO_TERM_ABLATION"):
    cases.insert(1, ("x3-w_lo", "bf16x3", "144"))
for name, prec, abl in cases:
    out, dt = run(prec, abl)
    m = metrics.ade_fde(out, ref)
    a = m["ade8"]
    print(f"{name:7s} ADE vs fp32 engine: mean {float(a.mean()):.3e} m  median {float(a.median()):.3e} m  worst {float(a.max()):.3e} m   "
          f"FDE worst {float(m['fde8'].max()):.3e} m   ({dt:.1f} s)", flush=True)
