#!/usr/bin/env python3
"""ADE against the fp32 engine and time of one generate() for the arithmetic forms of the MFMA tail's maps (ADX_TAIL_FORMS = memory-row form,
trajectory-token form: 0 fp32, 1 split-bf16, 2 split-fp16, 3 fp16 rounded once), in the fp16 engine and in the split-bf16 engine; bench
configuration (frame stack 10, DDIM 50 of 100, reference initialisation).

usage: tail_forms.py [scenarios]"""
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


def run(prec, forms):
    if forms:
        os.environ["ADX_TAIL_FORMS"] = forms
    else:
        os.environ.pop("ADX_TAIL_FORMS", None)
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


ref, _ = run("fp32", None)
for prec, forms in (("f16", "10"), ("f16", "22"), ("f16", "20"), ("bf16x3", "10"), ("bf16x3", "22"), ("bf16x3", "20"), ("f16", "22")):
    out, dt = run(prec, forms)
    ade = metrics.ade_fde(out, ref)["ade8"]
    q = torch.quantile(ade, torch.tensor([0.5, 0.9, 0.99]))
    print(f"{prec:6s} tail forms {forms} B={B}: {dt * 1e3:7.2f} ms per generate ({B / dt:7.0f} traj/s)   ADE vs fp32 engine  median {q[0]:.2e}  "
          f"p90 {q[1]:.2e}  p99 {q[2]:.2e}  max {ade.max():.2e} m  (> 1e-2 m: {(ade > 1e-2).sum().item()} of {B})", flush=True)
