#!/usr/bin/env python3
"""Time of one generate() at the bench configuration under tuning switches given on the command line (one engine per process: most
switches are read once).

usage: knob_probe.py PRECISION [B] [KEY=VALUE ...]"""
import os
import sys
import time

args = sys.argv[1:]
prec = args.pop(0) if args else "f16"
B = int(args.pop(0)) if args and "=" not in args[0] else 1024
for kv in args:
    k, v = kv.split("=", 1)
    os.environ[k] = v

import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from autodiffusion_amd import synth  # noqa: E402
import test_gpu_parity as T  # noqa: E402

fs, N, ns = 10, 100, 50
dev = torch.device("cuda:0")
maps = synth.maps_info(B, T.ISEED).to(dev)
prior = synth.trajectory_prior(B, T.ISEED).to(dev)
noise = synth.noise(B, (8, ns), (fs, 4), T.NSEED).to(dev)
da = T._make_da(fs, N, prec, "reference", dev, ns)
out = da.generate(trajectory_prior=prior, maps_info=maps, noise=noise)["traj_logits"].clone()
for _ in range(2):
    da.generate(trajectory_prior=prior, maps_info=maps, noise=noise)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(8):
    da.generate(trajectory_prior=prior, maps_info=maps, noise=noise)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 8
print(f"{prec} B={B} {' '.join(args) or '(defaults)'}: {dt * 1e3:7.2f} ms per generate ({B / dt:7.0f} traj/s)   checksum {out.double().sum().item():.9e}", flush=True)
