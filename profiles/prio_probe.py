#!/usr/bin/env python3
"""Wave priority of the per-step row / attention kernels against the tail that runs beside them on the second stream (ADX_A_PRIO, read per
call; a graph captured under one value keeps it, so every value gets its own engine): time of one generate() at the bench configuration.

usage: prio_probe.py [precision]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from autodiffusion_amd import synth  # noqa: E402
import test_gpu_parity as T  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "f16"
B, fs, N, ns = 1024, 10, 100, 50
dev = torch.device("cuda:0")
maps = synth.maps_info(B, T.ISEED).to(dev)
prior = synth.trajectory_prior(B, T.ISEED).to(dev)
noise = synth.noise(B, (8, ns), (fs, 4), T.NSEED).to(dev)
ref = None
for prio in (0, 1, 2, 3, 0, 3):
    os.environ["ADX_A_PRIO"] = str(prio)
    da = T._make_da(fs, N, prec, "reference", dev, ns)
    out = da.generate(trajectory_prior=prior, maps_info=maps, noise=noise)["traj_logits"].clone()
    ref = out if ref is None else ref
    for _ in range(2):
        da.generate(trajectory_prior=prior, maps_info=maps, noise=noise)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        da.generate(trajectory_prior=prior, maps_info=maps, noise=noise)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 8
    print(f"{prec} priority {prio}: {dt * 1e3:7.2f} ms per generate ({B / dt:7.0f} traj/s)   same bits as priority 0: {torch.equal(out, ref)}", flush=True)
