#!/usr/bin/env python3
"""Probe: does the fused tail (stage 3) overlap with the encoder-side kernels (stages 1, 2) when launched on two streams?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
da = bench.build_model(dev, "bf16x3")
maps, prior, noise = bench.make_inputs(B, 0, dev, 1)
out = torch.empty(B, 80, 4, device=dev)
da.generate(trajectory_prior=prior, maps_info=maps, noise=noise, out=out)
torch.cuda.synchronize()
eng = da.diffusion.diffusion.model.engine()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
N = 20


def run(enc, tail):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    sa.wait_stream(torch.cuda.current_stream())
    sb.wait_stream(torch.cuda.current_stream())
    for _ in range(N):
        if enc:
            with torch.cuda.stream(sa):
                eng.debug_stage_a(B, 1, 1)
                eng.debug_stage_a(B, 2, 1)
        if tail:
            with torch.cuda.stream(sb):
                eng.debug_stage_a(B, 3, 1)
    torch.cuda.current_stream().wait_stream(sa)
    torch.cuda.current_stream().wait_stream(sb)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / N * 1e3


for _ in range(2):
    print(f"B={B} waves={os.environ.get('ADX_ENC_ROWS_WAVES', '3')}: enc+attn {run(True, False):7.1f} us   tail {run(False, True):7.1f} us   both streams {run(True, True):7.1f} us")
