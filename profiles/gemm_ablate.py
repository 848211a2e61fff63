#!/usr/bin/env python3
"""Where the time of the staged GEMM goes: the kernel with phases switched off (ADX_GLDS_ABLATE bits: 1 no fragment reads,
2 no copies, 4 no MFMAs, 8 no epilogue; results are wrong, timings only), both forms, alternating in one warm process."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# the ablation instances live in the measurement library only (python -m autodiffusion_amd.build --measure)
os.environ.setdefault("ADX_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "autodiffusion_amd", "libadx_measure.so"))
from autodiffusion_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M, N, K = [int(v) for v in os.environ.get("ADX_GEMM_SHAPE", "20480,4096,1024").split(",")]
os.environ["ADX_GEMM_DMA_ANY"] = "1"
for prec in sys.argv[1:] or ["bf16x3", "bf16"]:
    x = torch.randn(M, K, device=dev)
    w = ops.prepare_weight(torch.randn(N, K, device=dev) * 0.02, prec)
    b = torch.randn(N, device=dev)
    ws = torch.empty(ops.linear_stage_bytes(M, N, K, prec), dtype=torch.uint8, device=dev)
    y = torch.empty(M, N, device=dev)
    cases = [(f, a) for f in os.environ.get("ADX_FORMS", "1,2").split(",") for a in os.environ.get("ADX_ABLS", "0,8,11,3,2,1,4").split(",")]

    def run(form, abl, n):
        os.environ["ADX_GLDS_FORM"], os.environ["ADX_GLDS_ABLATE"] = form, abl
        for _ in range(n):
            ops.linear(x, w, b, precision=prec, prepared_n=N, workspace=ws)
    for c in cases:
        run(*c, 10)
    best = {c: float("inf") for c in cases}
    for _ in range(3):
        for c in cases:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            s.record()
            run(*c, 10)
            e.record()
            torch.cuda.synchronize()
            best[c] = min(best[c], s.elapsed_time(e) * 100)
    names = {"0": "everything", "8": "no epilogue", "11": "MFMAs + barriers only", "3": "MFMAs + epilogue", "2": "no copies", "1": "no fragment reads",
             "4": "no MFMAs", "16": "interleaved", "24": "interleaved, no epilogue", "17": "interleaved, no fragment reads",
             "18": "interleaved, no copies", "19": "interleaved, MFMAs + epilogue", "20": "interleaved, no MFMAs", "27": "interleaved, MFMAs + barriers only", "80": "interleaved, copies re-read K-step 0"}
    for (f, a), us in best.items():
        print(f"{prec:7s} form {f}  {names[a]:24s} {us:8.1f} us (incl. ~48 us of re-layout kernels)", flush=True)
