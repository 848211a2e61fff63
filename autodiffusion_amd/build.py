"""Build libadx.so (HIP kernels + engines + C ABI) in-tree for gfx950.

``python -m autodiffusion_amd.build`` cross-compiles without a GPU.  The shared
library lands next to this file (``autodiffusion_amd/libadx.so``); it is what
``autodiffusion_amd._lib`` loads and what travels to the GPU box.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libadx.so")
OBJ = os.path.join(HERE, "csrc", "build")
SOURCES = ["kernels.hip", "gemm.hip", "attn_mfma.hip", "modes.hip", "metrics.hip", "extras.hip", "fused_a.hip", "tail_a.hip", "tail_mfma.hip", "engine_base.hip", "engine_a.hip", "engine_dit.hip", "noalloc.hip", "backward.hip", "raster.hip"]
HEADERS = ["adx_ops.h", "fastmath.h", "engine.h", os.path.join("..", "..", "include", "adx.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, measure: bool = False) -> str:
    """``measure``: the measurement library ``libadx_measure.so`` (-DADX_MEASURE): the same code plus the kernel instances with
    parts switched off that the ablation scripts under ``profiles/`` select through ``ADX_*_ABLATE`` -- their results are wrong
    on purpose, so they are compiled out of ``libadx.so``; ``ADX_LIB_PATH`` makes ``_lib.load()`` take the other library."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    obj_dir = OBJ + ("_measure" if measure else "")
    out = os.path.join(HERE, "libadx_measure.so") if measure else OUT
    flags = FLAGS + (["-DADX_MEASURE"] if measure else [])
    os.makedirs(obj_dir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(obj_dir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([hipcc, *flags, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _stale(out, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs])
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, measure="--measure" in sys.argv))
