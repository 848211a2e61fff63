"""Build libadx.so (HIP kernels + engines + C ABI) in-tree for gfx950.

``python -m autodiffusion_amd.build`` cross-compiles without a GPU.  The shared
library lands next to this file (``autodiffusion_amd/libadx.so``); it is what
``autodiffusion_amd._lib`` loads and what travels to the GPU box.
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libadx.so")
OBJ = os.path.join(HERE, "csrc", "build")
SOURCES = ["kernels.hip", "gemm.hip", "attn_mfma.hip", "modes.hip", "metrics.hip", "extras.hip", "fused_a.hip", "tail_a.hip", "tail_mfma.hip", "engine_base.hip", "engine_a.hip", "engine_dit.hip", "noalloc.hip", "backward.hip", "raster.hip", "moe.hip"]
HEADERS = ["adx_ops.h", "fastmath.h", "engine.h", os.path.join("..", "..", "include", "adx.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
         "-Rpass-analysis=kernel-resource-usage"]
# Kernels that must not touch scratch memory: their accumulators or row state live in registers by design, and a loop that stops
# unrolling (the accumulators are indexed by its counter) silently moves them to scratch -- same results, a quarter slower.
NO_SCRATCH = re.compile(r"linear_\w*kernel|mlp256_kernel|enc_rows_kernel|attn_g_kernel|tail_mfma_kernel|attn_bf_kernel|attn_mfma\w*kernel|"
                        r"layernorm_\w*kernel")


# known and accepted: the 4-waves-per-SIMD instances of the row kernel (a tuning alternative, not the default) spill a few registers
SCRATCH_OK = re.compile(r"enc_rows_kernelILi\dELi4E")


def _scratch_users(remarks: str):
    """(kernel, bytes per lane) for every kernel whose resource remark reports scratch."""
    out, name = [], None
    for line in remarks.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name and int(m.group(1)) > 0:
            out.append((name, int(m.group(1))))
    return out


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
        spilled = [(k, b) for k, b in _scratch_users(r.stderr) if NO_SCRATCH.search(k) and not SCRATCH_OK.search(k)]
        if spilled and not measure:         # the measurement library's ablation instances are exempt
            raise RuntimeError("kernels that must stay in registers use scratch memory:\n" + "\n".join(f"  {k}: {b} bytes per lane" for k, b in spilled))
        other = "\n".join(l for l in r.stderr.splitlines() if "-Rpass-analysis=kernel-resource-usage" not in l and "remark:" not in l)
        if verbose and other.strip():
            print(other, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _stale(out, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs])
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, measure="--measure" in sys.argv))
