#!/usr/bin/env python3
"""Throughput of the trajectory-diffusion sampling path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]

Workload (BASELINE.json configs[1]): the strDiffDiT-style refiner -- encoder/decoder Transformer denoiser,
d=32, 8 heads, 2 encoder + 4 decoder layers, frame_stack 10 (8 chunks of 10 trajectory tokens, 809
conditioning tokens), in-repo DDIM sampler 50 of 100 steps (eta 0, cosine schedule, pred_noise), bf16 MFMA GEMMs
(split operands by default, see --precision) / fp32 state, B = 1024 scenarios per GPU (the configured batch), random-init
weights, synthetic nuPlan-shaped conditioning already resident in HBM.  One "step" = one generate() pass over the batch:
8 chunks x 50 denoiser calls + scheduler updates, replayed from one hipGraph, plus (N > 1) the all-gather of the sampled
trajectories.

With N > 1 there is one process per GPU; every rank samples its own B scenarios (weak scaling) and the only collective is
the final all-gather.  Started by torch.distributed.run (RANK / WORLD_SIZE in the environment) the process is one rank;
started plainly with --gpus N > 1 it launches the N ranks itself (a fresh torch.distributed.run child, before this process
touches a GPU) and relays rank 0's line.

Rank 0 prints one JSON line.  ``roofline`` describes the kernel that dominates the step (measured live with events on the
launch stream); ``cpu_baseline`` times the CPU oracle on a bounded sample of BASELINE.json configs[0]; ``knee`` is the same
workload at 4096 scenarios per GPU (where every kernel fills the chip); ``dit_gemm`` is the dominant GEMM of the scaled DiT
decoder (M = 256 x 80 tokens, N = 4096, K = 1024) timed live, the second half of the metric; ``train_step`` is one optimisation
step of the same refiner on the library's training path (forward + backward + update) at the reference's training batch.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FS, N_TRAIN, N_SAMPLE, L_COND, D_MODEL, N_HEAD = 10, 100, 50, 809, 32, 8
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak


def build_model(dev, precision="bf16"):
    from autodiffusion_amd import refiner, synth
    cfg = refiner.refiner_config(frame_stack=FS, diffusion_timesteps=N_TRAIN, sampling_timesteps=N_SAMPLE if N_SAMPLE < N_TRAIN else None,
                                 objective="pred_noise", beta_schedule="cosine", ddim=0.0, normalize=True, residual=False, precision=precision)
    da = refiner.DiffusionAugmentation(cfg)
    synth.reference_init_(da, 11)
    return da.to(dev).eval()


def make_inputs(B, row0, dev, n_noise):
    """Conditioning and noise of B scenarios whose global indices start at row0 (a shard regenerates exactly its rows)."""
    from autodiffusion_amd import synth
    maps = synth.maps_info(B, 7, row0).to(dev)
    prior = synth.trajectory_prior(B, 7, row0).to(dev)
    noise = synth.noise(B, (80 // FS, n_noise), (FS, 4), 1234, row0).to(dev)      # the initial draw + what the schedule reads
    return maps, prior, noise


def time_kernel(fn, iters=20):
    """Average duration (s) of one launch, events recorded on the stream the kernel runs on."""
    for _ in range(3):
        fn(1)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    fn(iters)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / iters


def pmc_counters(kernel_prefix="enc_rows_ke", precision="bf16x3"):
    """Per-launch hardware counters of one kernel from the newest committed rocprofv3 --pmc summary (profiles/rNN_pmc_refiner.txt,
    collected at B = 1024 scenarios = 828 416 conditioning rows in separate passes per counter group).  Counter collection needs the
    profiler, so this run reads the committed figures and scales them by the row count; the line names the file."""
    import glob
    import re
    suffix = "" if precision == "bf16x3" else "_" + precision           # counters are per kernel form: one file per precision
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_pmc_refiner{suffix}.txt")))
    if not files:
        return None, {}
    out, cur = {}, None
    for ln in open(files[-1]):
        if ln.startswith("void "):
            cur = ln
            continue
        m = re.match(r"\s+(\w+)\s+n=\s*\d+\s+avg=\s*([0-9.]+)", ln)
        if m and cur and kernel_prefix in cur:
            out[m.group(1)] = float(m.group(2))
    return os.path.relpath(files[-1], ROOT), out


PMC_ROWS = 1024 * 809           # conditioning rows of the launch the committed counters were collected on
SIMDS, CLOCK_GHZ = 1024, 2.4    # 256 CUs x 4 SIMDs; peak engine clock
def gate_table_fits(B, n_steps):
    """The engine's rule for the per-chunk gate table (ADX_GATE_TABLE_MAX_MB included)."""
    budget = int(os.environ.get("ADX_GATE_TABLE_MAX_MB", "16384")) << 20
    need = n_steps * ((B * L_COND + 31) // 32) * 1024
    return 0 < need <= budget


def enc_rows_read_bytes_per_row(B, n_steps, fs, precision="bf16x3"):
    """x0 (128) + the cached attention output o / l (128) + this step's gates (32) while the per-chunk gate table fits the
    engine's budget; otherwise x0 + layer-0 query (128) + normalised state (160).  The fp16 engine keeps o / l and the gates as
    fp16 (64 + 16) when it has the gate table and the LayerNorm-row attention."""
    if not gate_table_fits(B, n_steps):
        return 416
    return 208 if precision == "f16" and (fs + 1) * N_HEAD <= 96 else 288


def enc_rows_write_bytes_per_row(B, n_steps, fs, precision="bf16x3"):
    """The layer-1 LayerNorm row (128 B; 64 B as fp16 in the fp16 engine with the gate table) while the visible rows fit the
    matrix-core attention ((fs + 1) x 8 heads <= 96); layer-1 K | V (256 B) for longer chunks, whose last attention reads stored
    keys and values."""
    if (fs + 1) * N_HEAD > 96:
        return 256
    return 64 if precision == "f16" and gate_table_fits(B, n_steps) else 128


def roofline(da, B, step, precision="bf16x3"):
    """Dominant kernel of a denoiser call: the fused encoder row pipeline (enc_rows_kernel), one launch per call.

    It streams every conditioning row once -- 416 algorithmic bytes per row (288 read, 128 written; 272 = 208 + 64 in the fp16
    engine, which keeps o / l, gates and its output as fp16), B * 809 rows per
    launch -- and does 18.4 kflop per row on the matrix cores, so of the two roofs HBM bandwidth is the one that binds
    (43 us at 8 TB/s against 6 us at the dense bf16 MFMA peak, B = 1024); what actually limits the kernel is vector
    issue (GELU and operand conversions: profiles/r02_enc_rows_ablation.txt).  Its launches are timed inside one more pass of the workload: the engine brackets
    every marked launch with events on the launch stream (no graph capture during that pass).  ``stages_us`` lists the
    other kernels of a call from the same pass.
    """
    eng = da.diffusion.diffusion.model.engine()
    names = {1: "enc_rows", 2: "last_layer_attention", 3: "tail", 4: "chunk_cache_build", 5: "map_cache_build"}
    eng.profile(True)
    try:
        eng.profile_read(-1)
        step()
        t = {}
        for i, n in names.items():
            ms, cnt = eng.profile_read(i)
            t[n] = ms * 1e-3 / max(cnt, 1)
        launches = eng.profile_read(1)[1]
        eng.profile_read(-1)
    finally:
        eng.profile(False)
    rows = B * L_COND
    n_steps = N_SAMPLE if N_SAMPLE < N_TRAIN else N_TRAIN
    wr = enc_rows_write_bytes_per_row(B, n_steps, FS, precision)
    rd = enc_rows_read_bytes_per_row(B, n_steps, FS, precision)
    algo = float((rd + wr) * rows)
    achieved = algo / t["enc_rows"] / 1e9
    line = {"kernel": "enc_rows_kernel (encoder layer 0 after its attention + layer 1 LayerNorm over all conditioning rows)", "bound": "hbm",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": None, "bytes_per_launch": algo, "launch_us": round(t["enc_rows"] * 1e6, 1),
            "launches_timed": launches, "stages_us": {n: round(v * 1e6, 1) for n, v in t.items()},
            "note": "HBM is the binding one of the two roofs the contract offers; the split-bf16 form of the kernel is limited by vector issue (see `issue`): "
                    "with no HBM traffic at all (no stores, inputs from L2) it still takes 80 of its 96 us (profiles/r02_enc_rows_ablation.txt); the fp16 form "
                    "issues a quarter fewer vector and a third of the matrix instructions and sits correspondingly closer to the memory roof"}
    src, pmc = pmc_counters(precision=precision)
    same_form = (wr, rd) == ((64, 208) if precision == "f16" else (128, 288)) and FS == 10          # the kernel form the counters were collected on
    if same_form and "TCC_EA0_RDREQ_sum" in pmc and "TCC_EA0_WRREQ_sum" in pmc:
        # HBM bytes = read requests x 128 B (FETCH_SIZE doubled: the gfx950 correction) + write requests x 64 B, scaled by rows
        line["traffic"] = round((pmc["TCC_EA0_RDREQ_sum"] * 128 + pmc["TCC_EA0_WRREQ_sum"] * 64) * rows / PMC_ROWS)
        line["traffic_source"] = f"{src} (TCC_EA0_RDREQ x 128 B + TCC_EA0_WRREQ x 64 B per launch at B=1024, scaled by the row count; not collected in this run)"
    if same_form and "SQ_INSTS_VALU" in pmc and "SQ_INSTS_MFMA" in pmc:
        # The roof that binds.  A SIMD issues one vector instruction per 4 cycles and one 32x32x16 MFMA per 32 cycles, and in this
        # kernel the two do not overlap: T >= (4 N_valu + 32 N_mfma) / (SIMDs x clock).
        valu, mfma = pmc["SQ_INSTS_VALU"] * rows / PMC_ROWS, pmc["SQ_INSTS_MFMA"] * rows / PMC_ROWS
        model_us = (4 * valu + 32 * mfma) / SIMDS / (CLOCK_GHZ * 1e3)
        line["issue"] = {"valu_insts": round(valu), "mfma_insts": round(mfma), "model_us": round(model_us, 1),
                         "frac": round(model_us / (t["enc_rows"] * 1e6), 4), "clock_ghz": CLOCK_GHZ, "simds": SIMDS,
                         "model": "(4 cycles x vector instructions + 32 cycles x matrix instructions) / (SIMDs x clock): the time the kernel's own "
                                  "instruction stream needs at full issue rate; frac = model_us / launch_us.  A lower bound of the limit: every vector "
                                  "instruction is priced at 4 cycles, while the kernel's exponentials (66 per 32-row tile and lane) issue in ~3.6x and its "
                                  "packed fp32 instructions in ~1.9x that (profiles/r03_valu_issue.txt), and the chip holds ~2.0 GHz, not 2.4, under this load",
                         "source": f"{src} (SQ_INSTS_VALU, SQ_INSTS_MFMA per launch at B=1024, scaled by the row count; not collected in this run)"}
    return line


# ------------------------------------------------------------------------------------------------ DiT decoder workloads
DIT_WORKLOADS = {           # name: (backbone, d, layers, heads, diffusion steps, default batch)
    "dit256": ("DiT", 256, 4, 8, 10, 4096),
    "mmdit256": ("MM-DiT", 256, 4, 8, 10, 2048),
    "dit1024": ("DiT", 1024, 24, 16, 200, 256),
}


def dit_macs_per_step(d, L, S=80):
    """Dense multiply-accumulates of one DiT denoiser call on one trajectory (state encoder excluded: once per scenario)."""
    x_enc = 1942272 + 768 * 2048 + 2816 * d
    return L * (26 * S * d * d + 2 * S * S * d) + S * x_enc + (2 * S * d * d + 4 * S * d) + 8 * d * d


def mmdit_macs_per_step(d, L, h, S=80):
    """The same count for the MM-DiT backbone: per block the q/k/v maps of both streams (6 S d^2 h), joint attention over
    head width 2 d (4 S^2 d h), its projection (2 S d^2 h), both modulations (12 S d^2), linear_x/c (2 S d^2), both MLPs (16 S d^2)."""
    x_enc = 1942272 + 768 * 2048 + 2816 * d
    block = 8 * S * d * d * h + 4 * S * S * d * h + 30 * S * d * d
    return L * block + S * x_enc + (2 * S * d * d + 4 * S * d) + 8 * d * d


def bench_dit(args, dev):
    """DiffusionDecoder.generate with stochastic sampling (determin=False): B trajectories x N DDPM steps per pass.
    --mc-num m: the B trajectories are m samples of each of B / m scenarios (the reference's mc_num)."""
    from autodiffusion_amd import dit, ops, synth
    backbone, d, L, h, n_steps, b_def = DIT_WORKLOADS[args.workload]
    n_steps = args.dit_steps or n_steps
    B = args.batch or b_def
    mc = max(args.mc_num, 1)
    if B % mc:
        raise SystemExit("--batch must be a multiple of --mc-num")
    cfg = dit.dit_config(d, L, h, backbone, k=mc, mc_num=mc, diffusion_timesteps=n_steps, precision=args.precision)
    dec = dit.DiffusionDecoder(cfg).to(dev).eval()
    torch.manual_seed(11)
    with torch.no_grad():
        for p in dec.parameters():          # every tensor N(0, 0.02): the reference zero-inits adaLN, which would idle the blocks
            p.normal_(0.0, 0.02)
    state = synth.hidden_state(B // mc, cfg.n_embd, 7).to(dev)
    noise = synth.noise(B, (n_steps + 1,), (80, 4), 1234).to(dev)

    def step():
        return dec.model(state, batch_size=B // mc, determin=False, noise=noise, mc_num=mc)
    for _ in range(max(args.warmup, 1)):
        out = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    assert bool(torch.isfinite(out).all())
    # dominant GEMM: the 4d-wide MLP / FFN expansions, M = B * 80 tokens, through the entry point the engine itself takes for
    # this shape: the product kernel on K-blocked operand planes when K >= 512 and the launch has >= 320 tiles of 256 x 128
    # (the producing kernels write the planes inside the decoder), the dispatching adx_op_linear otherwise (K = 256: the
    # row-stationary kernel)
    M, N, K = B * 80, 4 * d, d
    x = torch.randn(M, K, device=dev)
    w32 = torch.randn(N, K, device=dev) * 0.02
    bias = torch.zeros(N, device=dev)
    mult = {"bf16x3": 3, "f16x2": 2}.get(args.precision, 1)         # matrix instructions per product
    planes = args.precision != "fp32" and K >= 512 and K % 64 == 0 and ((M + 255) // 256) * ((N + 127) // 128) >= 320
    w = ops.prepare_weight(w32, args.precision)
    if planes:
        xp, wp, y = ops.stage_planes(x, args.precision), ops.block_weight(w, N, K, args.precision), torch.empty(M, N, device=dev)
        gemm = lambda k: [ops.linear_planes(xp, wp, bias, M, N, K, precision=args.precision, out=y) for _ in range(k)]  # noqa: E731
        kname = "linear_glds2_kernel (operand planes)"
    else:
        gemm = lambda k: [ops.linear(x, w, bias, precision=args.precision, prepared_n=N) for _ in range(k)]  # noqa: E731
        kname = "adx_op_linear dispatch (row-stationary kernel at K <= 256)"
    gemm(150)                                # an idle GPU needs about this many launches before its clocks settle
    tg = min(time_kernel(gemm, 10) for _ in range(3))
    useful = 2.0 * M * N * K / tg / 1e12
    achieved = useful * (mult if args.precision != "fp32" else 1)
    flops = 2.0 * (dit_macs_per_step(d, L) if backbone == "DiT" else mmdit_macs_per_step(d, L, h)) * B * n_steps
    line = {
        "metric": "denoised trajectories/sec (8s, 80-pt)", "value": round(B / dt, 2), "unit": "trajectories/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": {"fp32": "fp32", "f16x2": "fp16", "f16": "fp16"}.get(args.precision, "bf16"), "data": "synthetic",
        "config": {"workload": f"{backbone} decoder d={d} L={L} h={h}, {n_steps} DDPM steps (cosine), stochastic sampling, random init, "
                               "synthetic backbone hidden states" + (f", {mc} samples per scenario" if mc > 1 else ""), "batch_per_gpu": B, "global_batch": B, "parallelism": "dp1",
                   "precision": args.precision, "dense_tflops_whole_pass": round(flops / dt / 1e12, 2)},
        "roofline": {"kernel": f"{kname} M={M} N={N} K={K}", "bound": "mfma", "achieved": round(achieved, 2),
                     "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / MFMA_BF16_PEAK_TFLOPS, 4),
                     "traffic": None, "launch_us": round(tg * 1e6, 1), "useful_tflops": round(useful, 2),
                     "note": "achieved = flops the kernel executes on the matrix cores (3 MFMAs per product with split operands); "
                             "useful_tflops = 2 M N K per second"},
    }
    print(json.dumps(line), flush=True)


def bench_transition(args, dev):
    """config/strDiffDiT.yaml as written: the latent-carrying wrapper (StrDiffDiT), frame_stack 10 -> 8 chunks, latent 10 x 768,
    10 diffusers-style DDPM steps of 100 (squaredcos_cap_v2), B scenarios; all chunks x steps replayed from one captured graph.
    Parity of this wrapper is unpinned (its denoiser class is missing from the reference tree)."""
    from autodiffusion_amd import schedules, synth, transition
    B = args.batch or 1024
    cfg = transition.transition_config(frame_stack=10, z_dim=768, num_train_timesteps=100, num_inference_steps=10, precision=args.precision)
    m = transition.StrDiffDiT(cfg)
    synth.reference_init_(m, 11)
    m = m.to(dev).eval()
    maps, prior = synth.maps_info(B, 7).to(dev), (synth.trajectory_prior(B, 7) * 0.01).to(dev)
    draws = schedules.n_noise_draws(m.transition_model.schedule_rows())
    noise = synth.noise(B, (8, draws), (10, 4), 1234).to(dev)

    def step():
        return m.generate(trajectory_prior=prior, maps_info=maps, noise=noise)["traj_logits"]
    for _ in range(max(args.warmup, 1)):
        out = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    assert bool(torch.isfinite(out).all())
    print(json.dumps({
        "metric": "denoised trajectories/sec (8s, 80-pt)", "value": round(B / dt, 2), "unit": "trajectories/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "fp32" if args.precision == "fp32" else "bf16", "data": "synthetic",
        "config": {"workload": "strDiffDiT.yaml literal: latent-carrying refiner (enc/dec Transformer d=32 with a 768-wide latent head), frame_stack 10 x 8 "
                               "chunks, 10 DDPM steps of 100, random init, synthetic nuPlan-shaped conditioning", "batch_per_gpu": B, "global_batch": B,
                   "denoiser_calls_per_step": 80, "parallelism": "dp1", "precision": args.precision,
                   "captures": m.transition_model.diffusion.engine().capture_count()}}), flush=True)


def bench_str(args, dev):
    """Phase 1 of generate(): rasters -> trajectory prior + map tokens through STR_Mixtral (ViT raster encoder, the 'mixtral-small'
    mixture-of-experts decoder with its autoregressive key-point loop, trajectory head), then optionally the refiner on top
    (--with-refiner).  B = the reference's evaluation batch per device (8) unless --batch is given."""
    from autodiffusion_amd import refiner, str_backbone, synth
    B = args.batch or 8
    prec = args.precision
    model = synth.fill_module_(str_backbone.STR_Mixtral(str_backbone.str_config("mixtral-small", precision=prec)), 11).to(dev).eval()
    batch = dict(high_res_raster=synth.occupancy_raster(B, 58, 7).to(dev), low_res_raster=synth.occupancy_raster(B, 58, 8).to(dev),
                 context_actions=(synth.noise(B, (), (4, 4), 1241) * torch.tensor([2.0, 0.3, 0.0, 0.1])).to(dev),
                 trajectory_label=synth.trajectory_prior(B, 7).to(dev))

    def timed(fn):
        for _ in range(max(args.warmup, 1)):
            out = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.steps, out
    dt, out = timed(lambda: model.generate(**batch))
    assert bool(torch.isfinite(out["traj_logits"]).all())
    enc_dt, _ = timed(lambda: model.encoder(is_training=False, **batch))
    graphed = str_backbone.GraphedGenerate(model)
    gdt, gout = timed(lambda: graphed(**batch))
    assert torch.equal(gout["traj_logits"], out["traj_logits"])
    line = {
        "metric": "trajectory priors/sec (phase 1 of generate())", "value": round(B / dt, 2), "unit": "trajectories/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "fp32" if prec == "fp32" else "bf16", "data": "synthetic",
        "config": {"workload": "STR_Mixtral.generate, mixtral-small (n_embd 256, 4 layers, 8 heads, 8 experts of width 1024, top 2), ViT raster "
                               "encoder (hidden 128, 12 layers, 2 x 4 x 197 tokens per scenario), 5 autoregressive key points with cached keys, "
                               "873 tokens per scenario, random init, synthetic rasters", "batch_per_gpu": B, "global_batch": B, "parallelism": "dp1",
                   "precision": prec},
        "stages_ms": {"raster_encoder": round(enc_dt * 1e3, 3), "decoder_and_heads": round((dt - enc_dt) * 1e3, 3)},
        "graph_replay": {"ms_per_step": round(gdt * 1e3, 3), "value": round(B / gdt, 2), "unit": "trajectories/s",
                         "note": "the same call captured once per input shape and replayed (str_backbone.GraphedGenerate): same bits"},
    }
    if not args.no_cpu_baseline:
        from oracle import str_backbone as osb
        sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
        ocfg = dict(n_layer=4, n_head=8, n_kv_head=8, n_experts=8, top_k=2, rms_eps=1e-5, rope_theta=1e6)
        cb = {k: v[:1].cpu() for k, v in batch.items()}
        with torch.no_grad():
            t0 = time.perf_counter()
            osb.str_generate(sd, ocfg, cb["high_res_raster"], cb["low_res_raster"], cb["context_actions"], cb["trajectory_label"], [79, 39, 19, 9, 4])
            cdt = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": round(1.0 / cdt, 3), "unit": "trajectories/s", "cores": torch.get_num_threads(), "kind": "port",
                                "sample": f"fp32 CPU oracle of the same model on ONE scenario, the reference's six full decoder passes ({cdt:.2f} s)"}
    if args.with_refiner:
        da = refiner.DiffusionAugmentation(refiner.refiner_config(frame_stack=10, diffusion_timesteps=100, sampling_timesteps=50, objective="pred_noise",
                                                                    beta_schedule="cosine", precision="bf16x3"), str_model=model)
        synth.reference_init_(da.diffusion, 11)
        da = da.to(dev).eval()
        rdt, rout = timed(lambda: da.generate(seed=1, **batch))
        assert bool(torch.isfinite(rout["traj_logits"]).all())
        line["with_refiner"] = {"ms_per_step": round(rdt * 1e3, 3), "value": round(B / rdt, 2), "unit": "trajectories/s",
                                "note": "rasters -> refined trajectories: phase 1 + the refiner's chunk loop (frame_stack 10, DDIM 50 of 100), device noise"}
    print(json.dumps(line), flush=True)


def cpu_baseline():
    """BASELINE.json configs[0] on the host cores with the fp32 CPU oracle: B = 16, frame_stack 10, 100-step DDPM (cosine,
    pred_noise).  Bounded sample: ONE of the 8 chunks with all 100 steps on every allowed thread (the 8 chunks are
    sequential and cost the same, so the whole job is 8x that), plus 10 steps of the same chunk on one thread."""
    from autodiffusion_amd import refiner, synth
    from oracle import nets, sampling, schedules as osched
    B, fs, N = 16, 10, 100
    cfg = refiner.refiner_config(frame_stack=fs, diffusion_timesteps=N, objective="pred_noise", beta_schedule="cosine", precision="fp32")
    da = refiner.DiffusionAugmentation(cfg)
    synth.reference_init_(da, 11)
    sd = {k[len("diffusion.diffusion.model."):]: v.detach().float() for k, v in da.state_dict().items()
          if k.startswith("diffusion.diffusion.model.")}
    maps = synth.maps_info(B, 7)
    prior = synth.trajectory_prior(B, 7)[:, :fs].contiguous()
    noise = synth.noise(B, (N,), (fs, 4), 1234)
    bufs = osched.buffers("cosine", N, "pred_noise")
    n_chunks = 80 // fs

    def fn(x, t, tra_n, pri_n):
        tt = torch.full((x.shape[0],), int(t), dtype=torch.long)
        return nets.variant_a_forward(sd, x, tt, maps, tra_n, pri_n)
    threads = torch.get_num_threads()
    with torch.no_grad():
        fn(noise[0], 5, prior, prior)     # warm-up call
        t0 = time.perf_counter()
        sampling.refiner_sample(fn, bufs, "pred_noise", prior * 0.5, prior, noise)      # DDPM: all 100 steps of one chunk
        dt = time.perf_counter() - t0
        torch.set_num_threads(1)
        try:
            fn(noise[0], 5, prior, prior)
            t0 = time.perf_counter()
            for k in range(10):
                fn(noise[k], 50 + k, prior, prior)
            dt1 = (time.perf_counter() - t0) / 10
        finally:
            torch.set_num_threads(threads)
    dit_row = cpu_baseline_dit()
    return {"dit256": dit_row, "value": round(B / (dt * n_chunks), 4), "unit": "trajectories/s", "cores": threads, "kind": "port",
            "sample": f"fp32 CPU oracle on configs[0] (B=16, frame_stack 10, 100-step DDPM): 1 of {n_chunks} chunks x {N} steps timed "
                      f"({dt:.1f} s on {threads} threads), whole job = {n_chunks}x",
            "ms_per_denoiser_call": round(dt / N * 1e3, 1),
            "one_thread": {"value": round(B / (dt1 * N * n_chunks), 4), "cores": 1, "ms_per_denoiser_call": round(dt1 * 1e3, 1),
                           "sample": "10 denoiser calls of the same chunk on 1 thread, scaled to 8 chunks x 100 steps"}}


def cpu_baseline_dit(B=16, N=100, sample_steps=10):
    """The GEMM-heavy variant beside it: the decoder twin, DiT hidden 256 / 4 blocks / 8 heads, 100-step DDPM (cosine), B = 16,
    deterministic sampling (k = 1), fp32 CPU oracle on every allowed thread.  Bounded sample: `sample_steps` of the 100 steps are
    timed (every step costs the same: one denoiser call + the scheduler update), the whole job is N / sample_steps times that."""
    from autodiffusion_amd import dit, synth
    from oracle import nets
    dec = dit.DiffusionDecoder(dit.dit_config(256, 4, 8, "DiT", diffusion_timesteps=N))
    synth.fill_module_(dec, 11)
    sd = {k[len("model.model."):]: v.detach().float() for k, v in dec.state_dict().items() if k.startswith("model.model.")}
    state = synth.hidden_state(B, 256, 7)
    x = synth.noise(B, (), (80, 4), 1234)
    threads = torch.get_num_threads()
    with torch.no_grad():
        nets.dit_forward(sd, x, torch.full((B,), 5, dtype=torch.long), state, 8, "DiT")     # warm-up call
        t0 = time.perf_counter()
        for k in range(sample_steps):
            x = x - 0.01 * nets.dit_forward(sd, x, torch.full((B,), N - 1 - k, dtype=torch.long), state, 8, "DiT")
        dt = (time.perf_counter() - t0) / sample_steps
    return {"value": round(B / (dt * N), 4), "unit": "trajectories/s", "cores": threads, "kind": "port", "ms_per_denoiser_call": round(dt * 1e3, 1),
            "sample": f"fp32 CPU oracle, DiT d=256 L=4 h=8, B={B}, {N}-step DDPM: {sample_steps} denoiser calls timed on {threads} threads, "
                      f"whole job = {N} calls"}


def dit_gemm(dev):
    """The dominant GEMM of the scaled DiT decoder (BASELINE.json configs[4]: hidden 1024, 256 scenarios x 80 tokens): the MLP /
    FFN expansion M = 20480, N = 4096, K = 1024 through the library's staged entry point, both bf16 operand modes, timed with
    events on the launch stream.  ``us`` is the product kernel on operands already in K-blocked bf16 planes -- what runs inside
    the decoder, whose producing kernels write the planes; ``with_relayout`` adds the two kernels that turn fp32 X and bf16 W
    into planes, as a one-off caller of adx_op_linear_staged pays.  ``tflops`` is the product's 2 M N K per second; ``executed_tflops`` counts the MFMAs the
    kernel issues (three per product with split operands) and ``frac`` prices that against the dense bf16 MFMA peak."""
    from autodiffusion_amd import ops
    M, N, K = 20480, 4096, 1024
    x = torch.randn(M, K, device=dev)
    w32 = torch.randn(N, K, device=dev) * 0.02
    bias = torch.zeros(N, device=dev)
    out = {"shape": {"M": M, "N": N, "K": K}, "peak_tflops": MFMA_BF16_PEAK_TFLOPS}
    for prec, mult in (("bf16x3", 3), ("f16x2", 2), ("f16", 1), ("bf16", 1)):      # f16x2 / f16: fp16 operand planes, two / one matrix instructions per product (same peak rate as bf16)
        w = ops.prepare_weight(w32, prec)
        xp, wp = ops.stage_planes(x, prec), ops.block_weight(w, N, K, prec)
        y = torch.empty(M, N, device=dev)
        product = lambda k: [ops.linear_planes(xp, wp, bias, M, N, K, precision=prec, out=y) for _ in range(k)]  # noqa: E731
        product(200)                                         # an idle GPU needs ~150 of these launches (60 ms) before its clocks settle (profiles/gemm_warm_probe.py)
        t = min(time_kernel(product, 20) for _ in range(3))
        tf = 2.0 * M * N * K / t / 1e12
        out[prec] = {"us": round(t * 1e6, 1), "tflops": round(tf, 1), "executed_tflops": round(tf * mult, 1),
                     "frac": round(tf * mult / MFMA_BF16_PEAK_TFLOPS, 4)}
        if prec in ("bf16x3", "bf16"):                       # the one-off caller's form (adx_op_linear_staged) exists for the bf16 modes
            ws = torch.empty(ops.linear_stage_bytes(M, N, K, prec) + 64, dtype=torch.uint8, device=dev)
            whole = lambda k: [ops.linear(x, w, bias, precision=prec, prepared_n=N, workspace=ws) for _ in range(k)]  # noqa: E731
            t_whole = min(time_kernel(whole, 20) for _ in range(3))
            tf_whole = 2.0 * M * N * K / t_whole / 1e12
            out[prec]["with_relayout"] = {"us": round(t_whole * 1e6, 1), "executed_tflops": round(tf_whole * mult, 1),
                                          "frac": round(tf_whole * mult / MFMA_BF16_PEAK_TFLOPS, 4)}
            del ws
        del xp, wp, y
    return out


def train_step(dev, B=64, steps=5):
    """One optimisation step of the refiner on the HIP training path at the reference's training batch (script/trainingRefiner.sh:
    per_device_train_batch_size 64): DiffusionAugmentation.forward in gradient mode (the 8 chunks as one batch: q_sample, the
    denoiser with kept activations, min-SNR weighted loss), loss.backward() through the library's backward kernels, SGD update.
    fp32, dropout active as in the reference's training (p = 0.1 on embeddings, attention probabilities and the layers' outputs).
    Reported beside the sampling metric; it is not part of ``value``."""
    from autodiffusion_amd import refiner, synth
    cfg = refiner.refiner_config(frame_stack=FS, diffusion_timesteps=N_TRAIN, objective="pred_noise", beta_schedule="cosine")
    da = refiner.DiffusionAugmentation(cfg)
    synth.reference_init_(da, 11)
    da = da.to(dev).train()
    maps, prior = synth.maps_info(B, 7).to(dev), synth.trajectory_prior(B, 7).to(dev)
    label = prior + synth.noise(B, (), (80, 4), 1235).to(dev) * torch.tensor([1.5, 0.4, 0.0, 0.05], device=dev)
    opt = torch.optim.SGD(da.parameters(), lr=1e-4)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = da(trajectory_prior=prior, maps_info=maps, trajectory_label=label).loss
        loss.backward()
        opt.step()
        return loss
    for _ in range(2):
        loss = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    if not bool(torch.isfinite(loss.detach())):
        raise RuntimeError("training step produced a non-finite loss")
    return {"workload": f"refiner training step (forward + backward + SGD), frame_stack {FS} x {80 // FS} chunks, B={B}, fp32",
            "ms_per_step": round(dt * 1e3, 2), "samples_per_s": round(B / dt, 1), "steps": steps}


def launch_ranks(args):
    """--gpus N without a launcher: start the N ranks as a fresh child (never from a process that initialised a GPU) and
    relay its output; rank 0 of the child prints the JSON line."""
    from autodiffusion_amd.distributed import free_port
    port = free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def dry_run(args, rank, world):
    """No GPU: the launch, sharding and gather logic of the real run with gloo on the CPU.  Every rank "samples" its shard of
    a ragged total as a function of the global row index; the gathered tensor must equal the single-process one."""
    import torch.distributed as dist
    from autodiffusion_amd import synth
    from autodiffusion_amd.distributed import RowGatherer, shard_range
    if world > 1:
        dist.init_process_group("gloo")
    total = 5 * world + (1 if world > 1 else 0)              # ragged on purpose
    lo, hi = shard_range(total, rank, world)
    gather = RowGatherer(total, (80, 4), "cpu")
    t0 = time.perf_counter()
    for _ in range(max(args.steps, 1)):
        full = gather.gather(synth.noise(hi - lo, (), (80, 4), 1234, row0=lo))
    dt = time.perf_counter() - t0
    ok = bool(torch.equal(full, synth.noise(total, (), (80, 4), 1234)))
    if rank == 0:
        print(json.dumps({"metric": "dry run (no GPU): launch + shard + gather", "value": round(total * args.steps / dt, 2),
                          "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "gathered_equals_single_process": ok,
                          "config": {"workload": "gloo dry run", "global_batch": total, "parallelism": f"dp{world}"}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=0, help="scenarios per GPU (default: 1024 for the refiner, the configured batch)")
    ap.add_argument("--workload", default="refiner", choices=["refiner", "strdiffdit", "str"] + sorted(DIT_WORKLOADS),
                    help="refiner = the headline configuration; the DiT decoder workloads run on one GPU only")
    ap.add_argument("--frame-stack", type=int, default=0, help="refiner: points per chunk (default 10; 80 = whole trajectory)")
    ap.add_argument("--n-train", type=int, default=0, help="refiner: training timesteps (default 100)")
    ap.add_argument("--n-sample", type=int, default=0, help="refiner: sampling steps (default 50; equal to --n-train = DDPM loop)")
    ap.add_argument("--mc-num", type=int, default=1, help="DiT workloads: samples per scenario (--batch counts trajectories)")
    ap.add_argument("--dit-steps", type=int, default=0, help="override the number of diffusion steps of a DiT workload (profiling)")
    ap.add_argument("--precision", default=None, choices=["bf16x3", "bf16", "fp32", "f16x2", "f16"],
                    help="default: f16 for the refiner (its per-step kernels on fp16 operands, the rest split-bf16), bf16x3 for the other workloads; "
                         "f16x2 / f16 on the DiT workloads: fp16 operand planes, two / one matrix instructions per product")
    ap.add_argument("--with-refiner", action="store_true", help="str workload: also time phase 1 + refiner end to end")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the knee and dit_gemm measurements (profiling runs)")
    ap.add_argument("--dry-run", action="store_true", help="no GPU: exercise launch, sharding and gather with gloo on the CPU")
    args = ap.parse_args()

    if args.precision is None:
        args.precision = "f16" if args.workload == "refiner" else "bf16x3"
    if args.gpus < 1:
        raise SystemExit("--gpus must be positive")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args))                  # nothing below ran in this process: no GPU was touched
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks")
    if args.dry_run:
        raise SystemExit(dry_run(args, rank, world))
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # ADX_BENCH_DIST=1: join the process group even as a single rank, so that a one-GPU box exercises RCCL's set-up, barrier,
    # all-reduce and all-gather exactly as an N-rank run does
    use_dist = world > 1 or (os.environ.get("ADX_BENCH_DIST") == "1" and "RANK" in os.environ)
    if use_dist:
        dist.init_process_group("nccl", device_id=dev)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the process group has {dist.get_world_size()} ranks")

    global FS, N_TRAIN, N_SAMPLE, L_COND
    FS, N_TRAIN, N_SAMPLE = args.frame_stack or FS, args.n_train or N_TRAIN, args.n_sample or N_SAMPLE
    L_COND = 1 + 788 + 2 * FS
    if args.workload != "refiner":
        if world > 1:
            raise SystemExit("the DiT and latent-carrying workloads are single-GPU measurements")
        if args.workload == "str":
            return bench_str(args, dev)
        return bench_transition(args, dev) if args.workload == "strdiffdit" else bench_dit(args, dev)

    from autodiffusion_amd import schedules
    from autodiffusion_amd.distributed import RowGatherer
    da = build_model(dev, args.precision)
    n_noise = 1 + schedules.n_noise_draws(da.diffusion.diffusion.schedule_rows())

    def measure(B, steps, warmup):
        """K timed generate() passes (+ gather) over B scenarios per rank; returns (seconds, step function, gathered rows)."""
        maps, prior, noise = make_inputs(B, rank * B, dev, n_noise)
        out = torch.empty(B, 80, 4, device=dev)
        gatherer = RowGatherer(B * world, (80, 4), dev, always_collective=use_dist)

        def sample():
            return da.generate(trajectory_prior=prior, maps_info=maps, noise=noise, out=out)["traj_logits"]

        def step():
            return gatherer.gather(sample())          # enqueued behind the sampling graph on the same stream

        def fence():
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()

        for _ in range(max(warmup, 1)):      # the first call captures the graph
            gathered = step()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            gathered = step()
        fence()
        dt = time.perf_counter() - t0
        if use_dist:
            tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        assert gathered.shape[0] == B * world and bool(torch.isfinite(gathered).all())
        return dt, sample

    B = args.batch or 1024
    total = B * world
    dt, sample = measure(B, args.steps, args.warmup)

    if rank == 0:
        line = {
            "metric": "denoised trajectories/sec (8s, 80-pt)", "value": round(total * args.steps / dt, 2), "unit": "trajectories/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": {"fp32": "fp32", "f16": "f16"}.get(args.precision, "bf16"), "data": "synthetic",
            "config": {"workload": f"strDiffDiT-style refiner (enc/dec Transformer d=32, {L_COND} cond tokens), frame_stack {FS} x {80 // FS} chunks, "
                                   + (f"DDIM {N_SAMPLE} of {N_TRAIN} steps" if N_SAMPLE < N_TRAIN else f"DDPM {N_TRAIN} steps")
                                   + ", random init, synthetic nuPlan-shaped conditioning",
                       "batch_per_gpu": B, "global_batch": total, "denoiser_calls_per_step": (80 // FS) * N_SAMPLE,
                       "parallelism": f"dp{world}",
                       "precision": {"bf16x3": "bf16 MFMA on split (hi+lo) operands, 3 MFMAs per product, f32 accumulate; the fused tail (decoder, scheduler step) on split-fp16 "
                                               "operands (fp16 hi + scaled lo, 3 MFMAs per product, ~2^-21 per product)",
                                     "bf16": "bf16 MFMA operands, f32 accumulate; trajectory path and scheduler f32",
                                     "f16": "split-bf16 engine whose two per-step kernels over all conditioning rows (encoder row pipeline, last encoder attention) "
                                            "multiply operands rounded once to fp16: one f16 MFMA per product, f32 accumulate; the fused tail (rest of the last encoder layer on "
                                            "the visible rows, decoder, head) on split-fp16 operands (fp16 hi + scaled lo, 3 MFMAs per product, ~2^-21 per product); "
                                            "per-chunk caches and the scheduler step f32",
                                     "fp32": "f32 MFMA"}[args.precision]},
        }
        if FS == 10:                                             # the fused kernels the roofline entry describes need T + 1 <= 16
            line["roofline"] = roofline(da, B, sample, args.precision)           # rank 0 only: the pass runs no collective
    if world == 1 and not args.no_extras and not args.batch and FS == 10:
        B2 = 4096
        dt2, sample2 = measure(B2, 3, 1)
        knee = {"batch_per_gpu": B2, "value": round(B2 * 3 / dt2, 2), "unit": "trajectories/s", "ms_per_step": round(dt2 / 3 * 1e3, 3),
                "note": "same workload at 4x the configured batch: every per-step kernel fills the 256 CUs"}
        r2 = roofline(da, B2, sample2, args.precision)
        knee["roofline_frac"], knee["launch_us"] = r2["frac"], r2["launch_us"]
        line["knee"] = knee
        if args.precision != "bf16x3":           # the same workload with every product on split-bf16 operands, for comparison
            da_main, da = da, build_model(dev, "bf16x3")
            dt3, sample3 = measure(B, 3, 1)
            r3 = roofline(da, B, sample3, "bf16x3")
            line["bf16x3"] = {"value": round(B * 3 / dt3, 2), "unit": "trajectories/s", "ms_per_step": round(dt3 / 3 * 1e3, 3), "roofline_frac": r3["frac"],
                              "launch_us": r3["launch_us"], "stages_us": r3["stages_us"],
                              "note": "precision bf16x3 (3 bf16 MFMAs per product in the kernels over all conditioning rows): ADE against fp32 4.7e-4 m median / "
                                      "1.2e-3 m worst over 1024 scenarios; the headline precision: 1.06e-3 / 3.7e-3 m (gate 1e-2 m, profiles/tail_forms.py)"}
            da = da_main
        line["dit_gemm"] = dit_gemm(dev)
        line["train_step"] = train_step(dev)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
