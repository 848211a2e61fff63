#!/usr/bin/env python3
"""Time one training step of the refiner on the HIP training path: DiffusionAugmentation.forward (every chunk: q_sample,
one denoiser call with kept activations, weighted loss) + loss.backward() + an SGD update, at the reference's training batch
(script/trainingRefiner.sh: per_device_train_batch_size 64).  Prints one JSON line per configuration; --cpu also times the
CPU oracle under torch autograd on one chunk of the same batch (the reference's own way of getting these gradients).

    python profiles/train_step.py [--batch 64] [--steps 10] [--cpu]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--cpu", action="store_true")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    from autodiffusion_amd import refiner, synth
    dev = torch.device("cuda:0")
    B = args.batch
    for tag, fs, N, obj, sched, residual in (("fs10", 10, 100, "pred_noise", "cosine", False), ("fs80", 80, 10, "pred_x0", "sigmoid", True)):
        if args.only and args.only != tag:
            continue
        da = refiner.DiffusionAugmentation(refiner.refiner_config(frame_stack=fs, diffusion_timesteps=N, objective=obj, beta_schedule=sched,
                                                                  residual=residual))
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
        for _ in range(3):
            loss = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        assert bool(torch.isfinite(loss))
        line = {"workload": f"refiner training step, frame_stack {fs} x {80 // fs} chunks, {obj}, {sched}, N={N}, B={B}, fp32",
                "ms_per_step": round(dt * 1e3, 2), "samples_per_s": round(B / dt, 1), "denoiser_fwd_bwd_per_step": 80 // fs, "loss": float(loss.detach())}
        if args.cpu:
            from oracle import nets, sampling, schedules
            sd = {k[len("diffusion.diffusion.model."):]: v.detach().float().cpu() for k, v in da.state_dict().items()
                  if k.startswith("diffusion.diffusion.model.")}
            b = schedules.buffers(sched, N, obj)
            mc, pc, lc = maps.cpu(), prior.cpu()[:, :fs].contiguous(), label.cpu()[:, :fs].contiguous()
            t = torch.randint(0, N, (B,))
            nz = torch.randn(B, fs, 4)
            sampling.refiner_loss_grads(sd, nets.variant_a_forward, b, obj, mc, pc, pc, lc, t, nz)
            t0 = time.perf_counter()
            sampling.refiner_loss_grads(sd, nets.variant_a_forward, b, obj, mc, pc, pc, lc, t, nz)
            dc = time.perf_counter() - t0
            line["cpu_autograd"] = {"ms_per_chunk": round(dc * 1e3, 1), "samples_per_s": round(B / (dc * (80 // fs)), 2),
                                    "threads": torch.get_num_threads(), "sample": "1 chunk forward + backward of the fp32 oracle, whole step = chunks x that"}
        print(json.dumps(line), flush=True)
    for tag, backbone in (("dit256", "DiT"), ("mmdit256", "MM-DiT")):
        if args.only and args.only != tag:
            continue
        # decoder twin as ldr_diffusion.yaml sizes it (hidden 256, 4 blocks, 8 heads), the reference's debug batch of 32
        from autodiffusion_amd import dit
        Bd = min(B, 32)
        dec = dit.DiffusionDecoder(dit.dit_config(256, 4, 8, backbone, diffusion_timesteps=10))
        with torch.no_grad():
            for p in dec.parameters():
                p.normal_(0.0, 0.02)
        dec = dec.to(dev).train()
        state = synth.hidden_state(Bd, 256, 7).to(dev)
        label = synth.trajectory_prior(Bd, 7).to(dev)
        opt = torch.optim.SGD(dec.parameters(), lr=1e-9)      # timing only: the summed loss of random weights has gradients of norm 1e4

        def dstep():
            opt.zero_grad(set_to_none=True)
            loss, _ = dec(label, state)
            loss.backward()
            opt.step()
            return loss
        for _ in range(3):
            loss = dstep()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = dstep()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        assert bool(torch.isfinite(loss.detach()))
        print(json.dumps({"workload": f"{backbone} decoder twin training step, hidden 256, 4 blocks, 8 heads, B={Bd}, fp32",
                          "ms_per_step": round(dt * 1e3, 2), "samples_per_s": round(Bd / dt, 1)}), flush=True)


if __name__ == "__main__":
    main()
