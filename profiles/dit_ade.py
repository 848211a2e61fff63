import sys, torch
sys.path.insert(0, "/root/repo")
from autodiffusion_amd import dit, synth
dev = torch.device("cuda:0")
for backbone in ("DiT", "MM-DiT"):
    outs = {}
    for prec in ("fp32", "bf16x3", "bf16"):
        cfg = dit.dit_config(256, 4, 8, backbone, diffusion_timesteps=10, precision=prec)
        dec = dit.DiffusionDecoder(cfg).to(dev).eval()
        torch.manual_seed(11)
        with torch.no_grad():
            for p in dec.parameters():
                p.normal_(0.0, 0.02)
        B = 64
        state = synth.hidden_state(B, 256, 7).to(dev)
        noise = synth.noise(B, (11,), (80, 4), 1234).to(dev)
        outs[prec] = dec.model(state, batch_size=B, determin=False, noise=noise).cpu()
    for prec in ("bf16x3", "bf16"):
        d = outs[prec] - outs["fp32"]
        ade = torch.sqrt(d[..., 0] ** 2 + d[..., 1] ** 2).mean().item()
        print(backbone, prec, "ADE vs fp32 engine: %.3e m" % ade, "max |x|", outs["fp32"].abs().max().item())
