import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
dev = torch.device("cuda:0")
da = bench.build_model(dev, "bf16x3")
B = 1024
maps, prior, noise = bench.make_inputs(B, 0, dev)
for i in range(4):
    m2, p2, n2 = maps.clone(), prior.clone(), noise.clone()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = da.generate(trajectory_prior=p2, maps_info=m2, noise=n2)["traj_logits"]
    torch.cuda.synchronize(); print(f"fresh tensors, call {i}: {1e3*(time.perf_counter()-t0):.1f} ms")
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = da.generate(trajectory_prior=prior, maps_info=maps, noise=noise, out=out)["traj_logits"]
    torch.cuda.synchronize(); print(f"same tensors, call {i}: {1e3*(time.perf_counter()-t0):.1f} ms")
