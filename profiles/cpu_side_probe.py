import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
bench.FS, bench.N_SAMPLE = 80, 10
bench.L_COND = 1 + 788 + 160
dev = torch.device("cuda:0")
da = bench.build_model(dev, "bf16x3")
B = 512
maps, prior, noise = bench.make_inputs(B, 0, dev)
out = torch.empty(B, 80, 4, device=dev)
for _ in range(3):
    da.generate(trajectory_prior=prior, maps_info=maps, noise=noise, out=out)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    da.generate(trajectory_prior=prior, maps_info=maps, noise=noise, out=out)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host call {1e3 * (t1 - t0):.2f} ms, until done {1e3 * (t2 - t0):.2f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    da.generate(trajectory_prior=prior, maps_info=maps, noise=noise, out=out)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
