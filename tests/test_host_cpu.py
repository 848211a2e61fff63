"""CPU-only checks of the host side: schedule tables, C-ABI surface, module layout, sharding."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from autodiffusion_amd import _lib, schedules
from oracle import sampling, schedules as osched

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def apply_row(row, x, out, z):
    """The scheduler update of include/adx.h (adx_step_coef), in float32 torch."""
    t, a0, a1, clip, eps_mode, b0, b1, c0, c1, c2, c3, zi = [np.float32(v) for v in row]
    f = lambda v: torch.tensor(v, dtype=torch.float32)  # noqa: E731
    x0 = (f(a0) * x + f(a1) * out).clamp(-float(clip), float(clip))
    eps = (f(b0) * x - x0) * f(b1) if eps_mode != 0 else out
    r = f(c0) * x0 + f(c1) * x + f(c2) * eps
    if c3 != 0:
        r = r + f(c3) * z
    return r


def run_rows(rows, denoise, noise):
    x = noise[0].clone()
    for row in rows:
        zi = int(row[11])
        x = apply_row(row, x, denoise(x, int(row[0])), noise[1 + zi] if zi >= 0 else None)
    return x


def fake_denoiser(x, t):
    return torch.tanh(x * 0.7 + 0.01 * t) - 0.2 * x


@pytest.mark.parametrize("sched", ["sigmoid", "cosine", "linear"])
@pytest.mark.parametrize("n", [10, 50, 100, 200])
def test_product_buffers_match_golden(golden, sched, n):
    g = golden("g1_schedules")
    b = schedules.diffusion_buffers(schedules.make_betas(sched, n))
    for k in schedules.BUFFER_NAMES:
        np.testing.assert_allclose(b[k].numpy(), g[f"{sched}_{n}_{k}"], rtol=2e-6, atol=1e-12, err_msg=k)


@pytest.mark.parametrize("obj", ["pred_noise", "pred_x0", "pred_v"])
def test_ddpm_rows_reproduce_the_loop(obj):
    n = 20
    b = schedules.diffusion_buffers(schedules.make_betas("cosine", n))
    rows = schedules.ddpm_rows(b, obj)
    assert [int(r[0]) for r in rows] == list(reversed(range(n)))
    assert schedules.n_noise_draws(rows) == n - 1 and rows[-1][11] == -1
    noise = torch.randn(n, 3, 10, 4, generator=torch.Generator().manual_seed(0))
    got = run_rows(rows, fake_denoiser, noise)
    ref = sampling.ddpm_sample(fake_denoiser, osched.buffers("cosine", n, obj), obj, noise)
    assert torch.allclose(got, ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("obj", ["pred_noise", "pred_x0", "pred_v"])
@pytest.mark.parametrize("eta", [0.0, 1.0])
def test_ddim_rows_reproduce_the_loop(obj, eta):
    n, ns = 100, 50
    b = schedules.diffusion_buffers(schedules.make_betas("cosine", n))
    rows = schedules.ddim_rows(b, obj, ns, eta)
    # every step but the last draws a tensor in the reference; with eta = 0 the draws are multiplied by zero and get no index
    assert len(rows) == ns and schedules.n_noise_draws(rows) == (ns - 1 if eta else 0)
    assert schedules.ddim_time_pairs(10, 5) == sampling.ddim_pairs(10, 5)
    noise = torch.randn(ns, 3, 10, 4, generator=torch.Generator().manual_seed(0))
    got = run_rows(rows, fake_denoiser, noise)
    ref = sampling.ddim_sample(fake_denoiser, osched.buffers("cosine", n, obj), obj, noise, ns, eta)
    assert torch.allclose(got, ref, rtol=2e-5, atol=2e-5)


def test_twin_and_diffusers_rows():
    n = 10
    b = schedules.diffusion_buffers(schedules.cosine_beta_schedule(n))
    ob = osched.buffers_from_betas(osched.twin_betas("cosine", n))
    noise = torch.randn(n + 1, 2, 80, 4, generator=torch.Generator().manual_seed(0))
    rows = schedules.ddpm_rows(b, "pred_noise", clamp=100.0, noise_at_every_step=True, stochastic=True)
    assert schedules.n_noise_draws(rows) == n
    got = sampling.denormalize(run_rows(rows, fake_denoiser, noise))
    ref, _ = sampling.twin_sample(fake_denoiser, ob, (2, 80, 4), noise, determin=False)
    assert torch.allclose(got, ref, rtol=1e-5, atol=1e-4)
    rows = schedules.ddpm_rows(b, "pred_noise", clamp=100.0, noise_at_every_step=True, stochastic=False)
    assert schedules.n_noise_draws(rows) == 0
    got = sampling.denormalize(run_rows(rows, fake_denoiser, torch.zeros(1, 2, 80, 4)))
    ref, _ = sampling.twin_sample(fake_denoiser, ob, (2, 80, 4), None, determin=True)
    assert torch.allclose(got, ref, rtol=1e-5, atol=1e-4)
    # diffusers-style rows against the oracle's step functions
    betas = schedules.diffusers_betas("squaredcos_cap_v2", 100)
    ac = np.cumprod(1.0 - osched.squaredcos_cap_v2(100).astype(np.float64))
    for kind in ("ddim", "ddpm"):
        rows = schedules.diffusers_rows(betas, 10, kind)
        assert [int(r[0]) for r in rows] == sampling.diffusers_timesteps(100, 10)
        x = noise[0, :, :1, :].clone()
        xr = x.clone()
        k = 1
        for row in rows:
            t = int(row[0])
            eps = fake_denoiser(x, t)
            zi = int(row[11])
            x = apply_row(row, x, eps, noise[1 + zi, :, :1, :] if zi >= 0 else None)
            epr = fake_denoiser(xr, t)
            if kind == "ddim":
                xr = sampling.diffusers_ddim_step(ac, 10, epr, t, xr)
            else:
                xr = sampling.diffusers_ddpm_step(ac, 10, epr, t, xr, noise[k, :, :1, :])
                k += 1
        assert torch.allclose(x, xr, rtol=1e-4, atol=1e-5), kind


def test_library_exports_every_declared_symbol():
    """libadx.so loads without a GPU and exports exactly what include/adx.h declares."""
    assert os.path.exists(_lib.LIB_PATH), "build libadx.so first: python -m autodiffusion_amd.build"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    header = open(os.path.join(ROOT, "include", "adx.h")).read()
    declared = sorted(set(re.findall(r"\b(adx_[a-z0-9_]+)\s*\(", header)))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in adx.h but not exported"
    assert sorted(declared) == sorted(_lib.EXPORTS)
    lib.adx_abi_version.restype = ctypes.c_int
    assert lib.adx_abi_version() == 1
    assert ctypes.sizeof(_lib.StepCoef) == 48 and ctypes.sizeof(_lib.ModelConfig) == 17 * 4


def test_product_does_not_import_the_oracle():
    """No CPU fallback: the package never references oracle/ and fails loudly without a GPU."""
    pkg = os.path.join(ROOT, "autodiffusion_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "import oracle" not in src and "from oracle" not in src, fn
    if not torch.cuda.is_available():
        from autodiffusion_amd.engine import Engine
        with pytest.raises(_lib.AdxError, match="no GPU"):
            Engine(variant=0)


def test_module_layouts_match_reference_names():
    from autodiffusion_amd import dit, refiner
    cfg = refiner.refiner_config(frame_stack=10, diffusion_timesteps=10)
    da = refiner.DiffusionAugmentation(cfg)
    keys = set(da.state_dict().keys())
    for k in ("init_x", "diffusion.diffusion.betas", "diffusion.diffusion.loss_weight", "diffusion.diffusion.posterior_mean_coef2",
              "diffusion.diffusion.model.cond_pos_emb", "diffusion.diffusion.model.memory_mask",
              "diffusion.diffusion.model.decoder.layers.2.multihead_attn.in_proj_weight"):
        assert k in keys, k
    assert sum(p.numel() for p in da.diffusion.diffusion.model.parameters()) == 128324   # measured on the reference at fs=10
    m = dit.Transformer_TrajDiffusionModel(dit.dit_config(64, 2, 4, "MM-DiT"))
    names = dict(m.named_parameters())
    for k in ("blocks.1.attn.Wq_c.weight", "blocks.0.attn.ResNorm_x_q.weight", "blocks.0.adaLN_modulation_c.1.bias",
              "blocks.0.mlp_x.2.weight", "state_encoder.1.mlp.layer_norm.weight", "x_encoder.3.fc.weight", "time_encoder.5.bias",
              "final_layer.adaLN_modulation.1.weight", "position_embedding"):
        assert k in names, k
    assert tuple(names["blocks.0.attn.proj.weight"].shape) == (64, 2 * 64 * 4)
    m = dit.Transformer_TrajDiffusionModel(dit.dit_config(256, 4, 8, "DiT"))
    n = sum(p.numel() for p in m.parameters())
    assert abs(n - 13.1e6) < 0.1e6, n                                                   # 13.1 M measured on the reference
    from autodiffusion_amd import explicit
    ex = explicit.ExplicitDiT(explicit.explicit_config(frame_stack=10))
    ek = set(ex.state_dict().keys())
    for k in ("init_x", "data_mean", "data_std", "diffusion.diffusion.cond_pos_emb", "diffusion.diffusion.head.weight"):
        assert k in ek, k
    assert tuple(ex.diffusion.diffusion.cond_pos_emb.shape) == (1, 791, 32)               # the reference's "# 791"
    assert tuple(ex.init_x.shape) == (40,)
    assert [int(r[0]) for r in ex.diffusion.schedule_rows()] == [90, 80, 70, 60, 50, 40, 30, 20, 10, 0]
    from autodiffusion_amd import transition
    st = transition.StrDiffDiT(transition.transition_config(frame_stack=10, z_dim=768))
    sk = set(st.state_dict().keys())
    for k in ("init_z", "data_mean", "data_std", "transition_model.x_from_z.weight", "transition_model.diffusion.head.weight"):
        assert k in sk, k
    assert tuple(st.init_z.shape) == (10, 768)
    assert tuple(st.transition_model.x_from_z.weight.shape) == (4, 768)
    assert tuple(st.transition_model.diffusion.head.weight.shape) == (768, 32)
    assert tuple(st.transition_model.diffusion.cond_pos_emb.shape) == (1, 809, 32)
    assert [int(r[0]) for r in st.transition_model.schedule_rows()] == [90, 80, 70, 60, 50, 40, 30, 20, 10, 0]
    with pytest.raises(ValueError):
        dit.Transformer_TrajDiffusionModel(dit.dit_config(64, 2, 4, "DDPM"))
    with pytest.raises(AssertionError):
        refiner.DiffusionWrapper(torch.nn.Identity(), timesteps=10, objective="pred_eps")
    with pytest.raises(ValueError):
        refiner.DiffusionWrapper(torch.nn.Identity(), timesteps=10, beta_schedule="quadratic")


def test_shard_range():
    from autodiffusion_amd.distributed import shard_range
    for total in (1, 7, 8, 1024, 4099):
        for world in (1, 2, 4, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from autodiffusion_amd import synth
from autodiffusion_amd.distributed import shard_range, all_gather_rows
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=int(sys.argv[3]), world_size=2)
rank, world = dist.get_rank(), 2
from autodiffusion_amd.distributed import RowGatherer
for total in (7, 8, 1):
    lo, hi = shard_range(total, rank, world)
    # each rank builds only its own rows (global-row-indexed noise), "samples" them, then gathers
    local = synth.noise(hi - lo, (), (80, 4), 1234, row0=lo) * 2.0 + 1.0
    full = all_gather_rows(local, total)
    ref = synth.noise(total, (), (80, 4), 1234) * 2.0 + 1.0
    assert full.shape == ref.shape and torch.equal(full, ref), "gathered shards differ from the single-process result"
    g = RowGatherer(total, (80, 4), "cpu")
    for rep in range(2):                                   # the preallocated buffers are reused call after call
        assert torch.equal(g.gather(local * (rep + 1)), ref * (rep + 1))
dist.barrier()
dist.destroy_process_group()
print("ok", rank)
'''


def test_two_rank_gloo_shard_and_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    from autodiffusion_amd.distributed import free_port
    port = str(free_port())
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    for p in procs:
        out, _ = p.communicate(timeout=120)
        assert p.returncode == 0, out.decode()


def test_bench_launches_its_own_ranks():
    """`bench.py --gpus 2` with no launcher in the environment starts two ranks itself (a fresh torch.distributed.run child) and
    reports the rank count the process group saw; --dry-run keeps it on the CPU (gloo) with a ragged total, and checks the
    gathered shards against the single-process tensor."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "2"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                       # rank 0 alone prints
    assert lines[0]["n_gpus"] == 2 and lines[0]["gathered_equals_single_process"] is True
    assert lines[0]["config"]["global_batch"] == 11        # ragged: 6 + 5 rows
    # a launcher that started a different number of ranks than --gpus is an error, not a silently smaller run
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"], env=env, capture_output=True, text=True,
                       timeout=120)
    assert r.returncode != 0 and "--gpus 4" in (r.stdout + r.stderr)


REF_CONFIG = "/root/reference/config"


@pytest.mark.skipif(not os.path.isdir(REF_CONFIG), reason="the reference's yaml files are only present in the development container")
def test_reference_yaml_files_load():
    """config/*.yaml of the reference, read as they are, configure the drop-in classes (module construction needs no GPU)."""
    from autodiffusion_amd import config, dit, explicit, refiner, transition
    c = config.explicit_config_from(os.path.join(REF_CONFIG, "explicitDiT_whole.yaml"))
    assert c.frame_stack == 80 and c.data_mean == [50, 5, 0, 0] and c.data_std == [100, 10, 1, 1] and c.weight_decay == 1e-4
    assert c.scheduler.num_train_timesteps == 100 and c.scheduler.num_inference_steps == 10
    assert c.scheduler.beta_schedule == "squaredcos_cap_v2" and c.scheduler.clip_sample is True and c["scheduler"]["prediction_type"] == "epsilon"
    m = explicit.ExplicitDiT(c)
    assert tuple(m.init_x.shape) == (320,) and tuple(m.diffusion.diffusion.cond_pos_emb.shape) == (1, 791, 32)
    assert [int(r[0]) for r in m.diffusion.schedule_rows()] == [90, 80, 70, 60, 50, 40, 30, 20, 10, 0]
    assert m.data_std.tolist() == [100.0, 10.0, 1.0, 1.0]
    c = config.explicit_config_from(os.path.join(REF_CONFIG, "explicitDiT.yaml"))
    assert c.frame_stack == 10 and c.scheduler.num_train_timesteps == 10 and c.data_std == [100, 20, 1, 1]
    assert len(explicit.ExplicitDiT(c).diffusion.schedule_rows()) == 10
    c = config.transition_config_from(os.path.join(REF_CONFIG, "strDiffDiT.yaml"))
    assert c.z_shape == (10, 768) and c.x_shape == (4,) and c.cond_dim == 256 and c.prior_dim == 4 and c.data_std == [150, 20, 1, 1]
    st = transition.StrDiffDiT(c)
    assert tuple(st.init_z.shape) == (10, 768) and tuple(st.transition_model.x_from_z.weight.shape) == (4, 768)
    c = config.refiner_config_from(os.path.join(REF_CONFIG, "ldr_diffusion.yaml"), diffusion_timesteps=100, sampling_timesteps=50)
    assert c.frame_stack == 10 and c.trajectory_loss_rescale == 1.0 and c.visualize_diffusion_intermedians is False
    da = refiner.DiffusionAugmentation(c)
    assert da.diffusion.diffusion.is_ddim_sampling and tuple(da.init_x.shape) == (10, 4)
    c = config.dit_config_from(os.path.join(REF_CONFIG, "ldr_diffusion.yaml"), mc_num=16, k=6)
    dec = dit.DiffusionDecoder(c)
    assert dec.k == 6 and dec.config.mc_num == 16 and dec.config.trajectory_loss_rescale == 1.0
    with pytest.raises(ValueError, match="scheduler"):
        config.explicit_config_from(os.path.join(REF_CONFIG, "ldr_diffusion.yaml"))


def test_yaml_loader_on_inline_text(tmp_path):
    """The same loader without the reference tree: nested blocks, lists, YAML 1.1 exponent scalars, missing keys."""
    from autodiffusion_amd import config
    f = tmp_path / "c.yaml"
    f.write_text("frame_stack: 20\nmap_cond: True\ndata_mean: [1, 2, 0, 0]\ndata_std: [3, 4, 1, 1]\nlr: 2e-4\nweight_decay: 1e-4\n"
                 "scheduler:\n  num_train_timesteps: 50\n  num_inference_steps: 5\n  beta_start: 0.0001\n  beta_end: 0.02\n"
                 "  beta_schedule: squaredcos_cap_v2\n  clip_sample: True\n  prediction_type: epsilon\n")
    c = config.explicit_config_from(str(f), precision="bf16x3")
    assert c.frame_stack == 20 and c.lr == 2e-4 and c.weight_decay == 1e-4 and c.precision == "bf16x3" and c.n_maps_token == 788
    assert c.scheduler.num_inference_steps == 5 and "scheduler" in c and c.get("nothing", 7) == 7
    assert c.to_dict()["scheduler"]["beta_schedule"] == "squaredcos_cap_v2"
    f.write_text("- 1\n- 2\n")
    with pytest.raises(ValueError, match="mapping"):
        config.load_config(str(f))


def test_ltm_output_record():
    """The forward() result reads like the reference's ModelOutput: attributes, mapping items, None fields left out."""
    from autodiffusion_amd.refiner import LTMOutput
    o = LTMOutput(loss=torch.tensor(1.5), logits=torch.zeros(2, 80, 4), loss_items=torch.ones(8), pred_dict={"traj_logits": torch.zeros(2, 80, 4)})
    assert float(o.loss) == 1.5 and o["loss_items"].shape == (8,) and "pred_dict" in o and "trajectory_label" not in o
    assert o.trajectory_label is None
    assert [k for k, _ in o.items() if k not in ("past_key_values", "loss_items", "loss")] == ["logits", "pred_dict"]   # trainer.py:647


def test_metrics_host_formulas(golden):
    """metrics.py (torch path) against the golden ADE/FDE vectors and the oracle's restatement of compute_metrics."""
    from autodiffusion_amd import metrics
    from oracle import sampling
    m = golden("g10_metrics")
    got = metrics.ade_fde(torch.from_numpy(m["pred"]), torch.from_numpy(m["label"]))
    for k in ("ade1", "ade3", "ade5", "ade8", "fde1", "fde3", "fde5", "fde8"):
        np.testing.assert_allclose(got[k].numpy(), m[k], rtol=1e-6, atol=1e-6)
    g = np.random.default_rng(9)
    for S, C in ((80, 4), (8, 4), (80, 2)):
        pred = (g.standard_normal((33, S, C)) * np.array([9.0, 4.0, 0.0, 2.5][:C])).astype(np.float32)
        label = (g.standard_normal((33, S, C)) * np.array([9.0, 4.0, 0.0, 2.5][:C])).astype(np.float32)
        a = metrics.compute_metrics(torch.from_numpy(pred), torch.from_numpy(label))
        b = sampling.evaluation_metrics(pred.astype(np.float64), label.astype(np.float64))
        assert a.keys() == b.keys()
        for k in a:
            assert abs(a[k] - b[k]) < 1e-5 * max(1.0, abs(b[k])), (S, C, k, a[k], b[k])


def test_product_library_has_no_ablation_instances():
    """The kernel instances with parts switched off (wrong results on purpose) and the environment variables that select them
    are compiled into the measurement library only; the product library does not even hold their names."""
    from autodiffusion_amd import _lib
    blob = open(os.path.join(os.path.dirname(_lib.__file__), "libadx.so"), "rb").read()
    assert b"_ABLATE" not in blob


def test_compute_scores_groups_frames_by_scenario():
    """metrics.compute_scores (frames grouped by scenario15s_id, the clamp on the GROUP mean, five misses zero a group, the
    combined score) against the loop restatement of utils/nuplan_utils.py:229-314 and against cases worked by hand.  Parity
    unpinned: the reference module cannot be imported here."""
    from autodiffusion_amd import metrics
    from oracle import sampling
    g = np.random.default_rng(21)
    B = 57
    label = np.cumsum(g.standard_normal((B, 80, 4)) * np.array([1.0, 0.2, 0.0, 0.02]), axis=1).astype(np.float32)
    scale = g.choice([0.3, 2.0, 9.0], size=(B, 1, 1)) * np.array([1.0, 0.5, 0.0, 0.3])
    pred = (label + g.standard_normal((B, 80, 4)) * scale).astype(np.float32)
    ids = g.integers(0, 6, size=B) * 1000003 + 17          # six groups, uneven sizes, ids far apart, some groups with >= 5 misses
    a = metrics.compute_metrics(torch.from_numpy(pred), torch.from_numpy(label), scenario15s_id=torch.from_numpy(ids))
    ref = sampling.evaluation_scores(pred.astype(np.float64), label.astype(np.float64), ids)
    assert 0.0 < ref[1] < 1.0                              # the case has zeroed and surviving groups
    assert abs(a["average_score"] - ref[0]) < 1e-6 and abs(a["miss_score"] - ref[1]) < 1e-12
    # by hand: one group of two frames, constant offsets of 2 m and 6 m in x, no heading error, no miss:
    # ade = fde = 1 - ((2 + 6) / 2) / 8 = 0.5, heading scores 1 -> (0.5 + 0.5 + 2 + 2) / 6
    lab = torch.zeros(2, 80, 4)
    prd = lab.clone()
    prd[0, :, 0], prd[1, :, 0] = 2.0, 5.9
    s, m = metrics.compute_scores(metrics.per_scenario(prd, lab), [4, 4])
    assert abs(s - (2 * (1 - (2.0 + 5.9) / 2 / 8) + 4) / 6) < 1e-6 and m == 1.0
    # clamp on the group mean, not per frame: offsets 1 m and 17 m in ONE group give max(1 - 9/8, 0) = 0 (per-frame clamping
    # would give 0.4375); as two groups the first keeps 0.875 and the second is a miss-free zero
    prd[0, :, 0], prd[1, :, 0] = 1.0, 17.0
    t = metrics.per_scenario(prd, lab)
    t[:, 15:] = 0                                          # no misses, to isolate the displacement scores
    s1, _ = metrics.compute_scores(t, [9, 9])
    s2, _ = metrics.compute_scores(t, [9, 10])
    assert abs(s1 - 4 / 6) < 1e-6 and abs(s2 - ((2 * 0.875 + 4) / 6 + 4 / 6) / 2) < 1e-6
    # five missed frames zero their group's score whatever the errors of its other frames
    lab6, prd6 = torch.zeros(6, 80, 4), torch.zeros(6, 80, 4)
    prd6[:5, 10, 1] = 7.0                                  # 7 m at 1 s > the 6 m threshold of the 3 s horizon
    s, m = metrics.compute_scores(metrics.per_scenario(prd6, lab6), [1] * 6)
    assert s == 0.0 and m == 0.0
    s, m = metrics.compute_scores(metrics.per_scenario(prd6[1:], lab6[1:]), [1] * 5)     # four misses: the group survives
    assert s > 0.9 and m == 1.0
    with pytest.raises(ValueError):
        metrics.compute_scores(metrics.per_scenario(prd6, lab6), [1, 2])


def test_mlp256_rejects_mismatched_operands():
    """The Python wrapper of the fused MLP derives the hidden width from the weight buffers' BYTE size and refuses operands that do
    not fit it (a wrong width would send the kernel's weight copies out of bounds)."""
    from autodiffusion_amd import ops
    x = torch.zeros(128, 256)
    w1 = torch.zeros(64 * 256 * 2 * 2, dtype=torch.uint8)           # block_weight form of a (64, 256) weight, two planes
    w2 = torch.zeros(64 * 256 * 2 * 2, dtype=torch.uint8)
    with pytest.raises(ValueError):
        ops.mlp256(x, w1, torch.zeros(65), 1, w2, None)              # bias of the wrong length
    with pytest.raises(ValueError):
        ops.mlp256(x, w1, None, 1, w2[:-2], None)                    # second weight of another size
    with pytest.raises(ValueError):
        ops.mlp256(torch.zeros(128, 128), w1, None, 1, w2, None)     # not 256 wide
    with pytest.raises(ValueError):
        ops.mlp256(x, w1, None, 1, w2, None, res=torch.zeros(127, 256))


def test_training_path_host_logic():
    """The training path's host side without a GPU: the mode switch (gradients enabled and trainable parameters), the refusal of
    host tensors (there is no CPU fallback), and the flat gradient store that hands out zero-filled views by parameter name."""
    import torch
    from autodiffusion_amd import _lib, refiner, train
    tr = refiner.TrajectoryRefiner(refiner.refiner_config(frame_stack=10, diffusion_timesteps=10))
    net = tr.diffusion.model
    assert train.wants_grad(net)
    with torch.no_grad():
        assert not train.wants_grad(net)
    for p in net.parameters():
        p.requires_grad_(False)
    assert not train.wants_grad(net)
    names = train.trainable_names(net)
    assert len(names) == 108 and "cond_pos_emb" in names and "decoder.layers.3.multihead_attn.in_proj_weight" in names
    with pytest.raises(_lib.AdxError, match="device memory"):
        train.add(torch.zeros(4), torch.zeros(4))
    with pytest.raises(ValueError):
        train.add(torch.zeros(4), torch.zeros(5))
    params = {"a.weight": torch.ones(3, 2), "a.bias": torch.ones(3)}
    g = train._Grads(params)
    w = g.of("a.weight")
    assert w.shape == (3, 2) and not w.any() and g.of("a.weight") is w and g.of("a.bias").shape == (3,)
    w += 1
    assert float(g.flat.sum()) == 6.0 and set(g.g) == {"a.weight", "a.bias"}


def test_training_operators_reject_bad_arguments():
    """Argument checks of the training entry points through the raw ABI: they answer before any device work, so this runs
    without a GPU (null pointers, empty shapes, unsupported head widths, scratch buffers that are too small)."""
    import ctypes as C
    from autodiffusion_amd import _lib
    lib = _lib.load()
    fake = 4096                                  # never dereferenced: every call below is refused by its checks
    assert lib.adx_op_add(None, fake, fake, 8, None) == 1
    assert lib.adx_op_act(fake, fake, 8, 99, None) == 1 and b"activation" in lib.adx_last_error()
    assert lib.adx_op_linear_wgrad(fake, fake, 0, 4, 4, fake, None, 0, fake, 1 << 20, None) == 1
    need = lib.adx_reduce_ws_bytes(1000, 32, 32)
    assert need == ((1000 + 31) // 32) * (32 * 32 + 32) * 4          # 32 chunks of 32 rows, a dW tile and a bias row each
    assert lib.adx_op_linear_wgrad(fake, fake, 1000, 32, 32, fake, None, 0, fake, need - 1, None) == 1
    assert b"workspace" in lib.adx_last_error()
    assert lib.adx_op_colsum(fake, 10, 4, fake, 0, fake, 0, None) == 1
    args = (fake, fake, fake, fake, fake, 2, 8, 10, 10)
    assert lib.adx_op_attention_bwd(*args, 12, C.c_float(0.5), -1, fake, fake, fake, fake, 0, None) == 1
    assert b"head width" in lib.adx_last_error()
    assert lib.adx_op_attention_stats(fake, fake, fake, fake, fake, 2, 8, 200, 200, 64, C.c_float(0.5), -1, None) == 1   # 200 tokens: no LDS room
    assert lib.adx_op_group_colsum(fake, 70000, 80, 64, fake, None) == 1
    assert lib.adx_op_layernorm_bwd(fake, None, fake, 0, 32, C.c_float(1e-5), fake, None, None) == 1


def test_mixtral_state_dict_layouts():
    """MixtralModel holds its parameters under the names of a reference checkpoint (transformers 4.42.3: per-expert w1 / w2 / w3 under
    block_sparse_moe); a state dict in the layout of newer transformers releases (fused mlp.experts.gate_up_proj / down_proj, mlp.gate)
    loads into the same tensors; the size table and the refusals of STR_Mixtral follow build_models / the scope of this package."""
    import torch
    from autodiffusion_amd import str_backbone
    m = str_backbone.MixtralModel(hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                                  num_local_experts=4, num_experts_per_tok=2, vocab_size=16)
    names = set(m.state_dict().keys())
    assert "layers.1.block_sparse_moe.experts.3.w2.weight" in names and "layers.0.block_sparse_moe.gate.weight" in names
    assert "layers.0.self_attn.k_proj.weight" in names and m.state_dict()["layers.0.self_attn.k_proj.weight"].shape == (16, 32)
    assert "norm.weight" in names and "embed_tokens.weight" in names and not any("bias" in n for n in names)
    g = torch.Generator().manual_seed(0)
    newer = {}
    for k, v in m.state_dict().items():
        if ".block_sparse_moe.experts." in k:
            continue
        newer[k.replace(".block_sparse_moe.gate.", ".mlp.gate.")] = torch.randn(v.shape, generator=g)
    for i in range(2):
        newer[f"layers.{i}.mlp.experts.gate_up_proj"] = torch.randn(4, 128, 32, generator=g)
        newer[f"layers.{i}.mlp.experts.down_proj"] = torch.randn(4, 32, 64, generator=g)
    want = {k: v.clone() for k, v in newer.items()}
    m.load_state_dict(newer)
    sd = m.state_dict()
    for i in range(2):
        for e in range(4):
            p = f"layers.{i}.block_sparse_moe.experts.{e}."
            assert torch.equal(sd[p + "w1.weight"], want[f"layers.{i}.mlp.experts.gate_up_proj"][e, :64])       # gate rows first ...
            assert torch.equal(sd[p + "w3.weight"], want[f"layers.{i}.mlp.experts.gate_up_proj"][e, 64:])       # ... then up rows
            assert torch.equal(sd[p + "w2.weight"], want[f"layers.{i}.mlp.experts.down_proj"][e])
        assert torch.equal(sd[f"layers.{i}.block_sparse_moe.gate.weight"], want[f"layers.{i}.mlp.gate.weight"])
    assert str_backbone.mixtral_sizes("scratch-mixtral-small-wide") == dict(n_layer=4, n_embd=512, n_inner=2048, n_head=8)
    assert str_backbone.mixtral_sizes("mixtral-small") == dict(n_layer=4, n_embd=256, n_inner=1024, n_head=8)
    assert str_backbone.mixtral_sizes("mixtral-3b-wide")["n_embd"] == 1536
    with pytest.raises(AssertionError):
        str_backbone.mixtral_sizes("gpt-small")
    for bad in (dict(use_proposal=1), dict(autoregressive=True), dict(kp_tokenizer="cluster"), dict(task="waymo"), dict(kp_decoder_type="diffusion"),
                dict(raster_encoder_type="resnet18")):
        with pytest.raises(NotImplementedError):
            str_backbone.STR_Mixtral(str_backbone.str_config("mixtral-small", n_layer=1, **bad))
    with pytest.raises(ValueError):
        str_backbone.MixtralModel(hidden_size=30, intermediate_size=64, num_hidden_layers=1, num_attention_heads=4)
    # a module built from the configuration helper exposes the reference's sub-module names
    model = str_backbone.STR_Mixtral(str_backbone.str_config("mixtral-small", n_layer=1, use_key_points="no"))
    top = {n.split(".")[0] for n, _ in model.named_parameters()}
    assert top == {"encoder", "transformer", "traj_decoder"}            # no key-point head without key points
