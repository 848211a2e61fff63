"""Conditioning producer, first stage (raster encoder -> maps_info) on the GPU, through the C ABI, against the recording of the
reference's own NuplanRasterizeEncoder (tests/golden/g18_raster_encoder.npz) and the CPU oracle."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import ISEED, NSEED, WSEED, rel_err

pytestmark = pytest.mark.gpu
GATE = 1e-3


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def encoder_config(n_embd, n_head, inter, key_points, precision="fp32"):
    return SimpleNamespace(use_proposal=0, use_key_points=key_points, n_embd=n_embd, raster_encoder_type="vit", raster_channels=34,
                           vit_intermediate_size=inter, n_head=n_head, use_speed=False, separate_kp_encoder=True, camera_image_encoder=None,
                           x_random_walk=0.0, y_random_walk=0.0, arf_x_random_walk=0.0, arf_y_random_walk=0.0, with_traffic_light=True,
                           precision=precision)


def make_encoder(cfg, dev):
    from autodiffusion_amd import raster_encoder, synth
    enc = raster_encoder.NuplanRasterizeEncoder(cfg)
    synth.fill_module_(enc, WSEED)               # parameter names are the reference checkpoint's: same seeded values as the recording
    return enc.to(dev).eval()


@pytest.mark.parametrize("B,F,route,traffic", [(2, 4, 2, True), (1, 9, 1, True), (3, 2, 2, False)])
def test_op_raster_patches(dev, B, F, route, traffic):
    """The gather against the reference's own sequence of steps in torch: permute(0, 3, 2, 1), cat_raster_seq, unfold of the
    stride-16 convolution.  Values are copied, so the comparison is exact."""
    from autodiffusion_amd import raster_encoder
    from oracle import raster as oracle_raster
    C = route + 24 + 8 * F
    g = torch.Generator().manual_seed(B * 100 + F)
    raster = torch.randn(B, 64, 48, C, generator=g)            # two different spatial sizes: catches a swapped axis
    got = raster_encoder.raster_patches(raster.to(dev), F, traffic).cpu()
    seq = oracle_raster.cat_raster_seq(raster.permute(0, 3, 2, 1), F, traffic)          # (B, F, C', 48, 64)
    b, f, c, h, w = seq.shape
    ref = torch.nn.functional.unfold(seq.reshape(b * f, c, h, w), kernel_size=16, stride=16)      # (B F, C' 256, patches)
    ref = ref.transpose(1, 2).reshape(-1, c * 256)
    assert got.shape == ref.shape and torch.equal(got, ref)
    from autodiffusion_amd import _lib
    with pytest.raises(_lib.AdxError):
        raster_encoder.raster_patches(torch.zeros(1, 40, 48, C, device=dev), F, traffic)           # height not a multiple of 16
    with pytest.raises(_lib.AdxError):
        raster_encoder.raster_patches(torch.zeros(1, 32, 32, 8 * F + 20, device=dev), F, traffic)  # too few channels


def test_raster_encoder_small_against_reference(dev, golden):
    """n_embd 64 (ViT hidden 32, 4 heads of width 8, intermediate 96, 12 layers), 'specified_backward' key points, B = 2:
    maps_info and input_embeds against the reference recording (fp32, 1e-3 gate); newer-release state-dict keys load."""
    from autodiffusion_amd import synth
    g = golden("g18_raster_encoder")
    enc = make_encoder(encoder_config(64, 4, 96, "specified_backward"), dev)
    high, low = synth.occupancy_raster(2, 58, ISEED).to(dev), synth.occupancy_raster(2, 58, ISEED + 1).to(dev)
    label = synth.trajectory_prior(2, ISEED).to(dev)
    embeds, info, maps = enc(high_res_raster=high, low_res_raster=low, context_actions=T(g["small_context_actions"]).to(dev),
                             trajectory_label=label, is_training=False)
    e = rel_err(maps.cpu(), g["small_maps_info"])
    print(f"raster encoder (small): maps_info rel err {e:.2e}")
    assert maps.shape == (2, 788, 64) and e < GATE
    assert rel_err(embeds[:, 788:].cpu(), g["small_input_embeds_tail"]) < GATE and torch.equal(embeds[:, :788], maps)
    assert info["context_length"] == 788 and info["pred_length"] == 80 and info["selected_indices"] == [79, 39, 19, 9, 4]
    assert torch.equal(info["future_key_points"], label[:, [79, 39, 19, 9, 4]])
    # a state dict with the key layout of newer transformers releases
    sd = {k: v.detach().clone() for k, v in enc.image_downsample.state_dict().items()}
    newer = {}
    for k, v in sd.items():
        k2 = k
        for old, new in (("attention.attention.query", "attention.q_proj"), ("attention.attention.key", "attention.k_proj"),
                         ("attention.attention.value", "attention.v_proj"), ("attention.output.dense", "attention.o_proj"),
                         ("intermediate.dense", "mlp.fc1"), ("output.dense", "mlp.fc2")):
            k2 = k2.replace(old, new)
        newer[k2.replace("encoder.layer.", "layers.")] = v.flip(0) if "q_proj.weight" in k2 else v
    enc.image_downsample.load_state_dict(newer)
    assert torch.equal(enc.image_downsample.encoder.layer[5].attention.attention.query.weight, sd["encoder.layer.5.attention.attention.query.weight"].flip(0))


def test_raster_encoder_full_size_feeds_the_refiner(dev, golden):
    """The mixtral-small configuration (n_embd 256: ViT hidden 128, 8 heads of width 16, intermediate 3072), no key points.
    fp32 against the reference recording (every 7th token); split-bf16 close to it; and the map tokens handed to the refiner's
    generate() without leaving the device: trajectories sampled from the split-bf16 encoder's tokens stay within 1e-2 m ADE of
    those sampled from the fp32 encoder's."""
    from autodiffusion_amd import metrics, refiner, synth
    g = golden("g18_raster_encoder")
    B = 2
    high, low = synth.occupancy_raster(B, 58, ISEED).to(dev), synth.occupancy_raster(B, 58, ISEED + 1).to(dev)
    actions = torch.cat([T(g["full_context_actions"]), T(g["full_context_actions"]) * 0.5]).to(dev)
    label = synth.trajectory_prior(B, ISEED).to(dev)
    maps = {}
    for prec in ("fp32", "bf16x3"):
        enc = make_encoder(encoder_config(256, 8, 3072, "no", prec), dev)
        embeds, info, m = enc(high_res_raster=high, low_res_raster=low, context_actions=actions, trajectory_label=label, is_training=False)
        e = rel_err(m[:1, ::7].cpu(), g["full_maps_info_every7"])
        print(f"raster encoder (n_embd 256) {prec}: maps_info rel err {e:.2e}")
        assert m.shape == (B, 788, 256) and embeds.shape == (B, 868, 256) and e < (GATE if prec == "fp32" else 5e-3)
        assert not embeds[:, 788:].any()
        maps[prec] = m
    fs, N = 10, 10
    da = refiner.DiffusionAugmentation(refiner.refiner_config(frame_stack=fs, diffusion_timesteps=N, objective="pred_noise", beta_schedule="cosine",
                                                                precision="bf16x3"))
    synth.reference_init_(da, WSEED)
    da = da.to(dev).eval()
    prior = synth.trajectory_prior(B, ISEED).to(dev)
    noise = synth.noise(B, (80 // fs, N), (fs, 4), NSEED).to(dev)
    out = {k: da.generate(trajectory_prior=prior, maps_info=v, noise=noise)["traj_logits"].clone() for k, v in maps.items()}
    ade = float(metrics.ade_fde(out["bf16x3"], out["fp32"])["ade8"].max())
    print(f"refiner on the encoder's map tokens: ADE (split-bf16 encoder vs fp32 encoder) {ade:.2e} m")
    assert out["fp32"].shape == (B, 80, 4) and bool(torch.isfinite(out["fp32"]).all()) and ade < 1e-2
