"""Conditioning producer, middle stage and the whole of phase 1 on the GPU, through the C ABI: RMSNorm, rotary embedding and the
mixture-of-experts block against the CPU oracle; the decoder with cached keys against its own full pass; STR_Mixtral.generate against
the recording of the reference's (tests/golden/g21_str_generate.npz); rasters -> refined trajectories without leaving the device."""
import numpy as np
import pytest
import torch

from conftest import ISEED, NSEED, WSEED, rel_err
from test_oracle_golden import str_model_from_recording

pytestmark = pytest.mark.gpu
GATE = 1e-3


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.mark.parametrize("rows,d", [(7, 64), (1000, 256), (33, 100), (5, 30)])
def test_op_rmsnorm(dev, rows, d):
    from autodiffusion_amd import str_backbone
    from oracle import str_backbone as osb
    g = torch.Generator().manual_seed(rows + d)
    x, w = torch.randn(rows, d, generator=g) * 3, torch.randn(d, generator=g)
    got = str_backbone.rmsnorm(x.to(dev), w.to(dev), 1e-5).cpu()
    assert rel_err(got, osb.rms_norm(x, w, 1e-5)) < 2e-6


@pytest.mark.parametrize("B,S,H,dh,pos0", [(2, 5, 4, 16, 0), (1, 3, 2, 32, 790), (3, 1, 8, 8, 17)])
def test_op_rotary(dev, B, S, H, dh, pos0):
    """In-place rotary embedding of tokens at positions pos0 .. pos0 + S - 1 against the oracle's rotate-half form."""
    from autodiffusion_amd import _lib, str_backbone
    from oracle import str_backbone as osb
    g = torch.Generator().manual_seed(B + S + H)
    x = torch.randn(B, S, H * dh, generator=g)
    cos, sin = str_backbone.rotary_tables(pos0 + S, dh, 1e6, dev)
    got = str_backbone.rotary_(x.to(dev).clone(), cos, sin, H, pos0).cpu()
    oc, osn = osb.rotary_tables(pos0 + S, dh, 1e6)
    ref = osb.rotate(x.reshape(B, S, H, dh).transpose(1, 2), oc[pos0:], osn[pos0:]).transpose(1, 2).reshape(B, S, H * dh)
    assert rel_err(got, ref) < 1e-6
    with pytest.raises(_lib.AdxError):
        str_backbone.rotary_(x.to(dev), cos, sin, H, pos0 + 1)            # tables too short for the last position


def expert_state(d, inter, E, seed=WSEED):
    from autodiffusion_amd import synth
    shapes = [("gate.weight", (E, d))]
    for e in range(E):
        shapes += [(f"experts.{e}.w1.weight", (inter, d)), (f"experts.{e}.w3.weight", (inter, d)), (f"experts.{e}.w2.weight", (d, inter))]
    return synth.seeded_state(shapes, seed)


@pytest.mark.parametrize("prec,tol", [("fp32", 3e-6), ("bf16x3", 5e-5), ("bf16", 2e-2)])
@pytest.mark.parametrize("T_,d,inter,E,k", [(1746, 64, 256, 8, 2), (5, 32, 64, 8, 2), (300, 256, 1024, 8, 2), (130, 64, 96, 5, 3), (64, 32, 32, 4, 1)])
def test_op_moe_block(dev, prec, tol, T_, d, inter, E, k):
    """The expert block against the oracle: the experts chosen for every token are the oracle's, the weights agree, and the output
    (with the residual) agrees to the operand precision.  Shapes: a decoder pass of the small recording, fewer tokens than experts
    (empty experts), the mixtral-small widths, five experts with three per token, and one expert per token."""
    from autodiffusion_amd import str_backbone
    from oracle import str_backbone as osb
    sd = expert_state(d, inter, E)
    g = torch.Generator().manual_seed(T_ + d)
    x, res = torch.randn(T_, d, generator=g), torch.randn(T_, d, generator=g)
    w13, w2 = str_backbone.prepare_expert_weights([sd[f"experts.{e}.w1.weight"].to(dev) for e in range(E)], [sd[f"experts.{e}.w3.weight"].to(dev) for e in range(E)],
                                                  [sd[f"experts.{e}.w2.weight"].to(dev) for e in range(E)], prec)
    y, idx, wts = str_backbone.moe_block(x.to(dev), res.to(dev), sd["gate.weight"].to(dev), w13, w2, inter, E, k, prec, return_routing=True)
    prob = torch.softmax(x @ sd["gate.weight"].T, dim=-1)
    val, ref_idx = torch.topk(prob, k, dim=-1)
    assert torch.equal(idx.cpu().long(), ref_idx)
    assert rel_err(wts.cpu(), val / val.sum(-1, keepdim=True)) < 1e-5
    ref = res + osb.moe_block(sd, "", x, E, k)
    e = rel_err(y.cpu(), ref)
    print(f"moe block {prec} T={T_} d={d} inter={inter} E={E} k={k}: rel err {e:.1e}")
    assert e < tol
    y2 = str_backbone.moe_block(x.to(dev), res.to(dev), sd["gate.weight"].to(dev), w13, w2, inter, E, k, prec)
    assert torch.equal(y, y2)
    # a token's result does not depend on its neighbours: the first rows alone give the same bits
    y3 = str_backbone.moe_block(x[:3].to(dev), res[:3].to(dev), sd["gate.weight"].to(dev), w13, w2, inter, E, k, prec)
    assert torch.equal(y[:3], y3)


def test_op_moe_block_argument_checks(dev):
    from autodiffusion_amd import _lib, str_backbone
    sd = expert_state(32, 64, 4)
    w13, w2 = str_backbone.prepare_expert_weights([sd[f"experts.{e}.w1.weight"].to(dev) for e in range(4)], [sd[f"experts.{e}.w3.weight"].to(dev) for e in range(4)],
                                                  [sd[f"experts.{e}.w2.weight"].to(dev) for e in range(4)], "fp32")
    x = torch.zeros(8, 32, device=dev)
    with pytest.raises(_lib.AdxError):
        str_backbone.moe_block(x, None, sd["gate.weight"].to(dev), w13, w2, 64, 4, 5)                      # more experts per token than experts
    with pytest.raises(_lib.AdxError):
        str_backbone.moe_block(x, None, sd["gate.weight"].to(dev), w13, w2, 64, 4, 2, precision="f16")       # no fp16 form of the expert products
    with pytest.raises(_lib.AdxError):
        str_backbone.moe_block(torch.zeros(8, 40, device=dev), None, sd["gate.weight"].to(dev), w13, w2, 64, 4, 2)    # width not a multiple of 32
    lib = _lib.load()
    assert lib.adx_moe_workspace_bytes(0, 32, 64, 4, 2) == 0 and lib.adx_moe_workspace_bytes(8, 32, 64, 4, 2) > 0


def test_mixtral_cached_passes_equal_a_full_pass(dev, golden):
    """The decoder on a prefix, then on the remaining tokens one group at a time with cached keys and values, returns what one pass
    over the whole sequence returns (the reference re-runs the whole prefix every time)."""
    g = golden("g21_str_generate")
    model, _, _ = str_model_from_recording(g, "small")
    tr = model.transformer.to(dev)
    emb = torch.randn(2, 40, 64, generator=torch.Generator().manual_seed(3)).to(dev)
    full = tr(emb)["last_hidden_state"]
    cache = tr.new_cache()
    parts = [tr.extend(emb[:, :25], cache), tr.extend(emb[:, 25:26], cache), tr.extend(emb[:, 26:27], cache), tr.extend(emb[:, 27:], cache)]
    assert cache.length == 40
    e = rel_err(torch.cat(parts, dim=1).cpu(), full.cpu())
    print(f"cached passes vs full pass: rel err {e:.1e}")
    assert e < 2e-6
    from oracle import str_backbone as osb
    _, ocfg, _ = str_model_from_recording(g, "small")
    ref = osb.mixtral_forward({k: v.detach().cpu() for k, v in model.state_dict().items()}, emb.cpu(), ocfg)
    assert rel_err(full.cpu(), ref) < 1e-5


@pytest.mark.parametrize("tag,prec,gate", [("small", "fp32", GATE), ("full", "fp32", GATE), ("full", "bf16x3", 5e-3)])
def test_str_generate_against_reference(dev, golden, tag, prec, gate):
    """STR_Mixtral.generate on the device against the recording of the reference's: trajectory prior, key points and map tokens.
    The device runs the decoder once over the context and then only over new tokens; the reference ran it six times over the
    growing prefix."""
    g = golden("g21_str_generate")
    model, _, batch = str_model_from_recording(g, tag, prec)
    model = model.to(dev)
    out = model.generate(**{k: v.to(dev) for k, v in batch.items()})
    errs = {k: rel_err(out[k].cpu(), g[f"{tag}_{k}"]) for k in ("traj_logits", "key_points_logits")}
    errs["maps_info"] = rel_err(out["maps_info"][:, ::7].cpu(), g[tag + "_maps_info_every7"])
    print(f"STR generate ({tag}, {prec}): " + ", ".join(f"{k} {v:.1e}" for k, v in errs.items()))
    B = batch["trajectory_label"].shape[0]
    assert out["traj_logits"].shape == (B, 80, 4) and out["key_points_logits"].shape == (B, 5, 2) and out["maps_info"].shape[1] == 788
    assert max(errs.values()) < gate
    again = model.generate(**{k: v.to(dev) for k, v in batch.items()})
    assert torch.equal(again["traj_logits"], out["traj_logits"])
    # the final pass's hidden states against the recording, through the public full-pass entry
    embeds, info, _ = model.encoder(is_training=False, **{k: v.to(dev) for k, v in batch.items()})
    kp_emb = model._embed_key_points(out["key_points_logits"])
    embeds = embeds.clone()
    embeds[:, 788:793] = kp_emb
    hid = model.embedding_to_hidden(embeds)["last_hidden_state"]
    e_tail, e_16 = rel_err(hid[:, 780:].cpu(), g[tag + "_hidden_tail"]), rel_err(hid[:, :780:16].cpu(), g[tag + "_hidden_every16"])
    print(f"  last_hidden_state: tail {e_tail:.1e}, every 16th {e_16:.1e}")
    assert max(e_tail, e_16) < gate


def test_rasters_to_refined_trajectories_on_the_device(dev, golden):
    """Phase 1 and phase 2 chained as the reference chains them (DiffusionAugmentation.generate calls str_model.generate and refines
    its traj_logits conditioned on its maps_info): rasters in, refined trajectories out, nothing leaves the device; equal to handing
    the same prior and map tokens over by hand."""
    from autodiffusion_amd import refiner, str_backbone, synth
    B, fs, N = 2, 10, 10
    cfg = str_backbone.str_config("mixtral-small", n_layer=2, vit_intermediate_size=256, precision="bf16x3")
    model = synth.fill_module_(str_backbone.STR_Mixtral(cfg), WSEED).to(dev).eval()
    batch = dict(high_res_raster=synth.occupancy_raster(B, 58, ISEED).to(dev), low_res_raster=synth.occupancy_raster(B, 58, ISEED + 1).to(dev),
                 context_actions=(synth.noise(B, (), (4, 4), NSEED + 7) * torch.tensor([2.0, 0.3, 0.0, 0.1])).to(dev),
                 trajectory_label=synth.trajectory_prior(B, ISEED).to(dev))
    da = refiner.DiffusionAugmentation(refiner.refiner_config(frame_stack=fs, diffusion_timesteps=N, objective="pred_noise", beta_schedule="cosine",
                                                                precision="bf16x3"), str_model=model)
    synth.reference_init_(da.diffusion, WSEED)
    da = da.to(dev).eval()
    noise = synth.noise(B, (80 // fs, N), (fs, 4), NSEED).to(dev)
    out = da.generate(noise=noise, **batch)
    phase1 = model.generate(**batch)
    by_hand = da.generate(trajectory_prior=phase1["traj_logits"], maps_info=phase1["maps_info"], noise=noise)
    assert out["traj_logits"].shape == (B, 80, 4) and bool(torch.isfinite(out["traj_logits"]).all())
    assert torch.equal(out["traj_logits"], by_hand["traj_logits"])


@pytest.mark.parametrize("B,H,Sq,Skv,dh,mask_off", [(2, 8, 788, 788, 32, 0), (3, 4, 1, 793, 32, 792), (2, 2, 81, 873, 64, 792), (1, 3, 200, 200, 48, -1),
                                                    (2, 2, 37, 300, 128, 263), (1, 2, 130, 130, 32, 1)])
def test_op_attention_long_sequences(dev, B, H, Sq, Skv, dh, mask_off):
    """Attention over more than 128 keys and / or under the 'key j visible to query i iff j <= i + mask_off' mask with head widths
    that are multiples of 16 (keys walked in chunks with an online softmax on the fp32 matrix cores): the decoder's context pass,
    its one-token passes over cached keys, its final pass, an unmasked case and wide heads, against fp64."""
    from autodiffusion_amd import ops
    g = torch.Generator().manual_seed(Sq + Skv + dh)
    q, k, v = (torch.randn(B, s, H * dh, generator=g) for s in (Sq, Skv, Skv))
    got = ops.attention(q.to(dev), k.to(dev), v.to(dev), H, mask_off=mask_off).cpu()
    qd, kd, vd = (t.double().reshape(B, -1, H, dh).transpose(1, 2) for t in (q, k, v))
    s = qd @ kd.transpose(-1, -2) / dh ** 0.5
    if mask_off >= 0:
        vis = torch.arange(Skv)[None, :] <= torch.arange(Sq)[:, None] + mask_off
        s = s.masked_fill(~vis, float("-inf"))
    ref = (torch.softmax(s, dim=-1) @ vd).transpose(1, 2).reshape(B, Sq, H * dh)
    e = rel_err(got, ref)
    print(f"attention B={B} H={H} Sq={Sq} Skv={Skv} dh={dh} mask_off={mask_off}: rel err {e:.1e}")
    assert e < 5e-6


@pytest.mark.parametrize("key_points", ["no", "specified_two_backward"])
def test_str_generate_other_key_point_settings(dev, key_points):
    """generate() without key points (one decoder pass, the prior read from the trajectory slots) and with two key points, against
    the CPU oracle on the same seeded weights (the oracle is pinned by the recording at five key points)."""
    from autodiffusion_amd import str_backbone, synth
    from oracle import str_backbone as osb
    B = 2
    cfg = str_backbone.str_config("mixtral-small", n_embd=64, n_layer=2, n_head=4, n_inner=128, num_key_value_heads=2, vit_intermediate_size=96,
                                  use_key_points=key_points)
    model = synth.fill_module_(str_backbone.STR_Mixtral(cfg), WSEED).eval()
    batch = dict(high_res_raster=synth.occupancy_raster(B, 58, ISEED + 5), low_res_raster=synth.occupancy_raster(B, 58, ISEED + 6),
                 context_actions=synth.noise(B, (), (4, 4), NSEED + 3) * torch.tensor([2.0, 0.3, 0.0, 0.1]), trajectory_label=synth.trajectory_prior(B, ISEED))
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    ocfg = dict(n_layer=2, n_head=4, n_kv_head=2, n_experts=8, top_k=2, rms_eps=1e-5, rope_theta=1e6)
    sel = [] if key_points == "no" else [79, 4]
    ref = osb.str_generate(sd, ocfg, batch["high_res_raster"], batch["low_res_raster"], batch["context_actions"], batch["trajectory_label"], sel)
    out = model.to(dev).generate(**{k: v.to(dev) for k, v in batch.items()})
    e = rel_err(out["traj_logits"].cpu(), ref["traj_logits"])
    print(f"STR generate, key points {key_points!r}: traj_logits rel err {e:.1e}")
    assert e < GATE and rel_err(out["maps_info"].cpu(), ref["maps_info"]) < GATE
    if sel:
        assert out["key_points_logits"].shape == (B, 2, 2) and rel_err(out["key_points_logits"].cpu(), ref["key_points_logits"]) < GATE
    else:
        assert "key_points_logits" not in out


def test_str_generate_replayed_from_a_graph(dev, golden):
    """The whole of phase 1 captured once and replayed: the same bits as the eager call, for new inputs of the same shape too."""
    from autodiffusion_amd import str_backbone
    g = golden("g21_str_generate")
    model, _, batch = str_model_from_recording(g, "small")
    model = model.to(dev)
    batch = {k: v.to(dev) for k, v in batch.items()}
    eager = {k: v.clone() for k, v in model.generate(**batch).items()}
    graphed = str_backbone.GraphedGenerate(model)
    out = graphed(**batch)
    assert all(torch.equal(out[k], eager[k]) for k in eager)
    other = dict(batch, context_actions=batch["context_actions"] * 0.5, high_res_raster=batch["low_res_raster"], low_res_raster=batch["high_res_raster"])
    eager2 = {k: v.clone() for k, v in model.generate(**other).items()}
    out2 = graphed(**other)
    assert len(graphed._graphs) == 1 and all(torch.equal(out2[k], eager2[k]) for k in eager2)
    assert not torch.equal(eager2["traj_logits"], eager["traj_logits"])
