"""GPU parity of the refiner's training path: each backward operator (through the C ABI) against reverse-mode differentiation
of the same operator in float64 on the CPU, the whole backward against the gradients the reference's own ``loss.backward()``
left in its parameters (fixture G14), and the gradient that reaches ``init_x`` against the oracle's.

Tolerance: 1e-3 relative (L2 norm per tensor), the per-step gate of the fp32 path; observed values are printed.
"""
import numpy as np
import pytest
import torch

from conftest import ISEED, NSEED, WSEED, rel_err

pytestmark = pytest.mark.gpu

GATE = 1e-3


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def l2_rel(got, want):
    got, want = torch.as_tensor(got).double().cpu(), torch.as_tensor(want).double().cpu()
    return float((got - want).norm() / want.norm().clamp_min(1e-30))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from autodiffusion_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


# ------------------------------------------------------------------ operators
@pytest.mark.parametrize("M,N,K", [(20, 4, 32), (20, 32, 4), (1618, 96, 32), (1000, 32, 256), (5003, 128, 32), (300000, 32, 128), (7, 33, 70)])
def test_op_linear_wgrad(dev, M, N, K):
    from autodiffusion_amd import train
    g = torch.Generator().manual_seed(M + N * 3 + K)
    dy, x = torch.randn(M, N, generator=g), torch.randn(M, K, generator=g)
    dw0, db0 = torch.randn(N, K, generator=g), torch.randn(N, generator=g)
    dw, db = dw0.to(dev), db0.to(dev)
    train.linear_wgrad(dy.to(dev), x.to(dev), dw, db, accumulate=True)
    assert l2_rel(dw, dw0.double() + dy.double().T @ x.double()) < 2e-6
    assert l2_rel(db, db0.double() + dy.double().sum(0)) < 2e-6
    dw2 = torch.full((N, K), 7.0, device=dev)
    train.linear_wgrad(dy.to(dev), x.to(dev), dw2, None, accumulate=False)
    assert l2_rel(dw2, dy.double().T @ x.double()) < 2e-6
    again = torch.empty_like(dw2)
    train.linear_wgrad(dy.to(dev), x.to(dev), again, None, accumulate=False)
    assert torch.equal(again, dw2)                         # fixed summation order: bit-identical re-run


@pytest.mark.parametrize("M,N", [(2, 25888), (16, 320), (1618, 32), (100001, 32), (33, 96), (5, 700)])
def test_op_colsum_transpose_add(dev, M, N):
    from autodiffusion_amd import train
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, N, generator=g)
    out = torch.ones(N, device=dev)
    train.colsum(x.to(dev), out, accumulate=True)
    assert l2_rel(out, 1.0 + x.double().sum(0)) < 2e-6
    if M * N < 200000:
        assert torch.equal(train.transpose(x.to(dev)).cpu(), x.T.contiguous())
    assert torch.equal(train.add(x.to(dev), (2 * x).to(dev)).cpu(), x + 2 * x)


def test_op_gelu_and_loss_gradient(dev):
    from autodiffusion_amd import ops, train
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(4096, generator=g) * 3).requires_grad_(True)
    dy = torch.randn(4096, generator=g)
    y = torch.nn.functional.gelu(x.double())
    y.backward(dy.double())
    assert rel_err(train.gelu(x.detach().to(dev)).cpu(), y.detach()) < 2e-6
    assert rel_err(train.gelu_bwd(x.detach().to(dev), dy.to(dev)).cpu(), x.grad) < 2e-6
    out, target = torch.randn(5, 10, 4, generator=g), torch.randn(5, 10, 4, generator=g)
    t, w = torch.tensor([0, 3, 9, 3, 1]), torch.rand(10, generator=g) + 0.1
    o = out.double().requires_grad_(True)
    loss = (((o - target.double()) ** 2).reshape(5, -1).mean(1) * w.double()[t]).mean()
    (loss * 1.7).backward()
    got = train.weighted_mse_bwd(out.to(dev), target.to(dev), t.to(dev), w.to(dev), torch.tensor(1.7, device=dev))
    assert rel_err(got.cpu(), o.grad) < 2e-6
    assert abs(float(ops.weighted_mse(out.to(dev), target.to(dev), t.to(dev), w.to(dev))[1]) - float(loss.detach())) < 1e-6 * abs(float(loss.detach()))


@pytest.mark.parametrize("rows,d", [(1618, 32), (20, 32), (77, 128), (5, 1024), (9, 24)])
def test_op_layernorm_bwd(dev, rows, d):
    from autodiffusion_amd import train
    g = torch.Generator().manual_seed(rows + d)
    x = (torch.randn(rows, d, generator=g) * 2 + 0.5).double().requires_grad_(True)
    gamma = (torch.randn(d, generator=g) * 0.5 + 1).double().requires_grad_(True)
    beta = torch.zeros(d, dtype=torch.float64, requires_grad=True)
    dy = torch.randn(rows, d, generator=g)
    torch.nn.functional.layer_norm(x, (d,), gamma, beta, 1e-5).backward(dy.double())
    dx, dyx = train.layernorm_bwd(x.detach().float().to(dev), gamma.detach().float().to(dev), dy.to(dev), 1e-5)
    assert l2_rel(dx, x.grad) < 5e-6
    assert l2_rel(dyx.double().sum(0), gamma.grad) < 5e-6
    dx0, none = train.layernorm_bwd(x.detach().float().to(dev), None, dy.to(dev), 1e-5, want_dy_xhat=False)
    x2 = x.detach().clone().requires_grad_(True)
    torch.nn.functional.layer_norm(x2, (d,), None, None, 1e-5).backward(dy.double())
    assert none is None and l2_rel(dx0, x2.grad) < 5e-6


@pytest.mark.parametrize("B,H,Sq,Skv,dh,mask_off", [(2, 8, 809, 809, 4, -1), (3, 8, 10, 10, 4, 0), (3, 8, 10, 809, 4, 1), (2, 8, 80, 949, 4, 1),
                                                   (1, 2, 130, 257, 8, -1), (2, 4, 33, 33, 16, 0), (2, 8, 80, 80, 32, -1), (1, 3, 200, 140, 32, 2),
                                                   (2, 4, 80, 80, 128, -1), (1, 2, 80, 80, 512, -1), (2, 3, 37, 61, 64, 1)])
def test_op_attention_bwd(dev, B, H, Sq, Skv, dh, mask_off):
    from autodiffusion_amd import ops, train
    from oracle import nets
    g = torch.Generator().manual_seed(Sq * 3 + Skv + dh)
    q, k, v = (torch.randn(B, S, H * dh, generator=g) for S in (Sq, Skv, Skv))
    do = torch.randn(B, Sq, H * dh, generator=g)
    qd, kd, vd = (a.double().requires_grad_(True) for a in (q, k, v))
    allowed = None
    if mask_off >= 0:
        allowed = torch.arange(Skv)[None, :] <= torch.arange(Sq)[:, None] + mask_off
    nets.attention(qd, kd, vd, H, allowed).backward(do.double())
    o = ops.attention(q.to(dev), k.to(dev), v.to(dev), H, mask_off=mask_off)
    dq, dk, dv = train.attention_bwd(q.to(dev), k.to(dev), v.to(dev), o, do.to(dev), H, mask_off=mask_off)
    errs = [l2_rel(dq, qd.grad), l2_rel(dk, kd.grad), l2_rel(dv, vd.grad)]
    print("attention_bwd", (B, H, Sq, Skv, dh, mask_off), ["%.1e" % e for e in errs])
    assert max(errs) < 2e-5
    # the training forward (output + log-sum-exps) and the backward that starts from its statistics
    o2, stats = train.attention_stats(q.to(dev), k.to(dev), v.to(dev), H, mask_off=mask_off)
    want_o = nets.attention(q.double(), k.double(), v.double(), H, allowed)
    assert l2_rel(o2, want_o) < 2e-6 and l2_rel(o, want_o) < 2e-6
    s = torch.einsum("bihd,bjhd->bhij", q.double().reshape(B, Sq, H, dh), k.double().reshape(B, Skv, H, dh)) * dh ** -0.5
    if allowed is not None:
        s = s.masked_fill(~allowed, float("-inf"))
    assert float((stats[..., 0].cpu().double() - torch.logsumexp(s, -1) / np.log(2.0)).abs().max()) < 1e-4
    dq2, dk2, dv2 = train.attention_bwd(q.to(dev), k.to(dev), v.to(dev), o2, do.to(dev), H, mask_off=mask_off, stats=stats)
    assert max(l2_rel(dq2, qd.grad), l2_rel(dk2, kd.grad), l2_rel(dv2, vd.grad)) < 2e-5


# ------------------------------------------------------------------ the whole backward
def _refiner(dev, fs=10, N=100, obj="pred_noise", sched="cosine"):
    from autodiffusion_amd import refiner, synth
    tr = refiner.TrajectoryRefiner(refiner.refiner_config(frame_stack=fs, diffusion_timesteps=N, objective=obj, beta_schedule=sched))
    synth.fill_module_(tr, WSEED)
    tr = tr.to(dev).train()
    tr.diffusion.model.eval()           # the wrapper in training mode (it returns the loss), the denoiser without dropout: as G14 was recorded
    return tr


def _g14_inputs(dev, fs=10):
    from autodiffusion_amd import synth
    maps, prior = synth.maps_info(2, ISEED), synth.trajectory_prior(2, ISEED)
    label = prior + synth.noise(2, (), (80, 4), NSEED + 1) * torch.tensor([1.5, 0.4, 0.0, 0.05])
    init_x = synth.seeded_param("init_x", (fs, 4), WSEED)
    return dict(maps=maps.to(dev), prior=prior[:, :fs].contiguous().to(dev), label=label[:, :fs].contiguous().to(dev),
                transition=init_x[None].expand(2, fs, 4).contiguous().to(dev), noise=synth.noise(2, (), (fs, 4), NSEED + 3).to(dev))


def test_refiner_gradients_match_the_reference(dev, golden):
    """loss.backward() through TrajectoryRefiner.forward on the HIP training path against the gradients the reference's own
    backward() produced for the same weights, inputs, timesteps and noise: all 108 parameter tensors."""
    g = golden("g14_refiner_grads")
    tr = _refiner(dev)
    c = _g14_inputs(dev)
    loss, logits = tr(c["label"], c["transition"], c["prior"], c["maps"], t=T(g["t"]).to(dev), noise=c["noise"])
    assert loss.requires_grad and not logits.any()
    np.testing.assert_allclose(float(loss), float(g["loss"]), rtol=1e-4)
    loss.backward()
    want = {k[len("grad."):]: v for k, v in g.items() if k.startswith("grad.")}
    got = {n: p.grad for n, p in tr.diffusion.model.named_parameters()}
    assert set(want) == set(got) and len(want) == 108
    worst = max((l2_rel(got[k], v), k) for k, v in want.items())
    print("worst gradient error %.2e at %s" % worst)
    assert worst[0] < GATE
    # a second backward accumulates into .grad, as torch does
    loss2, _ = tr(c["label"], c["transition"], c["prior"], c["maps"], t=T(g["t"]).to(dev), noise=c["noise"])
    loss2.backward()
    assert l2_rel(tr.diffusion.model.head.weight.grad, 2 * want["head.weight"]) < GATE
    # the value-only evaluation entry (no_grad) reports the same loss from the sampling engine
    with torch.no_grad():
        val, _ = tr(c["label"], c["transition"], c["prior"], c["maps"], t=T(g["t"]).to(dev), noise=c["noise"])
    assert not val.requires_grad
    np.testing.assert_allclose(float(val), float(g["loss"]), rtol=1e-3)


def test_init_x_gradient_and_a_training_step(dev):
    """DiffusionAugmentation.forward in gradient mode: the gradient reaching init_x (through the transition rows of chunk 0)
    against reverse-mode differentiation of the oracle, and a few SGD steps on fixed draws lower the loss."""
    from autodiffusion_amd import refiner, synth
    from oracle import nets, sampling, schedules
    fs, N = 10, 100
    da = refiner.DiffusionAugmentation(refiner.refiner_config(frame_stack=fs, diffusion_timesteps=N))
    synth.fill_module_(da.diffusion, WSEED)
    with torch.no_grad():
        da.init_x.copy_(synth.seeded_param("init_x", (fs, 4), WSEED))
    da = da.to(dev).train()
    da.diffusion.diffusion.model.eval()             # no dropout: the comparison below is with a dropout-free computation
    maps, prior = synth.maps_info(2, ISEED), synth.trajectory_prior(2, ISEED)
    label = prior + synth.noise(2, (), (80, 4), NSEED + 1) * torch.tensor([1.5, 0.4, 0.0, 0.05])
    t = torch.from_numpy(np.random.default_rng(5).integers(0, N, size=(8, 2)).astype(np.int64))
    noise = synth.noise(2, (8,), (fs, 4), NSEED + 2)
    out = da(trajectory_prior=prior.to(dev), maps_info=maps.to(dev), trajectory_label=label.to(dev), t=t.to(dev), noise=noise.to(dev))
    out.loss.backward()
    sd = {k[len("diffusion.diffusion.model."):]: v.detach().float().cpu() for k, v in da.state_dict().items() if k.startswith("diffusion.diffusion.model.")}
    init_x = da.init_x.detach().cpu().clone().requires_grad_(True)
    fn = lambda x, tt, tra_n, pri_n: nets.variant_a_forward(sd, x, tt, maps, tra_n, pri_n)  # noqa: E731
    ref_loss, _ = sampling.refiner_forward(fn, schedules.buffers("cosine", N, "pred_noise"), "pred_noise", prior, label, init_x, t, noise, fs)
    ref_loss.backward()
    np.testing.assert_allclose(float(out.loss), float(ref_loss), rtol=1e-4)
    assert l2_rel(da.init_x.grad, init_x.grad) < GATE
    # plain SGD on the same draws
    opt = torch.optim.SGD(da.parameters(), lr=0.05)
    first = None
    for _ in range(4):
        opt.zero_grad()
        loss = da(trajectory_prior=prior.to(dev), maps_info=maps.to(dev), trajectory_label=label.to(dev), t=t.to(dev), noise=noise.to(dev)).loss
        loss.backward()
        opt.step()
        first = float(loss) if first is None else first
    assert float(loss) < first


def test_distributed_data_parallel_wraps_the_training_path(dev):
    """torch's DistributedDataParallel over RCCL (a process group of this one rank) around the accelerated module: its gradient
    hooks fire on the gradients the library's backward returns, and the averaged gradients equal the unwrapped module's."""
    import os
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel
    from autodiffusion_amd import refiner, synth
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    from autodiffusion_amd.distributed import free_port
    os.environ.setdefault("MASTER_PORT", str(free_port()))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    fs, N, B = 10, 100, 4
    maps, prior = synth.maps_info(B, ISEED).to(dev), synth.trajectory_prior(B, ISEED).to(dev)
    label = prior + synth.noise(B, (), (80, 4), NSEED + 1).to(dev) * torch.tensor([1.5, 0.4, 0.0, 0.05], device=dev)
    t = torch.from_numpy(np.random.default_rng(5).integers(0, N, size=(8, B)).astype(np.int64)).to(dev)
    noise = synth.noise(B, (8,), (fs, 4), NSEED + 2).to(dev)

    def build():
        da = refiner.DiffusionAugmentation(refiner.refiner_config(frame_stack=fs, diffusion_timesteps=N))
        synth.fill_module_(da.diffusion, WSEED)
        da = da.to(dev).train()
        da.diffusion.diffusion.model.eval()         # no dropout: the two modules must see the same computation
        return da
    plain = build()
    plain(trajectory_prior=prior, maps_info=maps, trajectory_label=label, t=t, noise=noise).loss.backward()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        wrapped = build()
        wrapped.init_x.data.copy_(plain.init_x.data)
        ddp = DistributedDataParallel(wrapped, device_ids=[dev.index])
        ddp(trajectory_prior=prior, maps_info=maps, trajectory_label=label, t=t, noise=noise).loss.backward()
        torch.cuda.synchronize()
        for (n, a), (_, b) in zip(plain.named_parameters(), wrapped.named_parameters()):
            assert a.grad is not None and b.grad is not None, n
            assert torch.equal(a.grad, b.grad), n
    finally:
        dist.destroy_process_group()


def test_explicit_wrapper_gradients(dev):
    """ExplicitDiT.forward in gradient mode (flattened chunks: one token of fs*4 channels; each chunk's prediction is the next
    chunk's transition, un-detached as in ExplicitDiT.py:107-114): loss and gradients against reverse-mode differentiation of
    the same computation built from the oracle network.  (The wrapper's noise schedule is third-party: parity unpinned.)"""
    from autodiffusion_amd import explicit, synth
    from oracle import nets
    fs, B = 20, 3
    cfg = explicit.explicit_config(frame_stack=fs, num_train_timesteps=100, num_inference_steps=10)
    m = explicit.ExplicitDiT(cfg)
    synth.fill_module_(m.diffusion, WSEED)
    with torch.no_grad():
        m.init_x.copy_(synth.seeded_param("init_x", (fs * 4,), WSEED))
    m = m.to(dev).train()
    m.diffusion.diffusion.eval()                    # no dropout: compared with reverse-mode differentiation of the dropout-free oracle
    n_chunks, W = 80 // fs, fs * 4
    maps, prior = synth.maps_info(B, ISEED), synth.trajectory_prior(B, ISEED)
    label = prior + synth.noise(B, (), (80, 4), NSEED + 1) * torch.tensor([1.5, 0.4, 0.0, 0.05])
    t = torch.from_numpy(np.random.default_rng(9).integers(1, 100, size=(n_chunks, B)).astype(np.int64))
    noise = synth.noise(B, (n_chunks,), (W,), NSEED + 2)
    out = m(trajectory_prior=prior.to(dev), maps_info=maps.to(dev), trajectory_label=label.to(dev), t=t.to(dev), noise=noise.to(dev))
    assert out.loss.requires_grad
    out.loss.backward()
    # CPU restatement of the same objective over the oracle network
    sd = {k[len("diffusion.diffusion."):]: v.detach().float().cpu().clone().requires_grad_(v.is_floating_point())
          for k, v in m.state_dict().items() if k.startswith("diffusion.diffusion.")}
    init_x = m.init_x.detach().cpu().clone().requires_grad_(True)
    acp = torch.cumprod(1.0 - m.diffusion.betas.detach().cpu().float(), dim=0)
    mean, std = m.data_mean.cpu(), m.data_std.cpu()
    lab = ((label - mean) / std).reshape(B, n_chunks, W)
    pri = prior.reshape(B, n_chunks, W)
    transition = init_x[None].expand(B, W)
    losses = []
    for c in range(n_chunks):
        a, s1 = acp[t[c]].sqrt()[:, None, None], (1.0 - acp[t[c]]).sqrt()[:, None, None]
        x = a * lab[:, c, None, :] + s1 * noise[c][:, None, :]
        pred = nets.variant_a_forward(sd, x, t[c], maps, transition[:, None, :], pri[:, c, None, :]).reshape(B, W)
        losses.append((pred - noise[c]) ** 2)
        transition = pred
    ref = torch.stack(losses).mean()
    ref.backward()
    np.testing.assert_allclose(float(out.loss.detach()), float(ref.detach()), rtol=1e-4)
    worst = max((l2_rel(p.grad, sd[n].grad), n) for n, p in m.diffusion.diffusion.named_parameters())
    print("explicit wrapper: worst gradient error %.2e at %s" % worst)
    assert worst[0] < GATE and l2_rel(m.init_x.grad, init_x.grad) < GATE
    with torch.no_grad():
        val = m(trajectory_prior=prior.to(dev), maps_info=maps.to(dev), trajectory_label=label.to(dev), t=t.to(dev), noise=noise.to(dev))
    assert not val.loss.requires_grad
    np.testing.assert_allclose(float(val.loss), float(ref.detach()), rtol=1e-3)


@pytest.mark.parametrize("code", [1, 2, 3, 4, 5])
def test_op_activations_mul_group_colsum(dev, code):
    from autodiffusion_amd import train
    fns = {1: torch.relu, 2: torch.nn.functional.gelu, 3: lambda v: torch.nn.functional.gelu(v, approximate="tanh"),
           4: lambda v: torch.nn.functional.leaky_relu(v, 0.01), 5: torch.nn.functional.silu}
    g = torch.Generator().manual_seed(code)
    x = (torch.randn(5000, generator=g) * 3).double().requires_grad_(True)
    dy = torch.randn(5000, generator=g)
    y = fns[code](x)
    y.backward(dy.double())
    assert rel_err(train.act(x.detach().float().to(dev), code).cpu(), y.detach()) < 3e-6
    assert rel_err(train.act_bwd(x.detach().float().to(dev), dy.to(dev), code).cpu(), x.grad) < 3e-6
    a, b = torch.randn(7, 80, 33, generator=g), torch.randn(7, 80, 33, generator=g)
    assert torch.equal(train.mul(a.to(dev), b.to(dev)).cpu(), a * b)
    assert l2_rel(train.group_colsum(a.to(dev)), a.double().sum(1)) < 2e-6


def check_grad_fixture(g, grads, tol):
    seen = 0
    for k, v in g.items():
        kind, _, name = k.partition(".")
        if kind == "grad":
            assert l2_rel(grads[name].reshape(-1), v) < tol, name
        elif kind == "gradnorm":
            flat = grads[name].reshape(-1)
            sample = g["gradsample." + name]
            assert abs(float(flat.norm()) - float(v)) < tol * float(v), name
            assert l2_rel(flat[::flat.numel() // 1024][:sample.shape[0]], sample) < tol, name
        else:
            continue
        seen += 1
    return seen


@pytest.mark.parametrize("backbone,fixture", [("DiT", "g15_twin_grads"), ("MM-DiT", "g16_mmdit_grads")])
def test_decoder_twin_gradients_match_the_reference(dev, golden, backbone, fixture):
    """loss.backward() through DiffusionDecoder.forward (hidden 64, 2 blocks, 4 heads, both backbones) on the HIP training path
    against the gradients of the reference's own backward(): every parameter tensor and the backbone hidden state."""
    from autodiffusion_amd import dit, synth
    g = golden(fixture)
    dec = dit.DiffusionDecoder(dit.dit_config(64, 2, 4, backbone, diffusion_timesteps=10))
    synth.fill_module_(dec, WSEED)
    dec = dec.to(dev).train()
    dec.model.model.eval()                          # the network without dropout, as the fixtures were recorded
    state = synth.hidden_state(2, 64, ISEED).to(dev).requires_grad_(True)
    label = (synth.trajectory_prior(2, ISEED) + synth.noise(2, (), (80, 4), NSEED + 1) * torch.tensor([1.5, 0.4, 0.0, 0.05])).to(dev)
    loss, logits = dec(label, state, t=T(g["t"]).to(dev), noise=synth.noise(2, (), (80, 4), NSEED + 2).to(dev))
    assert loss.requires_grad and not logits.any()
    np.testing.assert_allclose(float(loss.detach()), float(g["loss"]), rtol=1e-4)
    loss.backward()
    grads = {n: p.grad for n, p in dec.model.model.named_parameters() if p.grad is not None}
    n = check_grad_fixture(g, grads, GATE)
    assert n > 60
    in_fixture = {k.partition(".")[2] for k in g if k.startswith(("grad.", "gradnorm."))}
    # parameters the loss does not depend on (linear_c / mlp_c of the last MM-DiT block, whose condition-stream output is unused):
    # the reference's autograd leaves their .grad unset, and so does this path -- an optimizer then does not touch them either
    assert set(grads) == in_fixture, sorted(set(grads) ^ in_fixture)
    if backbone == "MM-DiT":
        assert "blocks.1.mlp_c.0.weight" not in grads and "blocks.1.linear_c.0.bias" not in grads and "blocks.0.mlp_c.0.weight" in grads
    assert l2_rel(state.grad, g["grad_state"]) < GATE
    with torch.no_grad():
        val, _ = dec(label, state, t=T(g["t"]).to(dev), noise=synth.noise(2, (), (80, 4), NSEED + 2).to(dev))
    assert not val.requires_grad
    np.testing.assert_allclose(float(val), float(g["loss"]), rtol=1e-3)


# ------------------------------------------------------------------ dropout
DSEED = 4242


def _seeded_multipliers(dev, seed):
    from autodiffusion_amd import synth
    return lambda k, p, shape: synth.dropout_multipliers(shape, p, seed, k).to(dev)


def _hash_multipliers(dev, seed, log=None):
    """The multipliers the device hash gives, materialised layer by layer (attention layers are the 4-D ones)."""
    from autodiffusion_amd import train

    def make(k, p, shape):
        site = train.DropSite(p, seed, k, None)
        m = train.attention_dropout_mask(*shape, site, dev) if len(shape) == 4 else train.dropout_mask(shape, site, dev)
        if log is not None:
            log.append(m)
        return m
    return make


def test_dropout_operator(dev):
    """adx_op_dropout: multipliers are 0 or 1 / (1 - p), the keep rate is 1 - p, the draw is a function of (seed, layer, position)
    only, y = x * multipliers exactly; the attention form gives the same multipliers as the kernels use."""
    from autodiffusion_amd import train
    n, p = (64, 1000, 32), 0.1
    a = train.dropout_mask(n, train.DropSite(p, 77, 3, None), dev)
    vals = torch.unique(a)
    assert vals.numel() == 2 and float(vals[0]) == 0.0 and abs(float(vals[1]) - 1.0 / 0.9) < 1e-6
    keep = float((a > 0).float().mean())
    assert abs(keep - 0.9) < 2e-3, keep
    assert torch.equal(a, train.dropout_mask(n, train.DropSite(p, 77, 3, None), dev))
    for other in (train.DropSite(p, 78, 3, None), train.DropSite(p, 77, 4, None)):
        b = train.dropout_mask(n, other, dev)
        agree = float(((a > 0) == (b > 0)).float().mean())
        assert abs(agree - 0.82) < 5e-3, agree              # independent draws agree with probability 0.9^2 + 0.1^2
    # no structure along rows or columns: keep rates per row and per column stay within sampling noise
    flat = (a > 0).float().reshape(-1, 32)
    assert float(flat.mean(0).std()) < 3e-3 and abs(float(flat.mean(1).std()) - (0.09 / 32) ** 0.5) < 5e-3
    x = torch.randn(n, device=dev)
    assert torch.equal(train.dropout(x, train.DropSite(p, 77, 3, None)), x * a)
    assert train.dropout(x, None) is x
    # attention probabilities: out = (softmax(q k^T) * multipliers) v with the multipliers of attention_dropout_mask
    B, H, Sq, Skv, dh = 2, 4, 37, 150, 16
    q, k, v = (torch.randn(B, s, H * dh, device=dev) for s in (Sq, Skv, Skv))
    site = train.DropSite(0.25, 5, 11, None)
    m = train.attention_dropout_mask(B, H, Sq, Skv, site, dev)
    assert abs(float((m > 0).float().mean()) - 0.75) < 5e-3
    o, _ = train.attention_stats(q, k, v, H, drop=site)
    s = torch.einsum("bihd,bjhd->bhij", q.double().reshape(B, Sq, H, dh), k.double().reshape(B, Skv, H, dh)) * dh ** -0.5
    want = torch.einsum("bhij,bjhd->bihd", torch.softmax(s, -1) * m.double(), v.double().reshape(B, Skv, H, dh)).reshape(B, Sq, H * dh)
    assert l2_rel(o, want) < 2e-6
    o2, _ = train.attention_stats(q, k, v, H, drop=train.DropSite(0.25, 0, 0, m))
    assert torch.equal(o, o2)
    with pytest.raises(Exception):
        train.dropout_mask(n, train.DropSite(1.0, 1, 1, None), dev)


@pytest.mark.parametrize("mask_off", [-1, 0, 1])
def test_attention_backward_with_dropout(dev, mask_off):
    """dq, dk, dv of softmax -> dropout -> weighted sum against float64 autograd with the same multipliers, for the three mask
    rules; hash mode and supplied-multiplier mode are the same computation bit for bit."""
    from autodiffusion_amd import train
    B, H, Sq, Skv, dh = 2, 4, 45, 45 if mask_off == 0 else 130, 16
    g = torch.Generator().manual_seed(3)
    q, k, v, do = (torch.randn(B, s, H * dh, generator=g) for s in (Sq, Skv, Skv, Sq))
    site = train.DropSite(0.1, 9, 2, None)
    m = train.attention_dropout_mask(B, H, Sq, Skv, site, dev)
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    s = torch.einsum("bihd,bjhd->bhij", qd.reshape(B, Sq, H, dh), kd.reshape(B, Skv, H, dh)) * dh ** -0.5
    if mask_off >= 0:
        i, j = torch.arange(Sq)[:, None], torch.arange(Skv)[None, :]
        s = s.masked_fill(~(j <= i + mask_off), float("-inf"))
    out = torch.einsum("bhij,bjhd->bihd", torch.softmax(s, -1) * m.cpu().double(), vd.reshape(B, Skv, H, dh)).reshape(B, Sq, H * dh)
    out.backward(do.double())
    o, stats = train.attention_stats(q.to(dev), k.to(dev), v.to(dev), H, mask_off=mask_off, drop=site)
    assert l2_rel(o, out.detach()) < 2e-6
    got = train.attention_bwd(q.to(dev), k.to(dev), v.to(dev), o, do.to(dev), H, mask_off=mask_off, stats=stats, drop=site)
    errs = [l2_rel(a, b.grad) for a, b in zip(got, (qd, kd, vd))]
    print("attention_bwd with dropout", mask_off, ["%.1e" % e for e in errs])
    assert max(errs) < 2e-5
    again = train.attention_bwd(q.to(dev), k.to(dev), v.to(dev), o, do.to(dev), H, mask_off=mask_off, stats=stats, drop=train.DropSite(0.1, 0, 0, m))
    assert all(torch.equal(a, b) for a, b in zip(got, again))
    nostats = train.attention_bwd(q.to(dev), k.to(dev), v.to(dev), o, do.to(dev), H, mask_off=mask_off, drop=site)
    assert max(l2_rel(a, b.grad) for a, b in zip(nostats, (qd, kd, vd))) < 2e-5


def test_refiner_gradients_with_dropout_match_the_reference(dev, golden):
    """The whole refiner in training mode (p_drop_emb = p_drop_attn = 0.1): loss and all 108 gradients against the reference's
    backward() recorded with the same dropout multipliers (34 layers' worth, in the order the reference draws them)."""
    from autodiffusion_amd import refiner, synth, train
    g = golden("g19_dropout_grads")
    tr = refiner.TrajectoryRefiner(refiner.refiner_config(frame_stack=10, diffusion_timesteps=100, objective="pred_noise", beta_schedule="cosine"))
    synth.fill_module_(tr, WSEED)
    tr = tr.to(dev).train()
    c = _g14_inputs(dev)
    drawn = []
    seeded = _seeded_multipliers(dev, DSEED)
    train.set_dropout(tr.diffusion.model, masks=lambda k, p, shape: (drawn.append(k), seeded(k, p, shape))[1])
    loss, _ = tr(c["label"], c["transition"], c["prior"], c["maps"], t=T(g["refiner_t"]).to(dev), noise=c["noise"])
    assert len(drawn) == int(g["refiner_draws"]) == 34
    np.testing.assert_allclose(float(loss), float(g["refiner_loss"]), rtol=1e-4)
    loss.backward()
    sub = {k.replace("refiner.", "", 1): v for k, v in g.items() if k.startswith(("grad.refiner.", "gradnorm.refiner.", "gradsample.refiner."))}
    grads = {n: p.grad for n, p in tr.diffusion.model.named_parameters()}
    assert check_grad_fixture(sub, grads, GATE) == 108
    # the loss differs from the dropout-free one (the layers are really active), and eval mode is the dropout-free computation
    assert abs(float(loss) - float(golden("g14_refiner_grads")["loss"])) > 1e-3
    tr.diffusion.model.eval()
    loss0, _ = tr(c["label"], c["transition"], c["prior"], c["maps"], t=T(g["refiner_t"]).to(dev), noise=c["noise"])
    np.testing.assert_allclose(float(loss0), float(golden("g14_refiner_grads")["loss"]), rtol=1e-4)


def test_decoder_twin_gradients_with_dropout_match_the_reference(dev, golden):
    """DiT backbone in training mode: torch's default dropout 0.1 inside each block's encoder layer (8 draws for 2 blocks)."""
    from autodiffusion_amd import dit, synth, train
    g = golden("g19_dropout_grads")
    dec = dit.DiffusionDecoder(dit.dit_config(64, 2, 4, "DiT", diffusion_timesteps=10))
    synth.fill_module_(dec, WSEED)
    dec = dec.to(dev).train()
    state = synth.hidden_state(2, 64, ISEED).to(dev).requires_grad_(True)
    label = (synth.trajectory_prior(2, ISEED) + synth.noise(2, (), (80, 4), NSEED + 1) * torch.tensor([1.5, 0.4, 0.0, 0.05])).to(dev)
    drawn = []
    seeded = _seeded_multipliers(dev, DSEED + 1)
    train.set_dropout(dec.model.model, masks=lambda k, p, shape: (drawn.append(k), seeded(k, p, shape))[1])
    loss, _ = dec(label, state, t=T(g["twin_t"]).to(dev), noise=synth.noise(2, (), (80, 4), NSEED + 2).to(dev))
    assert len(drawn) == int(g["twin_draws"]) == 8
    np.testing.assert_allclose(float(loss.detach()), float(g["twin_loss"]), rtol=1e-4)
    loss.backward()
    sub = {k.replace("twin.", "", 1): v for k, v in g.items() if k.startswith(("grad.twin.", "gradnorm.twin.", "gradsample.twin."))}
    grads = {n: p.grad for n, p in dec.model.model.named_parameters() if p.grad is not None}
    assert check_grad_fixture(sub, grads, GATE) > 60
    assert l2_rel(state.grad, g["twin_grad_state"]) < GATE


def test_hash_dropout_is_the_mask_computation(dev):
    """A training step that draws from the device hash equals, bit for bit, the step that is handed the multipliers the hash gives:
    forward and backward regenerate the same draws, nothing is stored.  Seeds: same seed -> same gradients, a different seed or no
    seed (torch's generator) -> different ones, torch.manual_seed makes the unseeded path repeatable."""
    from autodiffusion_amd import refiner, synth, train
    tr = refiner.TrajectoryRefiner(refiner.refiner_config(frame_stack=10, diffusion_timesteps=100, objective="pred_noise", beta_schedule="cosine"))
    synth.fill_module_(tr, WSEED)
    tr = tr.to(dev).train()
    net = tr.diffusion.model
    c = _g14_inputs(dev)
    t = torch.tensor([37, 81], device=dev)

    def step(**how):
        net.zero_grad()
        if how:
            train.set_dropout(net, **how)
        loss, _ = tr(c["label"], c["transition"], c["prior"], c["maps"], t=t, noise=c["noise"])
        loss.backward()
        return float(loss), {n: p.grad.clone() for n, p in net.named_parameters()}
    l1, g1 = step(seed=123)
    log = []
    l2, g2 = step(masks=_hash_multipliers(dev, 123, log))
    assert len(log) == 34 and l1 == l2 and all(torch.equal(g1[n], g2[n]) for n in g1)
    for m in log:                                   # every layer keeps 90 % of its elements, within 5 sigma of a binomial draw
        assert abs(float((m > 0).float().mean()) - 0.9) < 5 * (0.09 / m.numel()) ** 0.5 + 1e-4, tuple(m.shape)
    l3, g3 = step(seed=123)
    assert l3 == l1 and all(torch.equal(g1[n], g3[n]) for n in g1)
    l4, _ = step(seed=124)
    assert l4 != l1
    torch.manual_seed(5)
    l5, _ = step()
    l6, _ = step()
    torch.manual_seed(5)
    l7, _ = step()
    assert l5 != l6 and l5 == l7
