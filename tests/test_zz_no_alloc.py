"""The allocation-free entry point (adx_op_linear_noalloc) against the entry point it is a variant of (adx_op_linear_staged):
bit-identical outputs on two calls in a row, the second with a smaller input, through the C ABI.  The library's fixed device
array is shared by every call on the GPU, so the second call also shows that nothing left behind by the first one leaks in."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _call(lib, name, x, w, b, M, N, K, act, prec, ws=None):
    from autodiffusion_amd import _lib
    y = torch.empty(M, N, device=x.device, dtype=torch.float32)
    s = _lib.current_stream()
    if ws is None:
        _lib.check(getattr(lib, name)(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), M, N, K, act, prec, s), name)
    else:
        _lib.check(getattr(lib, name)(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), M, N, K, act, prec, _lib.ptr(ws),
                                      ws.numel(), s), name)
    return y


@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
def test_linear_noalloc_equals_staged(precision):
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from autodiffusion_amd import _lib, ops
    lib = _lib.load()
    dev = torch.device("cuda:0")
    prec = _lib.PRECISIONS[precision]
    cap = int(lib.adx_noalloc_scratch_bytes())
    assert cap == 64 << 20
    g = torch.Generator().manual_seed(5)
    # first call: 16384 x 2048 x 512 (ragged neither way); second call: smaller and ragged in M and N
    for M, N, K in ((16384, 2048, 512), (4100, 1000, 256)):
        x = torch.randn(M, K, generator=g).to(dev)
        w32 = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
        b = torch.randn(N, generator=g).to(dev)
        w = ops.prepare_weight(w32, precision)
        need = ops.linear_stage_bytes(M, N, K, precision)
        assert need <= cap
        ws = torch.empty(need + 64, dtype=torch.uint8, device=dev)
        free_before = torch.cuda.mem_get_info()[0]
        got = _call(lib, "adx_op_linear_noalloc", x, w, b, M, N, K, _lib.ACT_GELU_TANH, prec)
        torch.cuda.synchronize()
        if M == 4100:       # (the first call on a device also loads the library's code object, which holds the fixed array)
            assert torch.cuda.mem_get_info()[0] >= free_before - M * N * 4 - (4 << 20)      # at most the output tensor appeared
        import os
        os.environ["ADX_GEMM_DMA_ANY"] = "1"            # the staged entry point must take the same kernel for the comparison
        try:
            ref = _call(lib, "adx_op_linear_staged", x, w, b, M, N, K, _lib.ACT_GELU_TANH, prec, ws)
        finally:
            del os.environ["ADX_GEMM_DMA_ANY"]
        assert torch.equal(got, ref), (M, N, K)
        xr, wr = (x.bfloat16().double(), w32.bfloat16().double()) if precision == "bf16" else (x.double(), w32.double())
        want = torch.nn.functional.gelu(xr @ wr.T + b.double(), approximate="tanh").float()
        assert float((got - want).abs().max() / want.abs().max()) < 2e-5
    # beyond the fixed array: a clear error, nothing launched
    M, N, K = 65536, 4096, 1024
    assert ops.linear_stage_bytes(M, N, K, precision) > cap
    x = torch.empty(M, K, device=dev)
    w = torch.empty(2 * N * K, dtype=torch.int16, device=dev)
    with pytest.raises(_lib.AdxError, match="scratch"):
        _call(lib, "adx_op_linear_noalloc", x, w, None, M, N, K, 0, prec)
    with pytest.raises(_lib.AdxError, match="staged kernel needs"):
        _call(lib, "adx_op_linear_noalloc", x, w, None, 100, 64, 64, 0, prec)


def test_linear_noalloc_under_graph_capture():
    """The callers this entry point exists for: after adx_noalloc_init() (outside any capture) the call is recorded into a graph
    on a capturing stream and replayed; the replay writes the same bits as an eager call, also after the inputs changed in place."""
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from autodiffusion_amd import _lib, ops
    lib = _lib.load()
    dev = torch.device("cuda:0")
    _lib.check(lib.adx_noalloc_init(), "adx_noalloc_init")
    M, N, K = 8192, 1024, 512
    g = torch.Generator().manual_seed(9)
    x = torch.randn(M, K, generator=g).to(dev)
    w = ops.prepare_weight((torch.randn(N, K, generator=g) / K ** 0.5).to(dev), "bf16x3")
    b = torch.randn(N, generator=g).to(dev)
    y = torch.zeros(M, N, device=dev)
    prec = _lib.PRECISIONS["bf16x3"]
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            _lib.check(lib.adx_op_linear_noalloc(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), M, N, K, _lib.ACT_NONE, prec,
                                                 _lib.current_stream()), "adx_op_linear_noalloc (capturing)")
    torch.cuda.current_stream().wait_stream(side)
    for scale in (1.0, -0.5):
        x.mul_(scale)
        y.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(y, _call(lib, "adx_op_linear_noalloc", x, w, b, M, N, K, _lib.ACT_NONE, prec))
