"""Conditioning producer, complete: the model that turns rasters into ``traj_logits`` (the trajectory prior) and ``maps_info`` --
phase 1 of every refiner's ``generate()`` -- backed by libadx.so.

Interfaces mirrored (paths relative to the reference tree):

* ``STR_Mixtral`` / ``STR.generate``          transformer4planning/models/backbone/mixtral.py:12-22, str_base.py:457-813
  (nuplan task, no proposals, ``k`` = 1, key points through the separate embedding and the MLP key-point head; the waymo task,
  proposals, key-point tokenizers, the diffusion key-point head and the point-by-point ``autoregressive`` trajectory raise
  NotImplementedError)
* ``STR.embedding_to_hidden``                 str_base.py:379-387
* ``build_models`` size table                 str_base.py:929-1036 (``mixtral_sizes``)
* ``KeyPointMLPDeocder.generate_keypoints``   models/decoder/base.py:395-413, 543-557
* the ``transformers.MixtralModel`` the reference instantiates (third-party, pinned 4.42.3 in requirements.txt:212): parameter
  names are those of that version, i.e. of a reference checkpoint (``transformer.layers.{i}.block_sparse_moe.experts.{e}.w1.weight``
  ...); state dicts written by newer ``transformers`` releases (fused ``mlp.experts.gate_up_proj`` / ``down_proj``) are split and
  renamed on load.

What differs from the reference by design: the reference re-runs the whole decoder on the growing prefix once per key point
(str_base.py:583-591: five full passes plus the final one).  The decoder is causal, so the hidden states of a prefix do not change
when tokens are appended: here keys and values are cached, the context is processed once and every later pass processes only its new
tokens (one per key point, then the last key point together with the trajectory slots) -- the same numbers from one sixth of the
arithmetic.  Every arithmetic step is a libadx kernel (``adx_op_rmsnorm``, ``adx_op_rotary``, ``adx_op_linear``,
``adx_op_attention`` with ``mask_off`` = cached length, ``adx_op_moe_block``); torch concatenates, slices and holds the cache.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import check, current_stream, f32c, ptr
from .raster_encoder import NuplanRasterizeEncoder, _add, _tanh
from .refiner import _Lin
from .trajectory_decoder import TrajectoryDecoder


def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    x = f32c(x)
    y = torch.empty_like(x)
    check(_lib.load().adx_op_rmsnorm(ptr(x), ptr(f32c(weight)), float(eps), x.numel() // x.shape[-1], x.shape[-1], ptr(y), current_stream()), "rmsnorm")
    return y


def rotary_(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, n_head: int, pos0: int) -> torch.Tensor:
    """In place on x (B, S, n_head * dh); tables (n_pos, dh / 2)."""
    B, S, D = x.shape
    check(_lib.load().adx_op_rotary(ptr(x), ptr(cos), ptr(sin), B, S, n_head, D // n_head, int(pos0), cos.shape[0], current_stream()), "rotary")
    return x


def rotary_tables(n_pos: int, dh: int, theta: float, device) -> tuple:
    """cos / sin (n_pos, dh / 2), fp32, computed on the host the way the rotary embedding defines them (inverse frequencies in fp32,
    position times frequency in fp32) and uploaded: a few kilobytes, once per (length, head width)."""
    inv = 1.0 / (theta ** (torch.arange(0, dh, 2, dtype=torch.int64).float() / dh))
    ang = torch.arange(n_pos, dtype=torch.float32)[:, None] * inv[None, :]
    return ang.cos().to(device).contiguous(), ang.sin().to(device).contiguous()


def moe_block(x: torch.Tensor, res: Optional[torch.Tensor], router_w: torch.Tensor, w13: torch.Tensor, w2: torch.Tensor, inter: int, n_experts: int,
              top_k: int, precision: str = "fp32", workspace: Optional[torch.Tensor] = None, return_routing: bool = False):
    """x, res (T, d) -> res + sum of the top_k experts' outputs; w13 / w2 as ``prepare_expert_weights`` returns them."""
    lib = _lib.load()
    x = f32c(x)
    T, d = x.shape
    need = int(lib.adx_moe_workspace_bytes(T, d, inter, n_experts, top_k))
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=x.device)
    y = torch.empty_like(x)
    check(lib.adx_op_moe_block(ptr(x), ptr(f32c(res)) if res is not None else None, ptr(router_w), ptr(w13), ptr(w2), T, d, inter, n_experts, top_k,
                               _lib.PRECISIONS[precision], ptr(workspace), workspace.numel(), ptr(y), current_stream()), "moe_block")
    if not return_routing:
        return y
    n = T * top_k
    off = (n * 4 + 255) // 256 * 256
    idx = workspace[:n * 4].view(torch.int32).reshape(T, top_k).clone()
    wts = workspace[off:off + n * 4].view(torch.float32).reshape(T, top_k).clone()
    return y, idx, wts


def prepare_expert_weights(w1: List[torch.Tensor], w3: List[torch.Tensor], w2: List[torch.Tensor], precision: str):
    """Per-expert (inter, d) gate / up and (d, inter) down matrices -> the two stacked operands of ``adx_op_moe_block``."""
    w13 = torch.stack([torch.cat([f32c(a), f32c(b)], dim=0) for a, b in zip(w1, w3)])          # (E, 2 inter, d)
    wd = torch.stack([f32c(a) for a in w2])                                                     # (E, d, inter)
    return (ops.prepare_weight(w13.reshape(-1, w13.shape[-1]), precision).reshape(-1),
            ops.prepare_weight(wd.reshape(-1, wd.shape[-1]), precision).reshape(-1))


class _RMS(nn.Module):
    def __init__(self, d: int):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d))


class _Expert(nn.Module):
    def __init__(self, d: int, inter: int):
        super().__init__()
        self.w1, self.w2, self.w3 = _Lin(d, inter, bias=False), _Lin(inter, d, bias=False), _Lin(d, inter, bias=False)


class _SparseMoe(nn.Module):
    def __init__(self, d: int, inter: int, n_experts: int):
        super().__init__()
        self.gate = _Lin(d, n_experts, bias=False)
        self.experts = nn.ModuleList([_Expert(d, inter) for _ in range(n_experts)])


class _Attention(nn.Module):
    def __init__(self, d: int, n_head: int, n_kv: int):
        super().__init__()
        dh = d // n_head
        self.q_proj, self.k_proj = _Lin(d, n_head * dh, bias=False), _Lin(d, n_kv * dh, bias=False)
        self.v_proj, self.o_proj = _Lin(d, n_kv * dh, bias=False), _Lin(n_head * dh, d, bias=False)


class _DecoderLayer(nn.Module):
    def __init__(self, d: int, inter: int, n_head: int, n_kv: int, n_experts: int):
        super().__init__()
        self.self_attn = _Attention(d, n_head, n_kv)
        self.block_sparse_moe = _SparseMoe(d, inter, n_experts)
        self.input_layernorm, self.post_attention_layernorm = _RMS(d), _RMS(d)


class KVCache:
    """Per layer the rotated keys and the values of every position processed so far, (B, S, n_head * dh) (key / value heads already
    repeated for their query heads)."""

    def __init__(self, n_layer: int):
        self.k: List[Optional[torch.Tensor]] = [None] * n_layer
        self.v: List[Optional[torch.Tensor]] = [None] * n_layer
        self.length = 0


class MixtralModel(nn.Module):
    """Parameter holder + device forward of the sparse mixture-of-experts decoder on input embeddings (causal mask, positions
    0, 1, 2, ...).  ``precision``: operand type of the products ("fp32" | "bf16" | "bf16x3"); the router computes in fp32."""

    def __init__(self, hidden_size: int, intermediate_size: int, num_hidden_layers: int, num_attention_heads: int, num_key_value_heads: int = 8,
                 num_local_experts: int = 8, num_experts_per_tok: int = 2, rms_norm_eps: float = 1e-5, rope_theta: float = 1e6,
                 vocab_size: int = 32000, sliding_window: Optional[int] = 4096, precision: str = "fp32"):
        super().__init__()
        if hidden_size % num_attention_heads or num_attention_heads % num_key_value_heads or (hidden_size // num_attention_heads) % 2:
            raise ValueError("MixtralModel: hidden size divisible by the head count, head count by the key / value head count, even head width")
        if hidden_size % 32 or intermediate_size % 32:
            raise ValueError("MixtralModel: hidden and expert widths must be multiples of 32")
        self.embed_tokens = nn.Module()                                  # in the checkpoint; the model is driven by embeddings
        self.embed_tokens.weight = nn.Parameter(torch.zeros(vocab_size, hidden_size))
        self.layers = nn.ModuleList([_DecoderLayer(hidden_size, intermediate_size, num_attention_heads, num_key_value_heads, num_local_experts)
                                     for _ in range(num_hidden_layers)])
        self.norm = _RMS(hidden_size)
        self.d, self.inter, self.n_head, self.n_kv = hidden_size, intermediate_size, num_attention_heads, num_key_value_heads
        self.n_experts, self.top_k, self.eps, self.theta = num_local_experts, num_experts_per_tok, float(rms_norm_eps), float(rope_theta)
        self.sliding_window, self.precision = sliding_window, precision
        self._prepared: Dict[str, torch.Tensor] = {}
        self._stamp = None
        self._tables = None
        self._ws: Optional[torch.Tensor] = None
        self._register_load_state_dict_pre_hook(self._rename_keys)

    def _rename_keys(self, state_dict, prefix, *args):
        """load_state_dict pre-hook: accept the fused expert tensors and the ``mlp`` naming of newer ``transformers`` releases."""
        for k in [k for k in state_dict if k.startswith(prefix) and ".mlp." in k]:
            t = state_dict.pop(k)
            base = k.replace(".mlp.", ".block_sparse_moe.")
            if k.endswith(".mlp.experts.gate_up_proj"):
                stem, inter = base[:-len("gate_up_proj")], t.shape[1] // 2
                for e in range(t.shape[0]):
                    state_dict[f"{stem}{e}.w1.weight"], state_dict[f"{stem}{e}.w3.weight"] = t[e, :inter], t[e, inter:]
            elif k.endswith(".mlp.experts.down_proj"):
                stem = base[:-len("down_proj")]
                for e in range(t.shape[0]):
                    state_dict[f"{stem}{e}.w2.weight"] = t[e]
            else:
                state_dict[base] = t

    def _weights(self) -> Dict[str, torch.Tensor]:
        params = [(n, p) for n, p in self.named_parameters() if p.dim() == 2 and not n.startswith("embed_tokens")]
        stamp = (self.precision,) + tuple((n, p.data_ptr(), p._version) for n, p in params)
        if stamp != self._stamp:
            W: Dict[str, torch.Tensor] = {}
            for i, layer in enumerate(self.layers):
                a, m = layer.self_attn, layer.block_sparse_moe
                for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
                    W[f"{i}.{n}"] = ops.prepare_weight(f32c(getattr(a, n).weight), self.precision)
                W[f"{i}.gate"] = f32c(m.gate.weight)
                W[f"{i}.w13"], W[f"{i}.w2"] = prepare_expert_weights([e.w1.weight for e in m.experts], [e.w3.weight for e in m.experts],
                                                                      [e.w2.weight for e in m.experts], self.precision)
            self._prepared, self._stamp = W, stamp
        return self._prepared

    def _rope(self, n_pos: int, device):
        if self._tables is None or self._tables[0].shape[0] < n_pos or self._tables[0].device != device:
            self._tables = rotary_tables(max(n_pos, 1024), self.d // self.n_head, self.theta, device)
        return self._tables

    def new_cache(self) -> KVCache:
        return KVCache(len(self.layers))

    @torch.no_grad()
    def extend(self, new_embeds: torch.Tensor, cache: KVCache, return_routing: bool = False):
        """Process ``new_embeds`` (B, n, d) as positions cache.length .. cache.length + n - 1 on top of the cached prefix; returns their
        last_hidden_state (B, n, d) and grows the cache."""
        if not new_embeds.is_cuda:
            raise RuntimeError("MixtralModel runs on the GPU only")
        x = f32c(new_embeds)
        B, n, d = x.shape
        past, H, Hkv = cache.length, self.n_head, self.n_kv
        if self.sliding_window is not None and past + n > self.sliding_window:
            raise NotImplementedError("sequences longer than the sliding window are not built (the reference's are below 900 tokens)")
        W = self._weights()
        cos, sin = self._rope(past + n, x.device)
        lin = lambda t, name, N: ops.linear(t, W[name], None, precision=self.precision, prepared_n=N)  # noqa: E731
        dh, rep = d // H, H // Hkv
        routing = []
        for i, layer in enumerate(self.layers):
            n1 = rmsnorm(x, layer.input_layernorm.weight, self.eps)
            q, k, v = lin(n1, f"{i}.q_proj", H * dh), lin(n1, f"{i}.k_proj", Hkv * dh), lin(n1, f"{i}.v_proj", Hkv * dh)
            rotary_(q, cos, sin, H, past)
            rotary_(k, cos, sin, Hkv, past)
            if rep > 1:                                          # query head h reads key / value head h // rep
                k = k.reshape(B, n, Hkv, 1, dh).expand(B, n, Hkv, rep, dh).reshape(B, n, H * dh)
                v = v.reshape(B, n, Hkv, 1, dh).expand(B, n, Hkv, rep, dh).reshape(B, n, H * dh)
            K = k if cache.k[i] is None else torch.cat([cache.k[i], k], dim=1)
            V = v if cache.v[i] is None else torch.cat([cache.v[i], v], dim=1)
            cache.k[i], cache.v[i] = K, V
            att = ops.attention(q, K, V, H, mask_off=past, precision=self.precision)
            x = _add(x, lin(att, f"{i}.o_proj", d))
            n2 = rmsnorm(x, layer.post_attention_layernorm.weight, self.eps)
            r = moe_block(n2.reshape(B * n, d), x.reshape(B * n, d), W[f"{i}.gate"], W[f"{i}.w13"], W[f"{i}.w2"], self.inter, self.n_experts,
                          self.top_k, self.precision, self._workspace(B * n, x.device), return_routing)
            if return_routing:
                r, idx, wts = r
                routing.append((idx, wts))
            x = r.reshape(B, n, d)
        cache.length = past + n
        out = rmsnorm(x, self.norm.weight, self.eps)
        return (out, routing) if return_routing else out

    def _workspace(self, T: int, device) -> torch.Tensor:
        need = int(_lib.load().adx_moe_workspace_bytes(T, self.d, self.inter, self.n_experts, self.top_k))
        if self._ws is None or self._ws.numel() < need or self._ws.device != device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=device)
        return self._ws

    def forward(self, inputs_embeds: torch.Tensor, **unused):
        return {"last_hidden_state": self.extend(inputs_embeds, self.new_cache())}


class KeyPointMLPDeocder(TrajectoryDecoder):
    """The MLP key-point head for ``k`` = 1 (the class name is the reference's): ``model`` = DecoderResCat(n_inner, n_embd, 2), the
    arithmetic of the trajectory head with two outputs."""

    def __init__(self, config, precision: Optional[str] = None):
        if int(getattr(config, "k", 1)) != 1:
            raise NotImplementedError("KeyPointMLPDeocder: only k = 1 is built")
        super().__init__(SimpleNamespace(n_embd=config.n_embd, n_inner=config.n_inner, trajectory_decoder_type="mlp", predict_yaw=False,
                                         loss_fn=getattr(config, "loss_fn", "mse")), precision or getattr(config, "precision", "fp32"))

    def generate_keypoints(self, hidden_state: torch.Tensor, info_dict: Optional[dict] = None):
        assert hidden_state.shape[1] == 1
        return self.decode(hidden_state).reshape(hidden_state.shape[0], 1, -1), None


def mixtral_sizes(model_name: str) -> Dict[str, int]:
    """The size table of build_models (str_base.py:935-1034): n_layer, n_embd, n_inner, n_head for a 'mixtral-*' model name."""
    table = [("mixtral-small-wide", (4, 512, 2048, 8)), ("mixtral-small", (4, 256, 1024, 8)), ("mixtral-narrow-medium", (16, 512, 2048, 16)),
             ("mixtral-medium", (16, 1024, 4096, 16)), ("mixtral-800m-deep", (32, 512, 2048, 32)), ("mixtral-3b-deep", (32, 1024, 4096, 32)),
             ("mixtral-3b-wide", (16, 1536, 6144, 16)), ("mixtral-6b-deep", (64, 1024, 4096, 64))]
    for key, (L, d, inner, h) in table:
        if key in model_name:
            return dict(n_layer=L, n_embd=d, n_inner=inner, n_head=h)
    raise AssertionError(f"Unsupported model name: {model_name}!")


def str_config(model_name: str = "mixtral-small", **kw) -> SimpleNamespace:
    """A configuration with the reference's defaults for the fields this model reads (utils/args.py ModelArguments; MixtralConfig:
    8 key / value heads, 8 experts, 2 per token, RMSNorm eps 1e-5, rope_theta 1e6, sliding window 4096)."""
    base = dict(model_name=model_name, task="nuplan", encoder_type="raster", raster_encoder_type="vit", raster_channels=34, vit_intermediate_size=3072,
                use_speed=False, use_key_points="specified_backward", separate_kp_encoder=True, kp_decoder_type="mlp", kp_tokenizer=None,
                use_proposal=0, autoregressive_proposals=False, autoregressive=False, k=1, with_traffic_light=True, camera_image_encoder=None,
                x_random_walk=0.0, y_random_walk=0.0, arf_x_random_walk=0.0, arf_y_random_walk=0.0, predict_yaw=True, loss_fn="mse",
                trajectory_decoder_type="mlp", traj_dropout=0.0, kp_dropout=0.0, mean_circular_loss=True, trajectory_prediction_mode="all_frames",
                pred_key_points_only=False, trajectory_loss_rescale=1.0, reverse_traj_index_order=False, num_key_value_heads=8,
                num_local_experts=8, num_experts_per_tok=2, rms_norm_eps=1e-5, rope_theta=1e6, sliding_window=4096)
    base.update(mixtral_sizes(model_name))
    base.update(kw)
    return SimpleNamespace(**base)


class STR_Mixtral(nn.Module):
    """encoder -> MixtralModel (``transformer``) -> key-point head (autoregressive) -> trajectory head, as ``STR.generate``."""

    def __init__(self, config, precision: Optional[str] = None):
        super().__init__()
        self.config = config
        cfg = config
        if getattr(cfg, "task", "nuplan") != "nuplan":
            raise NotImplementedError("STR_Mixtral: only the nuplan task is built")
        if getattr(cfg, "use_proposal", 0):
            raise NotImplementedError("STR_Mixtral: proposals are not built")
        if getattr(cfg, "autoregressive", False):
            raise NotImplementedError("STR_Mixtral: the point-by-point autoregressive trajectory is not built")
        if getattr(cfg, "kp_tokenizer", None) is not None:
            raise NotImplementedError("STR_Mixtral: key-point tokenizers are not built")
        self.k = int(getattr(cfg, "k", 1))
        self.use_proposal, self.use_key_points = 0, cfg.use_key_points
        precision = precision or getattr(cfg, "precision", "fp32")
        self.encoder = NuplanRasterizeEncoder(cfg, precision)
        self.transformer = MixtralModel(cfg.n_embd, cfg.n_inner, cfg.n_layer, cfg.n_head, getattr(cfg, "num_key_value_heads", 8),
                                        getattr(cfg, "num_local_experts", 8), getattr(cfg, "num_experts_per_tok", 2),
                                        getattr(cfg, "rms_norm_eps", 1e-5), getattr(cfg, "rope_theta", 1e6),
                                        sliding_window=getattr(cfg, "sliding_window", 4096), precision=precision)
        if self.use_key_points != "no":
            if getattr(cfg, "kp_decoder_type", "mlp") != "mlp":
                raise NotImplementedError("STR_Mixtral: only the MLP key-point head is built (kp_decoder_type='mlp')")
            self.key_points_decoder = KeyPointMLPDeocder(cfg, precision)
        self.traj_decoder = TrajectoryDecoder(cfg, precision)

    def embedding_to_hidden(self, input_embeds, attention_mask=None, position_ids=None, return_dict=True):
        return self.transformer(input_embeds)

    def _embed_key_points(self, kp: torch.Tensor) -> torch.Tensor:
        lin = getattr(self.encoder.kps_m_embed, "0")
        return _tanh(ops.linear(f32c(kp), f32c(lin.weight), f32c(lin.bias)))

    @torch.no_grad()
    def generate(self, **kwargs) -> Dict[str, torch.Tensor]:
        input_embeds, info_dict, maps_info = self.encoder(is_training=False, **kwargs)
        B = input_embeds.shape[0]
        device = input_embeds.device
        ctx = int(info_dict["context_length"])
        cache = self.transformer.new_cache()
        pred = {"maps_info": maps_info}
        if self.use_key_points != "no":
            pred_length = info_dict["pred_length"]
            n_kp = len(self.encoder.selected_indices) if "specified" in self.use_key_points else len(range(19, pred_length, 20))
            assert n_kp > 0, "future points not enough to sample"
            input_embeds = input_embeds.clone()
            input_embeds[:, ctx:ctx + n_kp] = self._embed_key_points(torch.zeros(B, n_kp, 2, device=device))
            last = self.transformer.extend(input_embeds[:, :ctx], cache)[:, -1:, :]
            kps = []
            for i in range(n_kp):
                kp, _ = self.key_points_decoder.generate_keypoints(last.contiguous())
                emb = self._embed_key_points(kp).reshape(B, 1, -1)
                input_embeds[:, ctx + i] = emb[:, 0]
                kps.append(kp[:, 0, :2].unsqueeze(1))
                if i < n_kp - 1:
                    last = self.transformer.extend(emb, cache)              # one new token: position ctx + i
            pred["key_points_logits"] = torch.cat(kps, dim=1).reshape(B, n_kp, -1)
            hidden = self.transformer.extend(input_embeds[:, ctx + n_kp - 1:], cache)     # the last key point and the trajectory slots
        else:
            hidden = self.transformer.extend(input_embeds, cache)
        traj = self.traj_decoder.generate_trajs(hidden, info_dict)
        if getattr(self.config, "reverse_traj_index_order", False):
            traj = traj.flip(-2)
        pred["traj_logits"] = traj
        return pred


class GraphedGenerate:
    """``generate(**batch)`` of a phase-1 model replayed from a captured graph.  Phase 1 is ~1 200 operator launches composed on the
    host, and at evaluation batch sizes their launch cost is most of its time; every one of them is enqueued on the current stream
    without a host round trip (the expert block buckets its tokens on the device), so the whole call can be captured once per input
    shape and replayed.  Inputs are copied into buffers the graph owns; the returned tensors belong to the graph and are overwritten
    by the next call of the same shape (clone what must outlive it)."""

    def __init__(self, model):
        self.model = model
        self._graphs: Dict[tuple, tuple] = {}

    @torch.no_grad()
    def __call__(self, **batch):
        tensors = {k: v for k, v in batch.items() if torch.is_tensor(v)}
        other = {k: v for k, v in batch.items() if not torch.is_tensor(v)}
        key = tuple((k, tuple(v.shape), v.dtype) for k, v in sorted(tensors.items())) + tuple(sorted((k, repr(v)) for k, v in other.items()))
        if key not in self._graphs:
            static_in = {k: v.detach().clone() for k, v in tensors.items()}
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                  # warm-up: weights prepared, kernel attributes set, workspaces sized
                self.model.generate(**static_in, **other)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_out = self.model.generate(**static_in, **other)
            self._graphs[key] = (graph, static_in, static_out)
        graph, static_in, static_out = self._graphs[key]
        for k, v in tensors.items():
            static_in[k].copy_(v)
        graph.replay()
        return static_out
