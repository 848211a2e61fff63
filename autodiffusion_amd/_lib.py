"""ctypes binding of libadx.so (the C ABI declared in include/adx.h).

There is no fallback: if the shared library is missing or a call fails, an
exception is raised.  Nothing in this package computes the hot path on the CPU.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ADX_LIB_PATH") or os.path.join(_HERE, "libadx.so")      # ADX_LIB_PATH: another build of the library (same-box A/B timings)

ACT_NONE, ACT_RELU, ACT_GELU_ERF, ACT_GELU_TANH, ACT_LEAKY, ACT_SILU, ACT_TANH = range(7)     # ACT_TANH: adx_op_act only
PREC_F32, PREC_BF16, PREC_BF16X3, PREC_F16X2, PREC_F16 = 0, 1, 2, 3, 4
PRECISIONS = {"fp32": PREC_F32, "bf16": PREC_BF16, "bf16x3": PREC_BF16X3, "f16x2": PREC_F16X2, "f16": PREC_F16}
VARIANT_A, VARIANT_DIT, VARIANT_MMDIT = 0, 1, 2
FLAG_NORMALIZE, FLAG_RESIDUAL, FLAG_INIT_ZERO, FLAG_NO_GRAPH, FLAG_KEEP_MAPS = 1, 2, 4, 8, 16

EXPORTS = [
    "adx_abi_version", "adx_last_error",
    "adx_op_timestep_embedding", "adx_op_linear", "adx_op_linear_staged", "adx_linear_stage_bytes", "adx_linear_planes_bytes", "adx_op_stage_planes", "adx_op_block_weight",
    "adx_op_linear_planes", "adx_op_mlp256", "adx_op_to_bf16", "adx_op_to_f16", "adx_op_layernorm", "adx_op_attention", "adx_op_attention_prec",
    "adx_op_normalize", "adx_op_denormalize", "adx_op_sched_step",
    "adx_engine_create", "adx_engine_destroy", "adx_engine_set_tensor", "adx_engine_finalize", "adx_engine_reserve",
    "adx_engine_workspace_bytes", "adx_denoise_a", "adx_denoise_dit", "adx_engine_set_schedule", "adx_sample_a",
    "adx_generate_a", "adx_sample_dit", "adx_sample_dit_grouped", "adx_debug_stage_a", "adx_op_select_modes", "adx_op_trajectory_metrics", "adx_engine_profile", "adx_engine_profile_read",
    "adx_engine_set_noise_seed", "adx_op_randn", "adx_engine_capture_count", "adx_op_q_sample", "adx_op_weighted_mse",
    "adx_op_linear_noalloc", "adx_noalloc_scratch_bytes", "adx_noalloc_init",
    "adx_op_add", "adx_op_mul", "adx_op_act", "adx_op_act_bwd", "adx_op_group_colsum", "adx_op_transpose", "adx_op_gelu", "adx_op_gelu_bwd", "adx_reduce_ws_bytes", "adx_op_linear_wgrad", "adx_op_colsum",
    "adx_op_layernorm_bwd", "adx_op_attention_bwd", "adx_op_attention_stats", "adx_op_weighted_mse_bwd",
    "adx_op_raster_patches", "adx_generate_transition", "adx_op_traj_loss",
    "adx_op_dropout", "adx_op_attention_stats_dropout", "adx_op_attention_bwd_dropout", "adx_op_attention_dropout_mask",
    "adx_op_rmsnorm", "adx_op_rotary", "adx_moe_workspace_bytes", "adx_op_moe_block",
]


class StepCoef(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("t", "a0", "a1", "clip", "eps_mode", "b0", "b1", "c0", "c1", "c2", "c3", "z_index")]


class ModelConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "variant", "precision", "elide", "d_model", "n_head", "n_enc_layers", "n_layers", "horizon", "in_dim", "out_dim",
        "n_prior_tokens", "prior_dim", "n_maps_token", "maps_dim", "map_cond", "n_inner", "connect_dim")]


class AdxError(RuntimeError):
    pass


_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load libadx.so and declare every prototype.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AdxError(f"{LIB_PATH} not found: build it with `python -m autodiffusion_amd.build` "
                       "(there is no CPU fallback for the sampling path)")
    lib = C.CDLL(LIB_PATH)
    p, i, i64, f, u = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint
    protos = {
        "adx_abi_version": ([], i),
        "adx_last_error": ([], C.c_char_p),
        "adx_op_timestep_embedding": ([p, i, i, p, p], i),
        "adx_op_linear": ([p, p, p, p, i, i, i, i, i, p], i),
        "adx_op_linear_staged": ([p, p, p, p, i, i, i, i, i, p, C.c_size_t, p], i),
        "adx_linear_stage_bytes": ([i, i, i, i], C.c_size_t),
        "adx_linear_planes_bytes": ([i, i, i], C.c_size_t),
        "adx_op_stage_planes": ([p, i, i, i, p, p], i),
        "adx_op_block_weight": ([p, i, i, i, p, p], i),
        "adx_op_linear_planes": ([p, p, p, p, i, i, i, i, i, p], i),
        "adx_op_to_bf16": ([p, p, i64, p], i),
        "adx_op_to_f16": ([p, p, i64, p], i),
        "adx_op_layernorm": ([p, p, p, p, p, i64, f, i, i64, i, p, p], i),
        "adx_op_mlp256": ([p, p, p, i, p, p, i, p, p, p, i, i, i, p], i),
        "adx_op_attention_prec": ([p, p, p, p, i, i, i, i, i, i, f, i, i, p], i),
        "adx_op_attention": ([p, p, p, p, i, i, i, i, i, i, f, i, p], i),
        "adx_op_normalize": ([p, p, i64, i, p], i),
        "adx_op_denormalize": ([p, p, p, i64, i, p], i),
        "adx_op_sched_step": ([p, p, p, p, p, i64, p, p], i),
        "adx_engine_create": ([C.POINTER(ModelConfig), C.POINTER(p)], i),
        "adx_engine_destroy": ([p], i),
        "adx_engine_set_tensor": ([p, C.c_char_p, p, C.POINTER(i64), i], i),
        "adx_engine_finalize": ([p], i),
        "adx_engine_reserve": ([p, i], i),
        "adx_engine_workspace_bytes": ([p], i64),
        "adx_denoise_a": ([p, i, p, p, p, p, p, p, p], i),
        "adx_denoise_dit": ([p, i, p, p, p, p, p], i),
        "adx_engine_set_schedule": ([p, C.POINTER(StepCoef), i], i),
        "adx_sample_a": ([p, i, p, p, p, p, i, p, p, p, u, p], i),
        "adx_generate_a": ([p, i, i, i, p, p, p, p, i, p, u, p], i),
        "adx_sample_dit": ([p, i, p, p, i, p, u, p], i),
        "adx_sample_dit_grouped": ([p, i, i, p, p, i, p, u, p], i),
        "adx_debug_stage_a": ([p, i, i, i, p], i),
        "adx_engine_profile": ([p, i], i),
        "adx_op_trajectory_metrics": ([p, p, i, i, i, p, p], i),
        "adx_op_select_modes": ([p, i, i, i, i, i, i, p, p, p], i),
        "adx_engine_profile_read": ([p, i, p, p], i),
        "adx_engine_set_noise_seed": ([p, C.c_uint64, i64], i),
        "adx_op_randn": ([p, i, i, i, C.c_uint64, i64, p, p], i),
        "adx_engine_capture_count": ([p], i64),
        "adx_op_linear_noalloc": ([p, p, p, p, i, i, i, i, i, p], i),
        "adx_noalloc_scratch_bytes": ([], C.c_size_t),
        "adx_noalloc_init": ([], i),
        "adx_op_q_sample": ([p, p, p, p, p, p, p, i, i, p], i),
        "adx_op_weighted_mse": ([p, p, p, p, p, p, i, i, p], i),
        "adx_op_add": ([p, p, p, i64, p], i),
        "adx_op_mul": ([p, p, p, i64, p], i),
        "adx_op_act": ([p, p, i64, i, p], i),
        "adx_op_act_bwd": ([p, p, p, i64, i, p], i),
        "adx_op_group_colsum": ([p, i, i64, i, p, p], i),
        "adx_op_transpose": ([p, p, i, i, p], i),
        "adx_op_gelu": ([p, p, i64, p], i),
        "adx_op_gelu_bwd": ([p, p, p, i64, p], i),
        "adx_reduce_ws_bytes": ([i64, i, i], C.c_size_t),
        "adx_op_linear_wgrad": ([p, p, i64, i, i, p, p, i, p, C.c_size_t, p], i),
        "adx_op_colsum": ([p, i64, i, p, i, p, C.c_size_t, p], i),
        "adx_op_layernorm_bwd": ([p, p, p, i64, i, f, p, p, p], i),
        "adx_op_attention_bwd": ([p, p, p, p, p, i, i, i, i, i, f, i, p, p, p, p, i, p], i),
        "adx_op_attention_stats": ([p, p, p, p, p, i, i, i, i, i, f, i, p], i),
        "adx_op_weighted_mse_bwd": ([p, p, p, p, p, p, i, i, p], i),
        "adx_op_traj_loss": ([p, p, i, i, i, i, i, p, p, p], i),
        "adx_op_raster_patches": ([p, i, i, i, i, i, i, p, p], i),
        "adx_generate_transition": ([p, i, i, i, p, p, p, p, i, p, p, u, p], i),
        "adx_op_dropout": ([p, p, i64, f, C.c_uint64, i, p], i),
        "adx_op_attention_stats_dropout": ([p, p, p, p, p, i, i, i, i, i, f, i, f, C.c_uint64, i, p, p], i),
        "adx_op_attention_bwd_dropout": ([p, p, p, p, p, i, i, i, i, i, f, i, p, p, p, p, i, f, C.c_uint64, i, p, p], i),
        "adx_op_attention_dropout_mask": ([p, i, i, i, i, f, C.c_uint64, i, p], i),
        "adx_op_rmsnorm": ([p, p, f, i64, i, p, p], i),
        "adx_op_rotary": ([p, p, p, i, i, i, i, i, i, p], i),
        "adx_moe_workspace_bytes": ([i64, i, i, i, i], C.c_size_t),
        "adx_op_moe_block": ([p, p, p, p, p, i64, i, i, i, i, i, p, C.c_size_t, p, p], i),
    }
    for name, (args, res) in protos.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = res
    if lib.adx_abi_version() != 1:
        raise AdxError("libadx.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().adx_last_error().decode("utf-8", "replace")
        if "divisible by frame stack" in msg:
            raise ValueError(msg)
        raise AdxError(f"{what} failed (code {rc}): {msg}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Device pointer of a contiguous CUDA tensor (None passes NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise AdxError("libadx operates on device memory only; got a CPU tensor")
    if not t.is_contiguous():
        raise AdxError("tensor must be contiguous")
    return t.data_ptr()


def current_stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def f32c(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.float32).contiguous()
