"""Read the reference's diffusion yaml files into the namespaces the drop-in classes take.

The reference loads ``config/*.yaml`` with OmegaConf (transformer4planning/utils/common_utils.py:10-12; consumed at
models/backbone/ExplicitDiT.py:34 and StrDiffDiT.py) and reads the result by attribute (``cfg.scheduler.num_train_timesteps``,
``cfg.data_mean`` ...).  ``load_config`` gives the same attribute view with PyYAML (YAML 1.1 scalars such as ``1e-4`` that
PyYAML leaves as strings are converted to numbers, as OmegaConf does), and the ``*_config_from`` helpers fill in the
constants the wrapper constructors hard-code on top of the file (DiffusionAugmentation.py:33-37) plus this package's own
switches (``precision``, ``elide``, ``kind``).

Which class a file configures:

=======================  ===================================================  =====================================
file                     reference consumer                                   here
=======================  ===================================================  =====================================
explicitDiT.yaml         ExplicitDiT (flattened chunks, diffusers DDIM)       explicit.ExplicitDiT
explicitDiT_whole.yaml   ExplicitDiT, frame_stack 80                          explicit.ExplicitDiT
strDiffDiT.yaml          StrDiffDiT (latent-carrying, diffusers DDPM)         transition.StrDiffDiT
ldr_diffusion.yaml       DiffusionAugmentation / DiffusionDecoder extras      refiner.DiffusionAugmentation, dit.DiffusionDecoder
=======================  ===================================================  =====================================
"""
from __future__ import annotations

import re
from types import SimpleNamespace
from typing import Any, Optional

import yaml

_NUMBER = re.compile(r"^[+-]?(\d+\.?\d*|\.\d+)([eE][+-]?\d+)?$")


class Config(SimpleNamespace):
    """Attribute and item access over a yaml mapping (the subset of OmegaConf's DictConfig the reference uses)."""

    def __getitem__(self, key):
        return getattr(self, key)

    def __contains__(self, key):
        return hasattr(self, key)

    def get(self, key, default=None):
        return getattr(self, key, default)

    def keys(self):
        return self.__dict__.keys()

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, Config) else v) for k, v in self.__dict__.items()}


def _convert(node: Any) -> Any:
    if isinstance(node, dict):
        return Config(**{str(k): _convert(v) for k, v in node.items()})
    if isinstance(node, list):
        return [_convert(v) for v in node]
    if isinstance(node, str) and _NUMBER.match(node.strip()):
        return float(node)
    return node


def load_config(config_path: str) -> Config:
    """The yaml file as a nested attribute namespace (transformer4planning/utils/common_utils.py:10-12)."""
    with open(config_path, "r") as f:
        raw = yaml.safe_load(f)
    if not isinstance(raw, dict):
        raise ValueError(f"{config_path}: the top level of a diffusion config must be a mapping")
    return _convert(raw)


def _with(cfg: Config, **defaults) -> Config:
    for k, v in defaults.items():
        if not hasattr(cfg, k):
            setattr(cfg, k, v)
    return cfg


def _check_scheduler(cfg: Config, path: str) -> None:
    sc = cfg.get("scheduler")
    if sc is None:
        raise ValueError(f"{path}: no 'scheduler' block")
    for k in ("num_train_timesteps", "num_inference_steps", "beta_start", "beta_end", "beta_schedule", "clip_sample", "prediction_type"):
        if not hasattr(sc, k):
            raise ValueError(f"{path}: scheduler.{k} is missing")
    sc.num_train_timesteps, sc.num_inference_steps = int(sc.num_train_timesteps), int(sc.num_inference_steps)


def explicit_config_from(path: str, kind: str = "ddim", precision: str = "fp32", elide: int = 3) -> Config:
    """config/explicitDiT.yaml, config/explicitDiT_whole.yaml -> the namespace ``explicit.ExplicitDiT`` takes."""
    cfg = load_config(path)
    _check_scheduler(cfg, path)
    return _with(cfg, trajectory_dim=4, maps_dim=256, n_maps_token=788, map_cond=True, kind=kind, precision=precision, elide=elide)


def transition_config_from(path: str, precision: str = "fp32") -> Config:
    """config/strDiffDiT.yaml -> the namespace ``transition.StrDiffDiT`` takes."""
    cfg = load_config(path)
    _check_scheduler(cfg, path)
    cfg.x_shape, cfg.z_shape = tuple(int(v) for v in cfg.x_shape), tuple(int(v) for v in cfg.z_shape)
    return _with(cfg, n_maps_token=788, cond_dim=256, prior_dim=4, map_cond=True, learnable_init_z=True, context_frames=0,
                 precision=precision)


def refiner_config_from(path: str, diffusion_timesteps: int = 100, objective: str = "pred_noise", beta_schedule: str = "cosine",
                        ddim: float = 0.0, normalize: bool = True, residual: bool = False, sampling_timesteps: Optional[int] = None,
                        precision: str = "fp32", elide: int = 3) -> Config:
    """A yaml without a scheduler block (config/ldr_diffusion.yaml: frame_stack, map_cond, data_mean/std,
    trajectory_loss_rescale ...) plus the diffusion flags the CLI copies onto the model config (utils/args.py:154-183;
    str_base.py:1045-1053) -> the namespace ``refiner.DiffusionAugmentation`` takes."""
    cfg = load_config(path)
    cfg.frame_stack = int(cfg.frame_stack)
    return _with(cfg, trajectory_dim=4, maps_dim=256, n_maps_token=788, map_cond=True, beta_schedule=beta_schedule,
                 diffusion_timesteps=diffusion_timesteps, objective=objective, ddim=ddim, normalize=normalize, residual=residual,
                 trajectory_loss_rescale=1.0, sampling_timesteps=sampling_timesteps, precision=precision, elide=elide)


def dit_config_from(path: str, n_embd: int = 256, n_layer: int = 4, n_head: int = 8, backbone: str = "DiT", n_inner: Optional[int] = None,
                    k: int = 1, mc_num: int = 1, diffusion_timesteps: int = 10, beta_schedule: str = "cosine",
                    precision: str = "fp32") -> Config:
    """config/ldr_diffusion.yaml (``trajectory_loss_rescale``, ``visualize_diffusion_intermedians``) plus the backbone
    sizes that come from the STR model config (str_base.py:929-1036) -> the namespace ``dit.DiffusionDecoder`` takes."""
    cfg = load_config(path)
    return _with(cfg, n_embd=n_embd, n_inner=n_inner or 4 * n_embd, n_head=n_head, n_layer=n_layer, diffusion_backbone=backbone,
                 predict_yaw=True, k=k, predict_range=80, beta_schedule=beta_schedule, diffusion_timesteps=diffusion_timesteps,
                 mc_num=mc_num, trajectory_loss_rescale=1.0, visualize_diffusion_intermedians=False, precision=precision)
