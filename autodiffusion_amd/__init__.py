"""MI355X-native trajectory-diffusion sampling path (HIP kernels behind the reference's model interface)."""
from . import synth  # noqa: F401

__all__ = ["synth", "refiner", "dit", "ops", "schedules", "engine", "distributed", "metrics"]
