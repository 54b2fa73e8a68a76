"""MI355X-native LBVH build + ray tracing hot path (gpurt C-ABI, HIP kernels for gfx950)."""
from . import capi  # noqa: F401

__all__ = ["capi", "runtime", "pipeline", "scenes"]
