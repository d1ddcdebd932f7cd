"""MI355X-native batched iLQR hot path (gfx950 HIP kernels behind the C ABI of include/ilqr_hip.h)."""
from . import capi  # noqa: F401

__all__ = ["capi", "workloads"]
