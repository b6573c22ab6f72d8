"""MI355X-native batched marinenav_env + IQN training loop.

Product path = hand-written gfx950 HIP kernels behind the C-ABI in include/marinenav_hip.h.
There is no CPU fallback; the CPU oracle under /oracle is test infrastructure only.
"""
from . import _capi  # noqa: F401

__all__ = ["_capi"]
