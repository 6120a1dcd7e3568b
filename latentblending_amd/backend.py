"""Mixing-primitive backends.

The product backend is :class:`HipBackend`: every mixing primitive is a launch
of a hand-written gfx950 kernel through the C-ABI library (``liblbhip.so``).
There is deliberately NO CPU implementation in this package: asking the HIP
backend to mix host tensors raises.  Host-logic tests that have to run without
a GPU inject a checker backend from ``oracle/`` (test infrastructure) through
``set_backend`` / the ``backend=`` constructor arguments.

Reference behaviour being replaced (file:line relative to /root/reference):
  * ``latentblending/utils.py:29-71``   interpolate_spherical  -> ``slerp``
  * ``latentblending/utils.py:74-102``  interpolate_linear     -> ``lerp``
"""
from __future__ import annotations

from typing import Optional, Protocol, Sequence

import torch


class MixBackend(Protocol):
    name: str

    def slerp(self, p0: torch.Tensor, p1: torch.Tensor, fract: float) -> torch.Tensor: ...

    def lerp(self, p0: torch.Tensor, p1: torch.Tensor, fract: float) -> torch.Tensor: ...

    def slerp_pairs(self, p0: Sequence[torch.Tensor], p1: Sequence[torch.Tensor],
                    fracts: Sequence[float]) -> list: ...


class HipBackend:
    """gfx950 kernels behind the C-ABI (``include/lb_hip.h``)."""

    name = "hip"

    def __init__(self):
        from .hip import ops  # raises loudly when liblbhip.so is missing
        self._ops = ops

    @staticmethod
    def _require_device(*tensors):
        for t in tensors:
            if not (isinstance(t, torch.Tensor) and t.is_cuda):
                raise RuntimeError(
                    "latentblending_amd: the HIP backend only mixes device tensors; got a "
                    f"{type(t).__name__} on {getattr(t, 'device', '?')}. There is no CPU "
                    "fallback in the product path (inject a checker backend in tests).")

    def slerp(self, p0, p1, fract):
        self._require_device(p0, p1)
        return self._ops.slerp(p0, p1, float(fract))

    def lerp(self, p0, p1, fract):
        self._require_device(p0, p1)
        return self._ops.lerp(p0, p1, float(fract))

    def slerp_pairs(self, p0, p1, fracts):
        self._require_device(*p0, *p1)
        return self._ops.slerp_pairs(list(p0), list(p1), [float(f) for f in fracts])


_ACTIVE: Optional[MixBackend] = None


def set_backend(backend: Optional[MixBackend]) -> None:
    """Install the process-wide mixing backend (``None`` -> lazily create HipBackend)."""
    global _ACTIVE
    _ACTIVE = backend


def get_backend() -> MixBackend:
    global _ACTIVE
    if _ACTIVE is None:
        _ACTIVE = HipBackend()
    return _ACTIVE
