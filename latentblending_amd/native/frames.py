"""Device-resident frames that still satisfy the reference's return type.

``run_transition`` must return ``list[PIL.Image]`` (blending_engine.py:365 of the reference) but
the engine itself only needs frames on the GPU (LPIPS).  ``DeviceImage`` is a real
``PIL.Image.Image`` whose pixels stay in HBM until somebody actually touches them (save, numpy,
show): the device->host copy — and the stream sync it implies — happens lazily in ``load()``.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch
from PIL import Image


class DeviceImage(Image.Image):
    def __init__(self, frame_u8: torch.Tensor):
        """frame_u8: [H, W, 3] uint8 tensor (device or host)."""
        super().__init__()
        h, w, _ = frame_u8.shape
        self._lb_u8 = frame_u8
        self._lb_feats: Optional[List[torch.Tensor]] = None
        self._mode = "RGB"
        self._size = (w, h)
        self._im = None
        self._lb_loaded = False

    def _materialise(self):
        if not self._lb_loaded:
            arr = np.ascontiguousarray(self._lb_u8.detach().cpu().numpy())
            self._im = Image.fromarray(arr, "RGB").im
            self._lb_loaded = True

    @property
    def im(self):
        self._materialise()
        return self._im

    @im.setter
    def im(self, value):
        self._im = value
        self._lb_loaded = value is not None

    def load(self):
        self._materialise()
        return super().load()
