"""Device-resident frames that still satisfy the reference's return type.

``run_transition`` must return ``list[PIL.Image]`` (blending_engine.py:365 of the reference) but
the engine itself only needs frames on the GPU (LPIPS).  ``DeviceImage`` is a real
``PIL.Image.Image`` whose pixels stay in HBM until somebody actually touches them (save, numpy,
show): the device->host copy — and the stream sync it implies — happens lazily in ``load()``.
"""
from __future__ import annotations

import os
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


_PINNED: dict = {}

_RGBX = os.environ.get("LB_FRAMES_RGBX", "1") != "0"      # (A/B switch: 0 = three bytes per pixel over the bus, PIL unpacks RGB -> RGBX itself)


def host_cores(host_u8) -> list:
    """PIL pixel cores of host uint8 frames [n, H, W, 3] or (round 6) [n, H, W, 4].  PIL stores an RGB image as four bytes per pixel
    (RGBX); from three-byte pixels `Image.fromarray` zero-fills a new image and then unpacks pixel by pixel.  With the pad byte already in
    place (added on the device: +1/3 on a 0.25 ms copy) the core is an UNINITIALISED image filled by PIL's raw "RGBX" decoder - a row memcpy:
    the same image, ~1/3 less host time per frame.  ONE host copy per frame either way (the source buffer is free again afterwards).
    (A thread pool was measured slower: allocation and unpack hold the GIL.)"""
    if host_u8.shape[-1] == 4:
        out = []
        for arr in host_u8:
            h, w, _ = arr.shape
            im = Image.new("RGB", (w, h), None)
            im.frombytes(memoryview(arr), "raw", "RGBX", 0, 1)
            out.append(im.im)
        return out
    return [Image.fromarray(arr, "RGB").im for arr in host_u8]


def materialise_frames(frames) -> int:
    """Bring every still device-resident ``DeviceImage`` of ``frames`` to the host in ONE device->host copy (one stream
    synchronisation for the whole transition instead of one per frame) and build their PIL pixel cores - what the
    reference's ``run_transition`` has done by the time it returns (blending_engine.py:347,575: real PIL images).
    Returns the number of frames copied."""
    todo = [f for f in frames if isinstance(f, DeviceImage) and not f._lb_loaded]
    dev = [f for f in todo if f._lb_u8.is_cuda]
    if dev:
        stacked = torch.stack([f._lb_u8 for f in dev])
        if _RGBX:
            stacked = torch.nn.functional.pad(stacked, (0, 1), value=255)      # [n, H, W, 4]: PIL's own RGB storage layout
        key = (tuple(stacked.shape), stacked.device.index)
        pinned = _PINNED.get(key)
        if pinned is None:                                # one page-locked staging buffer per batch shape: DMA at PCIe speed
            _PINNED.clear()
            pinned = _PINNED[key] = torch.empty(stacked.shape, dtype=torch.uint8, pin_memory=True)
        pinned.copy_(stacked, non_blocking=True)
        torch.cuda.current_stream(stacked.device).synchronize()
        host = pinned.numpy()
        cores = host_cores(host)
        for f, core in zip(dev, cores):
            f._im = core
            f._lb_loaded = True
    for f in todo:
        f._materialise()
    return len(todo)
