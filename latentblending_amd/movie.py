"""Frame in-betweening + movie output used by ``BlendingEngine.write_movie_transition``.

The reference takes ``MovieSaver`` and ``fill_up_frames_linear_interpolation`` from the
``lunar_tools`` package (``latentblending/blending_engine.py:13, 698, 703-706`` in
/root/reference), which drives an ffmpeg binary.  When ``lunar_tools`` is importable it is used
unchanged; otherwise a dependency-free fallback writes a Motion-JPEG AVI (every frame a JPEG
encoded by Pillow inside a RIFF container), playable by ffmpeg / VLC / mpv.  This is the step
AFTER the hot path and runs on the host.
"""
from __future__ import annotations

import io
import struct
from typing import List

import numpy as np
from PIL import Image

from .utils import add_frames_linear_interp

try:  # pragma: no cover - not installed in the build image
    from lunar_tools import MovieSaver, fill_up_frames_linear_interpolation  # type: ignore
    HAVE_LUNAR_TOOLS = True
except Exception:
    HAVE_LUNAR_TOOLS = False

    def fill_up_frames_linear_interpolation(list_imgs: List, a: float, b: float) -> List[np.ndarray]:
        """``a`` and ``b`` are (duration, fps) in either order — the reference passes them swapped
        relative to the callee's signature and only the product is used."""
        return add_frames_linear_interp(list(list_imgs), nmb_frames_target=int(round(a * b)))

    class MovieSaver:
        """Minimal MJPEG-in-AVI writer: ``write_frame(uint8 HxWx3)`` ... ``finalize()``."""

        def __init__(self, fp_out: str, fps: int = 30, shape_hw=None, quality: int = 92, **_):
            self.fp_out = fp_out
            self.fps = int(fps)
            self.shape_hw = list(shape_hw) if shape_hw is not None else None
            self.quality = quality
            self._jpegs: List[bytes] = []

        def write_frame(self, frame) -> None:
            img = frame if isinstance(frame, Image.Image) else Image.fromarray(np.asarray(frame, dtype=np.uint8))
            if self.shape_hw is None:
                self.shape_hw = [img.height, img.width]
            assert [img.height, img.width] == self.shape_hw, "frame size differs from shape_hw"
            buf = io.BytesIO()
            img.convert("RGB").save(buf, format="JPEG", quality=self.quality)
            self._jpegs.append(buf.getvalue())

        def finalize(self) -> None:
            h, w = self.shape_hw
            n = len(self._jpegs)

            def chunk(tag: bytes, payload: bytes) -> bytes:
                pad = b"\x00" if len(payload) % 2 else b""
                return tag + struct.pack("<I", len(payload)) + payload + pad

            def riff_list(kind: bytes, payload: bytes) -> bytes:
                return b"LIST" + struct.pack("<I", len(payload) + 4) + kind + payload

            biggest = max((len(j) for j in self._jpegs), default=0)
            avih = struct.pack("<14I", int(1e6 / self.fps), biggest * self.fps, 0, 0x10, n, 0, 1,
                               biggest, w, h, 0, 0, 0, 0)
            strh = struct.pack("<4s4sIHHIIIIIIIIhhhh", b"vids", b"MJPG", 0, 0, 0, 0, 1, self.fps,
                               0, n, biggest, 0xFFFFFFFF, 0, 0, 0, w, h)
            strf = struct.pack("<IiiHH4sIiiII", 40, w, h, 1, 24, b"MJPG", w * h * 3, 0, 0, 0, 0)
            hdrl = riff_list(b"hdrl", chunk(b"avih", avih) +
                             riff_list(b"strl", chunk(b"strh", strh) + chunk(b"strf", strf)))
            movi_body, index, offset = b"", b"", 4
            for j in self._jpegs:
                c = chunk(b"00dc", j)
                index += struct.pack("<4sIII", b"00dc", 0x10, offset, len(j))
                movi_body += c
                offset += len(c)
            body = b"AVI " + hdrl + riff_list(b"movi", movi_body) + chunk(b"idx1", index)
            with open(self.fp_out, "wb") as fh:
                fh.write(b"RIFF" + struct.pack("<I", len(body)) + body)
