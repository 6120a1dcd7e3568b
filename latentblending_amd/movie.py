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
import warnings
from typing import List

import numpy as np
from PIL import Image

from .utils import add_frames_linear_interp

try:  # pragma: no cover - not installed in the build image
    import lunar_tools as _lt  # type: ignore
    if getattr(_lt, "__lb_facade__", False):      # the repo-root facade forwards to THIS module
        raise ImportError("lunar_tools facade")
    from lunar_tools import MovieSaver, fill_up_frames_linear_interpolation, concatenate_movies  # type: ignore
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
            self.container = "avi-mjpeg"          # whatever the extension of fp_out says (the caller's path is kept: drop-in)
            if not str(fp_out).lower().endswith(".avi"):
                warnings.warn(f"lunar_tools / ffmpeg are not installed: '{fp_out}' will hold a Motion-JPEG AVI stream (RIFF 'AVI ' "
                              "header), not the container its extension names; ffmpeg / VLC / mpv detect it by content, "
                              "latentblending_amd.movie.read_movie_header / read_movie_jpegs read it back", UserWarning, stacklevel=2)
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


    def read_movie_jpegs(fp_movie: str) -> List[bytes]:
        """JPEG payloads of the ``00dc`` chunks of an MJPEG-AVI written by :class:`MovieSaver`."""
        with open(fp_movie, "rb") as fh:
            data = fh.read()
        if data[:4] != b"RIFF" or data[8:12] != b"AVI ":
            raise ValueError(f"{fp_movie}: not an AVI written by latentblending_amd.movie.MovieSaver")
        pos, out = 12, []
        while pos + 8 <= len(data):
            tag, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
            if tag == b"LIST":
                if data[pos + 8:pos + 12] == b"movi":
                    q, end = pos + 12, pos + 8 + size
                    while q + 8 <= end:
                        t2, s2 = data[q:q + 4], struct.unpack("<I", data[q + 4:q + 8])[0]
                        if t2 == b"00dc":
                            out.append(data[q + 8:q + 8 + s2])
                        q += 8 + s2 + (s2 & 1)
                pos += 8 + size + (size & 1)
            else:
                pos += 8 + size + (size & 1)
        return out

    def read_movie_header(fp_movie: str):
        """(fps, height, width, number of frames) of an AVI written by :class:`MovieSaver`."""
        with open(fp_movie, "rb") as fh:
            data = fh.read(256)
        i = data.index(b"avih")
        v = struct.unpack("<14I", data[i + 8:i + 8 + 56])
        return int(round(1e6 / v[0])), v[9], v[8], v[4]

    def concatenate_movies(fp_final: str, list_fp_movies: List[str]) -> None:
        """``lunar_tools.concatenate_movies(fp_final, parts)`` (reference: example_multi_trans.py:62): the
        parts' frames back to back in one file.  JPEG payloads are copied, not re-encoded."""
        assert len(list_fp_movies) > 0, "concatenate_movies: empty list"
        fps, h, w, _ = read_movie_header(list_fp_movies[0])
        saver = MovieSaver(fp_final, fps=fps, shape_hw=[h, w])
        for fp in list_fp_movies:
            f2, h2, w2, _ = read_movie_header(fp)
            assert (f2, h2, w2) == (fps, h, w), f"{fp}: fps/size differ from the first part"
            saver._jpegs.extend(read_movie_jpegs(fp))
        saver.finalize()
