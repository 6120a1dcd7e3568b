"""Public mixing helpers with the reference's names and call signatures.

Drop-in surface for ``latentblending/utils.py`` of the reference
(``latentblending/__init__.py:3`` re-exports exactly these seven names).
The two hot functions dispatch to the active mixing backend (HIP kernels on
gfx950, see ``backend.py``); the remaining helpers are small host utilities kept
for import compatibility (not on the hot path).

  interpolate_spherical  <- latentblending/utils.py:29-71   (kernel ``lb_slerp_*``)
  interpolate_linear     <- latentblending/utils.py:74-102  (kernel ``lb_lerp_*`` for tensors)
  add_frames_linear_interp <- latentblending/utils.py:105-178
  get_spacing            <- latentblending/utils.py:181-200
  get_time               <- latentblending/utils.py:203-221
  compare_dicts          <- latentblending/utils.py:224-242
  yml_load / yml_save    <- latentblending/utils.py:245-262
"""
from __future__ import annotations

import datetime as _dt
import time as _time
from typing import List, Optional, Union

import numpy as np
import torch
import yaml

from .backend import get_backend

Number = Union[int, float]


@torch.no_grad()
def interpolate_spherical(p0: torch.Tensor, p1: torch.Tensor, fract_mixing: float) -> torch.Tensor:
    """Whole-tensor slerp.  float64 reductions, fp16 in -> fp16 out, anything else -> fp32 out
    (reference ``utils.py:47-50,66-69``); ``fract_mixing`` 0 / 1 return p0 / p1 exactly."""
    return get_backend().slerp(p0, p1, float(fract_mixing))


def interpolate_linear(p0, p1, fract_mixing):
    """``(1-f)*p0 + f*p1``.  Tensors go to the backend kernel; numpy images (uint8 frames used by
    the movie in-betweening, outside the hot path) are blended on the host in float64 and clipped
    back to uint8 as the reference does (``utils.py:88-100``)."""
    if isinstance(p0, torch.Tensor) and isinstance(p1, torch.Tensor):
        return get_backend().lerp(p0, p1, float(fract_mixing))
    was_u8 = False
    a, b = p0, p1
    if isinstance(a, np.ndarray) and a.dtype == np.uint8:
        a, was_u8 = a.astype(np.float64), True
    if isinstance(b, np.ndarray) and b.dtype == np.uint8:
        b, was_u8 = b.astype(np.float64), True
    out = (1 - fract_mixing) * a + fract_mixing * b
    if was_u8:
        out = np.clip(out, 0, 255).astype(np.uint8)
    return out


def _device_frame_stack(list_imgs):
    """[n, H, W, 3] uint8 device tensor when every frame is a device-resident ``DeviceImage`` of one size whose byte
    count the kernel accepts (multiple of 16), else None (host path)."""
    frames = [getattr(im, "_lb_u8", None) for im in list_imgs]
    if not frames or any(f is None or not f.is_cuda or f.shape != frames[0].shape for f in frames):
        return None
    if frames[0].numel() % 16 != 0:
        return None
    return torch.stack([f.contiguous() for f in frames])


def add_frames_linear_interp(list_imgs: List[np.ndarray],
                             fps_target: Optional[Number] = None,
                             duration_target: Optional[Number] = None,
                             nmb_frames_target: Optional[int] = None) -> List[np.ndarray]:
    """Pad a frame list to an exact frame count with linearly blended in-betweens.

    Either ``fps_target`` and ``duration_target`` or ``nmb_frames_target``.  The per-gap insert
    counts are ``floor(mean)`` plus a random 0/1 so that the total is hit exactly (same scheme and
    the same use of ``np.random.rand`` as the reference, so a seeded numpy RNG reproduces it)."""
    if nmb_frames_target is not None and fps_target is not None:
        raise ValueError("You cannot specify both fps_target and nmb_frames_target")
    if nmb_frames_target is None:
        assert fps_target is not None and duration_target is not None, \
            "Either specify duration_target and fps_target OR nmb_frames_target"
        nmb_frames_target = fps_target * duration_target

    n_gaps = len(list_imgs) - 1
    n_missing = nmb_frames_target - n_gaps - 1
    if n_missing < 1:
        return list_imgs

    mean_insert = n_missing / n_gaps
    base = np.floor(mean_insert)
    threshold = 1 - (mean_insert - base)
    tries = 0
    while True:
        draw = np.random.rand(n_gaps)
        per_gap = np.where(draw > threshold, 1.0, 0.0) + base
        if per_gap.sum() == n_missing:
            break
        tries += 1
        if tries > 100000:
            print("add_frames_linear_interp: issue with inserting the right number of frames")
            break
    per_gap = per_gap.astype(np.int32)

    dev = _device_frame_stack(list_imgs)
    if dev is not None:                 # key frames resident in HBM (native pipe): blend them there, one copy back
        from .hip import ops
        left, weights = [], []
        for g in range(n_gaps):
            for w in np.linspace(0, 1, per_gap[g] + 2)[:-1]:        # w = 0 reproduces the key frame itself
                left.append(g)
                weights.append(float(w))
        blended = ops.frames_lerp_u8(dev, left, weights).cpu().numpy()
        return [blended[k] for k in range(blended.shape[0])] + [dev[-1].cpu().numpy()]

    frames = [np.asarray(im).astype(np.float32) for im in list_imgs]
    out: List[np.ndarray] = []
    for g in range(n_gaps):
        left, right = frames[g], frames[g + 1]
        out.append(left.astype(np.uint8))
        for w in np.linspace(0, 1, per_gap[g] + 2)[1:-1]:
            out.append(interpolate_linear(left, right, w).astype(np.uint8))
    out.append(frames[-1].astype(np.uint8))
    return out


def get_spacing(nmb_points: int, scaling: float):
    """Non-linear [0,1] spacing, denser around 0.5 for ``scaling >= 1.7``."""
    if scaling < 1.7:
        return np.linspace(0, 1, nmb_points)
    half = nmb_points // 2 + 1
    ramp = np.abs(np.linspace(1, 0, half) ** scaling / 2 - 0.5)
    if nmb_points % 2:
        left = ramp
        right = 1 - left[::-1][1:]
    else:
        left = ramp[:-1]
        right = 1 - left[::-1]
    return np.hstack([left, right])


def get_time(resolution: Optional[str] = None) -> str:
    """Timestamp string such as ``221117_1620``.  (The reference's "millisecond" branch calls
    ``datetime.utcnow`` on the module and raises, ``utils.py:218``; fixed here.)"""
    resolution = resolution or "second"
    now = _time.localtime()
    if resolution == "day":
        return _time.strftime('%y%m%d', now)
    if resolution == "minute":
        return _time.strftime('%y%m%d_%H%M', now)
    if resolution == "second":
        return _time.strftime('%y%m%d_%H%M%S', now)
    if resolution == "millisecond":
        ms = _dt.datetime.now(_dt.timezone.utc).microsecond // 1000
        return _time.strftime('%y%m%d_%H%M%S', now) + "_{:03d}".format(ms)
    raise ValueError("bad resolution provided: %s" % resolution)


def compare_dicts(a: dict, b: dict) -> dict:
    """Keys present in both dicts whose values differ -> ``{key: [a[key], b[key]]}``."""
    return {k: [a[k], b[k]] for k in a if k in b and a[k] != b[k]}


def yml_load(fp_yml, print_fields=False):
    with open(fp_yml) as fh:
        data = dict(yaml.load(fh, Loader=yaml.loader.SafeLoader))
    print("load: loaded {}".format(fp_yml))
    return data


def yml_save(fp_yml, dict_stuff):
    with open(fp_yml, 'w') as fh:
        yaml.dump(dict_stuff, fh, sort_keys=False, default_flow_style=False)
    print("yml_save: saved {}".format(fp_yml))
