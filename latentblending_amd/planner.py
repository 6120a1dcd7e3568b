"""Host-side schedules of the blending engine: branch budget planner, crossfeed coefficient
vectors and guidance mid-dampening.  Pure integer / float64 host arithmetic — nothing here
touches a device.

Each function states the reference lines whose observable behaviour it reproduces (paths are
relative to /root/reference); golden vectors produced by the reference's own functions are in
``tests/golden/planner.json``.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np


def damped_guidance(guidance_base: float, damper: float, fract_mixing: float) -> float:
    """Guidance scale lowered linearly towards the middle of the transition.

    ``blending_engine.py:155-164``: at fract 0/1 the base value, at 0.5 the value
    ``base - (base*(1-damper) - 1)``.
    """
    closeness_to_mid = 1 - np.abs(fract_mixing - 0.5) / 0.5
    largest_cut = guidance_base * (1 - damper) - 1
    return guidance_base - largest_cut * closeness_to_mid


def anchor_crossfeed_coeffs(num_steps: int, power: float, span: float, decay: float) -> List[float]:
    """Per-step slerp weights that pull the second anchor towards the first
    (``blending_engine.py:404-408``): a linear ramp ``power -> power*decay`` over the first
    ``round(num_steps*span)`` steps, zero afterwards."""
    stop = int(round(num_steps * span))
    coeffs = list(np.linspace(power, power * decay, stop))
    coeffs.extend((num_steps - stop) * [0])
    return coeffs


def parental_crossfeed_coeffs(num_steps: int, idx_injection: int, power: float, span: float,
                              decay: float) -> List[float]:
    """Per-step slerp weights for a mid branch (``blending_engine.py:452-457``): constant
    ``power`` below the injection index, then a ramp ``power -> power*decay`` up to
    ``round(num_steps*span)``, zero-padded to ``num_steps`` entries."""
    stop = int(round(num_steps * span))
    coeffs = idx_injection * [power]
    ramp_len = stop - idx_injection
    if ramp_len > 0:
        coeffs.extend(list(np.linspace(power, power * decay, ramp_len)))
    coeffs.extend((num_steps - len(coeffs)) * [0])
    return coeffs


def turbo_branching(num_steps: int, depth_strength: Optional[float],
                    nmb_max_branches: Optional[int]) -> Tuple[List[int], List[int]]:
    """Single injection level used for SDXL-Turbo (``blending_engine.py:273-283``).
    ``nmb_max_branches`` counts MID branches here (frames = nmb + 2)."""
    if depth_strength is not None:
        idx_inject = int(round(num_steps * depth_strength))  # python banker's round, as upstream
    else:
        idx_inject = 2
    if nmb_max_branches is None:
        nmb_max_branches = 10
    return [idx_inject], [nmb_max_branches]


def time_based_branching(num_steps: int, depth_strength: float, dt_unet_step: float, dt_vae: float,
                         t_compute_max_allowed: Optional[float] = None,
                         nmb_max_branches: Optional[int] = None):
    """Multi-level plan under a time budget or a frame budget (``blending_engine.py:467-529``).

    Levels start at ``floor(num_steps*depth)`` and are spaced ``ceil(num_steps/10)`` apart.  Stems
    are added one at a time, always to the first level that does not yet exceed its successor
    (else to the last level), until the estimated compute time passes the allowance or the stem
    total reaches ``nmb_max_branches - 2`` (here the budget counts the two anchors, unlike the
    turbo path).  If the frame budget is met on the very first pass the levels are re-sampled
    with ``linspace`` and one stem each.  Returns numpy int arrays like the reference.
    """
    first_level = int(np.floor(num_steps * depth_strength))
    stride = int(np.ceil(num_steps / 10))
    levels = np.arange(first_level, num_steps, stride)
    stems = np.ones(len(levels), dtype=np.int32)

    if nmb_max_branches is None:
        assert t_compute_max_allowed is not None, \
            "Either specify t_compute_max_allowed or nmb_max_branches"
        by_time = True
    elif t_compute_max_allowed is None:
        by_time = False
        nmb_max_branches -= 2  # the two anchors are part of the frame budget
    else:
        raise ValueError("Either specify t_compute_max_allowed or nmb_max_branches")

    first_pass = True
    while True:
        # cost estimate of the plan BEFORE this pass's increment
        unet_steps = (num_steps - levels) * stems
        t_compute = np.sum(unet_steps) * dt_unet_step + dt_vae * np.sum(stems)
        t_compute += 2 * (num_steps * dt_unet_step + dt_vae)

        for lvl in range(len(stems) - 1):
            if stems[lvl + 1] / stems[lvl] >= 1:
                stems[lvl] += 1
                break
        else:
            stems[-1] += 1

        if by_time and t_compute > t_compute_max_allowed:
            break
        if (not by_time) and np.sum(stems) >= nmb_max_branches:
            if first_pass:
                levels = np.linspace(levels[0], levels[-1], nmb_max_branches).astype(np.int32)
                stems = np.ones(len(levels), dtype=np.int32)
            break
        first_pass = False
    return levels, stems


def transition_census(num_steps: int, levels: Sequence[int], stems: Sequence[int]) -> dict:
    """Work count of one ``run_transition`` with fresh anchors (SURVEY.md §3.6 schema)."""
    mid = int(np.sum(stems))
    unet = 2 * num_steps + int(sum((num_steps - int(l)) * int(s) for l, s in zip(levels, stems)))
    return {"unet_forwards": unet, "scheduler_steps": unet, "vae_decodes": mid + 2,
            "frames": mid + 2, "lpips_calls": 2 * mid}
