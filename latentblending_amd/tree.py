"""Transition tree: the ordered set of committed branches along the prompt1 -> prompt2 axis and
the greedy "split the least-similar gap" policy.

Restates the policy of the reference engine (paths relative to /root/reference):
  * state lists                ``blending_engine.py:57-61, 345-349``
  * pick gap / find parents    ``blending_engine.py:531-562``  (``get_mixing_parameters``)
  * neighbours of a fraction   ``blending_engine.py:767-789``  (``get_closest_idx``)
  * commit a branch            ``blending_engine.py:564-588``  (``insert_into_tree``)

A *gap* is the interval between two adjacent committed branches; ``similarities[g]`` is the
perceptual distance across gap ``g`` (higher = less similar).  A gap has exactly one possible
child — its midpoint — which is what makes speculative evaluation of several gaps at once safe
(``BlendingEngine`` frontier mode).
"""
from __future__ import annotations

from typing import Any, List, Optional, Tuple

import numpy as np


class _Unscored:
    """Placeholder for the single anchor-to-anchor gap right after a reset.  The reference stores
    a bound method there by accident (``blending_engine.py:349``) and never computes that
    distance; ``np.argmax`` over the one-element list still selects gap 0.  Same observable
    behaviour here, without calling the metric."""

    def __repr__(self):
        return "<unscored gap>"


UNSCORED = _Unscored()


class TransitionTree:
    def __init__(self):
        self.latents: List[Any] = [None, None]       # per branch: list (len = steps) of latents / None
        self.fracts: Optional[List[float]] = None
        self.frames: List[Any] = []                  # decoded final image per branch
        self.idx_injection: List[int] = []
        self.similarities: List[Any] = []

    # -- lifecycle ---------------------------------------------------------------------------
    def reset(self, latents_first, latents_last, frame_first, frame_last) -> None:
        self.latents = [latents_first, latents_last]
        self.fracts = [0.0, 1.0]
        self.frames = [frame_first, frame_last]
        self.idx_injection = [0, 0]
        self.similarities = [UNSCORED]

    def __len__(self) -> int:
        return 0 if self.fracts is None else len(self.fracts)

    # -- policy ------------------------------------------------------------------------------
    def widest_gap(self) -> int:
        """Index of the gap to split next: first maximum of the similarity list."""
        if len(self.similarities) == 1 and self.similarities[0] is UNSCORED:
            return 0
        return int(np.argmax(self.similarities))

    def gap_child(self, gap: int, idx_injection: int) -> Tuple[float, int, int]:
        """Midpoint fraction of ``gap`` and the two parents a branch injected at ``idx_injection``
        must mix: walking outwards from the gap's ends, the first branches that were themselves
        injected strictly earlier."""
        fract = (self.fracts[gap] + self.fracts[gap + 1]) / 2
        left = gap
        while self.idx_injection[left] >= idx_injection:
            left -= 1
        right = gap + 1
        while self.idx_injection[right] >= idx_injection:
            right += 1
        return fract, left, right

    def next_split(self, idx_injection: int) -> Tuple[float, int, int]:
        return self.gap_child(self.widest_gap(), idx_injection)

    def neighbours(self, fract: float) -> Tuple[int, int]:
        """Closest committed branch at or below ``fract`` and closest strictly above it."""
        delta = fract - np.asarray(self.fracts)
        below = np.where(delta < 0, np.inf, delta)
        above = np.where(-delta <= 0, np.inf, -delta)
        lo, hi = int(np.argmin(below)), int(np.argmin(above))
        return (lo, hi) if lo <= hi else (hi, lo)

    def commit(self, fract: float, idx_injection: int, latents, frame,
               sim_left: float, sim_right: float) -> int:
        lo, _ = self.neighbours(fract)
        pos = lo + 1
        self.latents.insert(pos, latents)
        self.frames.insert(pos, frame)
        self.fracts.insert(pos, fract)
        self.idx_injection.insert(pos, idx_injection)
        self.similarities[lo] = sim_left
        self.similarities.insert(pos, sim_right)
        return pos
