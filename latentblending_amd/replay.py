"""Multi-transition driver and the movie JSON replay format.

The reference has no function for this: the loop lives three times in its scripts
(``example_multi_trans.py:39-62``, ``example_multi_trans_json.py:24-71`` and
``latentblending/gradio_ui.py:222-262`` in /root/reference) — ``set_prompt1/2`` for the first
segment, then ``swap_forward`` + ``set_prompt2`` + ``run_transition(recycle_img1=True)`` so that every
key frame is diffused once, one movie part per segment, ``concatenate_movies`` at the end.  This module
is that loop behind two calls, with the same observable order of engine calls (so seeds, recycled
trajectories and the part files are what the scripts produce), plus reader / writer of the JSON file
the gradio UI saves (``gradio_ui.py:168-190``):

    [ {"settings": "sdxl", "width": W, "height": H, "num_inference_steps": S},
      {"iteration": i, "seed": s, "prompt": "...", "negative_prompt": "...", "preview_image": ...}, ... ]

Host-side control only; the hot path below it is ``BlendingEngine.run_transition``.
"""
from __future__ import annotations

import json
import os
from typing import Callable, Dict, List, Optional, Sequence, Tuple


def load_movie_json(fp_json: str) -> Tuple[Dict, List[Dict]]:
    """Return ``(settings, items)`` of a movie JSON (header item first, example_multi_trans_json.py:24-45)."""
    with open(fp_json, "r") as fh:
        data = json.load(fh)
    if not isinstance(data, list) or len(data) < 1 or "width" not in data[0]:
        raise ValueError(f"{fp_json}: not a latentblending movie JSON (header item with width/height missing)")
    header, items = data[0], data[1:]
    for k in ("width", "height", "num_inference_steps"):
        if k not in header:
            raise ValueError(f"{fp_json}: header lacks {k!r}")
    for it in items:
        for k in ("prompt", "negative_prompt", "seed"):
            if k not in it:
                raise ValueError(f"{fp_json}: item {it.get('iteration', '?')} lacks {k!r}")
    return header, items


def save_movie_json(fp_json: str, be, items: Sequence[Dict]) -> None:
    """Write the file the UI writes (gradio_ui.py:168-173): header from the engine's holder, then the items."""
    header = {"settings": "sdxl", "width": be.dh.width_img, "height": be.dh.height_img,
              "num_inference_steps": be.dh.num_inference_steps}
    with open(fp_json, "w") as fh:
        json.dump([header] + [dict(it) for it in items], fh, indent=4)


def run_multi_transition(be, list_prompts: Sequence[str], list_seeds: Sequence[int], fp_movie: Optional[str] = None,
                         duration_single_trans: float = 10, list_negative_prompts: Optional[Sequence[str]] = None,
                         fps: int = 30, dp_parts: str = ".", keep_parts: bool = True,
                         on_segment: Optional[Callable[[int, List], None]] = None,
                         pipeline_keyframes: bool = False) -> List[List]:
    """Chain ``len(list_prompts) - 1`` transitions, recycling the shared key frame of neighbouring segments.

    Engine calls are those of example_multi_trans.py:39-62; with ``list_negative_prompts`` the negative prompt
    is set as in example_multi_trans_json.py:49-58 (item ``i`` for the first segment, ``i + 1`` afterwards —
    the reference's indexing, kept).  ``fp_movie=None`` skips all file output (frames are still returned and
    handed to ``on_segment(i, frames)``).  Returns the frames of every segment.

    ``pipeline_keyframes=True`` (cross-transition pipelining): all key frames are denoised and
    decoded AHEAD of the transitions by ``BlendingEngine.precompute_keyframes`` - one lock-step batch instead of one
    latency-bound batch-1 trajectory per transition on one GPU, key frame k on rank k % world under a branch farm - and
    every transition then runs with both anchors recycled.  Prompts, negative prompts and seeds are paired exactly as
    in the sequential loop; see ``precompute_keyframes`` for the two deliberate differences (guidance scale of the key
    frames, order of ancestral noise draws).
    """
    n = len(list_prompts)
    if n < 2:
        raise ValueError("run_multi_transition needs at least two prompts")
    if len(list_seeds) < n:
        raise ValueError("run_multi_transition needs one seed per prompt")
    if list_negative_prompts is not None and len(list_negative_prompts) < n:
        raise ValueError("run_multi_transition needs one negative prompt per prompt")
    parts, segments = [], []
    keys = _precompute_chain(be, list_prompts, list_seeds, list_negative_prompts) if pipeline_keyframes else None
    for i in range(n - 1):
        if keys is not None:
            embs, trajs, key_frames = keys
            # the engine state the sequential loop would have at this point, without re-encoding anything
            be.prompt1, be.text_embedding1 = list_prompts[i].replace("_", " "), embs[i]
            be.prompt2, be.text_embedding2 = list_prompts[i + 1].replace("_", " "), embs[i + 1]
            if list_negative_prompts is not None:
                be.set_negative_prompt(list_negative_prompts[0 if i == 0 else i + 1])
            be.preset_anchors(trajs[i], trajs[i + 1], key_frames[i], key_frames[i + 1])
            frames = be.run_transition(recycle_img1=True, recycle_img2=True, fixed_seeds=[int(s) for s in list_seeds[i:i + 2]])
        elif i == 0:
            be.set_prompt1(list_prompts[i])
            if list_negative_prompts is not None:
                be.set_negative_prompt(list_negative_prompts[i])
            be.set_prompt2(list_prompts[i + 1])
            recycle_img1 = False
        else:
            be.swap_forward()
            if list_negative_prompts is not None:
                be.set_negative_prompt(list_negative_prompts[i + 1])
            be.set_prompt2(list_prompts[i + 1])
            recycle_img1 = True
        if keys is None:
            fixed_seeds = [int(s) for s in list_seeds[i:i + 2]]
            frames = be.run_transition(recycle_img1=recycle_img1, fixed_seeds=fixed_seeds)
        segments.append(frames)
        if on_segment is not None:
            on_segment(i, frames)
        if fp_movie is not None:
            fp_part = os.path.join(dp_parts, f"tmp_part_{str(i).zfill(3)}.mp4")
            be.write_movie_transition(fp_part, duration_single_trans, fps=fps)
            parts.append(fp_part)
    if fp_movie is not None:
        from .movie import concatenate_movies
        concatenate_movies(fp_movie, parts)
        if not keep_parts:
            for fp in parts:
                os.remove(fp)
    return segments


def _precompute_chain(be, list_prompts, list_seeds, list_negative_prompts):
    """Embeddings of every prompt with the negative prompt the sequential loop pairs it with (prompt 0: the engine's
    current one; prompt 1: item 0; prompt k >= 2: item k), then all key frames in one ``precompute_keyframes`` call."""
    n = len(list_prompts)
    be.set_prompt1(list_prompts[0])
    embs = [be.text_embedding1]
    for k in range(1, n):
        if list_negative_prompts is not None:
            be.set_negative_prompt(list_negative_prompts[0 if k == 1 else k])
        be.set_prompt2(list_prompts[k])
        embs.append(be.text_embedding2)
    trajs, frames = be.precompute_keyframes(embs, [int(s) for s in list_seeds[:n]])
    return embs, trajs, frames


def run_movie_json(be, fp_json: str, fp_movie: Optional[str] = None, duration_single_trans: float = 10,
                   fps: int = 30, dp_parts: str = ".", keep_parts: bool = True, pipeline_keyframes: bool = False) -> List[List]:
    """``example_multi_trans_json.py`` as a call: size and step count from the header, then the chain."""
    header, items = load_movie_json(fp_json)
    be.set_dimensions((header["width"], header["height"]))
    be.set_num_inference_steps(header["num_inference_steps"])
    return run_multi_transition(be, [it["prompt"] for it in items], [it["seed"] for it in items], fp_movie,
                                duration_single_trans, [it["negative_prompt"] for it in items], fps=fps,
                                dp_parts=dp_parts, keep_parts=keep_parts, pipeline_keyframes=pipeline_keyframes)
