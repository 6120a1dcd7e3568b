"""Per-user engine state and the multi-user router (minimal form: no widgets).

The reference's gradio front end shares ONE ``BlendingEngine`` per model between all users
(``latentblending/gradio_ui.py:29-54`` in /root/reference: ``MultiUserRouter.register_new_user`` hands every user a
``BlendingVariableHolder`` around the same engine object) and the engine keeps its settings in mutable fields of itself,
of the holder and of the pipe (``dh.guidance_scale``, scheduler tables, negative prompt, image size:
``diffusers_holder.py:53,224-229``) - so two users' calls overwrite each other's state (SURVEY.md §8b "Threading").

Here the weights, launch programs and hipGraphs stay shared (one engine per model, as upstream), but everything a call
READS OR LEAVES BEHIND that belongs to a user lives in an :class:`EngineSession`: prompts and their embeddings, negative
prompt, seeds, size, step count, guidance, crossfeed settings, the branching plan, the transition tree (latents, frames,
fractions, similarities) and an optional private noise source.  ``with session.bound() as be:`` takes the router's lock,
installs the session's state on the shared engine, runs the user's calls, stores what they left behind back into the
session and RESTORES the engine - calls of different users can interleave in any order and each sees an engine that nobody
else touched.  New users start from the defaults the router captured when it was built (round 5: a session used to copy
the live engine at registration and bound() left the last user's state on it - a user registered after, or during,
somebody's block inherited that user's prompts, embeddings and preset frames).

Persistence is the reference's ``get_state_dict`` + ``yml_save`` (``blending_engine.py:709-728``, ``utils.py:245-262``);
``BlendingEngine.load_state_dict`` is the missing inverse, so ``get_state_dict -> yml_save -> yml_load -> load_state_dict``
on a fresh engine reproduces the same ``run_transition``.
"""
from __future__ import annotations

import contextlib
import copy
import threading
import uuid
from typing import Dict, Optional

from .tree import TransitionTree

# engine attributes that are per-user state (everything run_transition / compute_latents* read or write)
_ENGINE_FIELDS = ("prompt1", "prompt2", "text_embedding1", "text_embedding2", "negative_prompt", "seed1", "seed2",
                  "guidance_scale_base", "guidance_scale", "guidance_scale_mid_damper", "mid_compression_scaler",
                  "branch1_crossfeed_power", "branch1_crossfeed_range", "branch1_crossfeed_decay",
                  "parental_crossfeed_power", "parental_crossfeed_range", "parental_crossfeed_decay",
                  "num_inference_steps", "list_idx_injection", "list_nmb_stems", "image1_lowres", "image2_lowres",
                  "multi_transition_img_first", "multi_transition_img_last", "_preset_anchor_frames", "stats")
# holder attributes that follow them
_HOLDER_FIELDS = ("negative_prompt", "guidance_scale", "num_inference_steps", "width_img", "height_img", "width_latent",
                  "height_latent")


# of those, what a finished or prepared RUN leaves behind (never part of a new user's defaults)
_TRANSIENT_FIELDS = ("image1_lowres", "image2_lowres", "multi_transition_img_first", "multi_transition_img_last",
                     "_preset_anchor_frames")


def _snapshot(be):
    """(engine fields, holder fields, tree) of ``be`` as they are now; mutable containers are copied."""
    return ({k: copy.copy(getattr(be, k)) for k in _ENGINE_FIELDS}, {k: getattr(be.dh, k) for k in _HOLDER_FIELDS}, be._tree)


_ENGINE_LOCKS_GUARD = threading.Lock()


def _install(be, engine, holder, tree):
    for k, v in engine.items():
        setattr(be, k, v)
    for k, v in holder.items():
        setattr(be.dh, k, v)
    be._tree = tree
    be.dh.set_num_inference_steps(holder["num_inference_steps"])           # scheduler tables are pipe state


class EngineSession:
    """One user's view of a shared ``BlendingEngine``.

    A session starts from ``defaults`` - the (engine fields, holder fields) snapshot its router took of the engine when the
    ROUTER was built - never from whatever the shared engine happens to hold when the user registers; run leftovers
    (preset anchor frames, multi-transition images) are cleared and the tree is empty.  Without a router
    (``EngineSession(be)``) the snapshot is taken here, under the session's lock.  ``bound()`` puts the engine back the way
    it found it, so nothing a user did stays on the shared object between blocks."""

    def __init__(self, be, lock: Optional[threading.RLock] = None, noise_source=None, defaults=None):
        self.be = be
        if lock is None:        # ONE lock per engine, shared by every router-less session of it (a lock per session would let two such
            #                     sessions install themselves on the same engine from two threads at once)
            with _ENGINE_LOCKS_GUARD:
                lock = getattr(be, "_session_lock", None)
                if lock is None:
                    lock = be._session_lock = threading.RLock()
        self._lock = lock
        self.noise_source = noise_source            # optional private ancestral-noise source (native pipes)
        if defaults is None:
            with self._lock:
                if getattr(be, "_bound_session", None) is not None:
                    raise RuntimeError("EngineSession: the engine is inside another session's bound() block; build sessions "
                                       "from a SessionRouter (defaults taken once) or outside bound()")
                defaults = _snapshot(be)[:2]
        self._engine = {k: copy.copy(v) for k, v in defaults[0].items()}
        for k in _TRANSIENT_FIELDS:
            self._engine[k] = None
        self._engine["stats"] = {}
        self._holder = dict(defaults[1])
        self._tree = TransitionTree()               # a user never sees another user's tree

    @contextlib.contextmanager
    def bound(self):
        """Install this session on the shared engine for the duration of the block (exclusive), then put the engine back.
        Re-entering the SAME session on the same thread is a no-op; binding a second session inside the block raises."""
        be = self.be
        with self._lock:
            active = getattr(be, "_bound_session", None)
            if active is self:                      # (re-entrant use of one session: already installed)
                yield be
                return
            if active is not None:
                raise RuntimeError("EngineSession.bound(): another session is bound to this engine on this thread - "
                                   "sessions do not nest (leave the first block before entering the second)")
            saved = _snapshot(be)
            sched = getattr(be.dh.pipe, "scheduler", None)
            swap_noise = self.noise_source is not None and hasattr(sched, "noise_source")
            previous = sched.noise_source if swap_noise else None
            be._bound_session = self
            installed = False
            try:
                # (inside the try: an install that raises half way - e.g. a step count the scheduler rejects - must not leave
                #  part of this user's state on the shared engine)
                _install(be, self._engine, self._holder, self._tree)
                installed = True
                if swap_noise:
                    sched.noise_source = self.noise_source
                yield be
            finally:
                if swap_noise:
                    sched.noise_source = previous
                if installed:                       # (a failed install leaves the session as it was)
                    self._engine = {k: getattr(be, k) for k in _ENGINE_FIELDS}
                    self._holder = {k: getattr(be.dh, k) for k in _HOLDER_FIELDS}
                    self._tree = be._tree
                be._bound_session = None
                _install(be, *saved)                # the shared engine keeps nothing of this user

    # conveniences mirroring what the UI holder calls (gradio_ui.py:139-149, 238-256)
    def run_transition(self, **kw):
        with self.bound() as be:
            return be.run_transition(**kw)

    def get_state_dict(self) -> Dict:
        with self.bound() as be:
            return be.get_state_dict()

    def load_state_dict(self, state: Dict):
        with self.bound() as be:
            be.load_state_dict(state)


class SessionRouter:
    """``MultiUserRouter`` without the widgets (gradio_ui.py:29-54): engines per model name, sessions per user id, one
    lock per engine so that users of different models do not wait for each other."""

    def __init__(self, engines: Dict[str, object]):
        self.dict_blendingengines = dict(engines)
        self._locks = {m: threading.RLock() for m in self.dict_blendingengines}
        # every user's starting point: the engines as the OPERATOR configured them, captured once, here
        self._defaults = {}
        for m, be in self.dict_blendingengines.items():
            with self._locks[m]:
                self._defaults[m] = _snapshot(be)[:2]
        self.user_sessions: Dict[str, EngineSession] = {}

    def register_new_user(self, model: str, width: int, height: int, noise_source=None) -> str:
        user_id = str(uuid.uuid4().hex.upper()[0:8])
        be = self.dict_blendingengines[model]
        session = EngineSession(be, self._locks[model], noise_source=noise_source, defaults=self._defaults[model])
        with session.bound() as engine:
            engine.set_dimensions((width, height))
        self.user_sessions[user_id] = session
        return user_id

    def session(self, user_id: str) -> EngineSession:
        return self.user_sessions[user_id]

    def drop_user(self, user_id: str) -> None:
        self.user_sessions.pop(user_id, None)
