"""latentblending_amd — MI355X-native latent-blending transition engine.

Import surface mirrors the reference package (``latentblending/__init__.py:1-3``).
"""
from .blending_engine import BlendingEngine
from .diffusers_holder import DiffusersHolder
from .utils import (interpolate_spherical, add_frames_linear_interp, interpolate_linear,
                    get_spacing, get_time, yml_load, yml_save)

from . import replay  # noqa: E402  (multi-transition driver + movie JSON)
from .session import EngineSession, SessionRouter  # noqa: E402  (per-user state + router)
from .frontend import BlendingVariableHolder, MultiUserRouter  # noqa: E402  (the Gradio page's logic without the widgets, §8f rank 4)

__all__ = ["BlendingEngine", "replay", "EngineSession", "SessionRouter", "BlendingVariableHolder", "MultiUserRouter", "DiffusersHolder", "interpolate_spherical",
           "add_frames_linear_interp", "interpolate_linear", "get_spacing", "get_time",
           "yml_load", "yml_save"]
