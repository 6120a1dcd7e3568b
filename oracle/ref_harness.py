"""ORACLE — test infrastructure, NOT product code.

Imports the UNCHANGED reference package from /root/reference (read-only) so that its own host
classes can act as the oracle for the tree policy, the restartable denoising loop and the mixing
helpers.  The reference needs three third-party packages that are not installed here
(``lpips``, ``lunar_tools``, ``diffusers``); they are replaced by minimal stubs in
``sys.modules`` for the duration of the import.  Nothing is copied: the modules
are executed from where they lie.  /root/reference does not exist on the GPU box, so everything
derived from it is committed as fixtures under ``tests/golden/`` by ``oracle/make_golden.py``.
"""
from __future__ import annotations

import contextlib
import importlib
import os
import sys
import types
from typing import Optional

import torch

REFERENCE_ROOT = os.environ.get("LB_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "latentblending", "blending_engine.py"))


def _stub_modules(lpips_factory):
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        return m

    class _Unused:  # placeholder for names the reference imports but never touches on this path
        def __init__(self, *a, **k):
            raise RuntimeError("stubbed third-party class instantiated")

    def retrieve_timesteps(scheduler, num_inference_steps=None, device=None, timesteps=None, **_):
        scheduler.set_timesteps(num_inference_steps, device=device)
        return scheduler.timesteps, num_inference_steps

    def fill_up_frames_linear_interpolation(imgs, a, b):
        return list(imgs)

    class MovieSaver:
        def __init__(self, *a, **k):
            self.frames = []

        def write_frame(self, f):
            self.frames.append(f)

        def finalize(self):
            pass

    stubs = {
        "lpips": mod("lpips", LPIPS=lambda net='alex': lpips_factory()),
        "lunar_tools": mod("lunar_tools", MovieSaver=MovieSaver,
                           fill_up_frames_linear_interpolation=fill_up_frames_linear_interpolation,
                           concatenate_movies=lambda *a, **k: None),
        "diffusers": mod("diffusers", DiffusionPipeline=_Unused,
                         StableDiffusionControlNetPipeline=_Unused, ControlNetModel=_Unused,
                         AutoPipelineForText2Image=_Unused),
        "diffusers.models": mod("diffusers.models"),
        "diffusers.models.attention_processor": mod(
            "diffusers.models.attention_processor", AttnProcessor2_0=_Unused,
            LoRAAttnProcessor2_0=_Unused, LoRAXFormersAttnProcessor=_Unused,
            XFormersAttnProcessor=_Unused),
        "diffusers.pipelines": mod("diffusers.pipelines"),
        "diffusers.pipelines.stable_diffusion_xl": mod("diffusers.pipelines.stable_diffusion_xl"),
        "diffusers.pipelines.stable_diffusion_xl.pipeline_stable_diffusion_xl": mod(
            "diffusers.pipelines.stable_diffusion_xl.pipeline_stable_diffusion_xl",
            retrieve_timesteps=retrieve_timesteps),
    }
    return stubs


_CACHE = {}


def load_reference(lpips_factory=None):
    """Returns a namespace with the reference's ``BlendingEngine``, ``DiffusersHolder`` and
    ``utils`` module, imported in isolation (our own ``latentblending`` shim, if imported, is put
    back into ``sys.modules`` afterwards)."""
    if not reference_available():
        raise FileNotFoundError(f"reference not found under {REFERENCE_ROOT}")
    if "ns" in _CACHE:
        _CACHE["lpips_factory"][0] = lpips_factory or _CACHE["lpips_factory"][0]
        return _CACHE["ns"]
    if lpips_factory is None:
        from .sdxl_ref import OracleLPIPS
        lpips_factory = OracleLPIPS
    holder = [lpips_factory]
    stubs = _stub_modules(lambda: holder[0]())

    saved = {k: v for k, v in sys.modules.items()
             if k == "latentblending" or k.startswith("latentblending.") or k in stubs}
    for k in saved:
        del sys.modules[k]
    sys.modules.update(stubs)
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        eng = importlib.import_module("latentblending.blending_engine")
        hold = importlib.import_module("latentblending.diffusers_holder")
        utils = importlib.import_module("latentblending.utils")
    finally:
        sys.path.remove(REFERENCE_ROOT)
        for k in [k for k in sys.modules if k == "latentblending" or k.startswith("latentblending.")
                  or k in stubs]:
            del sys.modules[k]
        sys.modules.update(saved)
    torch.set_grad_enabled(False)
    ns = types.SimpleNamespace(BlendingEngine=eng.BlendingEngine, DiffusersHolder=hold.DiffusersHolder,
                               utils=utils, engine_module=eng, holder_module=hold)
    _CACHE["ns"] = ns
    _CACHE["lpips_factory"] = holder
    return ns


@contextlib.contextmanager
def cuda_is_identity():
    """The reference calls ``.cuda(device)`` unconditionally on the LPIPS path
    (blending_engine.py:76,750,753); on a GPU-less box make it a no-op while it runs."""
    if torch.cuda.is_available():
        yield
        return
    original = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        yield
    finally:
        torch.Tensor.cuda = original
