"""Stand-in for the ``diffusers`` package, used ONLY when the real one is not installed.

Scripts of the reference start with ``from diffusers import AutoPipelineForText2Image`` and call
``AutoPipelineForText2Image.from_pretrained("stabilityai/sdxl-turbo", torch_dtype=torch.float16,
variant="fp16")`` followed by ``pipe.to("cuda")`` (example_single_trans.py:3,11-12).  This facade
makes those lines return the MI355X-native SDXL pipe, so the scripts run unchanged:

* weights: HF-layout safetensors under ``$LB_WEIGHTS_DIR/{unet,vae}/`` when present (plus ``text_encoder/``,
  ``text_encoder_2/``, ``tokenizer*/`` for real prompt conditioning and ``lpips/`` for the real metric), otherwise
  seeded synthetic SDXL-shaped stand-ins, each announced with a ``UserWarning`` (no network / checkpoint here);
* if a real ``diffusers`` distribution exists further down ``sys.path`` it is loaded instead and
  this module gets out of the way.
"""
import importlib.machinery
import importlib.util
import os
import sys


def _real_diffusers():
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    paths = [p for p in sys.path if os.path.abspath(p or ".") != here]
    spec = importlib.machinery.PathFinder.find_spec("diffusers", paths)
    return spec if spec is not None and spec.origin and os.path.dirname(spec.origin) != os.path.dirname(__file__) else None


_spec = _real_diffusers()
if _spec is not None:                                   # defer to the real package
    _mod = importlib.util.module_from_spec(_spec)
    sys.modules[__name__] = _mod
    _spec.loader.exec_module(_mod)
else:
    class AutoPipelineForText2Image:
        @staticmethod
        def from_pretrained(pretrained_model_name_or_path, torch_dtype=None, variant=None, **kwargs):
            import latentblending_amd.native as N
            name = str(pretrained_model_name_or_path)
            turbo = "turbo" in name
            root = os.environ.get("LB_WEIGHTS_DIR")
            unet_p = vae_p = None
            if root and os.path.isdir(os.path.join(root, "unet")):
                unet_p = N.from_safetensors(os.path.join(root, "unet"))
            if root and os.path.isdir(os.path.join(root, "vae")):
                vae_p = N.from_safetensors(os.path.join(root, "vae"))
            kw = {}
            if os.environ.get("LB_TINY_MODEL") == "1":      # smoke tests: SDXL-shaped but tiny
                kw = dict(unet_cfg=N.UNetConfig(block_channels=(64, 128, 256), transformer_depth=(0, 1, 2), cross_dim=256,
                                                pooled_dim=128, add_time_dim=32, sample_size=16),
                          vae_cfg=N.VAEConfig(block_channels=(32, 64, 128, 128)))
            # text conditioning and the perceptual metric from the same directory when it holds them: HF layout
            # text_encoder/ + text_encoder_2/ (+ tokenizer/, tokenizer_2/), and lpips/ = {alexnet*.safetensors | .pth ,
            # alex*.pth | lpips_lin*.safetensors} (torchvision AlexNet trunk + the lpips v0.1 linear layers)
            if root and os.path.isdir(os.path.join(root, "text_encoder")) and os.path.isdir(os.path.join(root, "text_encoder_2")):
                kw["text_encoder_fn"] = N.NativeTextEncoders.from_dir(root).encode
            lp_dir = os.path.join(root, "lpips") if root else None
            if lp_dir and os.path.isdir(lp_dir):
                kw["lpips_provider"] = _lpips_from_dir(N, lp_dir)
            return N.NativeSDXLPipe(turbo=turbo, unet_provider=unet_p, vae_provider=vae_p, name_or_path=name, **kw)

    def _load_state(path):
        if path.endswith(".safetensors"):
            from safetensors.torch import load_file
            return load_file(path)
        import torch
        return torch.load(path, map_location="cpu", weights_only=True)

    def _lpips_from_dir(N, lp_dir):
        files = sorted(os.listdir(lp_dir))
        trunk = [f for f in files if "alexnet" in f.lower()]
        lins = [f for f in files if f not in trunk and (f.lower().startswith("alex") or "lin" in f.lower())]
        if not trunk or not lins:
            raise FileNotFoundError(f"{lp_dir}: need the AlexNet trunk (alexnet*.pth|safetensors) and the lpips linear layers (alex.pth)")
        from latentblending_amd.native.weights import lpips_provider
        return lpips_provider(_load_state(os.path.join(lp_dir, trunk[0])), _load_state(os.path.join(lp_dir, lins[0])))

    DiffusionPipeline = AutoPipelineForText2Image
    __all__ = ["AutoPipelineForText2Image", "DiffusionPipeline"]
