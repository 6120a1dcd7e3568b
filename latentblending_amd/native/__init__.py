"""MI355X-native SDXL modules: UNet / VAE / LPIPS launch programs and the native pipe."""
from .pipe import NativeSDXLPipe, StableDiffusionXLPipeline
from .unet import NativeUNet, UNetConfig
from .vae import NativeVAEDecoder, VAEConfig
from .clip import CLIPTextConfig, NativeCLIPText, NativeTextEncoders
from .weights import DictProvider, SyntheticProvider, from_safetensors, lpips_provider

__all__ = ["NativeSDXLPipe", "StableDiffusionXLPipeline", "NativeUNet", "UNetConfig", "NativeVAEDecoder",
           "VAEConfig", "DictProvider", "SyntheticProvider", "from_safetensors", "lpips_provider", "CLIPTextConfig", "NativeCLIPText", "NativeTextEncoders"]
