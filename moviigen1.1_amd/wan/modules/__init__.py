from .attention import flash_attention
from .model import WanModel
from .vae import WanVAE

__all__ = ['WanVAE', 'WanModel', 'flash_attention']
