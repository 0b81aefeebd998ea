from .attention import flash_attention
from .model import WanModel
from .t5 import T5Encoder, T5EncoderModel
from .tokenizers import HuggingfaceTokenizer
from .vae import WanVAE

__all__ = ['WanVAE', 'WanModel', 'T5Encoder', 'T5EncoderModel', 'HuggingfaceTokenizer', 'flash_attention']
