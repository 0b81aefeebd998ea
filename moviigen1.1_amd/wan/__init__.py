from . import configs, distributed, modules
from .text2video import WanT2V

__all__ = ['WanT2V', 'configs', 'distributed', 'modules']
