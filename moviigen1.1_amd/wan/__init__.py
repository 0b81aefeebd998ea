import os

# 8 HIP hardware queues instead of 4: the compute stream, the sequence-parallel exchange stream and RCCL's stream must not
# alias onto one queue or the exchange serialises with the attention kernels (profiles/r02d_sp_overlap.txt).  Effective
# when this package is imported before the first HIP call of the process; launchers set it themselves as well.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

from . import configs, distributed, modules
from .text2video import WanT2V

__all__ = ['WanT2V', 'configs', 'distributed', 'modules']
