"""WAN_CONFIGS / SIZE_CONFIGS / MAX_AREA_CONFIGS / SUPPORTED_SIZES — values of the reference
wan/configs/{__init__,shared_config,wan_t2v_14B}.py, without the easydict dependency."""
import copy

import torch


class Config(dict):
    """attribute-access dict (what the reference gets from EasyDict)."""
    __setattr__ = dict.__setitem__

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name) from None


t2v_14B = Config(
    __name__='Config: Wan T2V 14B',
    # shared (reference shared_config.py)
    t5_model='umt5_xxl', t5_dtype=torch.bfloat16, text_len=512, param_dtype=torch.bfloat16,
    num_train_timesteps=1000, sample_fps=16,
    sample_neg_prompt='色调艳丽，过曝，静态，细节模糊不清，字幕，风格，作品，画作，画面，静止，整体发灰，最差质量，低质量，JPEG压缩残留，'
                      '丑陋的，残缺的，多余的手指，画得不好的手部，画得不好的脸部，畸形的，毁容的，形态畸形的肢体，手指融合，静止不动的画面，'
                      '杂乱的背景，三条腿，背景人很多，倒着走',
    # t2v-14B (reference wan_t2v_14B.py)
    t5_checkpoint='models_t5_umt5-xxl-enc-bf16.pth', t5_tokenizer='google/umt5-xxl',
    vae_checkpoint='Wan2.1_VAE.pth', vae_stride=(4, 8, 8),
    patch_size=(1, 2, 2), dim=5120, ffn_dim=13824, freq_dim=256, num_heads=40, num_layers=40,
    window_size=(-1, -1), qk_norm=True, cross_attn_norm=True, eps=1e-6,
)
t2i_14B = copy.deepcopy(t2v_14B)
t2i_14B['__name__'] = 'Config: Wan T2I 14B'

WAN_CONFIGS = {'t2v-14B': t2v_14B, 't2i-14B': t2i_14B}

SIZE_CONFIGS = {
    '1920*1056': (1920, 1056), '1920*1072': (1920, 1072), '1920*832': (1920, 832), '1280*560': (1280, 560),
    '560*1280': (560, 1280), '1056*1920': (1056, 1920), '832*1920': (832, 1920), '720*1280': (720, 1280),
    '1280*720': (1280, 720), '480*832': (480, 832), '832*480': (832, 480), '1024*1024': (1024, 1024),
}
MAX_AREA_CONFIGS = {'720*1280': 720 * 1280, '1280*720': 1280 * 720, '480*832': 480 * 832, '832*480': 832 * 480}
SUPPORTED_SIZES = {
    't2v-14B': ('720*1280', '1280*720', '480*832', '832*480', '1920*1056', '1056*1920', '1920*832', '832*1920',
                '1920*1072', '1072*1920', '1280*560', '560*1280'),
    't2i-14B': tuple(SIZE_CONFIGS.keys()),
}
