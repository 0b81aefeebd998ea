"""Video write-out — the step after the path (reference wan/utils/utils.py:23-60 `cache_video`,
called from scripts/inference/generate.py:300-313).

The arithmetic (clamp, make_grid normalisation of ONE video, x255, truncating cast to uint8 frames
[T,H,W,3]) runs on the GPU (mg_video_to_u8, bit-exact with the reference expression).  The container
encode is host-side as in the reference: `imageio` (libx264) when it is installed.  This image has
neither imageio nor an H.264 encoder: the requested .mp4 is then written by `mp4_mjpeg.write_mp4_mjpeg`
— a plain ISO base media file whose video track carries one JPEG per frame (PIL), readable by every
libavformat-based player; any other suffix, or no PIL either, leaves the uint8 frames as a `.npy`."""
import binascii
import logging
import os
import os.path as osp

import numpy as np
import torch

from ..backend import ops

__all__ = ['cache_video', 'cache_image', 'video_frames_uint8', 'str2bool']


def rand_name(length=8, suffix=''):
    name = binascii.b2a_hex(os.urandom(length)).decode('utf-8')
    if suffix:
        name += suffix if suffix.startswith('.') else '.' + suffix
    return name


def video_frames_uint8(tensor, value_range=(-1, 1)):
    """[1,3,T,H,W] or [3,T,H,W] fp32 device tensor -> uint8 [T,H,W,3] device tensor."""
    if tensor.dim() == 5:
        if tensor.shape[0] != 1:
            raise NotImplementedError('make_grid of several videos (nrow tiling) is not on the T2V path: one video per call')
        tensor = tensor[0]
    return ops.video_to_u8(tensor.to(torch.float32).contiguous(), min(value_range), max(value_range))


def cache_video(tensor, save_file=None, fps=30, suffix='.mp4', nrow=8, normalize=True, value_range=(-1, 1), retry=5):
    if not normalize:
        raise NotImplementedError('the reference always calls cache_video with normalize=True')
    cache_file = osp.join('/tmp', rand_name(suffix=suffix)) if save_file is None else save_file
    frames = video_frames_uint8(tensor, value_range).cpu().numpy()
    try:
        import imageio
    except ModuleNotFoundError:
        if osp.splitext(cache_file)[1].lower() in ('.mp4', '.m4v', '.mov'):
            try:
                from .mp4_mjpeg import write_mp4_mjpeg
                write_mp4_mjpeg(cache_file, frames, fps=fps)
                logging.warning(f'cache_video: imageio is not installed, wrote {frames.shape[0]} JPEG frames into {cache_file} — an mp4v / Motion-JPEG '
                                'track, NOT H.264: browsers and QuickTime will not play it (ffmpeg / VLC do)')
                return cache_file
            except ModuleNotFoundError:          # no PIL either
                pass
            except Exception as e:               # noqa: BLE001 — an encode / IO error of the fallback writer must not lose the frames
                logging.warning(f'cache_video: the Motion-JPEG fallback writer failed ({type(e).__name__}: {e}); writing the frames as .npy')
        path = osp.splitext(cache_file)[0] + '.npy'
        np.save(path, frames)
        logging.info(f'cache_video: no video writer available, wrote the uint8 frames {frames.shape} to {path}')
        return path
    error = None
    for _ in range(retry):
        try:
            writer = imageio.get_writer(cache_file, fps=fps, codec='libx264', quality=8)
            for frame in frames:
                writer.append_data(frame)
            writer.close()
            return cache_file
        except Exception as e:  # noqa: BLE001  (the reference retries on any writer error)
            error = e
    logging.error(f'cache_video failed, error: {error}')
    return None


def cache_image(tensor, save_file, nrow=8, normalize=True, value_range=(-1, 1), retry=5):
    """reference wan/utils/utils.py:64-91, called for the t2i task (generate.py:308-315) with
    `video.squeeze(1)[None]`: ONE image [1,3,H,W] -> PNG (or the suffix given).  torchvision's make_grid returns a
    single image unpadded, so the pixels are clamp, (x-lo)/max(hi-lo,1e-5), *255, +0.5, clamp, uint8 (mg_image_to_u8)."""
    if not normalize:
        raise NotImplementedError('the reference always calls cache_image with normalize=True')
    if tensor.dim() == 4:
        if tensor.shape[0] != 1:
            raise NotImplementedError('make_grid of several images (nrow tiling) is not on the T2V path: one image per call')
        tensor = tensor[0]
    if osp.splitext(save_file)[1].lower() not in ('.jpg', '.jpeg', '.png', '.tiff', '.gif', '.webp'):
        save_file = save_file + '.png'      # the reference would hand PIL an unknown suffix and fail; keep the image
    pixels = ops.image_to_u8(tensor.to(torch.float32).contiguous(), min(value_range), max(value_range)).cpu().numpy()
    error = None
    for _ in range(retry):
        try:
            from PIL import Image
            Image.fromarray(pixels).save(save_file)
            return save_file
        except Exception as e:  # noqa: BLE001  (the reference retries on any writer error)
            error = e
    logging.error(f'cache_image failed, error: {error}')
    return None


def str2bool(v):
    """argparse helper of the reference CLI (utils.py:100-118)."""
    if isinstance(v, bool):
        return v
    lv = v.lower()
    if lv in ('yes', 'true', 't', 'y', '1'):
        return True
    if lv in ('no', 'false', 'f', 'n', '0'):
        return False
    import argparse
    raise argparse.ArgumentTypeError('Boolean value expected (True/False)')
