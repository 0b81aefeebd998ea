"""Launcher with the flag set of the reference CLI (scripts/inference/generate.py:66-179; SURVEY.md
§8(f) rank 4): same names, defaults, validation messages, seed broadcast, offload default and output
file naming (generate.py:300-306), on the MI355X engine.

    python scripts/inference/generate.py --task t2v-14B --size 1280*720 --ckpt_dir ./MoviiGen1.1 \\
        --prompt "A cat walks on the grass, realistic style." --base_seed 42
    torchrun --nproc_per_node 8 --master-addr 127.0.0.1 scripts/inference/generate.py ... \\
        --dit_fsdp --t5_fsdp --ulysses_size 8          (the reference's inference.sh line)

Differences: no xfuser (sequence parallelism is built in; `--ulysses_size U` x `--ring_size R`
must equal the world size as in the reference), `--use_prompt_extend` is not built and says so,
`--t5_fsdp` / `--t5_cpu` are accepted (the 9.4 GB encoder is simply replicated on the GPU).  Extra:
`--cfg_parallel`, `--vae_parallel` (this engine's multi-GPU layouts) and
`--prompt_embeds FILE` (a torch file {'prompt': [len,4096], 'negative': [len,4096]} instead of running
umT5 — for boxes without the tokenizer files)."""
import argparse
import logging
import os
import random
import sys
from datetime import datetime

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')   # before HIP initialises: see moviigen1.1_amd/wan/__init__.py

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'moviigen1.1_amd'))

import wan  # noqa: E402
from wan.configs import SIZE_CONFIGS, SUPPORTED_SIZES, WAN_CONFIGS  # noqa: E402
from wan.utils.utils import cache_image, cache_video, str2bool  # noqa: E402

EXAMPLE_PROMPT = {   # one short default per task (the reference ships long showcase prompts here)
    't2v-14B': {'prompt': 'A cat walks on the grass, realistic style.'},
    't2i-14B': {'prompt': 'A snow-capped mountain above a quiet lake at dawn, wide-angle photograph.'},
}

# (flag, kwargs) in the reference's order
FLAGS = [
    ('--task', dict(type=str, default='t2v-14B', choices=list(WAN_CONFIGS.keys()), help='The task to run.')),
    ('--size', dict(type=str, default='1280*720', choices=list(SIZE_CONFIGS.keys()),
                    help='The area (width*height) of the generated video.')),
    ('--frame_num', dict(type=int, default=None, help='How many frames to sample. The number should be 4n+1')),
    ('--ckpt_dir', dict(type=str, default=None, help='The path to the checkpoint directory.')),
    ('--offload_model', dict(type=str2bool, default=None,
                             help='Whether to offload the model to CPU after the denoising loop.')),
    ('--ulysses_size', dict(type=int, default=1, help='The size of the ulysses parallelism in DiT.')),
    ('--ring_size', dict(type=int, default=1, help='The size of the ring attention parallelism in DiT.')),
    ('--t5_fsdp', dict(action='store_true', default=False, help='Whether to use FSDP for T5.')),
    ('--t5_cpu', dict(action='store_true', default=False, help='Whether to place T5 model on CPU.')),
    ('--dit_fsdp', dict(action='store_true', default=False, help='Whether to use FSDP for DiT.')),
    ('--save_file', dict(type=str, default=None, help='The file to save the generated image or video to.')),
    ('--prompt', dict(type=str, default=None, help='The prompt to generate the image or video from.')),
    ('--use_prompt_extend', dict(action='store_true', default=False, help='Whether to use prompt extend.')),
    ('--prompt_extend_model', dict(type=str, default='ZuluVision/MoviiGen1.1_Prompt_Rewriter', help='The prompt extend model to use.')),
    ('--prompt_extend_target_lang', dict(type=str, default='en', help='The target language of prompt extend.')),
    ('--base_seed', dict(type=int, default=-1, help='The seed to use for generating the image or video.')),
    ('--sample_solver', dict(type=str, default='unipc', choices=['unipc', 'dpm++'], help='The solver used to sample.')),
    ('--sample_steps', dict(type=int, default=None, help='The sampling steps.')),
    ('--sample_shift', dict(type=float, default=None, help='Sampling shift factor for flow matching schedulers.')),
    ('--sample_guide_scale', dict(type=float, default=5.0, help='Classifier free guidance scale.')),
    # this engine's additions
    ('--cfg_parallel', dict(action='store_true', default=False,
                            help='cond / uncond forwards on the two halves of an even number of ranks, Ulysses inside each half (not with --ring_size > 1).')),
    ('--vae_parallel', dict(nargs='?', const='spatial', default=False, choices=['spatial', 'pipeline'],
                            help='VAE decode over all ranks instead of rank 0 alone: spatial (default) = every rank decodes a band of image columns, '
                                 'pipeline = the decoder layers cut into one segment per rank.')),
    ('--prompt_embeds', dict(type=str, default=None, help="torch file {'prompt','negative'} of umT5 embeddings, replaces the text encoder.")),
]


def _validate_args(args):
    assert args.ckpt_dir is not None, 'Please specify the checkpoint directory.'
    assert args.task in WAN_CONFIGS, f'Unsupport task: {args.task}'
    assert args.task in EXAMPLE_PROMPT, f'Unsupport task: {args.task}'
    if args.sample_steps is None:
        args.sample_steps = 50
    if args.sample_shift is None:
        args.sample_shift = 5.0
    if args.frame_num is None:
        args.frame_num = 1 if 't2i' in args.task else 81
    if 't2i' in args.task:
        assert args.frame_num == 1, f'Unsupport frame_num {args.frame_num} for task {args.task}'
    args.base_seed = args.base_seed if args.base_seed >= 0 else random.randint(0, sys.maxsize)
    assert args.size in SUPPORTED_SIZES[args.task], \
        f"Unsupport size {args.size} for task {args.task}, supported sizes are: {', '.join(SUPPORTED_SIZES[args.task])}"


def _parse_args(argv=None):
    parser = argparse.ArgumentParser(description='Generate a image or video from a text prompt using Wan / MoviiGen1.1')
    for flag, kw in FLAGS:
        parser.add_argument(flag, **kw)
    args = parser.parse_args(argv)
    _validate_args(args)
    return args


def default_save_name(args, now=None):
    """reference generate.py:300-306."""
    stamp = (now or datetime.now()).strftime('%Y%m%d_%H%M%S')
    prompt = args.prompt.replace(' ', '_').replace('/', '_')[:50]
    size = args.size.replace('*', 'x') if sys.platform == 'win32' else args.size
    suffix = '.png' if 't2i' in args.task else '.mp4'
    return f'{args.task}_{size}_{args.ulysses_size}_{args.ring_size}_{prompt}_{stamp}' + suffix


def _init_logging(rank):
    if rank == 0:
        logging.basicConfig(level=logging.INFO, format='[%(asctime)s] %(levelname)s: %(message)s',
                            handlers=[logging.StreamHandler(stream=sys.stdout)])
    else:
        logging.basicConfig(level=logging.ERROR)


def generate(args):
    rank, world = int(os.getenv('RANK', 0)), int(os.getenv('WORLD_SIZE', 1))
    local = int(os.getenv('LOCAL_RANK', 0))
    _init_logging(rank)
    if args.offload_model is None:
        args.offload_model = world == 1
        logging.info(f'offload_model is not specified, set to {args.offload_model}.')
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local)
        dist.init_process_group(backend='nccl', init_method='env://', rank=rank, world_size=world,
                                device_id=torch.device(f'cuda:{local}'))
    else:
        assert not (args.t5_fsdp or args.dit_fsdp), \
            't5_fsdp and dit_fsdp are not supported in non-distributed environments.'
        assert not (args.ulysses_size > 1 or args.ring_size > 1), \
            'context parallel are not supported in non-distributed environments.'
    if args.ulysses_size > 1 or args.ring_size > 1:
        assert args.ulysses_size * args.ring_size == world, \
            'The number of ulysses_size and ring_size should be equal to the world size.'
    if args.use_prompt_extend:
        raise NotImplementedError('prompt extension (a 7B LLM rewriter) is outside this engine: pass the rewritten prompt')
    cfg = WAN_CONFIGS[args.task]
    if args.ulysses_size > 1:
        assert cfg.num_heads % args.ulysses_size == 0, \
            f'`{cfg.num_heads=}` cannot be divided evenly by `{args.ulysses_size=}`.'
    logging.info(f'Generation job args: {args}')
    logging.info(f'Generation model config: {cfg}')
    if dist.is_initialized():
        seed = [args.base_seed] if rank == 0 else [None]
        dist.broadcast_object_list(seed, src=0)
        args.base_seed = seed[0]
    if args.prompt is None:
        args.prompt = EXAMPLE_PROMPT[args.task]['prompt']
    logging.info(f'Input prompt: {args.prompt}')

    logging.info('Creating WanT2V pipeline.')
    pipe = wan.WanT2V(config=cfg, checkpoint_dir=args.ckpt_dir, device_id=local, rank=rank, t5_fsdp=args.t5_fsdp,
                      dit_fsdp=args.dit_fsdp, use_usp=(args.ulysses_size > 1 or args.ring_size > 1), t5_cpu=args.t5_cpu,
                      cfg_parallel=args.cfg_parallel, vae_parallel=args.vae_parallel,
                      sp_degrees=(args.ulysses_size, args.ring_size) if args.ring_size > 1 else None)
    prompt, n_prompt = args.prompt, ''
    if args.prompt_embeds:
        emb = torch.load(args.prompt_embeds, map_location='cpu', weights_only=True)
        prompt, n_prompt = emb['prompt'], emb['negative']
    logging.info(f"Generating {'image' if 't2i' in args.task else 'video'} ...")
    video = pipe.generate(prompt, size=SIZE_CONFIGS[args.size], frame_num=args.frame_num, shift=args.sample_shift,
                          sample_solver=args.sample_solver, sampling_steps=args.sample_steps,
                          guide_scale=args.sample_guide_scale, n_prompt=n_prompt, seed=args.base_seed,
                          offload_model=args.offload_model)
    if rank == 0:
        if args.save_file is None:
            args.save_file = default_save_name(args)
        if 't2i' in args.task:          # reference generate.py:308-315
            logging.info(f'Saving generated image to {args.save_file}')
            args.saved_as = cache_image(tensor=video.squeeze(1)[None], save_file=args.save_file, nrow=1,
                                        normalize=True, value_range=(-1, 1))
        else:
            logging.info(f'Saving generated video to {args.save_file}')
            args.saved_as = cache_video(tensor=video[None], save_file=args.save_file, fps=cfg.sample_fps, nrow=1,
                                        normalize=True, value_range=(-1, 1))
    logging.info('Finished.')
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    return args


if __name__ == '__main__':
    generate(_parse_args())
