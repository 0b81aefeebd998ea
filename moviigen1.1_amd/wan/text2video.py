"""WanT2V — the drop-in pipeline surface (reference wan/text2video.py:28-271).

Same constructor and `generate()` signature, defaults and return contract
(Tensor[3, frame_num, H, W] fp32 in [-1,1] on rank 0, None elsewhere).  What runs underneath is
the MI355X engine: WanModel (HIP kernels), the fused-update schedulers, WanVAE (HIP kernels),
Ulysses over RCCL when `use_usp`, block-sharded weights when `dit_fsdp`.

The umT5-XXL text encoder (wan/modules/t5.py on the same HIP kernels) is
created from `checkpoint_dir/config.t5_checkpoint` as in the reference when that file exists;
`text_encoder=` injects any callable with T5EncoderModel's contract (prompt list, device) -> list
of `[len<=512, 4096]` tensors, and `input_prompt` / `n_prompt` may also be pre-computed embeddings.
"""
import gc
import logging
import math
import os
import random
import sys

import torch
import torch.distributed as dist

from .backend import ops
from .modules.model import WanModel
from .modules.vae import WanVAE
from .utils.fm_solvers import FlowDPMSolverMultistepScheduler, get_sampling_sigmas, retrieve_timesteps
from .utils.fm_solvers_unipc import FlowUniPCMultistepScheduler


class WanT2V:

    def __init__(self, config, checkpoint_dir, device_id=0, rank=0, t5_fsdp=False, dit_fsdp=False, use_usp=False,
                 t5_cpu=False, text_encoder=None, model=None, vae=None, cfg_parallel=False, vae_parallel=False, use_ring=False, sp_degrees=None):
        self.device = torch.device(f'cuda:{device_id}')
        self.config = config
        self.rank = rank
        self.t5_cpu = t5_cpu
        self.num_train_timesteps = config.num_train_timesteps
        self.param_dtype = config.param_dtype
        if t5_fsdp or t5_cpu:   # both only trade memory for time in the reference; the outputs are the same
            logging.info('t5_fsdp / t5_cpu: the 9.4 GB bf16 text encoder is kept whole on the GPU (288 GB HBM)')
        t5_path = os.path.join(checkpoint_dir, config.t5_checkpoint) if checkpoint_dir else None
        if text_encoder is None and t5_path and os.path.exists(t5_path):
            from .modules.t5 import T5EncoderModel
            text_encoder = T5EncoderModel(text_len=config.text_len, dtype=config.t5_dtype, device=self.device,
                                          checkpoint_path=t5_path,
                                          tokenizer_path=os.path.join(checkpoint_dir, config.t5_tokenizer))
        self.text_encoder = text_encoder
        self.vae_stride = config.vae_stride
        self.patch_size = config.patch_size
        self.vae = vae if vae is not None else WanVAE(
            vae_pth=os.path.join(checkpoint_dir, config.vae_checkpoint), device=self.device)
        if model is None:
            logging.info(f'Creating WanModel from {checkpoint_dir}')
            model = WanModel.from_pretrained(checkpoint_dir)
        self.model = model
        self.model.eval().requires_grad_(False)
        if dist.is_initialized():
            dist.barrier()
        self.model.to(self.device)
        self.cfgp = None
        self.vae_parallel = bool(vae_parallel) and dist.is_initialized() and dist.get_world_size() > 1
        # vae_parallel: True / 'spatial' = every rank decodes its band of image columns (WanVAE.decode_spatial); 'pipeline' = the layer pipeline
        self.vae_parallel_kind = vae_parallel if vae_parallel in ('spatial', 'pipeline') else 'spatial'
        # flag combinations are validated BEFORE any process group is created
        want_ring = bool(use_ring or (sp_degrees and sp_degrees[1] > 1))
        if cfg_parallel and want_ring:
            raise NotImplementedError('cfg_parallel is built on plain Ulysses groups (no ring / hybrid layout)')
        if cfg_parallel and not (dist.is_initialized() and dist.get_world_size() % 2 == 0):
            logging.warning('cfg_parallel needs an even number of ranks under torch.distributed: ignored, both CFG '
                            'forwards run on every rank')
            cfg_parallel = False
        if cfg_parallel:
            # cond / uncond halves of the ranks, Ulysses inside each half when it has more than one rank
            # (wan/distributed/cfg_parallel.py) — independent of use_usp: 2 ranks need no sequence parallelism at all
            from .distributed.cfg_parallel import enable_cfg_parallel
            self.cfgp = enable_cfg_parallel(self.model)
        elif use_usp and (use_ring or sp_degrees):
            from .distributed.ring import enable_hybrid_sp, enable_ring_attention
            if sp_degrees and sp_degrees[0] > 1 and sp_degrees[1] > 1:      # (ulysses_size, ring_size)
                enable_hybrid_sp(self.model, *sp_degrees)
            elif want_ring:
                enable_ring_attention(self.model)
            else:
                from .distributed.xdit_context_parallel import enable_sequence_parallel
                enable_sequence_parallel(self.model)
        elif use_usp:
            from .distributed.xdit_context_parallel import enable_sequence_parallel
            enable_sequence_parallel(self.model)
        self.sp_size = self.model.sp_size
        if dit_fsdp:
            from .distributed.fsdp import shard_model
            self.model = shard_model(self.model, device_id=device_id)
        self.sample_neg_prompt = config.sample_neg_prompt

    def _encode(self, prompt):
        if torch.is_tensor(prompt):
            return [prompt.to(self.device)]
        if isinstance(prompt, (list, tuple)) and prompt and torch.is_tensor(prompt[0]):
            return [p.to(self.device) for p in prompt]
        if self.text_encoder is None:
            raise FileNotFoundError(
                'no text encoder: config.t5_checkpoint not found in the checkpoint dir; pass pre-computed '
                'umT5 embeddings ([len<=512, 4096] tensors) as `input_prompt`/`n_prompt`, or '
                'WanT2V(text_encoder=callable)')
        return [t.to(self.device) for t in self.text_encoder([prompt], self.device)]

    def generate(self, input_prompt, size=(1280, 720), frame_num=81, shift=5.0, sample_solver='unipc',
                 sampling_steps=50, guide_scale=5.0, n_prompt="", seed=-1, offload_model=True,
                 noise=None, callback=None):
        F = frame_num
        target_shape = (self.vae.model.z_dim, (F - 1) // self.vae_stride[0] + 1, size[1] // self.vae_stride[1],
                        size[0] // self.vae_stride[2])
        seq_len = math.ceil((target_shape[2] * target_shape[3]) / (self.patch_size[1] * self.patch_size[2]) *
                            target_shape[1] / self.sp_size) * self.sp_size
        if isinstance(n_prompt, str) and n_prompt == "":
            n_prompt = self.sample_neg_prompt
        seed = seed if seed >= 0 else random.randint(0, sys.maxsize)
        seed_g = torch.Generator(device=self.device)
        seed_g.manual_seed(seed)
        context = self._encode(input_prompt)
        context_null = self._encode(n_prompt)
        if noise is None:
            noise = torch.randn(*target_shape, dtype=torch.float32, device=self.device, generator=seed_g)
        else:
            noise = noise.to(self.device, torch.float32)
            assert tuple(noise.shape) == tuple(target_shape)

        with torch.no_grad():
            if sample_solver == 'unipc':
                sample_scheduler = FlowUniPCMultistepScheduler(
                    num_train_timesteps=self.num_train_timesteps, shift=1, use_dynamic_shifting=False)
                sample_scheduler.set_timesteps(sampling_steps, device=self.device, shift=shift)
                timesteps = sample_scheduler.timesteps
            elif sample_solver == 'dpm++':
                sample_scheduler = FlowDPMSolverMultistepScheduler(
                    num_train_timesteps=self.num_train_timesteps, shift=1, use_dynamic_shifting=False)
                sampling_sigmas = get_sampling_sigmas(sampling_steps, shift)
                timesteps, _ = retrieve_timesteps(sample_scheduler, device=self.device, sigmas=sampling_sigmas)
            else:
                raise NotImplementedError("Unsupported solver.")

            latent = noise
            timesteps_host = timesteps.tolist()          # ONE sync for the whole loop
            if next(self.model.parameters()).device != self.device:     # only after a real offload (a no-op move would
                self.model.to(self.device)                               # still drop the fused-weight views: 28 GB re-packed)
            noise_pred = torch.empty_like(latent)
            for i, t_host in enumerate(timesteps_host):
                t = timesteps[i:i + 1]
                pair = getattr(self.model, 'forward_pair', None)
                if self.cfgp is None and pair is not None:
                    # both branches in one call: what they share (everything in front of block 0's cross-attention) is computed once,
                    # the results are those of the two calls below bit for bit (WanModel.forward_pair)
                    cond, uncond = pair([latent], t, context, context_null, seq_len)
                    cond, uncond = cond[0], uncond[0]
                elif self.cfgp is None:      # a model object without forward_pair: the reference's two calls (text2video.py:237-240)
                    cond = self.model([latent], t=t, context=context, seq_len=seq_len)[0]
                    uncond = self.model([latent], t=t, context=context_null, seq_len=seq_len)[0]
                else:   # this half's branch only, then swap predictions with the partner rank
                    mine = self.model([latent], t=t, context=context_null if self.cfgp.branch else context,
                                      seq_len=seq_len)[0]
                    cond, uncond = self.cfgp.exchange(mine)
                ops.cfg_combine(noise_pred, uncond, cond, guide_scale)
                latent = sample_scheduler.step(noise_pred.unsqueeze(0), t_host, latent.unsqueeze(0),
                                               return_dict=False, generator=seed_g)[0].squeeze(0)
                if callback is not None:
                    callback(i, latent)
            x0 = [latent]
            offloaded = False
            if offload_model:
                # reference text2video.py:257-259 moves the DiT to the host before the VAE decode to make room on an
                # 80 GB device.  Here that is 28 GB over PCIe and back on the next call, for nothing, whenever the decode
                # fits beside the resident model — which it does on 288 GB: the flag then only drops the DiT's activation
                # workspace; the weights move only when the free memory would not hold 1.25 x the decode's estimated peak
                # (MOVIIGEN_FORCE_OFFLOAD=1 restores the reference's unconditional move), and a decode that still runs out
                # of memory is retried once with the DiT on the host.
                self.model._ws = {}
                torch.cuda.empty_cache()
                free_b, _ = torch.cuda.mem_get_info(self.device)
                need_b = self.vae.decode_peak_bytes(target_shape) if hasattr(self.vae, 'decode_peak_bytes') else 32 << 30
                need_b = need_b * 5 // 4
                if os.environ.get('MOVIIGEN_FORCE_OFFLOAD') == '1':
                    need_b = free_b + 1
                if free_b > need_b:
                    logging.info(f'offload_model: {free_b / 2**30:.0f} GiB free >= {need_b / 2**30:.0f} GiB for the VAE decode '
                                 '-> the DiT stays resident (nothing is moved to the host)')
                else:
                    self.model.cpu()
                    torch.cuda.empty_cache()
                    offloaded = True
            def decode():
                if self.vae_parallel:    # multi-GPU decode over all ranks, video assembled on rank 0: W bands (default), or the layer pipeline
                    out = self.vae.decode_pipelined(x0) if self.vae_parallel_kind == 'pipeline' else self.vae.decode_spatial(x0)
                    return out if self.rank == 0 else None
                return self.vae.decode(x0) if self.rank == 0 else None
            retry = False
            try:
                videos = decode()
            except torch.cuda.OutOfMemoryError:
                if not offload_model or offloaded or self.vae_parallel:      # (pipelined decode: one rank retrying alone would leave its peers in their send / recv)
                    raise
                retry = True       # decided here, done BELOW: inside the handler the live exception's traceback still holds the failed decode's
                #                    frames — its multi-GB fp32 activations — and neither model.cpu() nor empty_cache() could free them
            if retry:
                logging.warning('offload_model: the VAE decode ran out of memory beside the resident DiT -> moving the '
                                'DiT to the host (reference text2video.py:257-259) and decoding again')
                gc.collect()
                self.model.cpu()
                torch.cuda.empty_cache()
                offloaded = True
                videos = decode()
            self.last_offloaded = offloaded

        del noise, latent, sample_scheduler
        if offload_model:
            gc.collect()
            torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        return videos[0] if self.rank == 0 else None
