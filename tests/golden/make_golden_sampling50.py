"""Generate tests/golden/g9_sampling50.npz by IMPORTING the reference: the PRODUCTION sampling setting
(N = 50 steps, shift 5.0, guide scale 5.0 — the defaults of wan/text2video.py:114-124, tables built by
wan/utils/fm_solvers_unipc.py:160-227 and wan/utils/fm_solvers.py:22-66,226-289).

    python tests/golden/make_golden_sampling50.py

Two things are captured, both with the reference's own scheduler classes:
  (1) 50-step UniPC and DPM++ trajectories of a synthetic velocity v(x, t) — every step is stored, so the order ramp-up
      (step 1 first order) and `lower_order_final` (steps 49/50) are pinned (fm_solvers_unipc.py:656-742,
      fm_solvers.py:706-797);
  (2) the 50-step loop body of wan/text2video.py:228-254 (two WanModel forwards, u + 5.0 (c - u), scheduler.step) on the
      small head-dim-128 DiT of tests/golden/weights.py, in fp32 and under the bf16-autocast emulation of make_golden.py,
      the latent after steps 1, 10, 20, 30, 40, 50, and the video WanVAE_(dim=8) decodes from the fp32 final latent.
The bf16-emulated run is the reference's own drift from its fp32 arithmetic over 50 steps: the scale the product's
end-to-end tolerance is stated against.  A fixture is data; weights are regenerated from seeds by weights.py.
"""
import contextlib
import importlib
import io
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402
import weights as W  # noqa: E402

N, SHIFT, GUIDE = 50, 5.0, 5.0
KEEP = (1, 10, 20, 30, 40, 50)


def make_scheduler(unipc, dpm, solver):
    if solver == 'unipc':
        s = unipc.FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
        s.set_timesteps(N, device='cpu', shift=SHIFT)
        return s, s.timesteps
    s = dpm.FlowDPMSolverMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
    ts, _ = dpm.retrieve_timesteps(s, device='cpu', sigmas=dpm.get_sampling_sigmas(N, SHIFT))
    return s, ts


@torch.no_grad()
def main():
    torch.manual_seed(0)
    model, vae, unipc, dpm = G.load_ref()
    sink = io.StringIO()
    arrs = {}

    # ---- (1) synthetic-velocity trajectories, all 50 steps -----------------------------------------------------
    x0 = W.randn((1, 16, 2, 4, 4), 50)
    arrs['traj_x0'] = x0
    for name in ('unipc', 'dpm'):
        s, ts = make_scheduler(unipc, dpm, name)
        arrs[f'{name}_t'] = ts
        lat, traj = x0.clone(), []
        with contextlib.redirect_stdout(sink):
            for t in ts:
                v = 0.5 * torch.tanh(lat) + 0.1 * torch.sin(t.float() / 100.0)
                lat = s.step(v, t, lat, return_dict=False)[0]
                traj.append(lat.clone())
        arrs[f'traj_{name}'] = torch.stack(traj)

    # ---- (2) the generate loop, 50 steps, CFG 5.0 --------------------------------------------------------------
    cfg = W.SMALL_DIT_HD128
    m = G.build_ref_dit(model, cfg, W.make_dit_params(cfg, 0))
    noise = W.randn((16, 2, 8, 12), 80)
    ctx, ctxn = W.randn((33, cfg['text_dim']), 81), W.randn((9, cfg['text_dim']), 82)
    arrs.update(noise=noise, ctx=ctx, ctx_null=ctxn, keep=torch.tensor(KEEP))
    amp32 = importlib.import_module('torch.amp')
    for solver in ('unipc', 'dpm++'):
        for mode in ('fp32', 'bf16'):
            s, ts = make_scheduler(unipc, dpm, 'unipc' if solver == 'unipc' else 'dpm')
            model.amp = G.AmpCpu if mode == 'bf16' else amp32
            cm = torch.autocast('cpu', dtype=torch.bfloat16) if mode == 'bf16' else contextlib.nullcontext()
            lat, kept = [noise], []
            with contextlib.redirect_stdout(sink), cm:
                for i, t in enumerate(ts):          # the loop body of wan/text2video.py:233-254
                    tt = torch.stack([t])
                    c = m(lat, t=tt, context=[ctx], seq_len=48)[0]
                    u = m(lat, t=tt, context=[ctxn], seq_len=48)[0]
                    v = u + GUIDE * (c - u)
                    lat = [s.step(v.unsqueeze(0), t, lat[0].unsqueeze(0), return_dict=False)[0].squeeze(0)]
                    if i + 1 in KEEP:
                        kept.append(lat[0].float().clone())
            model.amp = amp32
            arrs[f'lat_{solver}_{mode}'] = torch.stack(kept)
    Pv = W.make_vae_params(8, 1)
    ref = vae.WanVAE_(dim=8, z_dim=16, dim_mult=[1, 2, 4, 4], num_res_blocks=2, attn_scales=[],
                      temperal_downsample=[False, True, True], dropout=0.0)
    sd = ref.state_dict()
    sd.update({k: v.reshape(sd[k].shape) for k, v in Pv.items()})
    ref.load_state_dict(sd)
    mean = torch.tensor([-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508,
                         0.4134, -0.0715, 0.5517, -0.3632, -0.1922, -0.9497, 0.2503, -0.2921])
    std = torch.tensor([2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743,
                        3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253, 2.8251, 1.9160])
    arrs['video_unipc_fp32'] = ref.eval().decode(arrs['lat_unipc_fp32'][-1][None],
                                                 [mean, 1.0 / std]).float().clamp_(-1, 1)[0]
    G.save('g9_sampling50', **arrs)
    for solver in ('unipc', 'dpm++'):
        a, b = arrs[f'lat_{solver}_fp32'], arrs[f'lat_{solver}_bf16']
        print(solver, 'reference bf16-vs-fp32 rel-L2 per kept step:',
              [f'{((a[i] - b[i]).norm() / a[i].norm()).item():.2e}' for i in range(len(KEEP))])


if __name__ == '__main__':
    main()
