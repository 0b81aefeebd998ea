"""ORACLE — CPU restatement of the two flow-matching samplers (test infrastructure, NOT product).

  wan/utils/fm_solvers_unipc.py:77-132   __init__ (training sigma table, sigma_min/max)
  wan/utils/fm_solvers_unipc.py:160-227  set_timesteps (linspace -> shift -> int64 timesteps)
  wan/utils/fm_solvers_unipc.py:279-349  convert_model_output (x0 = x - sigma * v)
  wan/utils/fm_solvers_unipc.py:351-485  UniP predictor (bh2), :487-627 UniC corrector
  wan/utils/fm_solvers_unipc.py:656-742  step (order warm-up 1->2, lower_order_final)
  wan/utils/fm_solvers.py:22-27          get_sampling_sigmas
  wan/utils/fm_solvers.py:226-289        set_timesteps(sigmas=...)
  wan/utils/fm_solvers.py:415-470,472-541,706-797  DPM-Solver++ (2M, midpoint) first/second order, step

Scalars are torch fp32 0-d tensors exactly as in the reference (its sigma table is a float32
tensor); the reference's stray print()s are not reproduced.  Only the configuration the pipeline
uses is restated: flow_prediction, predict_x0, solver_order 2, bh2 / midpoint, final sigma 0.
"""
import numpy as np
import torch


def _train_sigma_range(num_train_timesteps, shift):
    alphas = np.linspace(1, 1 / num_train_timesteps, num_train_timesteps)[::-1].copy()
    sig = torch.from_numpy(1.0 - alphas).to(torch.float32)
    sig = shift * sig / (1 + (shift - 1) * sig)
    return sig[0].item(), sig[-1].item()


def _schedule(sigmas64, num_train_timesteps):
    timesteps = torch.from_numpy(sigmas64 * num_train_timesteps).to(torch.int64)
    sig = torch.from_numpy(np.concatenate([sigmas64, [0]]).astype(np.float32))
    return timesteps, sig


def _lam(sigma):
    return torch.log(1 - sigma) - torch.log(sigma)


class UniPCOracle:
    def __init__(self, num_train_timesteps=1000, shift=1.0, solver_order=2):
        self.N = num_train_timesteps
        self.order = solver_order
        self.sigma_max, self.sigma_min = _train_sigma_range(num_train_timesteps, shift)
        self.cfg_shift = shift

    def set_timesteps(self, n, shift=None):
        s = np.linspace(self.sigma_max, self.sigma_min, n + 1).copy()[:-1]
        sh = self.cfg_shift if shift is None else shift
        s = sh * s / (1 + (sh - 1) * s)
        self.timesteps, self.sigmas = _schedule(s, self.N)
        self.m = [None] * self.order  # x0 predictions, newest last
        self.lower_order_nums = 0
        self.last_sample = None
        self.i = 0
        self.this_order = 1
        return self.timesteps

    def _coeffs(self, order, hh):
        """R, b of the B(h) system (shared by predictor and corrector)."""
        h_phi_1 = torch.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        B_h = torch.expm1(hh)
        fact = 1
        b = []
        for i in range(1, order + 1):
            b.append(h_phi_k * fact / B_h)
            fact *= i + 1
            h_phi_k = h_phi_k / hh - 1 / fact
        return h_phi_1, B_h, b

    def _predict(self, x, order):
        i = self.i
        sig_t, sig_s0 = self.sigmas[i + 1], self.sigmas[i]
        a_t = 1 - sig_t
        h = _lam(sig_t) - _lam(sig_s0)
        m0 = self.m[-1]
        rks, D1s = [], []
        for j in range(1, order):
            rk = (_lam(self.sigmas[i - j]) - _lam(sig_s0)) / h
            rks.append(rk)
            D1s.append((self.m[-(j + 1)] - m0) / rk)
        hh = -h
        h_phi_1, B_h, _ = self._coeffs(order, hh)
        x_t = sig_t / sig_s0 * x - a_t * h_phi_1 * m0
        if D1s:
            assert order == 2  # rhos_p = [0.5]
            x_t = x_t - a_t * B_h * (0.5 * D1s[0])
        return x_t.to(x.dtype)

    def _correct(self, model_t, last, order):
        i = self.i
        sig_t, sig_s0 = self.sigmas[i], self.sigmas[i - 1]
        a_t = 1 - sig_t
        h = _lam(sig_t) - _lam(sig_s0)
        m0 = self.m[-1]
        rks, D1s = [], []
        for j in range(1, order):
            rk = (_lam(self.sigmas[i - (j + 1)]) - _lam(sig_s0)) / h
            rks.append(rk)
            D1s.append((self.m[-(j + 1)] - m0) / rk)
        rks.append(1.0)
        rks = torch.tensor(rks)
        hh = -h
        h_phi_1, B_h, b = self._coeffs(order, hh)
        if order == 1:
            rhos = torch.tensor([0.5])
        else:
            R = torch.stack([torch.pow(rks, k) for k in range(order)])
            rhos = torch.linalg.solve(R, torch.tensor(b)).to(last.dtype)
        x_t = sig_t / sig_s0 * last - a_t * h_phi_1 * m0
        corr = 0
        for j, d in enumerate(D1s):
            corr = corr + rhos[j] * d
        x_t = x_t - a_t * B_h * (corr + rhos[-1] * (model_t - m0))
        return x_t.to(last.dtype)

    def step(self, v, sample):
        i = self.i
        x0 = sample - self.sigmas[i] * v
        if i > 0 and self.last_sample is not None:
            sample = self._correct(x0, self.last_sample, self.this_order)
        self.m = self.m[1:] + [x0]
        this_order = min(self.order, len(self.timesteps) - i)
        self.this_order = min(this_order, self.lower_order_nums + 1)
        self.last_sample = sample
        prev = self._predict(sample, self.this_order)
        if self.lower_order_nums < self.order:
            self.lower_order_nums += 1
        self.i += 1
        return prev


class DPMppOracle:
    def __init__(self, num_train_timesteps=1000, solver_order=2):
        self.N = num_train_timesteps
        self.order = solver_order

    @staticmethod
    def sampling_sigmas(steps, shift):
        s = np.linspace(1, 0, steps + 1)[:steps]
        return shift * s / (1 + (shift - 1) * s)

    def set_timesteps(self, n, shift):
        self.timesteps, self.sigmas = _schedule(self.sampling_sigmas(n, shift), self.N)
        self.m = [None] * self.order
        self.lower_order_nums = 0
        self.i = 0
        return self.timesteps

    def step(self, v, sample):
        i, n = self.i, len(self.timesteps)
        x0 = sample - self.sigmas[i] * v
        self.m = self.m[1:] + [x0]
        final = (i == n - 1)  # final_sigmas_type == "zero"
        sig_t, sig_s0 = self.sigmas[i + 1], self.sigmas[i]
        a_t = 1 - sig_t
        h = _lam(sig_t) - _lam(sig_s0)
        if self.lower_order_nums < 1 or final:
            prev = (sig_t / sig_s0) * sample - (a_t * (torch.exp(-h) - 1.0)) * x0
        else:
            sig_s1 = self.sigmas[i - 1]
            h0 = _lam(sig_s0) - _lam(sig_s1)
            r0 = h0 / h
            D0, D1 = self.m[-1], (1.0 / r0) * (self.m[-1] - self.m[-2])
            prev = ((sig_t / sig_s0) * sample - (a_t * (torch.exp(-h) - 1.0)) * D0
                    - 0.5 * (a_t * (torch.exp(-h) - 1.0)) * D1)
        if self.lower_order_nums < self.order:
            self.lower_order_nums += 1
        self.i += 1
        return prev.to(x0.dtype)


def sample_loop(model_fn, noise, ctx, ctx_null, steps, shift, guide_scale, solver='unipc'):
    """The denoise loop of wan/text2video.py:204-254 (one video): two model calls per step, CFG,
    scheduler step.  model_fn(latent, t, ctx) -> v.  Returns (x0 latent, per-step latents)."""
    if solver == 'unipc':
        sch = UniPCOracle(shift=1.0)
        ts = sch.set_timesteps(steps, shift=shift)
    elif solver == 'dpm++':
        sch = DPMppOracle()
        ts = sch.set_timesteps(steps, shift)
    else:
        raise NotImplementedError("Unsupported solver.")
    lat = noise
    traj = []
    for t in ts:
        vc = model_fn(lat, t, ctx)
        vu = model_fn(lat, t, ctx_null)
        v = vu + guide_scale * (vc - vu)
        lat = sch.step(v[None], lat[None])[0]
        traj.append(lat.clone())
    return lat, traj
