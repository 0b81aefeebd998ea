"""FlowUniPCMultistepScheduler — UniPC (B(h)=expm1, predictor-corrector, order <= 2) for flow
matching, the default sampler of WanT2V.generate (reference wan/text2video.py:206-213).

Same public surface as the reference class (wan/utils/fm_solvers_unipc.py:22-803) for the
configuration the pipeline uses: `set_timesteps(n, device=, shift=)`, `.timesteps`, `.sigmas`,
`.step(model_output, timestep, sample, return_dict=False, generator=None) -> (prev_sample, x0)`.

Execution differs: every UniPC update is a LINEAR COMBINATION of at most four resident tensors
(sample / last_sample / the two stored x0 predictions) with scalar coefficients.  The scalars are
evaluated on the host in fp32 exactly as the reference does (its sigma table is a float32 tensor);
the tensor algebra is ONE fused HIP kernel launch per update (mg_lincomb4_f32) instead of ~15
elementwise torch kernels.  `lincomb` can be injected (tests drive the host logic with numpy).
"""
import numpy as np
import torch

__all__ = ['FlowUniPCMultistepScheduler']


def _hip_lincomb(like, terms):
    from ..backend import ops
    out = torch.empty_like(like)
    return ops.lincomb(out, [(t.contiguous(), c) for t, c in terms])


def _lam(s):
    return torch.log(1 - s) - torch.log(s)


class FlowUniPCMultistepScheduler:
    order = 1

    def __init__(self, num_train_timesteps=1000, solver_order=2, prediction_type='flow_prediction', shift=1.0,
                 use_dynamic_shifting=False, thresholding=False, dynamic_thresholding_ratio=0.995,
                 sample_max_value=1.0, predict_x0=True, solver_type='bh2', lower_order_final=True,
                 disable_corrector=(), solver_p=None, timestep_spacing='linspace', steps_offset=0,
                 final_sigmas_type='zero', lincomb=None):
        if (prediction_type != 'flow_prediction' or use_dynamic_shifting or thresholding or not predict_x0
                or solver_type != 'bh2' or solver_p is not None or final_sigmas_type != 'zero'
                or solver_order not in (1, 2)):
            raise NotImplementedError('only the configuration used by WanT2V.generate is implemented: '
                                      'flow_prediction, predict_x0, bh2, order<=2, final sigma zero')
        self.num_train_timesteps = num_train_timesteps
        self.solver_order = solver_order
        self.shift = shift
        self.lower_order_final = lower_order_final
        self.disable_corrector = list(disable_corrector)
        self._lincomb = lincomb or _hip_lincomb
        alphas = np.linspace(1, 1 / num_train_timesteps, num_train_timesteps)[::-1].copy()
        sig = torch.from_numpy(1.0 - alphas).to(torch.float32)
        sig = shift * sig / (1 + (shift - 1) * sig)
        self.sigmas = sig
        self.timesteps = sig * num_train_timesteps
        self.sigma_min, self.sigma_max = sig[-1].item(), sig[0].item()
        self.num_inference_steps = None
        self._step_index = None

    @property
    def step_index(self):
        return self._step_index

    def set_timesteps(self, num_inference_steps=None, device=None, sigmas=None, mu=None, shift=None):
        if sigmas is None:
            sigmas = np.linspace(self.sigma_max, self.sigma_min, num_inference_steps + 1).copy()[:-1]
        if shift is None:
            shift = self.shift
        sigmas = shift * sigmas / (1 + (shift - 1) * sigmas)
        timesteps = sigmas * self.num_train_timesteps
        self.sigmas = torch.from_numpy(np.concatenate([sigmas, [0]]).astype(np.float32))
        self.timesteps = torch.from_numpy(timesteps).to(device=device, dtype=torch.int64)
        self._timesteps_host = [int(v) for v in timesteps.astype(np.int64)]
        self.num_inference_steps = len(timesteps)
        self.model_outputs = [None] * self.solver_order
        self.lower_order_nums = 0
        self.last_sample = None
        self.this_order = 1
        self._step_index = None

    def _init_step_index(self, timestep):
        t = int(timestep)
        hits = [i for i, v in enumerate(self._timesteps_host) if v == t]
        if not hits:
            raise ValueError(f'timestep {t} is not in the schedule')
        self._step_index = hits[1] if len(hits) > 1 else hits[0]

    # -- coefficient algebra (reference :351-485 predictor, :487-627 corrector) -----------------------
    def _bh(self, order, hh):
        h_phi_1 = torch.expm1(hh)
        B_h = torch.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        fact, b = 1, []
        for i in range(1, order + 1):
            b.append(h_phi_k * fact / B_h)
            fact *= i + 1
            h_phi_k = h_phi_k / hh - 1 / fact
        return h_phi_1, B_h, b

    def _predict(self, x, order):
        i = self._step_index
        sig_t, sig_s0 = self.sigmas[i + 1], self.sigmas[i]
        a_t = 1 - sig_t
        h = _lam(sig_t) - _lam(sig_s0)
        h_phi_1, B_h, _ = self._bh(order, -h)
        m0 = self.model_outputs[-1]
        terms = [(x, (sig_t / sig_s0).item())]
        c_m0 = -a_t * h_phi_1
        if order == 2:
            rk = (_lam(self.sigmas[i - 1]) - _lam(sig_s0)) / h
            c = a_t * B_h * 0.5 / rk          # rhos_p = [0.5]
            c_m0 = c_m0 + c
            terms.append((self.model_outputs[-2], (-c).item()))
        terms.append((m0, c_m0.item()))
        return self._lincomb(x, terms)

    def _correct(self, x0_t, last, order):
        i = self._step_index
        sig_t, sig_s0 = self.sigmas[i], self.sigmas[i - 1]
        a_t = 1 - sig_t
        h = _lam(sig_t) - _lam(sig_s0)
        h_phi_1, B_h, b = self._bh(order, -h)
        m0 = self.model_outputs[-1]
        rks = []
        for j in range(1, order):
            rks.append((_lam(self.sigmas[i - (j + 1)]) - _lam(sig_s0)) / h)
        if order == 1:
            rhos = torch.tensor([0.5])
        else:
            rk_all = torch.stack(rks + [torch.tensor(1.0)])
            R = torch.stack([torch.pow(rk_all, k) for k in range(order)])
            rhos = torch.linalg.solve(R, torch.stack(b))
        terms = [(last, (sig_t / sig_s0).item()), (x0_t, (-a_t * B_h * rhos[-1]).item())]
        c_m0 = -a_t * h_phi_1 + a_t * B_h * rhos[-1]
        for j, rk in enumerate(rks):
            c = a_t * B_h * rhos[j] / rk
            c_m0 = c_m0 + c
            terms.append((self.model_outputs[-(j + 2)], (-c).item()))
        terms.append((m0, c_m0.item()))
        return self._lincomb(last, terms)

    def step(self, model_output, timestep, sample, return_dict=True, generator=None):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after "
                             "creating the scheduler")
        if self._step_index is None:
            self._init_step_index(timestep)
        i = self._step_index
        use_corrector = i > 0 and (i - 1) not in self.disable_corrector and self.last_sample is not None
        x0 = self._lincomb(sample, [(sample, 1.0), (model_output, -self.sigmas[i].item())])  # x - sigma*v
        if use_corrector:
            sample = self._correct(x0, self.last_sample, self.this_order)
        self.model_outputs = self.model_outputs[1:] + [x0]
        this_order = self.solver_order
        if self.lower_order_final:
            this_order = min(self.solver_order, len(self._timesteps_host) - i)
        self.this_order = min(this_order, self.lower_order_nums + 1)
        self.last_sample = sample
        prev = self._predict(sample, self.this_order)
        if self.lower_order_nums < self.solver_order:
            self.lower_order_nums += 1
        self._step_index += 1
        if not return_dict:
            return (prev, x0)
        return type('SchedulerOutput', (), {'prev_sample': prev})()

    def scale_model_input(self, sample, *args, **kwargs):
        return sample

    def __len__(self):
        return self.num_train_timesteps
