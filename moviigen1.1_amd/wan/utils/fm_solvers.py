"""FlowDPMSolverMultistepScheduler — DPM-Solver++(2M, midpoint) for flow matching, the `dpm++`
sampler of WanT2V.generate (reference wan/text2video.py:214-223; class at
wan/utils/fm_solvers.py:69-860), plus get_sampling_sigmas / retrieve_timesteps (:22-66).

As for UniPC, each update is one fused linear-combination kernel launch with host-side fp32
scalar coefficients; see fm_solvers_unipc.py for the rationale."""
import numpy as np
import torch

from .fm_solvers_unipc import _hip_lincomb, _lam

__all__ = ['FlowDPMSolverMultistepScheduler', 'get_sampling_sigmas', 'retrieve_timesteps']


def get_sampling_sigmas(sampling_steps, shift):
    sigma = np.linspace(1, 0, sampling_steps + 1)[:sampling_steps]
    return shift * sigma / (1 + (shift - 1) * sigma)


def retrieve_timesteps(scheduler, num_inference_steps=None, device=None, timesteps=None, sigmas=None, **kwargs):
    if timesteps is not None and sigmas is not None:
        raise ValueError('Only one of `timesteps` or `sigmas` can be passed. Please choose one to set custom values')
    if timesteps is not None:
        raise ValueError(f"The current scheduler class {scheduler.__class__}'s `set_timesteps` does not support "
                         'custom timestep schedules.')
    if sigmas is not None:
        scheduler.set_timesteps(sigmas=sigmas, device=device, **kwargs)
    else:
        scheduler.set_timesteps(num_inference_steps, device=device, **kwargs)
    return scheduler.timesteps, len(scheduler.timesteps)


class FlowDPMSolverMultistepScheduler:
    order = 1

    def __init__(self, num_train_timesteps=1000, solver_order=2, prediction_type='flow_prediction', shift=1.0,
                 use_dynamic_shifting=False, thresholding=False, dynamic_thresholding_ratio=0.995,
                 sample_max_value=1.0, algorithm_type='dpmsolver++', solver_type='midpoint',
                 lower_order_final=True, euler_at_final=False, final_sigmas_type='zero',
                 lambda_min_clipped=-float('inf'), variance_type=None, invert_sigmas=False, lincomb=None):
        if (prediction_type != 'flow_prediction' or use_dynamic_shifting or thresholding
                or algorithm_type != 'dpmsolver++' or solver_type != 'midpoint' or final_sigmas_type != 'zero'
                or solver_order not in (1, 2)):
            raise NotImplementedError('only the configuration used by WanT2V.generate is implemented: '
                                      'flow_prediction, dpmsolver++, midpoint, order<=2, final sigma zero')
        self.num_train_timesteps = num_train_timesteps
        self.solver_order = solver_order
        self.shift = shift
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
        self._step_index = None

    def _init_step_index(self, timestep):
        t = int(timestep)
        hits = [i for i, v in enumerate(self._timesteps_host) if v == t]
        if not hits:
            raise ValueError(f'timestep {t} is not in the schedule')
        self._step_index = hits[1] if len(hits) > 1 else hits[0]

    def step(self, model_output, timestep, sample, generator=None, variance_noise=None, return_dict=True):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after "
                             "creating the scheduler")
        if self._step_index is None:
            self._init_step_index(timestep)
        i, n = self._step_index, len(self._timesteps_host)
        x0 = self._lincomb(sample, [(sample, 1.0), (model_output, -self.sigmas[i].item())])
        self.model_outputs = self.model_outputs[1:] + [x0]
        final = i == n - 1
        sig_t, sig_s0 = self.sigmas[i + 1], self.sigmas[i]
        a_t = 1 - sig_t
        h = _lam(sig_t) - _lam(sig_s0)
        c_x = sig_t / sig_s0
        c_d0 = -(a_t * (torch.exp(-h) - 1.0))
        if self.solver_order == 1 or self.lower_order_nums < 1 or final:
            prev = self._lincomb(sample, [(sample, c_x.item()), (x0, c_d0.item())])
        else:
            h0 = _lam(sig_s0) - _lam(self.sigmas[i - 1])
            r0 = h0 / h
            c_d1 = 0.5 * c_d0 * (1.0 / r0)     # D1 = (m0 - m1)/r0
            prev = self._lincomb(sample, [(sample, c_x.item()), (x0, (c_d0 + c_d1).item()),
                                          (self.model_outputs[-2], (-c_d1).item())])
        if self.lower_order_nums < self.solver_order:
            self.lower_order_nums += 1
        self._step_index += 1
        if not return_dict:
            return (prev,)
        return type('SchedulerOutput', (), {'prev_sample': prev})()

    def scale_model_input(self, sample, *args, **kwargs):
        return sample

    def __len__(self):
        return self.num_train_timesteps
