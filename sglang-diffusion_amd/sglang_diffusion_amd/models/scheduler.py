"""Flow-match Euler scheduler.

Mirrors python/sglang/multimodal_gen/runtime/models/schedulers/
scheduling_flow_match_euler_discrete.py:98-154 (constructor), :267-411 (set_timesteps),
:439-536 (step) and :628-639 (time shift); `calculate_shift`/`prepare_mu` mirror
runtime/pipelines/flux.py:22-51 and `prepare_sigmas` configs/pipeline_configs/base.py:741-747.
The schedule is host arithmetic (numpy, as in the reference); the latent update is one HIP
kernel.
"""
from __future__ import annotations

import math
from typing import List, Optional

import numpy as np
import torch

from .. import _C


def calculate_shift(image_seq_len, base_seq_len: int = 256, max_seq_len: int = 4096,
                    base_shift: float = 0.5, max_shift: float = 1.15):
    m = (max_shift - base_shift) / (max_seq_len - base_seq_len)
    b = base_shift - m * base_seq_len
    return image_seq_len * m + b


def prepare_mu(height: int, width: int, vae_scale_factor: int = 8) -> float:
    """runtime/pipelines/flux.py:35-51: NOTE image_seq_len = (H/8)*(W/8), not the packed
    (H/16)*(W/16) token count."""
    image_seq_len = (int(height) // vae_scale_factor) * (int(width) // vae_scale_factor)
    return calculate_shift(image_seq_len, 256, 4096, 0.5, 1.15)


def prepare_sigmas(num_inference_steps: int) -> np.ndarray:
    """configs/pipeline_configs/base.py:741-747: linspace(1, 1/N, N)."""
    return np.linspace(1.0, 1 / num_inference_steps, num_inference_steps)


class FlowMatchEulerDiscreteScheduler:
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, shift: float = 1.0,
                 use_dynamic_shifting: bool = False, base_shift: Optional[float] = 0.5,
                 max_shift: Optional[float] = 1.15, base_image_seq_len: Optional[int] = 256,
                 max_image_seq_len: Optional[int] = 4096, invert_sigmas: bool = False,
                 shift_terminal: Optional[float] = None, time_shift_type: str = "exponential",
                 stochastic_sampling: bool = False):
        if time_shift_type not in {"exponential", "linear"}:
            raise ValueError("`time_shift_type` must either be 'exponential' or 'linear'.")
        if stochastic_sampling:
            raise NotImplementedError("stochastic sampling is not on the Flux path")
        self.num_train_timesteps = num_train_timesteps
        self.use_dynamic_shifting = use_dynamic_shifting
        self.invert_sigmas = invert_sigmas
        self.shift_terminal = shift_terminal
        self.time_shift_type = time_shift_type
        timesteps = np.linspace(1, num_train_timesteps, num_train_timesteps,
                                dtype=np.float32)[::-1].copy()
        timesteps = torch.from_numpy(timesteps).to(dtype=torch.float32)
        sigmas = timesteps / num_train_timesteps
        if not use_dynamic_shifting:
            sigmas = shift * sigmas / (1 + (shift - 1) * sigmas)
        self.timesteps = sigmas * num_train_timesteps
        self._step_index: Optional[int] = None
        self._begin_index: Optional[int] = None
        self._shift = shift
        self.sigmas = sigmas.to("cpu")
        self.sigma_min = self.sigmas[-1].item()
        self.sigma_max = self.sigmas[0].item()

    @property
    def shift(self) -> float:
        return self._shift

    @property
    def step_index(self) -> Optional[int]:
        return self._step_index

    @property
    def begin_index(self) -> Optional[int]:
        return self._begin_index

    def set_begin_index(self, begin_index: int = 0) -> None:
        self._begin_index = begin_index

    def _sigma_to_t(self, sigma: float) -> float:
        return sigma * self.num_train_timesteps

    def _time_shift_exponential(self, mu, sigma, t):
        if isinstance(t, np.ndarray):
            return np.exp(mu) / (np.exp(mu) + (1 / t - 1) ** sigma)
        return math.exp(mu) / (math.exp(mu) + (1 / t - 1) ** sigma)

    def _time_shift_linear(self, mu, sigma, t):
        return mu / (mu + (1 / t - 1) ** sigma)

    def time_shift(self, mu, sigma, t):
        if self.time_shift_type == "exponential":
            return self._time_shift_exponential(mu, sigma, t)
        return self._time_shift_linear(mu, sigma, t)

    def set_timesteps(self, num_inference_steps: Optional[int] = None, device=None,
                      sigmas: Optional[List[float]] = None, mu: Optional[float] = None,
                      timesteps: Optional[List[float]] = None) -> None:
        if self.use_dynamic_shifting and mu is None:
            raise ValueError("`mu` must be passed when `use_dynamic_shifting` is set to be `True`")
        if sigmas is not None and timesteps is not None and len(sigmas) != len(timesteps):
            raise ValueError("`sigmas` and `timesteps` should have the same length")
        if num_inference_steps is not None:
            if (sigmas is not None and len(sigmas) != num_inference_steps) or (
                    timesteps is not None and len(timesteps) != num_inference_steps):
                raise ValueError("`sigmas` and `timesteps` should have the same length as "
                                 "num_inference_steps, if `num_inference_steps` is provided")
        else:
            if sigmas is not None:
                num_inference_steps = len(sigmas)
            elif timesteps is not None:
                num_inference_steps = len(timesteps)
            else:
                raise ValueError("Either num_inference_steps, sigmas, or timesteps must be provided")
        self.num_inference_steps = num_inference_steps
        is_timesteps_provided = timesteps is not None
        timesteps_array = np.array(timesteps).astype(np.float32) if is_timesteps_provided else None
        if sigmas is None:
            if timesteps_array is None:
                timesteps_array = np.linspace(self._sigma_to_t(self.sigma_max),
                                              self._sigma_to_t(self.sigma_min),
                                              num_inference_steps)
            sigmas_array = timesteps_array / self.num_train_timesteps
        else:
            sigmas_array = np.array(sigmas).astype(np.float32)
            num_inference_steps = len(sigmas_array)
        if self.use_dynamic_shifting:
            sigmas_array = self.time_shift(mu, 1.0, sigmas_array)
        else:
            sigmas_array = self.shift * sigmas_array / (1 + (self.shift - 1) * sigmas_array)
        if self.shift_terminal:
            one_minus_z = 1 - sigmas_array
            scale_factor = one_minus_z[-1] / (1 - self.shift_terminal)
            sigmas_array = 1 - (one_minus_z / scale_factor)
        sigmas_tensor = torch.from_numpy(np.asarray(sigmas_array)).to(dtype=torch.float32)
        if not is_timesteps_provided:
            timesteps_tensor = sigmas_tensor * self.num_train_timesteps
        else:
            timesteps_tensor = torch.from_numpy(timesteps_array).to(dtype=torch.float32)
        if self.invert_sigmas:
            sigmas_tensor = 1.0 - sigmas_tensor
            timesteps_tensor = sigmas_tensor * self.num_train_timesteps
            sigmas_tensor = torch.cat([sigmas_tensor, torch.ones(1)])
        else:
            sigmas_tensor = torch.cat([sigmas_tensor, torch.zeros(1)])
        # sigmas stay on the host (they parameterise kernel launches); timesteps go to `device`
        self.sigmas = sigmas_tensor
        self.timesteps = timesteps_tensor.to(device) if device is not None else timesteps_tensor
        self._timesteps_host = timesteps_tensor
        self._step_index = None
        self._begin_index = None

    def index_for_timestep(self, timestep, schedule_timesteps=None) -> int:
        st = self._timesteps_host if schedule_timesteps is None else schedule_timesteps
        t = float(timestep)
        indices = (st == t).nonzero()
        pos = 1 if len(indices) > 1 else 0
        return indices[pos].item()

    def _init_step_index(self, timestep) -> None:
        if self.begin_index is None:
            self._step_index = self.index_for_timestep(timestep)
        else:
            self._step_index = self._begin_index

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor,
             return_dict: bool = False, out: Optional[torch.Tensor] = None):
        """prev = (sample.float() + dt * model_output).to(model_output.dtype), dt = sigma_next -
        sigma (scheduling_flow_match_euler_discrete.py:495-531)."""
        if isinstance(timestep, int) or (isinstance(timestep, torch.Tensor)
                                         and not timestep.is_floating_point()):
            raise ValueError("Passing integer indices (e.g. from `enumerate(timesteps)`) as "
                             "timesteps to `FlowMatchEulerDiscreteScheduler.step()` is not "
                             "supported. Make sure to pass one of the `scheduler.timesteps` as a "
                             "timestep.")
        if self.step_index is None:
            self._init_step_index(timestep)
        i = self._step_index
        sigma, sigma_next = self.sigmas[i].item(), self.sigmas[i + 1].item()
        if sample.dtype != torch.bfloat16 or model_output.dtype != torch.bfloat16:
            raise TypeError("euler step expects bf16 latents and model output")
        if out is None:
            out = torch.empty_like(model_output)
        _C.euler_step(out, sample.contiguous(), model_output.contiguous(), sigma, sigma_next)
        self._step_index += 1
        return (out,)
