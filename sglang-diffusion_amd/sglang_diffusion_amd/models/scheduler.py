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


def shift_sigmas(sigmas: np.ndarray, *, mu: Optional[float], shift: float, kind: str) -> np.ndarray:
    """The schedule warp of the flow-match sampler.  With `mu` (dynamic shifting, what Flux
    uses): sigma' = e^mu / (e^mu + (1/sigma - 1)) ("exponential", scheduling_flow_match_euler_
    discrete.py:628-634) or mu / (mu + (1/sigma - 1)) ("linear", :636-639); without it the
    static form shift * sigma / (1 + (shift - 1) * sigma) (:375-379)."""
    if mu is None:
        return shift * sigmas / (1 + (shift - 1) * sigmas)
    lead = np.exp(mu) if kind == "exponential" else mu
    return lead / (lead + (1 / sigmas - 1) ** 1.0)


class FlowMatchEulerDiscreteScheduler:
    """The part of the reference scheduler the DiT denoise loop uses: a sigma schedule that is
    either handed in (`sigmas=`, the Flux pipeline's linspace(1, 1/N, N)) or the default ramp
    over the training range, warped by `shift_sigmas`, a trailing 0, and the Euler update.
    Options of the reference class that no pipeline on this path sets -- custom `timesteps`,
    `invert_sigmas`, `shift_terminal`, stochastic sampling -- raise instead of being carried
    along untested."""

    order = 1

    def __init__(self, num_train_timesteps: int = 1000, shift: float = 1.0,
                 use_dynamic_shifting: bool = False, base_shift: Optional[float] = 0.5,
                 max_shift: Optional[float] = 1.15, base_image_seq_len: Optional[int] = 256,
                 max_image_seq_len: Optional[int] = 4096, invert_sigmas: bool = False,
                 shift_terminal: Optional[float] = None, time_shift_type: str = "exponential",
                 stochastic_sampling: bool = False):
        if time_shift_type not in {"exponential", "linear"}:
            raise ValueError("`time_shift_type` must either be 'exponential' or 'linear'.")
        for name, on in (("stochastic_sampling", stochastic_sampling), ("invert_sigmas", invert_sigmas),
                         ("shift_terminal", shift_terminal)):
            if on:
                raise NotImplementedError(f"{name} is not on the Flux / HunyuanVideo denoise path")
        self.num_train_timesteps = num_train_timesteps
        self.use_dynamic_shifting = use_dynamic_shifting
        self.time_shift_type = time_shift_type
        self._shift = shift
        self._step_index: Optional[int] = None
        self._begin_index: Optional[int] = None
        # before set_timesteps: the training ramp 1 .. 1/T (statically shifted unless dynamic)
        ramp = np.linspace(1, num_train_timesteps, num_train_timesteps, dtype=np.float32)[::-1].copy()
        sig = torch.from_numpy(ramp) / num_train_timesteps
        if not use_dynamic_shifting:
            sig = shift * sig / (1 + (shift - 1) * sig)
        self.sigmas = sig
        self.timesteps = sig * num_train_timesteps
        self._timesteps_host = self.timesteps
        self.sigma_min, self.sigma_max = sig[-1].item(), sig[0].item()

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

    def time_shift(self, mu, sigma, t):
        if sigma != 1.0:
            raise NotImplementedError("time_shift with an exponent other than 1")
        return shift_sigmas(np.asarray(t, dtype=np.float64) if not isinstance(t, np.ndarray) else t,
                            mu=mu, shift=self._shift, kind=self.time_shift_type)

    def set_timesteps(self, num_inference_steps: Optional[int] = None, device=None,
                      sigmas: Optional[List[float]] = None, mu: Optional[float] = None,
                      timesteps: Optional[List[float]] = None) -> None:
        """scheduling_flow_match_euler_discrete.py:267-411 for the two callers on this path:
        set_timesteps(sigmas=..., mu=...) and set_timesteps(num_inference_steps)."""
        if timesteps is not None:
            raise NotImplementedError("custom `timesteps` are not on the Flux / HunyuanVideo path")
        if self.use_dynamic_shifting and mu is None:
            raise ValueError("`mu` must be passed when `use_dynamic_shifting` is set to be `True`")
        if sigmas is None:
            if num_inference_steps is None:
                raise ValueError("Either num_inference_steps, sigmas, or timesteps must be provided")
            # the default ramp, in timestep units as the reference builds it (float64)
            base = np.linspace(self.sigma_max * self.num_train_timesteps,
                               self.sigma_min * self.num_train_timesteps,
                               num_inference_steps) / self.num_train_timesteps
        else:
            base = np.array(sigmas).astype(np.float32)      # the reference narrows to fp32 here
            if num_inference_steps is not None and len(base) != num_inference_steps:
                raise ValueError("`sigmas` and `timesteps` should have the same length as "
                                 "num_inference_steps, if `num_inference_steps` is provided")
        self.num_inference_steps = len(base)
        warped = shift_sigmas(base, mu=mu if self.use_dynamic_shifting else None,
                              shift=self._shift, kind=self.time_shift_type)
        sig = torch.from_numpy(np.asarray(warped)).to(dtype=torch.float32)
        ts = sig * self.num_train_timesteps
        # sigmas stay on the host (they parameterise kernel launches); timesteps go to `device`
        self.sigmas = torch.cat([sig, torch.zeros(1)])
        self._timesteps_host = ts
        self.timesteps = ts.to(device) if device is not None else ts
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
