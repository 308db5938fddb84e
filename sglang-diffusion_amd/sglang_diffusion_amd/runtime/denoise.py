"""The denoising loop of the Flux pipeline on MI355X.

Mirrors python/sglang/multimodal_gen/runtime/pipelines_core/stages/denoising.py:966-1127
(DenoisingStage.forward), :340-350 (_build_guidance), :885-936 (timestep expansion) and
:1350-1507 (_predict_noise_with_cfg, serial CFG), plus stages/timestep_preparation.py:58-138
and stages/latent_preparation.py:38-105 for the inputs.  A `Req` carries one batch of samples
that share a shape and a step count.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from functools import lru_cache
from typing import Any, List, Optional

import torch

from .. import _C
from ..layers.attention import Mi355AttentionMetadataBuilder
from ..layers.forward_context import set_forward_context
from ..models.scheduler import (FlowMatchEulerDiscreteScheduler, prepare_mu, prepare_sigmas)
from . import pipeline_config as pc


@dataclass
class Req:
    """The fields of pipelines_core/schedule_batch.py:42- that the denoise stage reads."""

    latents: torch.Tensor                      # [B, tokens, C] packed, bf16
    prompt_embeds: torch.Tensor                # [B, 512, 4096]
    pooled_embeds: torch.Tensor                # [B, 768]
    negative_prompt_embeds: Optional[torch.Tensor] = None
    neg_pooled_embeds: Optional[torch.Tensor] = None
    height: int = 1024
    width: int = 1024
    num_inference_steps: int = 50
    guidance_scale: float = 1.0                # true CFG scale (Flux default 1.0 = off)
    embedded_cfg_scale: float = pc.EMBEDDED_CFG_SCALE
    cfg_normalization: float = 0.0             # > 0: cap ||noise|| at this factor * ||cond||
    guidance_rescale: float = 0.0              # > 0: std rescale of the combined prediction
    request_ids: List[Any] = field(default_factory=list)

    @property
    def do_classifier_free_guidance(self) -> bool:
        return self.guidance_scale > 1.0 and self.negative_prompt_embeds is not None

    @property
    def batch_size(self) -> int:
        return self.latents.shape[0]


class DenoisingStage:
    def __init__(self, transformer, scheduler: Optional[FlowMatchEulerDiscreteScheduler] = None,
                 fused: bool = True, cfg_batched: bool = True, cfg_group=None, sync_group=None,
                 value_half: Optional[bool] = None, half_fallback_limit: int = 2):
        """cfg_group: a 2-rank process group over which the conditional (rank 0) and the
        unconditional (rank 1) branch of true CFG are split (the reference's
        enable_cfg_parallel, pipelines_core/stages/denoising.py:1388-1446); None = both
        branches on this rank.  sync_group: any further group whose ranks run collectives with
        this one inside a step and therefore must take the half / bf16 V decision together.
        value_half: None follows the process default (_C.VALUE_HALF), False pins bf16 V."""
        self.transformer = transformer
        self.cfg_group = cfg_group
        self.sync_group = sync_group
        self.value_half = value_half
        self.half_fallback_limit = max(1, int(half_fallback_limit))
        self._half_failures = 0
        self._range_host = {}    # device -> pinned int64[2]
        self._range_event = {}   # device -> event behind the last snapshot
        # Flux ships use_dynamic_shifting=True in its scheduler config
        self.scheduler = scheduler or FlowMatchEulerDiscreteScheduler(use_dynamic_shifting=True)
        self.fused = fused
        self.cfg_batched = cfg_batched
        self.attn_metadata_builder = Mi355AttentionMetadataBuilder()
        self._freqs = {}
        self._ts_dev = {}  # (steps, height, width, device) -> (device timesteps, host copy)
        self._bufs = {}  # per-shape scratch and staging tensors: a step allocates nothing

    def _buf(self, name: str, shape, dtype, device) -> torch.Tensor:
        key = (name, tuple(shape), dtype, str(device))
        t = self._bufs.get(key)
        if t is None:
            t = self._bufs[key] = torch.empty(tuple(shape), dtype=dtype, device=device)
        return t

    def cfg_batched_inputs(self, batch: Req, guidance):
        """The step-invariant operands of the batched cond|uncond forward (text states, pooled
        vectors, guidance), concatenated once per request batch instead of once per step."""
        return dict(
            encoder_hidden_states=torch.cat([batch.prompt_embeds, batch.negative_prompt_embeds], 0),
            pooled_projections=torch.cat([batch.pooled_embeds, batch.neg_pooled_embeds], 0),
            guidance=None if guidance is None else torch.cat([guidance, guidance], 0))

    @staticmethod
    @lru_cache(maxsize=8)
    def _build_guidance(batch_size, device, guidance_val):
        """denoising.py:340-350: full(B, g, fp32).to(bf16) * 1000 (stays bf16)."""
        return torch.full((batch_size,), guidance_val, dtype=torch.float32,
                          device=device).to(torch.bfloat16) * 1000.0

    def _freqs_cis(self, text_len, width, height, device):
        key = (text_len, width, height, str(device))
        if key not in self._freqs:
            self._freqs[key] = pc.get_freqs_cis(text_len, width, height, device,
                                                self.transformer.rotary_emb)
        return self._freqs[key]

    def prepare_timesteps(self, batch: Req, device):
        """timestep_preparation.py:58-138 with the Flux extras: sigmas = linspace(1, 1/N, N),
        mu from the image size."""
        sigmas = prepare_sigmas(batch.num_inference_steps)
        mu = prepare_mu(batch.height, batch.width, pc.VAE_SCALE_FACTOR)
        # the schedule is built on the host; its device copy is kept per (steps, size, device),
        # so a batch that is being captured into a HIP graph (where a host-to-device copy from
        # pageable memory is not allowed) finds the copy its warm-up run left behind
        self.scheduler.set_timesteps(sigmas=list(sigmas), mu=mu, device=None)
        if torch.device(device).type == "cpu":
            return self.scheduler.timesteps
        key = (batch.num_inference_steps, batch.height, batch.width, str(device))
        ts = self._ts_dev.get(key)
        if ts is None or not torch.equal(ts[1], self.scheduler.timesteps):
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("DenoisingStage: run a batch of this shape once outside the "
                                   "graph capture first (its timestep table has to be on the device)")
            ts = self._ts_dev[key] = (self.scheduler.timesteps.to(device),
                                      self.scheduler.timesteps.clone())
        self.scheduler.timesteps = ts[0]
        return ts[0]

    def _dit(self, **kw):
        return self.transformer.forward_fused(**kw) if self.fused else self.transformer(**kw)

    def predict_noise_with_cfg(self, batch: Req, latent_model_input, timestep, guidance, freqs,
                               step_index: int, cfg_inputs=None):
        md = self.attn_metadata_builder.build(current_timestep=step_index)
        with set_forward_context(current_timestep=step_index, attn_metadata=md,
                                 forward_batch=batch):
            if not batch.do_classifier_free_guidance:
                out = self._dit(hidden_states=latent_model_input,
                                encoder_hidden_states=batch.prompt_embeds,
                                pooled_projections=batch.pooled_embeds, timestep=timestep,
                                guidance=guidance, freqs_cis=freqs)
                # the engine's output buffer is overwritten by its next call: hand out a copy
                # held in a per-shape buffer of the stage
                return self._buf("noise", out.shape, out.dtype, out.device).copy_(out)
            if self.cfg_group is not None:
                import torch.distributed as dist

                from .parallel import cfg_parallel_combine

                neg = dist.get_rank(self.cfg_group) != 0
                mine = self._dit(
                    hidden_states=latent_model_input,
                    encoder_hidden_states=batch.negative_prompt_embeds if neg else batch.prompt_embeds,
                    pooled_projections=batch.neg_pooled_embeds if neg else batch.pooled_embeds,
                    timestep=timestep, guidance=guidance, freqs_cis=freqs).clone()
                noise_pred = cfg_parallel_combine(mine, float(batch.guidance_scale), self.cfg_group)
                if self._wants_post_cfg(batch):
                    # the reference broadcasts rank 0's norm / std scalars (denoising.py:
                    # 1448-1478); here rank 0's conditional prediction itself travels (2 MB), so
                    # both ranks run the same two kernels on the same operands
                    from .parallel import _wire

                    cond = mine if not neg else torch.empty_like(mine)
                    wire = _wire(cond, self.cfg_group)   # host-staged under gloo
                    dist.broadcast(wire, src=dist.get_global_rank(self.cfg_group, 0),
                                   group=self.cfg_group)
                    if wire is not cond:
                        cond.copy_(wire)
                    self._post_cfg(batch, noise_pred, cond)
                return noise_pred
            if self.cfg_batched:
                # cond and uncond of every request ride in ONE forward of batch 2B: rows of a
                # batch never interact inside the DiT, so this equals two separate passes.
                if cfg_inputs is None:
                    cfg_inputs = self.cfg_batched_inputs(batch, guidance)
                B = latent_model_input.shape[0]
                x2 = self._buf("cfg_x", (2 * B,) + tuple(latent_model_input.shape[1:]),
                               latent_model_input.dtype, latent_model_input.device)
                x2[:B].copy_(latent_model_input)
                x2[B:].copy_(latent_model_input)
                t2 = self._buf("cfg_t", (2 * B,), timestep.dtype, timestep.device)
                t2[:B].copy_(timestep)
                t2[B:].copy_(timestep)
                out = self._dit(hidden_states=x2, timestep=t2, freqs_cis=freqs, **cfg_inputs)
                cond, uncond = out[:B], out[B:]          # contiguous halves of the engine's output
            else:
                cond = self._dit(hidden_states=latent_model_input,
                                 encoder_hidden_states=batch.prompt_embeds,
                                 pooled_projections=batch.pooled_embeds, timestep=timestep,
                                 guidance=guidance, freqs_cis=freqs).clone()
                uncond = self._dit(hidden_states=latent_model_input,
                                   encoder_hidden_states=batch.negative_prompt_embeds,
                                   pooled_projections=batch.neg_pooled_embeds, timestep=timestep,
                                   guidance=guidance, freqs_cis=freqs).clone()
        noise_pred = self._buf("cfg_noise", cond.shape, cond.dtype, cond.device)
        _C.cfg_combine(noise_pred, cond, uncond, float(batch.guidance_scale))
        self._post_cfg(batch, noise_pred, cond)
        return noise_pred

    @staticmethod
    def _wants_post_cfg(batch: Req) -> bool:
        return bool(batch.cfg_normalization and float(batch.cfg_normalization) > 0) or \
            batch.guidance_rescale > 0.0

    def _post_cfg(self, batch: Req, noise_pred, cond) -> None:
        """denoising.py:1490-1506, in place on noise_pred; the moment scratch is kept per batch
        size."""
        scratch = None
        if self._wants_post_cfg(batch) and noise_pred.is_cuda:
            scratch = self._buf("cfg_moments", (256 * max(1, noise_pred.shape[0]),), torch.float32,
                                noise_pred.device)
        if batch.cfg_normalization and float(batch.cfg_normalization) > 0:
            _C.cfg_normalize(noise_pred, cond, float(batch.cfg_normalization), scratch)
        if batch.guidance_rescale > 0.0:
            _C.guidance_rescale(noise_pred, cond, float(batch.guidance_rescale), scratch)

    # ---- the half-format V operand and its range guard -------------------------------------
    def _range_buffer(self, device) -> torch.Tensor:
        """Pinned host int64[2] that receives {overflow_blocks, tiny_blocks} of this stage's last
        request batch on `device`."""
        key = str(device)
        if key not in self._range_host:
            self._range_host[key] = torch.zeros(2, dtype=torch.int64).pin_memory()
        return self._range_host[key]

    def range_counts(self, device=None, wait: bool = True):
        """(overflow_blocks, tiny_blocks) of the bf16 -> half conversions of the last request batch
        this stage ran on `device`.  `forward` queues their copy to pinned host memory behind the
        batch; this call waits for that copy (an event, not a device synchronisation) and reads
        it.  A caller that captured `forward` into a HIP graph calls it after a replay has
        finished: non-zero counts mean that replay's latents must be discarded and the batch
        redone with bf16 V (`value_half=False`)."""
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        ev = self._range_event.get(str(dev))
        if ev is not None and wait:
            ev.synchronize()
        elif wait:
            torch.cuda.current_stream(dev).synchronize()   # captured batch: no event exists
        host = self._range_buffer(dev)
        return int(host[0]), int(host[1])

    def _agree(self, over: int, tiny: int, device):
        """Every rank that shares collectives with this one inside a step (the CFG pair, a
        sequence-parallel group) must take the same branch after the loop: the largest count
        of the group decides for all of it."""
        import torch.distributed as dist

        from .parallel import _wire

        for group in self._decision_groups():
            t = torch.tensor([over, tiny], dtype=torch.int64, device=device)
            w = _wire(t, group)
            dist.all_reduce(w, op=dist.ReduceOp.MAX, group=group)
            over, tiny = int(w[0]), int(w[1])
        return over, tiny

    def _decision_groups(self):
        groups = [g for g in (self.cfg_group, self.sync_group) if g is not None]
        sp = getattr(self.transformer, "sp_group", None)
        if sp is not None:
            groups.append(sp)
        return groups

    @torch.no_grad()
    def forward(self, batch: Req, num_steps: Optional[int] = None) -> Req:
        """Runs the loop in place on `batch.latents`; `num_steps` truncates it (benchmarks).

        The attention kernels take V in IEEE half (exact for 2^-14 <= |v| <= 65504).  Every
        conversion is range-checked on the device.  The two counters are zeroed before the loop
        and copied to pinned host memory behind it, both on the launch stream and without a
        synchronisation; the stage then waits for that copy (an event) and, if a value left the
        range in THIS batch, runs THIS batch again from its initial latents with bf16 V.  The
        process default and other stages are not touched; a stage whose batches keep failing
        (`half_fallback_limit` in a row) stops trying half for itself.  Ranks that share
        collectives inside a step agree on the decision first (`_agree`).

        Under HIP graph capture nothing can be read back: the zeroing, the loop and the copy are
        captured, and the caller polls `range_counts()` after each replay."""
        first = batch.latents
        use_half = (self.value_half if self.value_half is not None else _C.value_half()) \
            and first.is_cuda
        if not use_half:
            with _C.value_half_as(False):
                return self._loop(batch, num_steps)
        dev = first.device
        capturing = torch.cuda.is_current_stream_capturing()
        host = self._range_buffer(dev)
        _C.f16_range_reset(dev)        # counts of earlier, unrelated work do not matter
        with _C.value_half_as(True):
            out = self._loop(batch, num_steps)
        _C.f16_range_snapshot(host, reset=True, device=dev)
        if capturing:
            self._range_event.pop(str(dev), None)
            return out
        ev = self._range_event[str(dev)] = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        over, tiny = self._agree(*self.range_counts(dev), dev)
        if not (over or tiny):
            self._half_failures = 0
            return out
        import warnings

        self._half_failures += 1
        giving_up = self._half_failures >= self.half_fallback_limit
        warnings.warn(f"attention V left IEEE half's range in this request batch ({over} blocks "
                      f"beyond 65504, {tiny} blocks entirely below 2^-13): re-running the batch "
                      "with bf16 V" + (f"; {self._half_failures} batches in a row did, this stage "
                                       "keeps bf16 V from now on" if giving_up else ""))
        if giving_up:
            self.value_half = False
        batch.latents = first
        with _C.value_half_as(False):
            return self._loop(batch, num_steps)

    def _loop(self, batch: Req, num_steps: Optional[int] = None) -> Req:
        device = batch.latents.device
        timesteps = self.prepare_timesteps(batch, device)
        bsz = batch.batch_size
        guidance = (self._build_guidance(bsz, device, batch.embedded_cfg_scale)
                    if self.transformer.config.guidance_embeds else None)
        freqs = self._freqs_cis(batch.prompt_embeds.shape[1], batch.width, batch.height, device)
        latents = batch.latents
        ts_host = self.scheduler._timesteps_host
        n = len(ts_host) if num_steps is None else min(num_steps, len(ts_host))
        cfg_inputs = (self.cfg_batched_inputs(batch, guidance)
                      if (batch.do_classifier_free_guidance and self.cfg_batched
                          and self.cfg_group is None) else None)
        for i in range(n):
            latent_model_input = latents.to(torch.bfloat16)
            timestep = timesteps[i].repeat(bsz)  # sigma*1000, fp32; the model does not rescale
            noise_pred = self.predict_noise_with_cfg(batch, latent_model_input, timestep, guidance,
                                                     freqs, i, cfg_inputs)
            # the update ping-pongs between two buffers of the stage (never in place: the
            # input of step 0 belongs to the caller)
            nxt = (self._buf(f"latents{i & 1}", latents.shape, noise_pred.dtype, device)
                   if latents.is_cuda else None)
            latents = self.scheduler.step(noise_pred, ts_host[i], latents, out=nxt)[0]
        # a fresh tensor for the caller: the stage's buffers are reused by its next batch
        batch.latents = latents.clone() if n > 0 else latents
        return batch
