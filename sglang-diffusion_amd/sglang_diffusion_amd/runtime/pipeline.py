"""Stages, worker and event loop around the denoise stage: what a real request passes through.

The hot path of this package starts at encoded prompts and noise and ends at final latents.
This module is the glue on both sides of it, in the shapes the reference gives those pieces, so
that the data-parallel batch scheduler sits under the reference's request flow:

    text prompt --TextEncodingStage--> prompt / pooled embeddings          (stages/text_encoding.py:30-292)
                --LatentPreparationStage--> packed noise latents            (stages/latent_preparation.py:38-105)
                --DataParallelBatchScheduler + DenoisingStage--> latents    (the hot path)
                --DecodingStage--> pixels in [0, 1]                         (stages/decoding.py:52-240)

`GPUWorker.execute_forward` (managers/gpu_worker.py:207-290) runs these for a list of requests
and always returns an `OutputBatch` (errors are captured, not raised); `Scheduler.event_loop`
(managers/scheduler.py:326-418) is the receive / pick / execute / reply loop over a
`waiting_queue`, fed from an in-process inbox instead of a ZMQ socket (the transport is the
reference's control plane and stays there).

The text encoders, tokenizers and the VAE are NOT part of this package: the stages take them as
arguments (Hugging Face `CLIPTextModel` / `T5EncoderModel`-style modules, tokenizer callables,
any module with `decode`).  They run in eager PyTorch on whatever device they live on.
"""
from __future__ import annotations

import time
from collections import deque
from dataclasses import dataclass, field
from typing import Any, Callable, Deque, List, Optional, Sequence, Tuple

import torch

from . import pipeline_config as pc
from .batch_scheduler import DataParallelBatchScheduler
from .denoise import DenoisingStage, Req


# ------------------------------------------------------------------------------ requests --
@dataclass
class GenerationRequest:
    """A request before encoding: the fields of pipelines_core/schedule_batch.py:42-230 that
    the Flux image pipeline reads."""

    prompt: str | List[str]
    negative_prompt: Optional[str] = None
    height: int = 1024
    width: int = 1024
    num_inference_steps: int = 50
    guidance_scale: float = 1.0            # true CFG; > 1 needs negative_prompt
    embedded_cfg_scale: float = pc.EMBEDDED_CFG_SCALE
    guidance_rescale: float = 0.0
    cfg_normalization: float = 0.0
    num_outputs_per_prompt: int = 1
    seed: int = 0
    request_id: Any = None
    is_warmup: bool = False

    @property
    def do_classifier_free_guidance(self) -> bool:
        return self.guidance_scale > 1.0 and self.negative_prompt is not None

    @property
    def batch_size(self) -> int:
        """prompts x outputs per prompt (schedule_batch.py:231-243)."""
        n = len(self.prompt) if isinstance(self.prompt, list) else 1
        return n * self.num_outputs_per_prompt


@dataclass
class RequestMetrics:
    total_duration_ms: float = 0.0
    stages_ms: dict = field(default_factory=dict)


@dataclass
class OutputBatch:
    """pipelines_core/schedule_batch.py:327-346, the fields this pipeline fills."""

    output: Optional[torch.Tensor] = None
    error: Optional[str] = None
    metrics: RequestMetrics = field(default_factory=RequestMetrics)
    request_id: Any = None


# -------------------------------------------------------------------------------- stages --
# tokenizer arguments of the two Flux.1 encoders: CLIP (configs/pipeline_configs/flux.py:76-87)
# and T5 (configs/models/encoders/t5.py: text_len 512, padded to the maximum)
CLIP_TOKENIZER_KWARGS = dict(max_length=77, padding="max_length", truncation=True,
                             return_overflowing_tokens=False, return_length=False,
                             return_tensors="pt")
T5_TOKENIZER_KWARGS = dict(max_length=pc.TEXT_LEN, padding="max_length", truncation=True,
                           return_tensors="pt")


class TextEncodingStage:
    """Flux.1: encoder 0 is CLIP (its pooled output conditions the timestep embedding),
    encoder 1 is T5 (its last hidden state is the text stream).  The T5 attention mask is all
    ones for Flux.1, as in the reference (text_encoding.py:262-266)."""

    def __init__(self, text_encoders: Sequence[Any], tokenizers: Sequence[Callable],
                 tokenizer_kwargs: Optional[Sequence[dict]] = None):
        if len(text_encoders) != 2 or len(tokenizers) != 2:
            raise ValueError("Flux.1 has two text encoders (CLIP, T5) and two tokenizers")
        self.text_encoders = list(text_encoders)
        self.tokenizers = list(tokenizers)
        self.tokenizer_kwargs = list(tokenizer_kwargs or (CLIP_TOKENIZER_KWARGS, T5_TOKENIZER_KWARGS))

    @staticmethod
    def _device_of(module) -> torch.device:
        try:
            return next(module.parameters()).device
        except (AttributeError, StopIteration):
            return torch.device("cpu")

    @torch.no_grad()
    def encode_text(self, text: str | List[str]) -> Tuple[torch.Tensor, torch.Tensor]:
        """-> (prompt_embeds [B, text_len, joint_dim] from T5, pooled_embeds [B, pooled_dim]
        from CLIP)."""
        texts = [text] if isinstance(text, str) else list(text)
        outs = []
        for i, (tok, enc) in enumerate(zip(self.tokenizers, self.text_encoders)):
            dev = self._device_of(enc)
            inputs = tok(texts, **self.tokenizer_kwargs[i])
            input_ids = inputs["input_ids"].to(dev)
            if i == 1:
                mask = torch.ones(input_ids.shape[:2], device=dev)
            else:
                mask = inputs["attention_mask"].to(dev)
            outs.append(enc(input_ids=input_ids, attention_mask=mask, output_hidden_states=True))
        return outs[1].last_hidden_state, outs[0].pooler_output

    def forward(self, g: GenerationRequest, batch: dict) -> dict:
        """Fills prompt / pooled (and, under true CFG, negative) embeddings; one row per output
        image (a prompt's embeddings are repeated `num_outputs_per_prompt` times)."""
        emb, pooled = self.encode_text(g.prompt)
        rep = g.num_outputs_per_prompt
        batch["prompt_embeds"] = emb.repeat_interleave(rep, dim=0)
        batch["pooled_embeds"] = pooled.repeat_interleave(rep, dim=0)
        if g.do_classifier_free_guidance:
            n = len(g.prompt) if isinstance(g.prompt, list) else 1
            nemb, npooled = self.encode_text([g.negative_prompt] * n)
            batch["negative_prompt_embeds"] = nemb.repeat_interleave(rep, dim=0)
            batch["neg_pooled_embeds"] = npooled.repeat_interleave(rep, dim=0)
        return batch


class LatentPreparationStage:
    """Noise in the unpacked latent shape from a seeded generator, then packed to tokens
    (latent_preparation.py:38-105; the flow-match scheduler has no init_noise_sigma)."""

    def __init__(self, in_channels: int = 64, generator_device: str = "cpu"):
        self.in_channels = in_channels
        self.generator_device = generator_device

    def forward(self, g: GenerationRequest, batch: dict, device, dtype=torch.bfloat16) -> dict:
        shape = pc.prepare_latent_shape(g.batch_size, g.height, g.width, self.in_channels)
        # one generator per output image, seeds seed, seed + 1, ... (input_validation.py:60-90)
        rows = []
        for i in range(g.batch_size):
            gen = torch.Generator(self.generator_device).manual_seed(g.seed + i)
            rows.append(torch.randn((1,) + tuple(shape[1:]), generator=gen,
                                    device=self.generator_device, dtype=torch.float32))
        latents = torch.cat(rows, dim=0).to(device=device, dtype=dtype)
        batch["latents"] = pc.pack_latents(latents)
        return batch


class DecodingStage:
    """latents / scaling_factor + shift_factor -> vae.decode -> (x / 2 + 0.5).clamp(0, 1)
    (decoding.py:86-150); factors from the VAE's config (base.py:319-328)."""

    def __init__(self, vae, vae_dtype: torch.dtype = torch.float32):
        self.vae = vae
        self.vae_dtype = vae_dtype

    def _factors(self) -> Tuple[float, Optional[float]]:
        cfg = getattr(self.vae, "config", self.vae)
        scaling = getattr(cfg, "scaling_factor", None)
        if scaling is None:
            raise ValueError("the VAE (or its config) must carry scaling_factor")
        return scaling, getattr(cfg, "shift_factor", None)

    def scale_and_shift(self, latents: torch.Tensor) -> torch.Tensor:
        scaling, shift = self._factors()
        latents = latents / scaling
        if shift is not None:
            latents = latents + shift
        return latents

    @torch.no_grad()
    def decode(self, packed_latents: torch.Tensor, height: int, width: int) -> torch.Tensor:
        dev = TextEncodingStage._device_of(self.vae)
        latents = pc.unpack_latents(packed_latents, height, width).to(dev)      # post_denoising_loop
        latents = self.scale_and_shift(latents).to(self.vae_dtype)
        out = self.vae.decode(latents)
        image = out[0] if isinstance(out, tuple) else getattr(out, "sample", out)
        return (image / 2 + 0.5).clamp(0, 1)


# -------------------------------------------------------------------------------- worker --
class GPUWorker:
    """One process per GPU.  `execute_forward(batch)` takes the requests of one scheduler
    iteration, encodes them, runs the denoise loop on all of them as micro-batches of the batch
    scheduler, decodes, and returns one OutputBatch per request; a failing request yields an
    OutputBatch with `error` set and does not stop the others (gpu_worker.py:277-289)."""

    def __init__(self, transformer=None, text_encoding: Optional[TextEncodingStage] = None,
                 decoding: Optional[DecodingStage] = None,
                 denoising: Optional[DenoisingStage] = None, max_samples_per_step: int = 8,
                 rank: int = 0, world_size: int = 1, group=None, device=None,
                 latent_preparation: Optional[LatentPreparationStage] = None):
        self.denoising = denoising if denoising is not None else DenoisingStage(transformer, fused=True)
        self.text_encoding = text_encoding
        self.decoding = decoding
        self.latent_preparation = latent_preparation or LatentPreparationStage()
        self.rank, self.world_size = rank, world_size
        self.batcher = DataParallelBatchScheduler(self.denoising, rank=rank, world_size=world_size,
                                                  max_samples_per_step=max_samples_per_step,
                                                  group=group)
        self.device = device if device is not None else self.batcher.stage_device()

    def prepare(self, g: GenerationRequest) -> Req:
        """Everything before the denoise loop: one GenerationRequest -> one denoise-level Req."""
        if self.text_encoding is None:
            raise RuntimeError("this worker was built without text encoders")
        b: dict = {}
        self.text_encoding.forward(g, b)
        self.latent_preparation.forward(g, b, self.device)
        to = lambda x: None if x is None else x.to(device=self.device, dtype=torch.bfloat16)  # noqa: E731
        return Req(latents=b["latents"], prompt_embeds=to(b["prompt_embeds"]),
                   pooled_embeds=to(b["pooled_embeds"]),
                   negative_prompt_embeds=to(b.get("negative_prompt_embeds")),
                   neg_pooled_embeds=to(b.get("neg_pooled_embeds")), height=g.height, width=g.width,
                   num_inference_steps=g.num_inference_steps, guidance_scale=g.guidance_scale,
                   embedded_cfg_scale=g.embedded_cfg_scale, cfg_normalization=g.cfg_normalization,
                   guidance_rescale=g.guidance_rescale, request_ids=[g.request_id])

    def execute_forward(self, batch: List[GenerationRequest]) -> List[OutputBatch]:
        outs = [OutputBatch(request_id=g.request_id) for g in batch]
        t0 = time.monotonic()
        live = []
        for i, g in enumerate(batch):
            try:
                self.batcher.add_request(self.prepare(g), identity=i)
                live.append(i)
            except Exception as e:  # noqa: BLE001 - the reply must always be sent
                outs[i].error = f"Error executing request {g.request_id}: {e}"
        t1 = time.monotonic()
        try:
            done = {}
            while (items := self.batcher.get_next_batch_to_run()) is not None:
                for (ident, _), lat in zip(items, self.batcher.execute(items)):
                    done[ident] = lat
        except Exception as e:  # noqa: BLE001
            self.batcher.waiting_queue.clear()
            for i in live:
                outs[i].error = f"Error executing request {batch[i].request_id}: {e}"
            done = {}
        t2 = time.monotonic()
        for i in live:
            if outs[i].error is not None:
                continue
            try:
                g = batch[i]
                outs[i].output = (self.decoding.decode(done[i], g.height, g.width)
                                  if self.decoding is not None else done[i])
            except Exception as e:  # noqa: BLE001
                outs[i].error = f"Error executing request {batch[i].request_id}: {e}"
        t3 = time.monotonic()
        for o in outs:
            o.metrics.total_duration_ms = (t3 - t0) * 1e3
            o.metrics.stages_ms = dict(encode=(t1 - t0) * 1e3, denoise=(t2 - t1) * 1e3,
                                       decode=(t3 - t2) * 1e3)
        return outs


# ----------------------------------------------------------------------------- event loop --
class Scheduler:
    """The reference's scheduler loop (managers/scheduler.py:326-418) with two differences: the
    inbox is an in-process deque of (identity, request) instead of a ZMQ socket, and
    `get_next_batch_to_run` hands the worker up to `max_requests_per_iteration` requests at once
    (the reference pops exactly one, scheduler.py:182-190) so that the worker's batch scheduler
    can share denoise steps between them.  Replies go to `on_result(identity, OutputBatch)`."""

    def __init__(self, worker: GPUWorker, on_result: Callable[[Any, OutputBatch], None],
                 max_requests_per_iteration: int = 8, max_consecutive_errors: int = 3):
        self.worker = worker
        self.on_result = on_result
        self.inbox: Deque[Tuple[Any, GenerationRequest]] = deque()
        self.waiting_queue: Deque[Tuple[Any, GenerationRequest]] = deque()
        self.max_requests = max_requests_per_iteration
        self._max_consecutive_errors = max_consecutive_errors
        self._running = True

    def submit(self, identity: Any, req: GenerationRequest) -> None:
        self.inbox.append((identity, req))

    def recv_reqs(self) -> List[Tuple[Any, GenerationRequest]]:
        out = list(self.inbox)
        self.inbox.clear()
        return out

    def get_next_batch_to_run(self) -> Optional[List[Tuple[Any, GenerationRequest]]]:
        if not self.waiting_queue:
            return None
        n = min(self.max_requests, len(self.waiting_queue))
        return [self.waiting_queue.popleft() for _ in range(n)]

    def return_result(self, output_batch: OutputBatch, identity: Any, is_warmup: bool = False) -> None:
        if not is_warmup and identity is not None:
            self.on_result(identity, output_batch)

    def stop(self) -> None:
        self._running = False

    def event_loop(self, until_idle: bool = True) -> None:
        """Receive, pick, execute, reply.  `until_idle`: return once inbox and queue are empty
        (tests, batch jobs); otherwise spin until `stop()`."""
        errors = 0
        while self._running:
            try:
                self.waiting_queue.extend(self.recv_reqs())
                errors = 0
            except Exception as e:  # noqa: BLE001
                errors += 1
                if errors >= self._max_consecutive_errors:
                    raise RuntimeError(f"Scheduler terminated after {errors} consecutive errors. "
                                       f"Last error: {e}") from e
                continue
            items = self.get_next_batch_to_run()
            if not items:
                if until_idle and not self.inbox:
                    return
                continue
            identities = [it[0] for it in items]
            reqs = [it[1] for it in items]
            try:
                outputs = self.worker.execute_forward(reqs)
            except Exception as e:  # noqa: BLE001 - make sure a reply is always sent
                outputs = [OutputBatch(error=str(e), request_id=r.request_id) for r in reqs]
            for ident, req, out in zip(identities, reqs, outputs):
                self.return_result(out, ident, is_warmup=req.is_warmup)
