"""Prompt to pixels through the HIP denoise path: tiny text encoders and a one-layer VAE decoder
stand in for the real ones (they are arguments of the stages, not part of this package); the
transformer is the small Flux-shaped model on the fused path."""
import pytest
import torch

from tests.flux_small import build_small
from tests.pipeline_parts import TinyVAE, tiny_text_encoders

pytestmark = pytest.mark.gpu


def test_prompt_to_pixels_through_worker_and_event_loop(device):
    from sglang_diffusion_amd.runtime.denoise import DenoisingStage
    from sglang_diffusion_amd.runtime.pipeline import (DecodingStage, GenerationRequest, GPUWorker,
                                                       Scheduler, TextEncodingStage)

    model, _, cfg = build_small(device, fp8=True)
    encoders, tokenizers = tiny_text_encoders(cfg.joint_attention_dim, cfg.pooled_projection_dim)
    encoders = [e.to(device) for e in encoders]
    tok = (dict(max_length=77, padding="max_length", truncation=True),
           dict(max_length=32, padding="max_length", truncation=True))
    enc = TextEncodingStage(encoders, tokenizers, tok)
    dec = DecodingStage(TinyVAE().to(device))
    worker = GPUWorker(model, text_encoding=enc, decoding=dec, max_samples_per_step=4)

    def request(i, **kw):
        return GenerationRequest(prompt=f"picture number {i} of a fox", height=128, width=96, seed=i,
                                 num_inference_steps=3, request_id=i, **kw)

    reqs = [request(0), request(1), request(2, negative_prompt="blurry", guidance_scale=3.5),
            request(3, num_outputs_per_prompt=2)]
    got = {}
    sch = Scheduler(worker, on_result=lambda ident, out: got.__setitem__(ident, out),
                    max_requests_per_iteration=4)
    for r in reqs:
        sch.submit(f"id{r.request_id}", r)
    sch.event_loop()
    assert sorted(got) == ["id0", "id1", "id2", "id3"]
    for r in reqs:
        out = got[f"id{r.request_id}"]
        assert out.error is None, out.error
        assert tuple(out.output.shape) == (r.batch_size, 3, 128, 96)
        assert torch.isfinite(out.output).all() and 0 <= float(out.output.min()) <= float(out.output.max()) <= 1
        # the same request by hand: encode, prepare noise, denoise alone, decode
        req = worker.prepare(r)
        lat = DenoisingStage(model, fused=True).forward(req).latents
        want = dec.decode(lat, r.height, r.width)
        assert torch.equal(out.output, want), f"request {r.request_id} differs from the unbatched run"
    assert not torch.equal(got["id3"].output[0], got["id3"].output[1])      # two seeds, two images
