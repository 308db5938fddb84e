"""The stages, worker and event loop around the denoise stage, on CPU with a stand-in denoiser."""
import pytest
import torch

from sglang_diffusion_amd.runtime import pipeline_config as pc
from sglang_diffusion_amd.runtime.denoise import Req
from sglang_diffusion_amd.runtime.pipeline import (DecodingStage, GenerationRequest, GPUWorker,
                                                   LatentPreparationStage, OutputBatch, Scheduler,
                                                   TextEncodingStage)
from tests.pipeline_parts import TinyVAE, tiny_text_encoders

TOK = (dict(max_length=77, padding="max_length", truncation=True),
       dict(max_length=24, padding="max_length", truncation=True))


class _Denoiser:
    """latents <- 0.5 latents + mean(prompt) + 0.25 mean(pooled) (- negative under CFG), row by
    row, so a request's result cannot depend on its micro-batch; records the batch sizes."""

    def __init__(self):
        self.seen = []
        self.transformer = None

    def forward(self, req: Req, num_steps=None):
        self.seen.append(req.batch_size)
        if req.height % 16:
            raise ValueError("height must be a multiple of 16")
        x = req.latents.float() * 0.5
        x = x + req.prompt_embeds.float().mean(dim=(1, 2)).view(-1, 1, 1)
        x = x + 0.25 * req.pooled_embeds.float().mean(dim=1).view(-1, 1, 1)
        if req.do_classifier_free_guidance:
            x = x - req.negative_prompt_embeds.float().mean(dim=(1, 2)).view(-1, 1, 1)
        req.latents = x.to(torch.bfloat16)
        return req


@pytest.fixture(scope="module")
def parts():
    encoders, tokenizers = tiny_text_encoders(joint_dim=32, pooled_dim=16)
    return TextEncodingStage(encoders, tokenizers, TOK), DecodingStage(TinyVAE())


def test_text_encoding_stage_shapes_masks_and_cfg(parts):
    enc, _ = parts
    g = GenerationRequest(prompt=["a red fox", "two cats on a sofa"], num_outputs_per_prompt=2,
                          negative_prompt="blurry", guidance_scale=3.5)
    b = enc.forward(g, {})
    assert tuple(b["prompt_embeds"].shape) == (4, 24, 32)         # T5 last hidden state
    assert tuple(b["pooled_embeds"].shape) == (4, 16)             # CLIP pooled output
    assert torch.equal(b["prompt_embeds"][0], b["prompt_embeds"][1])      # outputs of one prompt
    assert not torch.equal(b["prompt_embeds"][0], b["prompt_embeds"][2])
    assert tuple(b["negative_prompt_embeds"].shape) == (4, 24, 32)
    # T5 runs with an all-ones mask: padding the prompt differently must not change its rows'
    # dependence on the mask (same ids, same output whatever the tokenizer's mask says)
    e1, _ = enc.encode_text("a red fox")
    tok = enc.tokenizers[1]
    ids = tok(["a red fox"], **TOK[1])
    o = enc.text_encoders[1](input_ids=ids["input_ids"],
                             attention_mask=torch.ones(ids["input_ids"].shape))
    assert torch.allclose(e1, o.last_hidden_state, atol=1e-5)       # (grad mode picks other kernels)
    masked = enc.text_encoders[1](input_ids=ids["input_ids"], attention_mask=ids["attention_mask"])
    assert not torch.allclose(e1, masked.last_hidden_state, atol=1e-3)   # the mask would matter
    g2 = GenerationRequest(prompt="x", guidance_scale=3.5)           # no negative prompt: no CFG
    assert "negative_prompt_embeds" not in enc.forward(g2, {})


def test_latent_preparation_is_seeded_per_image():
    st = LatentPreparationStage()
    g = GenerationRequest(prompt=["a", "b", "c"], height=64, width=96, seed=7)
    lat = st.forward(g, {}, "cpu")["latents"]
    assert tuple(lat.shape) == (3, (64 // 16) * (96 // 16), 64) and lat.dtype == torch.bfloat16
    # image i of a request draws from seed + i, whatever else is in the request
    one = st.forward(GenerationRequest(prompt="b", height=64, width=96, seed=8), {}, "cpu")["latents"]
    assert torch.equal(one[0], lat[1])
    ref = torch.randn((1, 16, 8, 12), generator=torch.Generator().manual_seed(7))
    assert torch.equal(pc.pack_latents(ref).to(torch.bfloat16)[0], lat[0])


def test_decoding_stage_formula(parts):
    _, dec = parts
    packed = torch.randn(2, 4 * 6, 64).bfloat16()
    img = dec.decode(packed, 64, 96)
    # scale and shift run in the latents' own dtype, then the cast to the VAE's (decoding.py:127-149)
    z = (pc.unpack_latents(packed, 64, 96) / 0.3611 + 0.1159).float()
    with torch.no_grad():
        want = (dec.vae.decode(z)[0] / 2 + 0.5).clamp(0, 1)
    assert tuple(img.shape) == (2, 3, 64, 96) and torch.equal(img, want)
    assert float(img.min()) >= 0 and float(img.max()) <= 1


def _worker(parts, **kw):
    enc, dec = parts
    den = _Denoiser()
    return GPUWorker(text_encoding=enc, decoding=dec, denoising=den, device="cpu", **kw), den


def test_worker_batches_requests_and_matches_each_alone(parts):
    w, den = _worker(parts, max_samples_per_step=4)
    reqs = [GenerationRequest(prompt=f"prompt number {i}", height=64, width=64, seed=i,
                              num_inference_steps=4, request_id=f"r{i}") for i in range(3)]
    reqs.append(GenerationRequest(prompt="with guidance", negative_prompt="bad", guidance_scale=3.5,
                                  height=64, width=64, num_inference_steps=4, request_id="r3"))
    outs = w.execute_forward(reqs)
    assert [o.request_id for o in outs] == ["r0", "r1", "r2", "r3"] and all(o.error is None for o in outs)
    assert den.seen == [3, 1]              # three plain requests share a micro-batch; CFG differs
    for g, o in zip(reqs, outs):
        w1, _ = _worker(parts)
        alone = w1.execute_forward([g])[0]
        assert torch.equal(alone.output, o.output) and tuple(o.output.shape) == (1, 3, 64, 64)
        assert o.metrics.total_duration_ms > 0 and set(o.metrics.stages_ms) == {"encode", "denoise", "decode"}


def test_worker_reports_errors_per_request(parts):
    w, _ = _worker(parts)
    good = GenerationRequest(prompt="fine", height=64, width=64, request_id="good")
    outs = w.execute_forward([GenerationRequest(prompt=5, request_id="bad"), good])   # not a string
    assert outs[0].error is not None and "bad" in outs[0].error and outs[0].output is None
    assert outs[1].error is None and outs[1].output is not None
    # an error inside the denoise loop fails the requests of that iteration, with a reply each
    outs = w.execute_forward([GenerationRequest(prompt="odd size", height=72, width=64, request_id="odd")])
    assert outs[0].error is not None and "multiple of 16" in outs[0].error


def test_scheduler_event_loop_replies_to_every_identity(parts):
    w, den = _worker(parts, max_samples_per_step=8)
    got = []
    sch = Scheduler(w, on_result=lambda ident, out: got.append((ident, out)),
                    max_requests_per_iteration=3)
    warm = GenerationRequest(prompt="", height=64, width=64, is_warmup=True)
    sch.submit(None, warm)
    for i in range(5):
        sch.submit(f"client-{i}".encode(), GenerationRequest(prompt=f"p {i}", height=64, width=64,
                                                             seed=i, request_id=i))
    sch.event_loop()
    assert [ident for ident, _ in got] == [f"client-{i}".encode() for i in range(5)]   # FIFO, no warmup reply
    assert all(isinstance(o, OutputBatch) and o.error is None for _, o in got)
    assert den.seen == [3, 3]              # iterations of 3 requests: (warmup + 2), then 3
    assert sch.get_next_batch_to_run() is None
    # requests submitted later are picked up by the next call
    sch.submit(b"late", GenerationRequest(prompt="late", height=64, width=64, request_id="late"))
    sch.event_loop()
    assert got[-1][0] == b"late"
