"""Would two half-batch pipelines on two streams fill each other's idle CUs (partial last
rounds of persistent GEMM / attention grids, HBM-bound passes beside MFMA-bound ones)?
One Flux.1-dev model, two FluxEngine instances (own workspaces, own side streams) over the same
weights: B = 4 in one call against 2 x (B = 2) issued on two streams, same box, alternating.
python tools/probes/two_pipelines.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "sglang-diffusion_amd"))
import torch

from sglang_diffusion_amd.layers.quantization import Fp8Config
from sglang_diffusion_amd.models.flux import FluxTransformer2DModel, flux_dev_config
from sglang_diffusion_amd.models.flux_engine import FluxEngine
from sglang_diffusion_amd.runtime.synthetic import init_synthetic_weights

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
cfg = flux_dev_config()
with torch.device(dev):
    model = FluxTransformer2DModel(cfg, quant_config=Fp8Config(is_checkpoint_fp8_serialized=False))
init_synthetic_weights(model, seed=0, host_rng=False)
model.process_weights_after_loading()
engines = [FluxEngine(model), FluxEngine(model), FluxEngine(model)]
from sglang_diffusion_amd.runtime.denoise import DenoisingStage

FREQS = DenoisingStage(model, fused=True)._freqs_cis(512, 1024, 1024, dev)


def inputs(B):
    g = torch.Generator(device=dev).manual_seed(B)
    lat = torch.randn(B, 4096, 64, device=dev, generator=g).bfloat16()
    txt = torch.randn(B, 512, 4096, device=dev, generator=g).bfloat16()
    pooled = torch.randn(B, 768, device=dev, generator=g).bfloat16()
    t = torch.full((B,), 0.5, device=dev)
    guid = torch.full((B,), 3.5, device=dev)
    return lat, txt, pooled, t, guid, FREQS


def call(engine, inp):
    lat, txt, pooled, t, guid, freqs = inp
    return engine.step(lat, txt, pooled, t, guid, freqs)


i4, i2a, i2b = inputs(4), inputs(2), inputs(2)
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)


def one(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        call(engines[0], i4)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def two(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        with torch.cuda.stream(s1):
            call(engines[1], i2a)
        with torch.cuda.stream(s2):
            call(engines[2], i2b)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    one(2)
    two(2)
    for r in range(3):
        print(f"B=4 in one call: {one(6):.1f} ms   2 x B=2 on two streams: {two(6):.1f} ms", flush=True)
