"""One GPU's share of the 8-GPU data-parallel configuration, at full size.

32 requests with true classifier-free guidance over 8 GPUs are 4 requests = 8 DiT samples per
GPU and denoise step.  This runs exactly that share -- the whole Flux.1-dev transformer (19 + 38
blocks, width 3072, 512 + 4096 tokens, W8A8 linears, synthetic weights), four guidance-3.5
requests with negative embeddings, through DataParallelBatchScheduler on one rank -- for two
denoise steps, and checks the size-independent properties the domain offers:

* the scheduler packs the four requests into ONE micro-batch of 8 samples, cond and uncond of a
  request side by side;
* every request's latents equal, bit for bit, what it gets when it runs alone through the
  denoise stage (batch rows are independent in every kernel of the path);
* the latents moved and are finite.

The oracle comparison of this arithmetic at full width lives in tests/test_real_width_gpu.py
(one block, teacher-forced); the oracle cannot run 57 blocks x 8 samples in test time.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

STEPS = 2


@pytest.fixture(scope="module")
def full_model(device):
    from sglang_diffusion_amd.layers.quantization import Fp8Config
    from sglang_diffusion_amd.models.flux import FluxTransformer2DModel, flux_dev_config
    from sglang_diffusion_amd.runtime.synthetic import init_synthetic_weights

    cfg = flux_dev_config()
    with torch.device(device):
        model = FluxTransformer2DModel(cfg, quant_config=Fp8Config(is_checkpoint_fp8_serialized=False))
    init_synthetic_weights(model, seed=0, host_rng=False)
    model.process_weights_after_loading()
    yield model, cfg
    del model
    torch.cuda.empty_cache()


def _request(device, seed):
    from sglang_diffusion_amd.runtime.synthetic import make_synthetic_request

    return make_synthetic_request(1, 1024, 1024, 50, device=device, seed=seed, true_cfg=True,
                                  guidance_scale=3.5)


def test_one_gpus_share_of_the_cfg_workload(device, full_model):
    from sglang_diffusion_amd.runtime.batch_scheduler import (DataParallelBatchScheduler,
                                                              request_cost)
    from sglang_diffusion_amd.runtime.denoise import DenoisingStage

    model, cfg = full_model
    stage = DenoisingStage(model, fused=True)
    sch = DataParallelBatchScheduler(stage, rank=0, world_size=1, max_samples_per_step=8)
    for seed in range(4):
        sch.add_request(_request(device, seed), identity=f"req-{seed}")
    # what the event loop would be handed: all four requests, 8 DiT samples
    peek = DataParallelBatchScheduler(stage, max_samples_per_step=8)
    for seed in range(4):
        peek.add_request(_request(device, seed), identity=f"req-{seed}")
    items = peek.get_next_batch_to_run()
    assert [ident for ident, _ in items] == ["req-0", "req-1", "req-2", "req-3"]
    assert sum(request_cost(r) for _, r in items) == 8
    assert peek.get_next_batch_to_run() is None

    timing = {}
    outs = sch.run(num_steps=STEPS, timing=timing)
    torch.cuda.synchronize()
    assert len(outs) == 4
    print(f"4 CFG requests (8 samples) x {STEPS} steps through the scheduler: "
          f"{timing['compute_s'] / STEPS * 1e3:.1f} ms per 8-sample step")
    for seed, got in enumerate(outs):
        start = _request(device, seed).latents
        assert got.shape == start.shape and torch.isfinite(got.float()).all()
        assert not torch.equal(got, start)
        alone = stage.forward(_request(device, seed), num_steps=STEPS).latents
        assert torch.equal(got, alone), f"request {seed} depends on its batch neighbours"
