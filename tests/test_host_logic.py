"""Host-side logic that runs without a GPU: request sharding and micro-batch planning of the
data-parallel batch scheduler, the flow-match schedule against the oracle's restatement, the
quantization / attention plugin registries, latent packing, and the op-registry schemas."""
import math

import numpy as np
import pytest
import torch

from oracle import flux_oracle
from sglang_diffusion_amd.layers import attention as A
from sglang_diffusion_amd.layers import quantization as Q
from sglang_diffusion_amd.models.scheduler import (FlowMatchEulerDiscreteScheduler, calculate_shift,
                                                   prepare_mu, prepare_sigmas)
from sglang_diffusion_amd.runtime import pipeline_config as P
from sglang_diffusion_amd.runtime.batch_scheduler import (DataParallelBatchScheduler,
                                                          assign_requests, merge_requests,
                                                          plan_microbatches, request_cost)
from sglang_diffusion_amd.runtime.denoise import Req
from sglang_diffusion_amd.runtime.parallel import shard_requests
from sglang_diffusion_amd.runtime.synthetic import make_synthetic_request


# ------------------------------------------------------------------------- DP scheduler --
@pytest.mark.parametrize("n,world", [(0, 8), (1, 8), (7, 8), (8, 8), (32, 8), (33, 8), (5, 2), (64, 3)])
def test_shard_requests_partitions_the_queue(n, world):
    parts = [shard_requests(n, world, r) for r in range(world)]
    flat = [i for p in parts for i in p]
    assert flat == list(range(n))                      # contiguous, ordered, complete, disjoint
    sizes = [len(p) for p in parts]
    assert max(sizes) - min(sizes) <= 1                # balanced
    assert sizes == sorted(sizes, reverse=True)        # the remainder goes to the first ranks


def _req(b=1, h=128, w=128, steps=4, cfg=False, seed=0):
    r = make_synthetic_request(b, height=h, width=w, num_inference_steps=steps, device="cpu",
                               seed=seed, text_len=8, joint_dim=16, pooled_dim=8)
    if cfg:
        r.guidance_scale = 3.5
        r.negative_prompt_embeds = torch.zeros_like(r.prompt_embeds)
        r.neg_pooled_embeds = torch.zeros_like(r.pooled_embeds)
    return r


def test_plan_microbatches_fifo_capacity_and_compat():
    reqs = [_req(1), _req(2), _req(1, cfg=True), _req(1), _req(1, h=256, w=256), _req(3), _req(1)]
    plans = plan_microbatches(reqs, max_samples=4)
    # every request exactly once
    assert sorted(i for p in plans for i in p) == list(range(len(reqs)))
    for p in plans:
        assert p == sorted(p)                          # FIFO inside a micro-batch
        keys = {(reqs[i].height, reqs[i].do_classifier_free_guidance) for i in p}
        assert len(keys) == 1                          # only compatible requests are merged
        cost = sum(reqs[i].batch_size * (2 if reqs[i].do_classifier_free_guidance else 1) for i in p)
        assert cost <= 4
    assert plans[0] == [0, 1, 3]                       # 1 + 2 + 1 samples fill the first batch
    assert [2] in plans and [4] in plans               # CFG pair and the other resolution stay apart
    with pytest.raises(ValueError):
        plan_microbatches([_req(5)], max_samples=4)
    # a CFG request costs two DiT samples
    assert plan_microbatches([_req(1, cfg=True)] * 3, max_samples=4) == [[0, 1], [2]]


def test_merge_requests_concatenates_batch_dim():
    a, b = _req(1, seed=1), _req(2, seed=2)
    m = merge_requests([a, b])
    assert m.batch_size == 3
    assert torch.equal(m.latents[:1], a.latents) and torch.equal(m.latents[1:], b.latents)
    assert torch.equal(m.prompt_embeds[1:], b.prompt_embeds)
    c, d = _req(1, cfg=True, seed=3), _req(1, cfg=True, seed=4)
    m2 = merge_requests([c, d])
    assert m2.do_classifier_free_guidance and m2.negative_prompt_embeds.shape[0] == 2


class _FakeStage:
    """Stands in for the GPU denoise loop: a deterministic per-sample function of the inputs,
    so the scheduler's slicing / merging / ordering can be checked without a device."""

    def forward(self, req: Req, num_steps=None):
        req.latents = req.latents * 2 + req.prompt_embeds.float().mean(dim=(1, 2)).view(-1, 1, 1)
        return req


def test_dp_scheduler_single_rank_matches_per_request_runs():
    reqs = [_req(1, seed=s) for s in range(5)] + [_req(2, seed=9)]
    expect = [_FakeStage().forward(merge_requests([r])).latents for r in reqs]
    sch = DataParallelBatchScheduler(_FakeStage(), max_samples_per_step=3)
    for r in reqs:
        sch.add_request(r)
    outs = sch.run()
    assert len(sch.waiting_queue) == 0 and len(outs) == len(reqs)
    for o, e in zip(outs, expect):
        assert torch.equal(o, e)


def test_assign_requests_balances_dit_sample_cost():
    # 4 CFG requests (2 samples each) + 4 plain ones on 2 ranks: 6 samples per rank, not 4 + 8
    costs = [2, 2, 2, 2, 1, 1, 1, 1]
    parts = assign_requests(costs, 2)
    assert sorted(i for p in parts for i in p) == list(range(8))
    assert [sum(costs[i] for i in p) for p in parts] == [6, 6]
    assert all(p == sorted(p) for p in parts)                    # queue order inside a rank
    # equal costs: the C4 case, 32 CFG pairs over 8 ranks = 4 requests = 8 samples each
    parts = assign_requests([2] * 32, 8)
    assert [len(p) for p in parts] == [4] * 8
    # uneven CFG mix: the heaviest rank is within one request of the mean
    costs = [2, 1, 1, 2, 2, 1, 2, 1, 1, 1, 2]
    for world in (2, 3, 4, 8):
        loads = [sum(costs[i] for i in p) for p in assign_requests(costs, world)]
        assert sum(loads) == sum(costs) and max(loads) - min(loads) <= 2
    assert assign_requests([], 4) == [[], [], [], []]
    assert request_cost(_req(3, cfg=True)) == 6 and request_cost(_req(3)) == 3


def test_merge_requests_carries_every_stage_option():
    a, b = _req(1, cfg=True, seed=1), _req(1, cfg=True, seed=2)
    for r, rid in ((a, "a"), (b, "b")):
        r.guidance_rescale, r.request_ids = 0.7, [rid]
    m = merge_requests([a, b])
    assert m.guidance_rescale == 0.7 and m.cfg_normalization == 0.0 and m.request_ids == ["a", "b"]
    b.guidance_rescale = 0.0                                     # differing options never merge
    with pytest.raises(ValueError):
        merge_requests([a, b])
    assert plan_microbatches([a, b], max_samples=8) == [[0], [1]]
    # the norm cap is a whole-batch norm: such requests run alone even when they match
    c, d = _req(1, cfg=True, seed=3), _req(1, cfg=True, seed=4)
    c.cfg_normalization = d.cfg_normalization = 1.0
    assert plan_microbatches([c, d], max_samples=8) == [[0], [1]]
    with pytest.raises(ValueError):
        merge_requests([c, d])
    assert merge_requests([c]).cfg_normalization == 1.0


class _OptionStage:
    """A stand-in denoise stage whose result depends on the two post-CFG options the way the
    real ones do: guidance_rescale per sample, cfg_normalization through a whole-batch norm."""

    def forward(self, req: Req, num_steps=None):
        x = req.latents.float() * (1.0 + req.guidance_rescale)
        if req.cfg_normalization and req.cfg_normalization > 0:
            x = x * req.cfg_normalization / x.norm()
        req.latents = x
        return req


def test_scheduler_equals_stage_forward_with_post_cfg_options():
    def queue():
        out = []
        for s in range(4):
            r = _req(1, cfg=True, seed=s)
            r.guidance_rescale, r.cfg_normalization = 0.7, 1.5
            out.append(r)
        return out

    expect = [_OptionStage().forward(r).latents for r in queue()]
    sch = DataParallelBatchScheduler(_OptionStage(), max_samples_per_step=8)
    for r in queue():
        sch.add_request(r)
    for o, e in zip(sch.run(), expect):
        assert torch.equal(o, e)


def test_get_next_batch_to_run_admits_between_micro_batches():
    sch = DataParallelBatchScheduler(_FakeStage(), max_samples_per_step=2)
    assert sch.get_next_batch_to_run() is None
    a, b, c, big = _req(1, seed=1), _req(1, h=256, w=256, seed=2), _req(1, seed=3), _req(2, seed=4)
    for name, r in (("a", a), ("b", b), ("c", c), ("big", big)):
        sch.add_request(r, identity=name)
    first = sch.get_next_batch_to_run()
    assert [i for i, _ in first] == ["a", "c"]                   # b (other shape) keeps its place
    sch.add_request(_req(1, h=256, w=256, seed=5), identity="late")   # admitted while "running"
    assert [i for i, _ in sch.get_next_batch_to_run()] == ["b", "late"]
    assert [i for i, _ in sch.get_next_batch_to_run()] == ["big"]
    assert sch.get_next_batch_to_run() is None
    outs = sch.execute(first)
    assert torch.equal(outs[1], _FakeStage().forward(merge_requests([_req(1, seed=3)])).latents)


# ----------------------------------------------------------------------------- schedule --
def test_mu_and_sigmas_follow_the_reference_rules():
    # runtime/pipelines/flux.py:22-51: image_seq_len = (H/8)(W/8) = 16384, NOT the 4096 packed tokens
    mu = prepare_mu(1024, 1024)
    m = (1.15 - 0.5) / (4096 - 256)
    assert mu == pytest.approx(16384 * m + (0.5 - m * 256), rel=1e-12)
    assert calculate_shift(256) == pytest.approx(0.5)
    assert calculate_shift(4096) == pytest.approx(1.15)
    s = prepare_sigmas(50)
    assert s.shape == (50,) and s[0] == 1.0 and s[-1] == pytest.approx(1 / 50)


@pytest.mark.parametrize("steps,h,w", [(50, 1024, 1024), (4, 1024, 1024), (28, 512, 768), (1, 256, 256)])
def test_scheduler_matches_oracle_schedule(steps, h, w):
    sch = FlowMatchEulerDiscreteScheduler(shift=3.0, use_dynamic_shifting=True)
    sch.set_timesteps(sigmas=prepare_sigmas(steps), mu=prepare_mu(h, w), device="cpu")
    sig_ref, ts_ref = flux_oracle.flux_schedule(steps, h, w)
    assert torch.equal(sch.sigmas.cpu(), sig_ref)
    assert torch.equal(sch.timesteps.cpu(), ts_ref)
    assert sch.sigmas.shape == (steps + 1,) and sch.sigmas[-1] == 0.0
    assert sch.sigmas.dtype == torch.float32
    # shifted sigma: e^mu / (e^mu + (1/sigma - 1))
    mu = prepare_mu(h, w)
    s0 = prepare_sigmas(steps)
    expect = np.exp(mu) / (np.exp(mu) + (1 / s0 - 1))
    np.testing.assert_allclose(sch.sigmas[:-1].numpy(), expect.astype(np.float32), rtol=1e-6)


# ------------------------------------------------------------------ quantization plugin --
def test_fp8_config_from_config_and_registry():
    cls = Q.get_quantization_config("fp8")
    assert cls is Q.Fp8Config and cls.get_name() == "fp8"
    cfg = cls.from_config({"quant_method": "fp8", "activation_scheme": "dynamic",
                           "modules_to_not_convert": ["model.proj_out", "ff.net"]})
    assert cfg.is_checkpoint_fp8_serialized and cfg.weight_block_size is None
    assert cfg.ignored_layers == ["proj_out", "ff.net"]
    blk = cls.from_config({"quant_method": "fp8", "activation_scheme": "dynamic",
                           "weight_block_size": [128, 128]})
    assert blk.weight_block_size == [128, 128]
    with pytest.raises(ValueError):
        Q.Fp8Config(activation_scheme="per-token")
    with pytest.raises(ValueError):   # block quantisation needs an fp8-serialised checkpoint
        Q.Fp8Config(is_checkpoint_fp8_serialized=False, weight_block_size=[128, 128])
    with pytest.raises(ValueError):
        Q.Fp8Config(is_checkpoint_fp8_serialized=True, weight_block_size=[128])
    with pytest.raises(ValueError):
        cls.from_config({"activation_scheme": "dynamic"})
    with pytest.raises(ValueError):
        Q.get_quantization_config("awq")


def test_fp8_config_picks_the_method_per_layer():
    from sglang_diffusion_amd.layers.linear import ReplicatedLinear

    cfg = Q.Fp8Config(is_checkpoint_fp8_serialized=True, ignored_layers=["ff.net"])
    lin = ReplicatedLinear(256, 128, quant_config=cfg, params_dtype=torch.bfloat16,
                           prefix="transformer_blocks.0.attn.to_q")
    assert isinstance(lin.quant_method, Q.Fp8LinearMethod)
    assert lin.weight.dtype == torch.float8_e4m3fn and lin.weight.shape == (128, 256)
    assert lin.weight_scale.dtype == torch.float32
    skip = ReplicatedLinear(256, 128, quant_config=cfg, params_dtype=torch.bfloat16,
                            prefix="transformer_blocks.0.ff.net.2")
    assert isinstance(skip.quant_method, Q.UnquantizedLinearMethod)
    assert skip.weight.dtype == torch.bfloat16
    assert cfg.get_quant_method(torch.nn.LayerNorm(8), "x") is None
    blk = Q.Fp8Config(is_checkpoint_fp8_serialized=True, weight_block_size=[128, 128])
    lb = ReplicatedLinear(384, 256, quant_config=blk, params_dtype=torch.bfloat16,
                          prefix="single_transformer_blocks.0.proj_mlp")
    assert lb.weight_scale_inv.shape == (2, 3)
    # the product path has no CPU implementation: applying on CPU tensors must raise
    with pytest.raises(RuntimeError):
        skip(torch.zeros(2, 256, dtype=torch.bfloat16))


# --------------------------------------------------------------------- attention plugin --
def test_attention_selector_and_backend_interface():
    be = A.get_attn_backend(128, torch.bfloat16)
    assert issubclass(be, A.AttentionBackend)
    assert be is A.get_attn_backend(128, torch.bfloat16)            # cached
    assert be.get_enum() in (A.AttentionBackendEnum.AITER, A.AttentionBackendEnum.FA)
    assert issubclass(be.get_impl_cls(), A.AttentionImpl)
    assert issubclass(be.get_builder_cls(), A.AttentionMetadataBuilder)
    md = be.get_builder_cls()().build(current_timestep=3)
    assert isinstance(md, be.get_metadata_cls()) and md.current_timestep == 3
    with pytest.raises(ValueError):
        A.get_attn_backend(128, torch.float32)
    with pytest.raises(ValueError):
        A.get_attn_backend(128, torch.bfloat16, (A.AttentionBackendEnum.TORCH_SDPA,))
    with pytest.raises(ValueError):
        A.get_attn_backend_cls_str(A.AttentionBackendEnum.TORCH_SDPA, 128, torch.bfloat16)
    impl = be.get_impl_cls()(num_heads=4, head_size=128, softmax_scale=128 ** -0.5)
    q = torch.zeros(1, 8, 4, 128, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError):                               # CPU tensors: no fallback
        impl.forward(q, q, q, None)
    attn = A.USPAttention(num_heads=4, head_size=128)
    assert attn.softmax_scale == pytest.approx(128 ** -0.5)


# ------------------------------------------------------------------------ latent layout --
def test_pack_unpack_latents_and_ids():
    x = torch.randn(2, 16, 32, 48)
    p = P.pack_latents(x)
    assert p.shape == (2, 16 * 24, 64)
    # token (r, c) holds the 2x2 patch [c, dy, dx] flattened (configs/pipeline_configs/flux.py:92-108)
    assert torch.equal(p[1, 5 * 24 + 7], x[1, :, 10:12, 14:16].reshape(64))
    back = P.unpack_latents(p, 32 * 8, 48 * 8)
    assert torch.equal(back, x)
    ids = P.prepare_latent_image_ids(4 * 16, 6 * 16, "cpu")     # pixel sizes -> 4 x 6 tokens
    assert ids.shape == (24, 3) and ids[:, 0].abs().sum() == 0
    assert ids[6 + 2].tolist() == [0.0, 1.0, 2.0]
    assert tuple(P.prepare_latent_shape(3, 1024, 1024)) == (3, 16, 128, 128)


# ------------------------------------------------------------------ op registry (no GPU) --
def test_sgl_kernel_schemas_and_fake_impls():
    import sgl_kernel
    from sgl_kernel._ops import SCHEMAS

    for name in SCHEMAS:
        op = getattr(torch.ops.sgl_kernel, name)
        assert str(op.default._schema).startswith(f"sgl_kernel::{name}(")
    s = str(torch.ops.sgl_kernel.fp8_scaled_mm.default._schema)
    assert "ScalarType out_dtype" in s and "Tensor? bias" in s
    s = str(torch.ops.sgl_kernel.sgl_per_token_quant_fp8.default._schema)
    assert "output_q" in s and "-> ()" in s
    a = torch.empty(5, 256, device="meta", dtype=torch.float8_e4m3fn)
    b = torch.empty(64, 256, device="meta", dtype=torch.float8_e4m3fn).t()
    sc = torch.empty(5, device="meta")
    out = torch.ops.sgl_kernel.fp8_scaled_mm(a, b, sc, torch.empty(64, device="meta"), torch.bfloat16, None)
    assert out.shape == (5, 64) and out.dtype == torch.bfloat16 and out.device.type == "meta"
    q = torch.empty(2, 9, 4, 128, device="meta", dtype=torch.bfloat16)
    o, lse, *_ = sgl_kernel.flash_attn_varlen_func(q, q, q, None, None, 9, 9, return_softmax_lse=True)
    assert o.shape == q.shape and lse.shape == (2, 4, 9) and lse.dtype == torch.float32
    with pytest.raises(NotImplementedError):           # CPU tensors are never computed on
        sgl_kernel.sgl_per_token_quant_fp8(torch.zeros(2, 128, dtype=torch.bfloat16),
                                           torch.zeros(2, 128).to(torch.float8_e4m3fn), torch.zeros(2))


def test_gemm_tile_rule_on_the_flux_shapes():
    """The engine's mirror of the GEMM launcher's tile choice (it labels launches for the
    per-class timing): 256x256 tiles unless they would cover under three quarters of the CUs
    or fill their rounds clearly worse than 128x128 tiles fill theirs."""
    from sglang_diffusion_amd.models.flux_engine import _big_tile_rule

    cus = 256
    # image stream and single blocks at B = 4: always the big tile
    for (M, N, fp8) in [(16384, 9216, True), (18432, 12288, True), (18432, 3072, True),
                        (16384, 12288, False), (16384, 3072, False)]:
        assert _big_tile_rule(M, N, fp8, cus)
    # text stream (M = 2048): q|k|v has 288 big tiles = 1.125 rounds -> small; FF1 stays big
    assert not _big_tile_rule(2048, 9216, True, cus)
    assert _big_tile_rule(2048, 12288, False, cus)
    # too few big tiles to cover the chip
    assert not _big_tile_rule(2048, 3072, True, cus)
    assert not _big_tile_rule(2048, 3072, False, cus)
    assert not _big_tile_rule(5, 64, True, cus)
    # exactly full rounds of big tiles beat anything
    assert _big_tile_rule(256 * 16, 256 * 16, True, cus)


# ------------------------------------------------------------------- bench.py's launcher --
def _bench(*argv, env_extra=None):
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], env=env,
                          capture_output=True, text=True, timeout=300)


def test_bench_starts_its_own_rank_processes():
    """`python bench.py --gpus N` must start N rank processes itself (no external launcher):
    they rendezvous on 127.0.0.1, see RANK / WORLD_SIZE, and rank 0 prints the one JSON line."""
    import json

    r = _bench("--gpus", "3", "--launch-check")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]   # gloo chats on stdout too
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out == {"launch_check": True, "n_gpus": 3, "rank_sum": 0 + 1 + 2, "local_rank": 0}
    # under an external launcher (WORLD_SIZE set) the process is a rank and starts nothing
    r = _bench("--gpus", "1", "--launch-check", env_extra={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode == 0 and json.loads(r.stdout.splitlines()[-1])["n_gpus"] == 1


def test_bench_launcher_reports_a_failing_rank():
    r = _bench("--gpus", "2", "--launch-check", "--launch-check-fail-rank", "1")
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    assert "rank 1 exited with 3" in r.stderr
