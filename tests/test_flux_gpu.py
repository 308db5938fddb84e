"""End-to-end parity of the Flux DiT forward and the denoise loop against the CPU oracle on a
small Flux-shaped model (same block code, same weights on both sides), plus size-independent
properties at the real Flux.1 shapes."""
import pytest
import torch

from tests.flux_small import build_oracle, build_small, small_inputs
from tests.helpers import bf16_ulp_diff, rel_l2

pytestmark = pytest.mark.gpu

# Every op except attention matches the oracle to <= 1e-3 relative (most bit-exactly, see
# test_kernels_gpu.py).  The attention kernel rounds the softmax probabilities to bf16 before
# the P.V MFMA, as every flash-attention implementation the reference calls does (FA3, aiter,
# its own CPU kernel); against an fp32-softmax oracle that costs ~2e-3 per attention call (the
# reference's own FA test allows 2x the bf16 round-trip error,
# sgl-kernel/tests/test_flash_attention.py:1273-1308), and a 4-block model stacks four of them.
TOL_BF16 = 8e-3
# W8A8: activations are re-quantised to e4m3 (3 mantissa bits) after every norm; a one-ulp
# bf16 difference upstream can move an fp8 rounding.
TOL_FP8 = 2e-2


def _freqs(oracle, Tt, H, W):
    return oracle.freqs_cis(Tt, H, W)


@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("guidance", [True, False])
def test_dit_forward_module_and_fused_vs_oracle(device, fp8, guidance):
    model, state, cfg = build_small(device, fp8=fp8, guidance_embeds=guidance)
    oracle = build_oracle(state, cfg, fp8)
    B, Hpx, Wpx, Tt = 2, 128, 96, 24
    Ti = (Hpx // 16) * (Wpx // 16)
    hid, enc, pooled, t, guid = small_inputs(B, Ti, Tt, cfg)
    if not guidance:
        guid = None
    freqs = _freqs(oracle, Tt, Hpx, Wpx)
    ref, inter = oracle.forward(hid, enc, pooled, t, guid, freqs, return_intermediates=True)
    dev = lambda x: None if x is None else x.to(device)  # noqa: E731
    fd = (freqs[0].to(device), freqs[1].to(device))
    # the product builds the same RoPE table as the reference formula
    from sglang_diffusion_amd.runtime import pipeline_config as pc
    pf = pc.get_freqs_cis(Tt, Wpx, Hpx, device, model.rotary_emb)
    assert torch.equal(pf[0].cpu(), freqs[0]) and torch.equal(pf[1].cpu(), freqs[1])
    with torch.no_grad():
        out_mod = model(dev(hid), dev(enc), dev(pooled), dev(t), dev(guid), fd)
        out_fused = model.forward_fused(dev(hid), dev(enc), dev(pooled), dev(t), dev(guid), fd).clone()
    tol = TOL_FP8 if fp8 else TOL_BF16
    e_mod, e_fused = rel_l2(out_mod, ref), rel_l2(out_fused, ref)
    print(f"fp8={fp8} guidance={guidance}: module rel {e_mod:.2e}, fused rel {e_fused:.2e}, "
          f"fused-vs-module {rel_l2(out_fused, out_mod):.2e}")
    assert e_mod <= tol and e_fused <= tol
    assert rel_l2(out_fused, out_mod) <= tol


def test_online_weight_quantisation_is_bit_exact(device):
    from oracle import ops
    model, state, cfg = build_small(device, fp8=True)
    lin = model.transformer_blocks[0].attn.to_q
    qr, sr = ops.quantize_weight_per_channel(state["transformer_blocks.0.attn.to_q.weight"])
    assert torch.equal(lin.weight_scale.cpu(), sr)
    assert torch.equal(lin.weight.data.cpu().view(torch.uint8), qr.view(torch.uint8))
    pm = model.single_transformer_blocks[1].proj_out
    qr, sr = ops.quantize_weight_per_channel(state["single_transformer_blocks.1.proj_out.weight"])
    assert torch.equal(pm.weight_scale.cpu(), sr)
    assert torch.equal(pm.weight.data.cpu().view(torch.uint8), qr.view(torch.uint8))
    # never-quantised layers keep bf16 weights (converter skip list, convert_hf_to_fp8.py:139-160)
    assert model.transformer_blocks[0].ff.net[2].weight.dtype == torch.bfloat16
    assert model.x_embedder.weight.dtype == torch.bfloat16
    assert model.transformer_blocks[0].norm1.linear.weight.dtype == torch.bfloat16


@pytest.mark.parametrize("fp8,true_cfg", [(False, False), (True, True)])
def test_denoise_loop_vs_oracle(device, fp8, true_cfg):
    """Identical latents + timestep schedule, 3 Euler steps (with serial-CFG semantics)."""
    from oracle import flux_oracle
    from sglang_diffusion_amd.runtime.denoise import DenoisingStage
    from sglang_diffusion_amd.runtime.synthetic import make_synthetic_request

    model, state, cfg = build_small(device, fp8=fp8)
    oracle = build_oracle(state, cfg, fp8)
    req = make_synthetic_request(2, height=96, width=128, num_inference_steps=3, device="cpu",
                                 seed=5, true_cfg=true_cfg, text_len=24,
                                 joint_dim=cfg.joint_attention_dim,
                                 pooled_dim=cfg.pooled_projection_dim)
    ref = flux_oracle.denoise(oracle, req.latents.clone(), req.prompt_embeds, req.pooled_embeds,
                              height=96, width=128, num_steps=3,
                              neg_embeds=req.negative_prompt_embeds, neg_pooled=req.neg_pooled_embeds,
                              guidance_scale=req.guidance_scale)
    for name in ("latents", "prompt_embeds", "pooled_embeds", "negative_prompt_embeds",
                 "neg_pooled_embeds"):
        v = getattr(req, name)
        if v is not None:
            setattr(req, name, v.to(device))
    stage = DenoisingStage(model, fused=True, cfg_batched=True)
    # schedule parity (sigmas / timesteps are host arithmetic: exact)
    stage.prepare_timesteps(req, device)
    sig, ts = flux_oracle.flux_schedule(3, 96, 128)
    assert torch.equal(stage.scheduler.sigmas, sig) and torch.equal(stage.scheduler._timesteps_host, ts)
    out = stage.forward(req).latents
    e = rel_l2(out, ref)
    print(f"denoise fp8={fp8} cfg={true_cfg}: rel {e:.2e}")
    # true CFG extrapolates u + s*(c-u) with s = 3.5: differences are amplified by ~s
    assert e <= (TOL_FP8 if fp8 else TOL_BF16) * (3.5 if true_cfg else 1.0)
    # batched CFG == two separate passes
    if true_cfg:
        req2 = make_synthetic_request(2, height=96, width=128, num_inference_steps=3, device=device,
                                      seed=5, true_cfg=True, text_len=24,
                                      joint_dim=cfg.joint_attention_dim,
                                      pooled_dim=cfg.pooled_projection_dim)
        out2 = DenoisingStage(model, fused=True, cfg_batched=False).forward(req2).latents
        assert torch.equal(out2, out)


def test_denoise_loop_with_cfg_normalization_and_guidance_rescale(device):
    """True CFG followed by the reference's two optional post-combine steps (norm cap, std
    rescale), 3 steps against the oracle's literal restatement of them."""
    from oracle import flux_oracle
    from sglang_diffusion_amd.runtime.denoise import DenoisingStage
    from sglang_diffusion_amd.runtime.synthetic import make_synthetic_request

    model, state, cfg = build_small(device, fp8=False)
    oracle = build_oracle(state, cfg, False)
    req = make_synthetic_request(2, height=96, width=128, num_inference_steps=3, device="cpu",
                                 seed=9, true_cfg=True, text_len=24,
                                 joint_dim=cfg.joint_attention_dim,
                                 pooled_dim=cfg.pooled_projection_dim)
    req.cfg_normalization, req.guidance_rescale = 1.0, 0.7
    kw = dict(height=96, width=128, num_steps=3, neg_embeds=req.negative_prompt_embeds,
              neg_pooled=req.neg_pooled_embeds, guidance_scale=req.guidance_scale)
    ref = flux_oracle.denoise(oracle, req.latents.clone(), req.prompt_embeds, req.pooled_embeds,
                              cfg_normalization=1.0, guidance_rescale=0.7, **kw)
    plain = flux_oracle.denoise(oracle, req.latents.clone(), req.prompt_embeds, req.pooled_embeds,
                                **kw)
    assert rel_l2(ref, plain) > 1e-2, "the two options must matter in this set-up"
    for name in ("latents", "prompt_embeds", "pooled_embeds", "negative_prompt_embeds",
                 "neg_pooled_embeds"):
        setattr(req, name, getattr(req, name).to(device))
    out = DenoisingStage(model, fused=True).forward(req).latents
    e = rel_l2(out, ref)
    print(f"denoise with cfg_normalization + guidance_rescale: rel {e:.2e}")
    assert e <= TOL_BF16 * 3.5


def test_batch_rows_are_independent_at_flux_shapes(device):
    """Size-independent property at the real sequence length (S = 512 + 4096) on a 1+1-block
    model of the real width: a sample's output does not depend on what else is in the batch,
    and permuting the batch permutes the output."""
    model, state, cfg = build_small(device, fp8=True, num_layers=1, num_single_layers=1,
                                    num_attention_heads=24, joint_attention_dim=4096,
                                    pooled_projection_dim=768)
    B, Ti, Tt = 3, 4096, 512
    hid, enc, pooled, t, guid = small_inputs(B, Ti, Tt, cfg, seed=9)
    from sglang_diffusion_amd.runtime import pipeline_config as pc
    freqs = pc.get_freqs_cis(Tt, 1024, 1024, device, model.rotary_emb)
    d = lambda x: x.to(device)  # noqa: E731
    with torch.no_grad():
        full = model.forward_fused(d(hid), d(enc), d(pooled), d(t), d(guid), freqs).clone()
        one = model.forward_fused(d(hid[1:2]), d(enc[1:2]), d(pooled[1:2]), d(t[1:2]), d(guid[1:2]),
                                  freqs).clone()
        perm = torch.tensor([2, 0, 1])
        permd = model.forward_fused(d(hid[perm]), d(enc[perm]), d(pooled[perm]), d(t[perm]),
                                    d(guid[perm]), freqs).clone()
    assert torch.isfinite(full.float()).all()
    assert torch.equal(full[1:2], one)
    assert torch.equal(permd, full[perm.to(device)])


def test_dp_batch_scheduler_on_the_hip_path(device):
    """The data-parallel batch scheduler (one rank here) drains a queue of requests of different
    batch sizes, with and without true CFG, in micro-batches; every request's latents equal the
    ones it gets when run alone (batch rows are independent, so this is exact)."""
    from sglang_diffusion_amd.runtime.batch_scheduler import DataParallelBatchScheduler
    from sglang_diffusion_amd.runtime.denoise import DenoisingStage
    from sglang_diffusion_amd.runtime.synthetic import make_synthetic_request

    model, state, cfg = build_small(device, fp8=True)

    def req(b, seed, cfg_pair=False):
        r = make_synthetic_request(b, height=128, width=128, num_inference_steps=3, device=device,
                                   seed=seed, text_len=32, joint_dim=cfg.joint_attention_dim,
                                   pooled_dim=cfg.pooled_projection_dim)
        if cfg_pair:
            r.guidance_scale = 3.5
            r.negative_prompt_embeds = torch.zeros_like(r.prompt_embeds)
            r.neg_pooled_embeds = torch.zeros_like(r.pooled_embeds)
        return r

    specs = [(1, 1, False), (2, 2, False), (1, 3, True), (1, 4, False), (1, 5, True), (1, 6, False)]
    alone = []
    for b, seed, pair in specs:
        alone.append(DenoisingStage(model, fused=True).forward(req(b, seed, pair)).latents.clone())
    sch = DataParallelBatchScheduler(DenoisingStage(model, fused=True), max_samples_per_step=4)
    for b, seed, pair in specs:
        sch.add_request(req(b, seed, pair))
    outs = sch.run()
    assert len(outs) == len(specs)
    for got, exp in zip(outs, alone):
        assert torch.equal(got, exp)


@pytest.mark.parametrize("overlap", [False, True])
def test_fused_step_replays_from_a_hip_graph(device, overlap):
    """A serving loop can capture the fused DiT forward once and replay it: every launch goes
    to the capturing stream, nothing synchronises, and the workspace is allocated before the
    capture.  The replay on NEW input values equals the eager call bit for bit."""
    model, state, cfg = build_small(device, fp8=True)
    B, Hpx, Wpx, Tt = 2, 128, 96, 24
    Ti = (Hpx // 16) * (Wpx // 16)
    oracle = build_oracle(state, cfg, True)
    freqs = _freqs(oracle, Tt, Hpx, Wpx)
    fd = (freqs[0].to(device), freqs[1].to(device))
    a = [x.to(device) for x in small_inputs(B, Ti, Tt, cfg, seed=1)]
    b = [x.to(device) for x in small_inputs(B, Ti, Tt, cfg, seed=2)]
    from sglang_diffusion_amd.models.flux_engine import FluxEngine

    model._engine = model._engine or FluxEngine(model)
    model._engine.overlap_text = False
    with torch.no_grad():
        want_b = model.forward_fused(*b, fd).clone()        # also warms up: workspace, attributes
        if overlap:
            # text chain of the double blocks on the engine's second stream: forks from and
            # rejoins the capturing stream, same bits as the one-stream schedule
            model._engine.overlap_text = True
            assert torch.equal(model.forward_fused(*b, fd), want_b)
        static = [x.clone() for x in a]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                         # warm-up on the capture stream
            model.forward_fused(*static, fd)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = model.forward_fused(*static, fd)
        want_a = model.forward_fused(*a, fd).clone()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want_a)
        for s, x in zip(static, b):
            s.copy_(x)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want_b)
