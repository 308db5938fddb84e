#!/usr/bin/env python
"""Benchmark of the Flux.1-dev denoise step on MI355X (BASELINE.json's metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one denoise step (DiT forward on the fused path + flow-match Euler update) of a
batch of B = 4 synthetic 1024x1024 requests per GPU, fp8 W8A8 attention/single-block linears,
bf16 everywhere else: the configuration the metric is quoted on.  With N > 1 every rank owns
its own batch (data parallel over requests, no collective inside a step): weak scaling, the
reported value is the whole-job images/s = N*B / (50 * step time).

Launch.  `python bench.py --gpus N` with N > 1 starts its own N rank processes (one per GPU,
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment, tcp rendezvous
on 127.0.0.1 as the reference's per-GPU scheduler processes do,
python/sglang/multimodal_gen/runtime/launch_server.py:61-187, managers/gpu_worker.py:92-111) and
waits for them; the parent never imports torch or touches the GPU.  Under an external launcher
(torch.distributed.run: WORLD_SIZE already set) the process IS a rank.

One JSON line is printed by rank 0.  Besides the contract fields it carries
  roofline     : the dominant kernel (the 256x256-tile fp8 MFMA GEMM), algorithmic FLOP per
                 launch / its average duration measured with events on the launch stream,
                 against the dense fp8 MFMA peak;
  cpu_baseline : the CPU oracle timed on this host on a bounded sample of the same workload;
  c4           : the fixed-size job "32 requests with true CFG (guidance 3.5, negative
                 prompt) = 64 DiT samples per step" drained through DataParallelBatchScheduler
                 over the N ranks, requests sent from rank 0 and final latents gathered back
                 inside the timed region: STRONG scaling (the same 32 images at every N).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sglang-diffusion_amd"))

# torch is imported by the rank processes only (inside main): the launcher parent must stay
# free of anything that could initialise the GPU before its children start

PEAK_FP8_TFLOPS = 5000.0   # dense fp8 MFMA, MI355X
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA
PEAK_HBM_GBS = 8000.0
NUM_INFERENCE_STEPS = 50


def flux_step_flops(B: int, S: int = 4608, D: int = 3072, n_double: int = 19, n_single: int = 38,
                    s_txt: int = 512):
    """Algorithmic FLOP of one DiT forward: linear 24*D^2*S per block, attention 4*S^2*D."""
    fp8 = (8 * D * D * S * n_double + 24 * D * D * S * n_single) * B
    bf16 = 16 * D * D * S * n_double * B
    attn = 4 * S * S * D * (n_double + n_single) * B
    return dict(fp8=float(fp8), bf16=float(bf16), attn=float(attn), total=float(fp8 + bf16 + attn))


def launch_ranks(n: int, argv, timeout_s: float = 3300.0, script: str = None) -> int:
    """Start `n` copies of this script (or of `script`) as rank processes and wait for them.  Returns the exit
    code for the parent: 0 when every rank exited 0, else the first failing rank's code (the
    remaining ranks are terminated by PID).  Rank 0 inherits stdout (it prints the JSON line);
    the other ranks' stdout goes to stderr."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(script or __file__)] + list(argv),
                                      env=env, stdout=None if r == 0 else sys.stderr))
    deadline = time.time() + timeout_s
    rc = 0
    rank0_done = False
    live = dict(enumerate(procs))
    while live:
        for r, p in list(live.items()):
            code = p.poll()
            if code is None:
                continue
            del live[r]
            if code != 0 and rc == 0:
                rc = code if code > 0 else 128 - code
                print(f"[bench] rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
            if r == 0 and code == 0:
                # the result line is out: the other ranks get a minute to leave their last
                # collective, not the whole time limit
                deadline = min(deadline, time.time() + 60.0)
                rank0_done = True
        if (rc != 0 or time.time() > deadline) and live:
            for p in live.values():
                p.terminate()
            for p in live.values():
                try:
                    p.wait(20)
                except subprocess.TimeoutExpired:
                    p.kill()
            if rc == 0 and not rank0_done:
                rc = 124
            break
        time.sleep(0.05)
    return rc


def cpu_baseline(seconds_hint: float = 20.0):
    """Oracle (CPU restatement of the reference eager path) on a bounded sample: one double +
    one single block of the FULL-SIZE Flux.1-dev DiT at B = 1, fp8 linears emulated as the
    reference does on CPU.  Scaled to a whole step by algorithmic FLOP."""
    import torch

    from oracle.flux_oracle import FluxOracle

    torch.manual_seed(0)
    ncores = os.cpu_count() or 1
    threads = min(ncores, int(os.environ.get("OMP_NUM_THREADS", ncores)))
    torch.set_num_threads(threads)
    D, H, dh, S_img, S_txt = 3072, 24, 128, 4096, 512
    g = torch.Generator().manual_seed(0)

    def w(n, k):
        return (torch.randn(n, k, generator=g) * k ** -0.5).bfloat16()

    sd = {}
    def lin(name, n, k):
        sd[name + ".weight"] = w(n, k)
        sd[name + ".bias"] = (torch.randn(n, generator=g) * 0.02).bfloat16()
    p = "transformer_blocks.0"
    for nm in ("norm1", "norm1_context"):
        lin(f"{p}.{nm}.linear", 6 * D, D)
    for nm in ("to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj", "to_out.0",
               "to_add_out"):
        lin(f"{p}.attn.{nm}", D, D)
    for nm in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
        sd[f"{p}.attn.{nm}.weight"] = torch.ones(dh)
    for ff in ("ff", "ff_context"):
        lin(f"{p}.{ff}.net.0.proj", 4 * D, D)
        lin(f"{p}.{ff}.net.2", D, 4 * D)
    q = "single_transformer_blocks.0"
    lin(f"{q}.norm.linear", 3 * D, D)
    for nm in ("to_q", "to_k", "to_v"):
        lin(f"{q}.attn.{nm}", D, D)
    for nm in ("norm_q", "norm_k"):
        sd[f"{q}.attn.{nm}.weight"] = torch.ones(dh)
    lin(f"{q}.proj_mlp", 4 * D, D)
    lin(f"{q}.proj_out", D, 5 * D)
    orc = FluxOracle(sd, num_layers=1, num_single_layers=1, num_heads=H, head_dim=dh,
                     guidance_embeds=True, fp8=True)
    hid = torch.randn(1, S_img, D, generator=g).bfloat16()
    enc = torch.randn(1, S_txt, D, generator=g).bfloat16()
    temb = torch.randn(1, D, generator=g).bfloat16()
    freqs = orc.freqs_cis(S_txt, 1024, 1024)
    S = S_img + S_txt
    pair_flops = 24 * D * D * S * 2 + 4 * S * S * D * 2
    reps = 0
    t0 = time.perf_counter()
    with torch.no_grad():
        # the block pair is repeated until about `seconds_hint / 2` of host time has gone by
        while reps < 16 and (reps == 0 or time.perf_counter() - t0 < 0.5 * seconds_hint):
            e1, h1 = orc.double_block(0, hid, enc, temb, freqs)
            orc.single_block(0, h1, e1, temb, freqs)
            reps += 1
    dt = time.perf_counter() - t0
    sample_flops = pair_flops * reps
    full = flux_step_flops(1)["total"]
    step_s = dt * full / sample_flops
    return {
        "value": 1.0 / (NUM_INFERENCE_STEPS * step_s), "unit": "images/s", "cores": threads,
        "kind": "port",
        "sample": (f"{reps} x (1 double + 1 single block) of the full-size Flux.1-dev DiT, B=1, S=4608, "
                   f"fp8-emulated linears: {dt:.1f} s for {sample_flops / 1e12:.2f} of "
                   f"{full / 1e12:.1f} TFLOP per step, scaled by FLOP to {step_s:.0f} s/step"),
    }


class PhaseTimer:
    """Per-phase device time of a step: `with timer(name):` brackets a region with two events
    on the current stream; `totals()` sums the regions by name after a synchronisation."""

    def __init__(self):
        self.spans = []

    def __call__(self, name):
        import contextlib

        import torch

        @contextlib.contextmanager
        def span():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            try:
                yield
            finally:
                e1.record()
                self.spans.append((name, e0, e1))
        return span()

    def totals(self):
        import torch

        torch.cuda.synchronize()
        out = {}
        for name, e0, e1 in self.spans:
            out[name] = out.get(name, 0.0) + e0.elapsed_time(e1)
        self.spans = []
        return out


def run_c5(args):
    """BASELINE config 5: one denoise step of the HunyuanVideo DiT (20 double + 40 single
    blocks, 24 heads of 128) on 720p x 129 frames -- latents [1, 16, 33, 90, 160], patch
    (1, 2, 2): 118 800 image tokens + 256 text tokens -- with Ulysses sequence parallelism over
    the N ranks (reference: models/dits/hunyuanvideo.py:49-410, layers/attention/layer.py:38-154,
    layers/usp.py:49-158).  Each rank holds a full weight replica and the tokens of its frames
    (the frame count is padded up to a multiple of N with zero frames, as the reference's
    shard_latents_for_sp does, configs/pipeline_configs/base.py:353-380); every block does one
    all-to-all of q|k|v (heads out, sequence in), attention of H / N heads over the whole
    sequence, one all-to-all back and an all-gather of the text heads.  A "step" is the DiT
    forward on the rank's shard plus the flow-match Euler update of that shard; nothing else
    is exchanged.  Strong scaling: the same video at every N."""
    import torch
    import torch.distributed as dist

    from sglang_diffusion_amd import _C
    from sglang_diffusion_amd.layers import attention as attn_mod
    from sglang_diffusion_amd.layers.attention import Mi355AttentionMetadataBuilder
    from sglang_diffusion_amd.layers.forward_context import set_forward_context
    from sglang_diffusion_amd.layers.quantization import Fp8Config
    from sglang_diffusion_amd.models.hunyuanvideo import (HunyuanVideoArchConfig,
                                                          HunyuanVideoTransformer3DModel,
                                                          process_weights_after_loading)
    from sglang_diffusion_amd.runtime.parallel import init_distributed
    from sglang_diffusion_amd.runtime.synthetic import hashed_tensor, init_synthetic_weights

    backend = os.environ.get("MI355_BENCH_BACKEND", "nccl")
    rank, world, local = init_distributed(backend if int(os.environ.get("WORLD_SIZE", "1")) > 1
                                          else None)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no HIP device visible)")
    dev = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    nd, ns = args.layers if args.layers else (20, 40)
    cfg = HunyuanVideoArchConfig(num_layers=nd, num_single_layers=ns, guidance_embeds=True)
    if cfg.num_attention_heads % world:
        raise SystemExit(f"{cfg.num_attention_heads} heads do not divide over {world} ranks")
    qc = None if args.c5_bf16 else Fp8Config(is_checkpoint_fp8_serialized=False,
                                            activation_scheme="dynamic")
    group = dist.group.WORLD if world > 1 else None
    with torch.device(dev):
        model = HunyuanVideoTransformer3DModel(cfg, quant_config=qc, sp_group=group)
    init_synthetic_weights(model, seed=0, host_rng=False)   # same seed on every rank: replicas
    process_weights_after_loading(model)
    for p in model.parameters():
        p.requires_grad_(False)

    frames = args.c5_frames
    padded = -(-frames // world) * world
    fl = padded // world
    h, w = args.c5_hw
    # the video's latents are a function of (frame, position): every rank draws the whole
    # tensor's values for ITS frames only
    full = hashed_tensor("c5.latents", (1, 16, padded, h, w), 0, 1.0, dev)
    if padded > frames:
        full[:, :, frames:] = 0
    x = full[:, :, rank * fl:(rank + 1) * fl].contiguous()
    del full
    enc = hashed_tensor("c5.text", (1, 1 + 256, cfg.text_embed_dim), 0, 0.1, dev)
    guid = torch.tensor([6016.0], device=dev).bfloat16()
    sig = [1.0 - 0.02 * i for i in range(51)]
    md = Mi355AttentionMetadataBuilder().build(current_timestep=0)
    timer = PhaseTimer()
    step_no = [0]

    @torch.no_grad()
    def one_step():
        nonlocal x
        i = step_no[0] % 50
        t = torch.tensor([sig[i] * 1000.0], device=dev)
        with set_forward_context(current_timestep=i, attn_metadata=md):
            v = model(x, enc, t, guid)
        nxt = torch.empty_like(x)
        _C.euler_step(nxt, x, v.contiguous(), sig[i], sig[i + 1])
        x = nxt
        step_no[0] += 1

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    barrier()
    elapsed = time.perf_counter() - t0
    # one more step with the phases bracketed by events (outside the timed region)
    attn_mod.PHASE_TIMER = timer
    one_step()
    phases = timer.totals()
    attn_mod.PHASE_TIMER = None
    ph = torch.tensor([elapsed, phases.get("all_to_all", 0.0), phases.get("attention", 0.0),
                       phases.get("all_gather", 0.0)], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(ph, op=dist.ReduceOp.MAX)
    elapsed, a2a_ms, attn_ms, ag_ms = ph.tolist()
    assert bool(torch.isfinite(x.float()).all())
    step_s = elapsed / args.steps
    S_img, S_txt, D = padded * (h // 2) * (w // 2), 256, cfg.hidden_size
    S = S_img + S_txt
    blocks = nd + ns
    flops_attn = 4.0 * S * S * D * blocks
    flops_lin = 24.0 * D * D * S * blocks
    if rank == 0:
        print(json.dumps({
            "metric": "ms_per_step_hunyuanvideo_720p_129f_ulysses", "value": step_s * 1e3,
            "unit": "ms", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": step_s * 1e3, "higher_is_better": False, "scaling": "strong",
            "vs_baseline": None,
            "dtype": "bf16" if args.c5_bf16 else "fp8_e4m3 (W8A8 linears) + bf16 (attention)",
            "data": "synthetic",
            "config": {"workload": f"HunyuanVideo DiT, latents [1,16,{frames},{h},{w}] "
                                   f"({frames} -> {padded} frames over {world} ranks), {S_img} image + "
                                   f"{S_txt} text tokens, {nd}+{ns} blocks [NOT the metric config]",
                       "parallelism": f"ulysses{world}", "seq_len": S,
                       "tokens_per_rank": S_img // world, "heads_per_rank": cfg.num_attention_heads // world},
            "step_tflops_total": (flops_attn + flops_lin) / step_s / 1e12,
            "phases_ms_per_step": {"all_to_all": a2a_ms, "attention": attn_ms, "all_gather": ag_ms,
                                   "linears_and_elementwise": max(0.0, step_s * 1e3 - a2a_ms - attn_ms
                                                                  - ag_ms)},
            "phases_note": "event-bracketed regions of one extra step after the timed ones, MAX "
                           "over ranks; linears_and_elementwise = ms_per_step minus the three",
            "attention_tflops_per_gpu": flops_attn / world / (attn_ms * 1e-3) / 1e12 if attn_ms else None,
        }), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=4, help="requests per GPU")
    ap.add_argument("--mode", default="fp8", choices=["fp8", "bf16", "fp8-block"],
                    help="fp8: per-channel W8A8 quantised on the device (the metric); fp8-block: an "
                         "fp8-serialised checkpoint with 128x128 weight blocks (NOT the metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap-text", action="store_true",
                    help="keep the double blocks' text-stream chain on the main stream (by default "
                         "it runs on a second HIP stream in the timed steps: same results; the "
                         "per-kernel accounting after the timed region is always un-overlapped)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the DiT forward of a step from a captured HIP graph (static input "
                         "buffers; the scheduler update stays outside the graph)")
    ap.add_argument("--schnell", action="store_true",
                    help="Flux.1-schnell: no guidance embedder, bf16, batch 1, 4-step schedule "
                         "(a parity-test configuration; the result is NOT the metric)")
    ap.add_argument("--checkpoint", default=None,
                    help="a diffusers `transformer/` directory (config.json + *.safetensors) to run "
                         "instead of random-init weights; \"data\" then says so")
    ap.add_argument("--layers", type=int, nargs=2, default=None,
                    help="(debug) override #double #single blocks; the result is then NOT the metric")
    ap.add_argument("--no-c4", action="store_true",
                    help="skip the fixed-size 32-CFG-request job through the batch scheduler")
    ap.add_argument("--c4-requests", type=int, default=32)
    ap.add_argument("--c4-timeout", type=float, default=600.0,
                    help="seconds after which the line is printed without the c4 object")
    ap.add_argument("--c4-steps", type=int, default=None,
                    help="denoise steps per micro-batch of the c4 job (default: --steps)")
    ap.add_argument("--workload", default="flux", choices=["flux", "c5"],
                    help="flux: the metric (and its c4 job); c5: one HunyuanVideo 720p x 129-frame "
                         "denoise step with Ulysses sequence parallelism over the N ranks (NOT the "
                         "metric; prints its own JSON line with per-phase times)")
    ap.add_argument("--c5-frames", type=int, default=33,
                    help="latent frames of the c5 workload (33 = 129 video frames; padded up to a "
                         "multiple of N as the reference does)")
    ap.add_argument("--c5-hw", type=int, nargs=2, default=[90, 160],
                    help="latent height and width of the c5 workload (720p: 90 x 160)")
    ap.add_argument("--c5-bf16", action="store_true", help="c5 with bf16 linears instead of W8A8")
    ap.add_argument("--launch-check", action="store_true",
                    help="only start the ranks, rendezvous over gloo on the host, all-reduce the "
                         "rank numbers and print {launch_check, n_gpus, rank_sum}: tests the "
                         "launcher where there is no GPU")
    ap.add_argument("--launch-check-fail-rank", type=int, default=-1, help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    if args.launch_check:
        from sglang_diffusion_amd.runtime.parallel import init_distributed

        rank, world, local = init_distributed("gloo")
        if rank == args.launch_check_fail_rank:
            raise SystemExit(3)
        t = torch.tensor([rank], dtype=torch.int64)
        if world > 1:
            dist.all_reduce(t)
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "rank_sum": int(t.item()),
                              "local_rank": local}))
        return

    if args.workload == "c5":
        return run_c5(args)

    from sglang_diffusion_amd.layers.quantization import Fp8Config
    from sglang_diffusion_amd.models.flux import FluxTransformer2DModel, flux_dev_config
    from sglang_diffusion_amd.runtime.denoise import DenoisingStage
    from sglang_diffusion_amd.runtime.parallel import init_distributed
    from sglang_diffusion_amd.runtime.synthetic import (init_synthetic_weights,
                                                        make_synthetic_request)

    # RCCL ("nccl") between the ranks of one node; MI355_BENCH_BACKEND=gloo lets several ranks
    # share one GPU for a rehearsal of the multi-rank control flow
    backend = os.environ.get("MI355_BENCH_BACKEND", "nccl")
    rank, world, local = init_distributed(backend if int(os.environ.get("WORLD_SIZE", "1")) > 1
                                          else None)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no HIP device visible)")
    dev = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(dev)

    cfg = flux_dev_config()
    global NUM_INFERENCE_STEPS
    if args.schnell:
        from sglang_diffusion_amd.models.flux import flux_schnell_config

        cfg = flux_schnell_config()
        args.mode, args.batch, NUM_INFERENCE_STEPS = "bf16", 1, 4
    if args.layers:
        cfg.num_layers, cfg.num_single_layers = args.layers
    qc = (Fp8Config(is_checkpoint_fp8_serialized=False) if args.mode == "fp8" else
          Fp8Config(is_checkpoint_fp8_serialized=True, weight_block_size=[128, 128])
          if args.mode == "fp8-block" else None)
    if args.checkpoint:
        from sglang_diffusion_amd.runtime.loader import get_component_config, load_flux_transformer

        declared = "quantization_config" in get_component_config(args.checkpoint)
        model, _ = load_flux_transformer(args.checkpoint, dev, quant_config=None if declared else qc)
        cfg = model.config
    else:
        with torch.device(dev):
            model = FluxTransformer2DModel(cfg, quant_config=qc)
        if args.mode == "fp8-block":
            from sglang_diffusion_amd.runtime.synthetic import init_synthetic_block_fp8

            init_synthetic_block_fp8(model, seed=0)
        else:
            init_synthetic_weights(model, seed=0, host_rng=False)
        model.process_weights_after_loading()
    stage = DenoisingStage(model, fused=True)
    from sglang_diffusion_amd.models.flux_engine import FluxEngine

    model._engine = model._engine or FluxEngine(model)
    if args.no_overlap_text:
        model._engine.overlap_text = False
        model._engine.overlap_mlp = False
    B = args.batch
    req = make_synthetic_request(B, 1024, 1024, NUM_INFERENCE_STEPS, device=dev, seed=rank)
    timesteps = stage.prepare_timesteps(req, dev)
    guidance = stage._build_guidance(B, dev, req.embedded_cfg_scale) if cfg.guidance_embeds else None
    freqs = stage._freqs_cis(req.prompt_embeds.shape[1], req.width, req.height, dev)
    ts_host = stage.scheduler._timesteps_host
    latents = req.latents
    step_no = [0]

    @torch.no_grad()
    def one_step():
        nonlocal latents
        i = step_no[0] % NUM_INFERENCE_STEPS
        if i == 0:
            stage.scheduler._step_index = None
        t = timesteps[i].repeat(B)
        noise = stage.predict_noise_with_cfg(req, latents.to(torch.bfloat16), t, guidance, freqs, i)
        latents = stage.scheduler.step(noise, ts_host[i], latents)[0]
        step_no[0] += 1

    if args.graph:
        # the forward is captured once on static buffers; a step copies its inputs in and replays
        static_lat = latents.to(torch.bfloat16).clone()
        static_t = timesteps[0].repeat(B).clone()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(2):
                stage.predict_noise_with_cfg(req, static_lat, static_t, guidance, freqs, 0)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph), torch.no_grad():
            static_noise = stage.predict_noise_with_cfg(req, static_lat, static_t, guidance, freqs, 0)
        eager_step = one_step

        @torch.no_grad()
        def one_step():
            nonlocal latents
            if model._engine.trace is not None:  # per-launch events cannot sit inside a replay
                return eager_step()
            i = step_no[0] % NUM_INFERENCE_STEPS
            if i == 0:
                stage.scheduler._step_index = None
            static_lat.copy_(latents)
            static_t.copy_(timesteps[i].expand(B))
            graph.replay()
            latents = stage.scheduler.step(static_noise, ts_host[i], latents)[0]
            step_no[0] += 1

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = tmax.item()
    step_s = elapsed / args.steps
    images_per_s = world * B / (NUM_INFERENCE_STEPS * step_s)

    # ---- per-kernel-class timing with events on the launch stream (2 traced steps) -----------
    eng = model._engine
    eng.trace = []
    one_step()
    one_step()
    torch.cuda.synchronize()
    classes = {}
    for kind, flops, nbytes, e0, e1 in eng.trace:
        c = classes.setdefault(kind, dict(ms=0.0, flops=0.0, bytes=0.0, n=0))
        c["ms"] += e0.elapsed_time(e1)
        c["flops"] += flops
        c["bytes"] += nbytes
        c["n"] += 1
    eng.trace = None
    nsteps_traced = 2
    breakdown = {k: dict(ms_per_step=v["ms"] / nsteps_traced, launches_per_step=v["n"] // nsteps_traced,
                         avg_launch_us=v["ms"] * 1e3 / v["n"],
                         tflops=(v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["ms"] > 0 else 0.0,
                         gbs=(v["bytes"] / (v["ms"] * 1e-3) / 1e9) if v["ms"] > 0 else 0.0)
                 for k, v in classes.items()}
    # dominant kernel: the 256x256-tile GEMM of the mode's main dtype
    dom = {"fp8": "gemm_fp8_256", "bf16": "gemm_bf16_256", "fp8-block": "gemm_fp8blk_256"}[args.mode]
    peak = PEAK_BF16_TFLOPS if args.mode == "bf16" else PEAK_FP8_TFLOPS
    d = classes[dom]
    ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
    roofline = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                "frac": ach / peak, "traffic": None,
                "kernel": {"fp8": "gemm_ring_kernel (fp8 W8A8 linear, 256x256x128B tile, 16x16x128 "
                                  "MFMAs, operands through a five-slot LDS ring)",
                           "bf16": "gemm_kernel<0,4,2,2,4,false,false,true,true> (bf16 linear, "
                                   "256x256x128B tile, 16x16x32 MFMAs, pipelined K loop)",
                           "fp8-block": "gemm_kernel<1,4,2,2,4,true,false,false,false> (block-scaled "
                                        "fp8 linear, 256x256x128B tile)"}[args.mode],
                "launches_per_step": d["n"] // nsteps_traced,
                "avg_launch_us": d["ms"] * 1e3 / d["n"],
                "flop_per_launch": d["flops"] / d["n"]}
    # HBM-side bytes per launch of that kernel come from separate rocprofv3 --pmc passes over
    # this same command (a running process cannot read them); the newest summary is reported
    try:
        import glob

        pmc = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[-1]
        # the profiler prints every template argument: <dtype, waves m, waves n, tm, tn, block
        # scales, half output, 16x16 MFMA shapes, pipelined K loop>; summaries taken before the
        # fused-key-norm argument was withdrawn carry one more `false` in sixth place
        key = {"fp8": "gemm_kernel<1, 4, 2, 2, 4, false", "bf16": "gemm_kernel<0, 4, 2, 2, 4, false",
               "fp8-block": "gemm_kernel<1, 4, 2, 2, 4, true"}[args.mode]
        tail = {"fp8": ", false, true, false>", "bf16": ", false, true, true>",
                "fp8-block": ", false, false, false>"}[args.mode]
        kernels = json.load(open(pmc))["kernels"]
        # the plain fp8 form runs as gemm_ring_kernel; older summaries hold the two-stage kernel
        cands = [key + tail, key + ", false" + tail]
        # (as two instantiations, with and without a half-tile tail: their launches together)
        ring = [v for k, v in kernels.items() if k.startswith("gemm_ring_kernel")] if args.mode == "fp8" else []
        if ring:
            roofline["traffic"] = (sum(v["traffic_bytes_per_launch"] * v["dispatches"] for v in ring)
                                   / sum(v["dispatches"] for v in ring))
        else:
            name = next(k for k in cands if k in kernels)
            roofline["traffic"] = kernels[name]["traffic_bytes_per_launch"]
        roofline["traffic_unit"] = "bytes per launch at the L2 memory side (2*FETCH_SIZE + WRITE_SIZE)"
        roofline["traffic_source"] = os.path.relpath(pmc, ROOT)
        roofline["algorithmic_bytes_per_launch"] = d["bytes"] / d["n"]
    except Exception:
        roofline["traffic"] = None
    fl = flux_step_flops(B, n_double=cfg.num_layers, n_single=cfg.num_single_layers)

    # ---- the fixed-size CFG job through the batch scheduler (strong scaling) -----------------
    def run_c4():
        c4 = None
        from sglang_diffusion_amd.runtime.batch_scheduler import DataParallelBatchScheduler

        n_req, c4_steps = args.c4_requests, args.c4_steps or args.steps
        sch = DataParallelBatchScheduler(stage, rank=rank, world_size=world, max_samples_per_step=8,
                                         queue_on_rank0=True)
        # rank 0 owns the queue: 32 requests of one image each, cond + uncond prompt, true CFG 3.5
        queue = [make_synthetic_request(1, 1024, 1024, NUM_INFERENCE_STEPS, device=dev,
                                        seed=1000 + i, true_cfg=True, guidance_scale=3.5)
                 for i in range(n_req)] if rank == 0 else []

        def drain(k, timing=None):
            for r in queue:
                sch.add_request(r)
            return sch.run(num_steps=k, timing=timing)

        drain(1)  # untimed: workspace for 8-sample micro-batches, communicator set-up
        phases = {}
        barrier()
        t0 = time.perf_counter()
        finals = drain(c4_steps, phases)
        barrier()
        c4_elapsed = time.perf_counter() - t0
        ph = torch.tensor([c4_elapsed, phases.get("scatter_s", 0.0), phases.get("compute_s", 0.0),
                           phases.get("gather_s", 0.0)], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(ph, op=dist.ReduceOp.MAX)
        c4_elapsed, sc_s, cp_s, ga_s = ph.tolist()
        if rank == 0:
            assert len(finals) == n_req and all(bool(torch.isfinite(f.float()).all()) for f in finals)
            full_job_s = sc_s + ga_s + cp_s * NUM_INFERENCE_STEPS / c4_steps
            c4 = {"workload": f"{n_req} requests x (cond + uncond), true CFG scale 3.5, 1024x1024: "
                              f"{2 * n_req} DiT samples per step, micro-batches of 8 samples, "
                              "requests sent from rank 0 and final latents gathered to rank 0",
                  "scaling": "strong", "n_gpus": world, "requests": n_req,
                  "dit_samples_per_gpu": 2 * n_req / world, "steps_run": c4_steps,
                  "elapsed_s": c4_elapsed, "scatter_s": sc_s, "compute_s": cp_s, "gather_s": ga_s,
                  "ms_per_step_per_microbatch": cp_s * 1e3 / c4_steps / max(1, -(-2 * n_req // world // 8)),
                  "images_per_s": n_req / full_job_s,
                  "images_per_s_note": (f"{n_req} images / (scatter + gather + compute x "
                                        f"{NUM_INFERENCE_STEPS}/{c4_steps}); each image costs 2 DiT "
                                        "samples per step, so x2 compares with `value`")}
        return c4

    want_c4 = not args.no_c4 and not args.schnell
    if rank != 0:
        if want_c4:
            import threading

            # rank 0 gives the job up after --c4-timeout and prints its line without it; a rank
            # still waiting in a collective then would keep the launcher waiting for ever (exit
            # code 3: the launcher and the driver must see the hang as a failure)
            dog = threading.Timer(args.c4_timeout + 30.0, lambda: os._exit(3))
            dog.daemon = True
            dog.start()
            try:
                run_c4()
            except Exception as e:  # rank 0 reports; a rank that cannot go on must not hang it
                print(f"[rank {rank}] c4 workload failed: {type(e).__name__}: {e}", file=sys.stderr)
                os._exit(1)
            dog.cancel()
        if world > 1:
            dist.destroy_process_group()
        return
    cpu = None
    if not args.no_cpu_baseline and world == 1:  # the host baseline belongs to the N = 1 line only
        try:
            cpu = cpu_baseline()
        except Exception as e:  # the baseline is informational; never lose the GPU line
            cpu = {"value": None, "unit": "images/s", "cores": 0, "kind": "port",
                   "sample": f"failed: {type(e).__name__}: {e}"}
    is_metric_config = args.layers is None and args.mode == "fp8" and B == 4 and not args.schnell
    out = {
        "metric": ("images_per_sec_flux1schnell_1024x1024_4step_bf16" if args.schnell
                   else "images_per_sec_flux1dev_1024x1024_50step_fp8"),
        "value": images_per_s, "unit": "images/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": step_s * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": ("bf16" if args.mode == "bf16" else
                  "fp8_e4m3 (W8A8 attention/single-block linears"
                  + (", 128x128 weight blocks" if args.mode == "fp8-block" else "")
                  + ") + bf16 (FF, attention)"),
        "data": ("synthetic inputs, weights from " + os.path.basename(os.path.normpath(args.checkpoint))
                 if args.checkpoint else "synthetic"),
        "config": {"workload": (("Flux.1-schnell" if args.schnell else "Flux.1-dev") + " DiT, 1024x1024 (4096 image + 512 text tokens), "
                                f"{NUM_INFERENCE_STEPS}-step schedule, batch {B} per GPU, "
                                f"{args.mode}" + ("" if is_metric_config else " [NOT the metric config]")),
                   "global_batch": world * B, "seq_len": 4608, "parallelism": f"dp{world}",
                   "text_stream_overlap": bool(model._engine.overlap_text),
                   "hip_graph": bool(args.graph),
                   "num_double_blocks": cfg.num_layers, "num_single_blocks": cfg.num_single_layers},
        "step_tflops_total": world * fl["total"] / step_s / 1e12,  # whole job
        "mfma_utilisation_vs_fp8_peak": fl["total"] / step_s / 1e12 / PEAK_FP8_TFLOPS,  # per GPU
        "mixed_roofline_min_ms": (fl["fp8"] / (PEAK_FP8_TFLOPS * 1e12)
                                  + (fl["bf16"] + fl["attn"]) / (PEAK_BF16_TFLOPS * 1e12)) * 1e3,
        "roofline": roofline, "breakdown": breakdown, "cpu_baseline": cpu, "c4": None,
    }
    # the step against the bound of its dtype mix (fp8 GEMMs at the fp8 peak, bf16 GEMMs and
    # attention at the bf16 peak), beside the dominant kernel's own fraction in `roofline`
    out["mixed_roofline_frac"] = out["mixed_roofline_min_ms"] / out["ms_per_step"]
    if want_c4:
        # The headline line above is complete; the fixed-size CFG job adds to it.  Whatever
        # happens in there (an exception here, a rank that died and left the others in a
        # collective), the line is printed: a watchdog prints it without `c4` and ends the
        # process if the job does not finish in time.
        import threading

        def give_up():
            out["c4"] = {"error": f"the CFG job did not finish within {args.c4_timeout} s"}
            print(json.dumps(out), flush=True)
            os._exit(3)  # the headline line is out, but a hung job is a failure for the launcher

        dog = threading.Timer(args.c4_timeout, give_up)
        dog.daemon = True
        dog.start()
        try:
            out["c4"] = run_c4()
        except Exception as e:
            out["c4"] = {"error": f"{type(e).__name__}: {e}"}
        dog.cancel()
    print(json.dumps(out), flush=True)
    if world > 1:
        try:
            dist.destroy_process_group()
        except Exception:
            pass


if __name__ == "__main__":
    main()
