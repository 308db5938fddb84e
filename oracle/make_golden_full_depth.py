"""Mint tests/golden/full_depth_{bf16,w8a8}.npz: ONE forward of the FULL-SIZE Flux.1-dev DiT
(19 double + 38 single blocks, D = 3072, 24 heads of 128, 512 text + 4096 image tokens, B = 1)
through the CPU oracle (oracle/flux_oracle.py, the restatement of
python/sglang/multimodal_gen/runtime/models/dits/flux.py:838-916).

TEST INFRASTRUCTURE ONLY.  Run in the build container (about 45 GB of host memory, 10-20
minutes on 8 cores):
    python oracle/make_golden_full_depth.py [bf16|w8a8|both]

Weights and inputs are `hashed_tensor` draws (sglang_diffusion_amd/runtime/synthetic.py): pure
functions of (parameter name, seed, element index) made of integer hashing and one fp32
multiply, so the GPU test regenerates the very same bits on the device without this script's
12 B-parameter state dict travelling anywhere.  The reference holds no fixture of a whole DiT
forward (its only output check is a mesh distance, test/server/test_server_utils.py:677-728),
so this fixture is minted from the oracle: "parity unpinned" end to end, pinned per operator.

Stored per mode: the model output [4096, 64] (bf16 bits), and after blocks d0, d9, d18, s18,
s37 the joint hidden state at SAMPLE_ROWS (16 text + 32 image rows, all 3072 channels, bf16
bits) with the fp64 L2 norm of each whole stream, so that a sampled relative error can be
read against the whole tensor's size.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sglang-diffusion_amd"))

SEED = 7
TT, TI = 512, 4096
CHECKPOINTS = ("d0", "d9", "d18", "s18", "s37")
TEXT_ROWS = [int(r) for r in np.linspace(0, TT - 1, 16).round()]
IMAGE_ROWS = [int(r) for r in np.linspace(0, TI - 1, 32).round()]
STEP_INDEX = 10            # of the 50-step schedule at 1024 x 1024


def bits(t: torch.Tensor) -> np.ndarray:
    return t.contiguous().view(torch.int16).numpy().view(np.uint16)


def full_depth_inputs(device="cpu"):
    """(latents [1,4096,64], text states [1,512,4096], pooled [1,768], timestep [1] fp32,
    guidance [1] bf16): hashed draws, identical on every device."""
    from oracle.flux_oracle import flux_schedule
    from sglang_diffusion_amd.runtime.synthetic import hashed_tensor

    lat = hashed_tensor("input.latents", (1, TI, 64), SEED, 1.0, device)
    txt = hashed_tensor("input.prompt_embeds", (1, TT, 4096), SEED, 0.1, device)
    pooled = hashed_tensor("input.pooled", (1, 768), SEED, 1.0, device)
    _, ts = flux_schedule(50, 1024, 1024)
    t = ts[STEP_INDEX].reshape(1).to(device)
    guid = (torch.full((1,), 3.5, dtype=torch.float32).to(torch.bfloat16) * 1000.0).to(device)
    return lat, txt, pooled, t, guid


def full_depth_state_dict():
    """The bf16 state dict of the full-size model, drawn by parameter name (no model instance:
    shapes come from the architecture on the meta device)."""
    from sglang_diffusion_amd.models.flux import FluxTransformer2DModel, flux_dev_config
    from sglang_diffusion_amd.runtime.synthetic import hashed_std, hashed_tensor

    cfg = flux_dev_config()
    with torch.device("meta"):
        model = FluxTransformer2DModel(cfg, quant_config=None)
    sd = {}
    for name, p in model.named_parameters():
        if p.dtype == torch.float32 and p.dim() == 1:
            sd[name] = torch.ones(p.shape, dtype=torch.float32)
        else:
            sd[name] = hashed_tensor(name, tuple(p.shape), SEED, hashed_std(p.shape), "cpu")
    return sd, cfg


def run(mode: str, sd, cfg) -> None:
    from oracle.flux_oracle import FluxOracle

    fp8 = mode == "w8a8"
    orc = FluxOracle(sd, num_layers=cfg.num_layers, num_single_layers=cfg.num_single_layers,
                     num_heads=cfg.num_attention_heads, head_dim=cfg.attention_head_dim,
                     guidance_embeds=True, fp8=fp8, axes_dims_rope=cfg.axes_dims_rope)
    lat, txt, pooled, t, guid = full_depth_inputs()
    freqs = orc.freqs_cis(TT, 1024, 1024)
    t0 = time.time()
    with torch.no_grad():
        out, inter = orc.forward(lat, txt, pooled, t, guid, freqs, return_intermediates=True)
    print(f"[{mode}] oracle forward: {time.time() - t0:.0f} s", flush=True)
    rec = {"out": bits(out[0]), "text_rows": np.array(TEXT_ROWS), "image_rows": np.array(IMAGE_ROWS),
           "timestep": t.numpy(), "seed": np.array(SEED), "step_index": np.array(STEP_INDEX),
           "out_norm": np.array(out.double().norm().item())}
    for key in CHECKPOINTS:
        enc, hid = inter[key]
        rec[f"{key}_text"] = bits(enc[0, TEXT_ROWS])
        rec[f"{key}_image"] = bits(hid[0, IMAGE_ROWS])
        rec[f"{key}_text_norm"] = np.array(enc.double().norm().item())
        rec[f"{key}_image_norm"] = np.array(hid.double().norm().item())
        print(f"[{mode}] {key}: |text| {enc.double().norm().item():.4e} |image| "
              f"{hid.double().norm().item():.4e} finite {bool(torch.isfinite(hid.float()).all())}",
              flush=True)
    path = os.path.join(ROOT, "tests", "golden", f"full_depth_{mode}.npz")
    np.savez_compressed(path, **rec)
    print(f"[{mode}] wrote {path} ({os.path.getsize(path)} bytes)", flush=True)


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    torch.set_num_threads(os.cpu_count() or 1)
    t0 = time.time()
    sd, cfg = full_depth_state_dict()
    print(f"state dict: {sum(v.numel() for v in sd.values()) / 1e9:.2f} B parameters in "
          f"{time.time() - t0:.0f} s", flush=True)
    for mode in (("bf16", "w8a8") if which == "both" else (which,)):
        run(mode, sd, cfg)


if __name__ == "__main__":
    main()
