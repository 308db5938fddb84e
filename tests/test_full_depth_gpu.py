"""The WHOLE Flux.1-dev DiT (19 double + 38 single blocks, D = 3072, 24 heads of 128, 512 text +
4096 image tokens, B = 1) on the fused HIP path against one forward of the CPU oracle at the
same size.

The oracle's forward (oracle/flux_oracle.py, restating
python/sglang/multimodal_gen/runtime/models/dits/flux.py:838-916) was run once in the build
container by oracle/make_golden_full_depth.py; tests/golden/full_depth_{bf16,w8a8}.npz hold its
output and, after blocks d0, d9, d18, s18 and s37, the joint hidden state at 48 sampled rows.
Weights and inputs are pure functions of (name, seed, index) -- `hashed_tensor` -- so the model
is rebuilt here on the GPU with the very same bits; nothing but the small fixture travels.
The reference itself holds no fixture of a DiT forward, so this is the oracle's word
("parity unpinned" end to end; every operator underneath is pinned, see test_golden_gpu.py).

Nothing is teacher-forced here: rounding differences travel down the residual stream through
all 57 blocks, which is exactly what is being measured.  The bars are set from the measured
figures and read against the floor between two faithful evaluations of the oracle itself
(tests/test_oracle_sensitivity.py).
"""
import os

import pytest
import torch

from tests.helpers import GOLDEN, bf16, load_golden

pytestmark = pytest.mark.gpu

TT, TI = 512, 4096
# relative L2 over the sampled rows of a checkpoint (text rows, image rows) and over the whole
# model output; the test prints the measured values (bf16 1.35e-2, W8A8 3.9e-2 at the output, which is
# also how far the oracle lands from itself with another GEMM summation order: oracle/full_depth_floor.py)
BAR_STATE = {"bf16": 2.5e-2, "w8a8": 6e-2}
BAR_OUT = {"bf16": 2.5e-2, "w8a8": 6e-2}


class _Sampler:
    """Engine probe: after the checkpoint blocks, keep the sampled rows of the joint state."""

    def __init__(self, text_rows, image_rows, keys):
        self.t = torch.as_tensor(text_rows)
        self.i = torch.as_tensor(image_rows) + TT
        self.keys, self.got = set(keys), {}

    def after(self, kind, i, joint):
        key = f"{kind}{i}"
        if key in self.keys:
            self.got[key] = (joint[0, self.t.to(joint.device)].cpu(),
                             joint[0, self.i.to(joint.device)].cpu())


def _rel(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / b.norm()).item()


@pytest.mark.parametrize("mode", ["bf16", "w8a8"])
def test_full_depth_forward_vs_oracle_fixture(device, mode):
    path = os.path.join(GOLDEN, f"full_depth_{mode}.npz")
    assert os.path.exists(path), f"{path} missing: run oracle/make_golden_full_depth.py"
    from oracle.make_golden_full_depth import CHECKPOINTS, SEED, full_depth_inputs
    from sglang_diffusion_amd.layers.quantization import Fp8Config
    from sglang_diffusion_amd.models.flux import FluxTransformer2DModel, flux_dev_config
    from sglang_diffusion_amd.models.flux_engine import FluxEngine
    from sglang_diffusion_amd.runtime import pipeline_config as pc
    from sglang_diffusion_amd.runtime.synthetic import init_hashed_weights

    z = load_golden(f"full_depth_{mode}.npz")
    assert int(z["seed"]) == SEED
    cfg = flux_dev_config()
    qc = Fp8Config(is_checkpoint_fp8_serialized=False) if mode == "w8a8" else None
    with torch.device(device):
        model = FluxTransformer2DModel(cfg, quant_config=qc)
    init_hashed_weights(model, seed=SEED)
    model.process_weights_after_loading()
    lat, txt, pooled, t, guid = full_depth_inputs(device)
    assert torch.equal(t.cpu(), torch.from_numpy(z["timestep"]))
    freqs = pc.get_freqs_cis(TT, 1024, 1024, device, model.rotary_emb)
    eng = model._engine = FluxEngine(model)
    probe = _Sampler(z["text_rows"], z["image_rows"], CHECKPOINTS)
    with torch.no_grad():
        out = eng.step(lat, txt, pooled, t, guid, freqs, probe=probe).clone()
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    errs = {}
    for key in CHECKPOINTS:
        gt, gi = probe.got[key]
        errs[key] = (_rel(gt, bf16(z[f"{key}_text"])), _rel(gi, bf16(z[f"{key}_image"])))
    e_out = _rel(out[0].cpu(), bf16(z["out"]))
    print(f"full depth {mode}: " + ", ".join(f"{k} (txt {a:.2e}, img {b:.2e})" for k, (a, b) in errs.items())
          + f", model output {e_out:.2e}")
    for key, (a, b) in errs.items():
        assert a <= BAR_STATE[mode] and b <= BAR_STATE[mode], (key, a, b)
    assert e_out <= BAR_OUT[mode], e_out
    del model, eng
    torch.cuda.empty_cache()
