"""Per-block distance between HunyuanVideoTransformer3DModel on the GPU and its CPU oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sglang-diffusion_amd"))
import torch
from oracle.hunyuan_oracle import HunyuanVideoOracle
from sglang_diffusion_amd.layers.attention import Mi355AttentionMetadataBuilder
from sglang_diffusion_amd.layers.forward_context import set_forward_context
from sglang_diffusion_amd.layers.quantization import Fp8Config
from sglang_diffusion_amd.models.hunyuanvideo import (HunyuanVideoArchConfig, HunyuanVideoTransformer3DModel,
                                                      process_weights_after_loading)
from sglang_diffusion_amd.runtime.synthetic import init_synthetic_weights

def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / b.norm())

for fp8 in (False, True):
    cfg = HunyuanVideoArchConfig(num_attention_heads=2, num_layers=2, num_single_layers=2,
                                 text_embed_dim=192, pooled_projection_dim=64, guidance_embeds=True)
    qc = Fp8Config(is_checkpoint_fp8_serialized=False, activation_scheme="dynamic") if fp8 else None
    model = HunyuanVideoTransformer3DModel(cfg, quant_config=qc)
    init_synthetic_weights(model, seed=5, host_rng=True)
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to("cuda"); process_weights_after_loading(model)
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2, 16, 3, 8, 12, generator=g).bfloat16()
    enc = (torch.randn(2, 21, 192, generator=g) * 0.5).bfloat16()
    t = torch.tensor([900.0, 417.0]); guid = torch.tensor([6016.0, 6016.0]).bfloat16()
    orc = HunyuanVideoOracle(state, num_heads=2, num_layers=2, num_single_layers=2, pooled_dim=64,
                             guidance_embeds=True, fp8=fp8)
    rec_o, rec_m, vec_o = {}, {}, []
    for name in ("double_block", "single_block", "token_refiner", "patch_embed", "final_layer", "timestep_embedder"):
        fn = getattr(orc, name)
        def wrap(*a, _fn=fn, _n=name, **k):
            out = _fn(*a, **k)
            if _n == "double_block" and a[0].endswith(".0"): vec_o.append(a[3])
            key = f"{_n}:{a[0] if isinstance(a[0], str) else ''}"
            rec_o.setdefault(key, []).append(out)
            return out
        setattr(orc, name, wrap)
    ref = orc.forward(x, enc, t, guid)
    def hook(n):
        def f(mod, inp, out): rec_m.setdefault(n, []).append(out)
        return f
    for n, m in [("patch_embed:", model.img_in), ("token_refiner:", model.txt_in), ("final_layer:", model.final_layer),
                 ("timestep_embedder:time_in", model.time_in), ("timestep_embedder:guidance_in", model.guidance_in)] + \
                [(f"double_block:double_blocks.{i}", b) for i, b in enumerate(model.double_blocks)] + \
                [(f"single_block:single_blocks.{i}", b) for i, b in enumerate(model.single_blocks)]:
        m.register_forward_hook(hook(n))
    blk0_in = []
    model.double_blocks[0].register_forward_pre_hook(lambda m, a: blk0_in.append(a))
    md = Mi355AttentionMetadataBuilder().build(current_timestep=0)
    with torch.no_grad(), set_forward_context(current_timestep=0, attn_metadata=md):
        out = model(x.cuda(), enc.cuda(), t.cuda(), guid.cuda())
    print("fp8 =", fp8, " final:", rel(out, ref))
    # block 0 on the MODEL's own inputs: isolates the block's arithmetic from upstream drift
    ins = blk0_in[0]
    cos, sin = orc.rope(3, 4, 6)
    oi, ot = orc.double_block.__wrapped__("double_blocks.0", ins[0].cpu(), ins[1].cpu(), ins[2].cpu(), cos, sin) \
        if hasattr(orc.double_block, "__wrapped__") else HunyuanVideoOracle.double_block(orc, "double_blocks.0", ins[0].cpu(), ins[1].cpu(), ins[2].cpu(), cos, sin)
    mo = rec_m["double_block:double_blocks.0"][0]
    print("  block 0 on identical inputs:", rel(mo[0], oi), rel(mo[1], ot), " vec drift:", rel(ins[2], vec_o[0]))
    for k in rec_m:
        o, m = rec_o.get(k, [None])[0], rec_m[k][0]
        if o is None: print(k, "no oracle record"); continue
        if isinstance(o, tuple): print(" ", k, [round(rel(a, b), 5) for a, b in zip(m, o)])
        else: print(" ", k, round(rel(m, o), 5))
