"""How far apart two FAITHFUL evaluations of one HunyuanVideo block land at the real width
(hidden 3072, 24 heads of 128, 1536 video + 128 text tokens): the measured floor under the bars
of tests/test_hunyuan_real_width_gpu.py.  As tests/test_oracle_sensitivity.py does for the Flux
blocks: the oracle (oracle/hunyuan_oracle.py; blocks: python/sglang/multimodal_gen/runtime/
models/dits/hunyuanvideo.py:177-270, 343-410) against itself with every linear's K dimension
walked in another order (the same products, another fp32 summation order -- what any GEMM
kernel does to the oracle's sums) and the softmax probabilities rounded to IEEE half before P.V
(as the HIP attention kernel does).  Under W8A8 the MLP of these blocks is quantised too
(fc_in, fc_out: a second quantiser behind a GELU), which the Flux double block does not have:
its floor is therefore measured here, not borrowed.

CPU only, about a minute."""
import pytest
import torch
import torch.nn.functional as F

from oracle import ops as O
from oracle.hunyuan_oracle import HunyuanBlocksOracle
from sglang_diffusion_amd.layers.quantization import Fp8Config
from sglang_diffusion_amd.models.hunyuanvideo import MMDoubleStreamBlock, MMSingleStreamBlock
from sglang_diffusion_amd.runtime.synthetic import init_synthetic_weights
from tests.helpers import rel_l2
from tests.hunyuan_width_bars import BAR, HEADS, HID, TI, TT, block_inputs
from tests.test_oracle_sensitivity import _attention_p16


class _Permuted(HunyuanBlocksOracle):
    def lin(self, name, x):
        b = self.sd.get(name + ".bias")
        K = x.shape[-1]
        pm = torch.randperm(K, generator=torch.Generator().manual_seed(K))
        if name in self.q:
            wq, ws = self.q[name]
            shp = x.shape
            q, s = O.per_token_quant_fp8(x.reshape(-1, K))      # the quantised VALUES are unchanged
            return O.fp8_scaled_mm(q[:, pm], wq[:, pm], s, ws, b, x.dtype).view(*shp[:-1], wq.shape[0])
        return F.linear(x[..., pm], self.sd[name + ".weight"][:, pm], b)


def _state(cls, fp8, prefix):
    qc = Fp8Config(is_checkpoint_fp8_serialized=False, activation_scheme="dynamic") if fp8 else None
    blk = cls(HID, HEADS, 4.0, dtype=torch.bfloat16, prefix=prefix, quant_config=qc)
    init_synthetic_weights(blk, seed=5, host_rng=True)
    return {f"{prefix}.{k}": v.detach().clone() for k, v in blk.state_dict().items()}


@pytest.mark.parametrize("fp8", [False, True])
def test_hunyuan_block_floor_at_real_width(fp8):
    img, txt, vec, cos, sin = block_inputs()
    keep = O.attention
    with torch.no_grad():
        sd = _state(MMDoubleStreamBlock, fp8, "double_blocks.0")
        base, perm = (c(sd, num_heads=HEADS, fp8=fp8) for c in (HunyuanBlocksOracle, _Permuted))
        ri, rt = base.double_block("double_blocks.0", img, txt, vec, cos, sin)
        try:
            O.attention = _attention_p16(torch.float16)
            pi, pt = perm.double_block("double_blocks.0", img, txt, vec, cos, sin)
        finally:
            O.attention = keep
        sd = _state(MMSingleStreamBlock, fp8, "single_blocks.0")
        base, perm = (c(sd, num_heads=HEADS, fp8=fp8) for c in (HunyuanBlocksOracle, _Permuted))
        x = torch.cat([ri, rt], dim=1)
        rs = base.single_block("single_blocks.0", x, vec, TT, cos, sin)
        try:
            O.attention = _attention_p16(torch.float16)
            ps = perm.single_block("single_blocks.0", x, vec, TT, cos, sin)
        finally:
            O.attention = keep
    ei, et, es = rel_l2(pi, ri), rel_l2(pt, rt), rel_l2(ps, rs)
    print(f"HunyuanVideo real-width floor fp8={fp8}: double (img {ei:.2e}, txt {et:.2e}) single {es:.2e}")
    # the floor is a measurement: asserted is only that it sits where the GPU bars assume it,
    # between 1e-4 and the bar of its block
    assert 1e-4 <= max(ei, et) <= BAR[("d", fp8)], (ei, et)
    assert es <= BAR[("s", fp8)], es
