"""How far apart two FAITHFUL evaluations of one Flux block land at the REAL width
(D = 3072, 24 heads of 128, 512 text + 4096 image tokens, B = 1): the measured floor under the
per-block bars of tests/test_real_width_gpu.py.

The oracle (oracle/flux_oracle.py; blocks: python/sglang/multimodal_gen/runtime/models/dits/
flux.py:499-559, 620-692) is evaluated four ways on identical block inputs:
  base      fp32 softmax, GEMM sums in the library's order;
  softmax   the same with an fp64 softmax(q k^T) v (strictly more accurate arithmetic);
  permuted  the same as base with every linear's K dimension walked in another order
            (x[:, perm] @ W[:, perm]^T: the same products, another fp32 summation order --
            what any other GEMM kernel, the HIP one included, does to the oracle's sums);
  flash     permuted, and the softmax probabilities rounded to 16 bits before P.V as every
            flash-attention kernel does: to bf16 (FA3 / aiter, the kernels the reference's own
            attention backends call, layers/attention/backends/flash_attn.py:517-532,
            aiter.py:84-92) and to IEEE half (the HIP kernel here).
Each is a legitimate rendering of the reference.  Whatever separates them is rounding noise
that no implementation can be asked to stay under; the GPU bars are read against it.

CPU only, about three minutes on 8 cores.
"""
import pytest
import torch
import torch.nn.functional as F

from oracle import ops as O
from oracle.flux_oracle import FluxOracle
from sglang_diffusion_amd.runtime.synthetic import hashed_std, hashed_tensor
from tests.helpers import rel_l2

D, H, DH, TT, TI = 3072, 24, 128, 512, 4096


def _state_dict(seed=5):
    sd = {}

    def lin(name, n, k):
        sd[name + ".weight"] = hashed_tensor(name + ".weight", (n, k), seed, hashed_std((n, k)))
        sd[name + ".bias"] = hashed_tensor(name + ".bias", (n,), seed, 0.02)

    p = "transformer_blocks.0"
    for nm in ("norm1", "norm1_context"):
        lin(f"{p}.{nm}.linear", 6 * D, D)
    for nm in ("to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj", "to_out.0", "to_add_out"):
        lin(f"{p}.attn.{nm}", D, D)
    for nm in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
        sd[f"{p}.attn.{nm}.weight"] = torch.ones(DH)
    for ff in ("ff", "ff_context"):
        lin(f"{p}.{ff}.net.0.proj", 4 * D, D)
        lin(f"{p}.{ff}.net.2", D, 4 * D)
    q = "single_transformer_blocks.0"
    lin(f"{q}.norm.linear", 3 * D, D)
    for nm in ("to_q", "to_k", "to_v"):
        lin(f"{q}.attn.{nm}", D, D)
    for nm in ("norm_q", "norm_k"):
        sd[f"{q}.attn.{nm}.weight"] = torch.ones(DH)
    lin(f"{q}.proj_mlp", 4 * D, D)
    lin(f"{q}.proj_out", D, 5 * D)
    return sd


class _PermutedOracle(FluxOracle):
    """Every linear contracts K in a fixed pseudo-random order."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self._perms = {}

    def _perm(self, K):
        if K not in self._perms:
            self._perms[K] = torch.randperm(K, generator=torch.Generator().manual_seed(K))
        return self._perms[K]

    def lin(self, name, x, act=None):
        b = self.sd.get(name + ".bias")
        K = x.shape[-1]
        pm = self._perm(K)
        if name in self.q:
            wq, ws = self.q[name]
            shp = x.shape
            q, s = O.per_token_quant_fp8(x.reshape(-1, K))      # the quantised VALUES are unchanged
            y = O.fp8_scaled_mm(q[:, pm], wq[:, pm], s, ws, b, x.dtype).view(*shp[:-1], wq.shape[0])
        else:
            y = F.linear(x[..., pm], self.sd[name + ".weight"][:, pm], b)
        if act == "gelu":
            y = F.gelu(y, approximate="tanh")
        return y


def _attention_fp64(q, k, v, scale=None, return_lse=False):
    out = torch.empty_like(q)
    sc = scale or q.shape[-1] ** -0.5
    for h in range(q.shape[2]):                                   # one head at a time: 170 MB
        qf, kf, vf = (x[:, :, h].double() for x in (q, k, v))
        p = torch.softmax(qf @ kf.transpose(-1, -2) * sc, -1)
        out[:, :, h] = (p @ vf).to(q.dtype)
    return out


def _attention_p16(dtype):
    """softmax in fp32, probabilities rounded to `dtype`, P.V accumulated in fp32."""
    def fn(q, k, v, scale=None, return_lse=False):
        out = torch.empty_like(q)
        sc = scale or q.shape[-1] ** -0.5
        for h in range(q.shape[2]):
            qf, kf, vf = (x[:, :, h].float() for x in (q, k, v))
            s_ = qf @ kf.transpose(-1, -2) * sc
            m = s_.amax(-1, keepdim=True)
            e = torch.exp(s_ - m)
            p = e.to(dtype).float()                         # the flash kernels' rounding point
            out[:, :, h] = ((p @ vf) / e.sum(-1, keepdim=True)).to(q.dtype)
        return out
    return fn


@pytest.mark.parametrize("fp8", [False, True])
def test_block_floor_at_real_width(fp8):
    torch.manual_seed(0)
    sd = _state_dict()
    kw = dict(num_layers=1, num_single_layers=1, num_heads=H, head_dim=DH, guidance_embeds=True,
              fp8=fp8)
    base, perm = FluxOracle(sd, **kw), _PermutedOracle(sd, **kw)
    hid = hashed_tensor("in.hid", (1, TI, D), 3, 1.0)
    enc = hashed_tensor("in.enc", (1, TT, D), 3, 1.0)
    temb = hashed_tensor("in.temb", (1, D), 3, 1.0)
    freqs = base.freqs_cis(TT, 1024, 1024)
    res = {}
    with torch.no_grad():
        d_enc, d_hid = base.double_block(0, hid, enc, temb, freqs)
        s_enc, s_hid = base.single_block(0, d_hid, d_enc, temb, freqs)
        # teacher-forced variants: every block gets the BASE run's input, as on the GPU
        pe, ph = perm.double_block(0, hid, enc, temb, freqs)
        pse, psh = perm.single_block(0, d_hid, d_enc, temb, freqs)
        res["permuted"] = dict(d=(rel_l2(pe, d_enc), rel_l2(ph, d_hid)),
                               s=(rel_l2(pse, s_enc), rel_l2(psh, s_hid)))
        keep = O.attention
        try:
            O.attention = _attention_fp64
            fe, fh = base.double_block(0, hid, enc, temb, freqs)
            fse, fsh = base.single_block(0, d_hid, d_enc, temb, freqs)
        finally:
            O.attention = keep
        res["softmax"] = dict(d=(rel_l2(fe, d_enc), rel_l2(fh, d_hid)),
                              s=(rel_l2(fse, s_enc), rel_l2(fsh, s_hid)))
        # (LayerNorm and tanh-GELU evaluated in fp64 on top of this were measured too: they move
        # the figures by less than 5e-5, so elementwise kernels are not where a block's
        # deviation comes from)
        for name, dt in (("flash_bf16", torch.bfloat16), ("flash_half", torch.float16)):
            try:
                O.attention = _attention_p16(dt)
                xe, xh = perm.double_block(0, hid, enc, temb, freqs)
                xse, xsh = perm.single_block(0, d_hid, d_enc, temb, freqs)
            finally:
                O.attention = keep
            res[name] = dict(d=(rel_l2(xe, d_enc), rel_l2(xh, d_hid)),
                             s=(rel_l2(xse, s_enc), rel_l2(xsh, s_hid)))
    for name, r in res.items():
        print(f"real-width floor fp8={fp8} {name}: double (txt {r['d'][0]:.2e}, img {r['d'][1]:.2e}) "
              f"single (txt {r['s'][0]:.2e}, img {r['s'][1]:.2e})")
    # The floor is a measurement, not a bar on a product: what is asserted is that it is where
    # the GPU bars assume it to be -- two faithful evaluations of a block sit between 1e-4 and
    # the GPU test's bar for that block, so that bar is not hiding an operator's error.
    from tests.real_width_bars import BAR

    for name, r in res.items():
        if name == "flash_bf16":
            continue   # the reference's own flash kernels are COARSER than the bars (reported only)
        for kind in ("d", "s"):
            for v in r[kind]:
                assert v <= BAR[(kind, fp8)], (name, kind, v)
    assert max(res["permuted"]["d"]) >= 1e-4
    # the half-precision P.V product of the HIP kernel is the finer of the two flash renderings
    assert max(res["flash_half"]["d"]) <= max(res["flash_bf16"]["d"])
