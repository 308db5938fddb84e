"""A/B of the attention kernel's MFMA shape: 32x32x16 (attn_fwd_kernel) against 16x16x32
(attn_fwd16_kernel), selected per call with MI355_ATTN_MFMA16.  `check` compares both with an
fp32 reference on ragged shapes (partial query / key tiles, grouped heads, the fused query
norm + rotation, both P.V precisions); `time` alternates them on the Flux and video shapes.
python tools/attn_mfma16_probe.py [check|time|both] [VAR]
VAR (default MI355_ATTN_MFMA16) names the switch: any other variant of the kernel that is selected
by VAR=1 and is not bit-identical to the shipped one (e.g. MI355_ATTN_WOVEN) is checked and timed
the same way ("16x16" in the output then stands for that variant)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sglang-diffusion_amd"))
import torch
from sglang_diffusion_amd import _C

dev = "cuda"
VAR = sys.argv[2] if len(sys.argv) > 2 else "MI355_ATTN_MFMA16"


def run(mode, q, k, v, o, lse, half, **kw):
    os.environ[VAR] = str(mode)
    _C.flash_attn_fwd(q, k, v, o, lse, q.shape[-1] ** -0.5, False, half, **kw)


def reference(q, k, v, lse=True):
    B, Sq, H, D = q.shape
    Hk = k.shape[2]
    qf = q.float().permute(0, 2, 1, 3)
    kf = k.float().permute(0, 2, 1, 3).repeat_interleave(H // Hk, 1)
    vf = v.float().permute(0, 2, 1, 3).repeat_interleave(H // Hk, 1)
    s = qf @ kf.transpose(-1, -2) * D ** -0.5
    return (torch.softmax(s, -1) @ vf).permute(0, 2, 1, 3), torch.logsumexp(s, -1)


def check():
    ok = True
    torch.manual_seed(0)
    cases = [(1, 64, 64, 2, 2), (2, 300, 333, 4, 2), (1, 1000, 4608, 3, 3), (2, 257, 129, 2, 1), (1, 31, 1, 2, 2),
             (1, 4608, 4608, 2, 2)]
    for B, Sq, Sk, H, Hk in cases:
        D = 128
        q = torch.randn(B, Sq, H, D, device=dev).bfloat16()
        k = torch.randn(B, Sk, Hk, D, device=dev).bfloat16()
        v = torch.randn(B, Sk, Hk, D, device=dev).bfloat16()
        ref, ref_lse = reference(q, k, v)
        for half in (True, False):
            vv = _C.value_to_f16(v) if half else v
            outs = []
            for mode in (0, 1):
                o = torch.full((B, Sq, H, D), float("nan"), device=dev, dtype=torch.bfloat16)
                lse = torch.zeros(B, H, Sq, device=dev)
                run(mode, q, k, vv, o, lse, half)
                torch.cuda.synchronize()
                outs.append((o.float(), lse))
            rel = [((o - ref).norm() / ref.norm()).item() for o, _ in outs]
            dl = [(l - ref_lse).abs().max().item() for _, l in outs]
            # (the fp32 reference is not rounded to bf16: 1.66e-3 of it is the output rounding itself)
            good = rel[1] <= rel[0] * 1.06 + 1e-5 and rel[1] < (1.8e-3 if half else 4e-3) and dl[1] < 1e-3 and \
                bool(torch.isfinite(outs[1][0]).all())
            ok = ok and good
            print(f"B={B} Sq={Sq} Sk={Sk} H={H} Hk={Hk} {'half' if half else 'bf16'} P.V: rel-L2 32x32 {rel[0]:.2e} "
                  f"16x16 {rel[1]:.2e}; lse err {dl[0]:.1e} {dl[1]:.1e}; max |32x32 - 16x16| "
                  f"{(outs[0][0] - outs[1][0]).abs().max().item():.2e} {'OK' if good else 'BAD'}", flush=True)
    # fused query norm + rotation: the two kernels against each other and the separate pass
    B, S, H, D = 2, 700, 3, 128
    qkv = torch.randn(B, S, 3, H, D, device=dev).bfloat16()
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    w0 = torch.rand(D, device=dev) + 0.5
    w1 = torch.rand(D, device=dev) + 0.5
    ang = torch.rand(S, D // 2, device=dev) * 6.28
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    vv = _C.value_to_f16(v)
    outs = []
    for mode in (0, 1):
        o = torch.empty(B, S, H, D, device=dev, dtype=torch.bfloat16)
        run(mode, q, k, vv, o, None, True, q_norm=(w0, w1, 100, 1e-6), q_rope=(cos, sin))
        torch.cuda.synchronize()
        outs.append(o.float())
    d = (outs[0] - outs[1]).abs().max().item()
    rel = ((outs[0] - outs[1]).norm() / outs[0].norm()).item()
    good = rel < 2e-3
    ok = ok and good
    print(f"fused query norm + rotation: 32x32 vs 16x16 rel-L2 {rel:.2e}, max abs {d:.2e} {'OK' if good else 'BAD'}",
          flush=True)
    return ok


def time_ab(iters=20):
    for B, S, H in [(4, 4608, 24), (1, 4608, 24), (1, 15106, 24)]:
        D = 128
        qkv = torch.randn(B, S, 3, H, D, device=dev).bfloat16()
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
        v16 = _C.value_to_f16(v)
        o = torch.empty(B, S, H, D, device=dev, dtype=torch.bfloat16)
        for half in (True, False):
            vv = v16 if half else v
            res = {0: [], 1: []}
            for rep in range(3):
                for mode in (0, 1):
                    for _ in range(5):
                        run(mode, q, k, vv, o, None, half)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(iters):
                        run(mode, q, k, vv, o, None, half)
                    e1.record()
                    torch.cuda.synchronize()
                    res[mode].append(e0.elapsed_time(e1) * 1e3 / iters)
            f = 4.0 * B * H * S * S * D / 1e6
            print(f"attention B={B} S={S} H={H} {'half' if half else 'bf16'} P.V: 32x32 " +
                  " ".join(f"{u:7.1f}" for u in res[0]) + f" us ({f / min(res[0]):6.0f} TF/s)   16x16 " +
                  " ".join(f"{u:7.1f}" for u in res[1]) + f" us ({f / min(res[1]):6.0f} TF/s)   ratio "
                  f"{min(res[1]) / min(res[0]):.3f}", flush=True)


mode = sys.argv[1] if len(sys.argv) > 1 else "both"
ok = True
if mode in ("check", "both"):
    ok = check()
if mode in ("time", "both") and ok:
    time_ab()
sys.exit(0 if ok else 1)
