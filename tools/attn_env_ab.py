"""A/B of one environment switch of the attention kernels, alternating in one process; outputs
and LSE of both settings are compared bit for bit first (ragged shapes included).
python tools/attn_env_ab.py VAR [A_VALUE B_VALUE]     ('-' = unset; default: unset against 1)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sglang-diffusion_amd"))
import torch
from sglang_diffusion_amd import _C

dev = "cuda"
var = sys.argv[1]
vals = (sys.argv[2], sys.argv[3]) if len(sys.argv) > 3 else ("-", "1")


def setv(v):
    if v == "-":
        os.environ.pop(var, None)
    else:
        os.environ[var] = v
    _C.reload_env()  # the library reads its switches when it is loaded, not per call


bad = 0
torch.manual_seed(0)
for B, Sq, Sk, H, Hk in [(1, 64, 64, 2, 2), (2, 300, 333, 4, 2), (2, 257, 129, 2, 1), (1, 31, 1, 2, 2), (1, 1000, 4608, 3, 3),
                         (4, 1000, 1000, 24, 24)]:
    D = 128
    q = torch.randn(B, Sq, H, D, device=dev).bfloat16()
    k = torch.randn(B, Sk, Hk, D, device=dev).bfloat16()
    v = torch.randn(B, Sk, Hk, D, device=dev).bfloat16()
    for half in (True, False):
        vv = _C.value_to_f16(v) if half else v
        outs = []
        for i in range(2):
            setv(vals[i])
            o = torch.full((B, Sq, H, D), float("nan"), device=dev, dtype=torch.bfloat16)
            lse = torch.zeros(B, H, Sq, device=dev)
            _C.flash_attn_fwd(q, k, vv, o, lse, D ** -0.5, False, half)
            torch.cuda.synchronize()
            outs.append((o, lse))
        nd = (outs[0][0].view(torch.int16) != outs[1][0].view(torch.int16)).sum().item()
        nl = (outs[0][1] != outs[1][1]).sum().item()
        bad += nd + nl
        print(f"B={B} Sq={Sq} Sk={Sk} H={H} Hk={Hk} {'half' if half else 'bf16'} P.V: differing outputs {nd}, lse {nl}",
              flush=True)
for B, S, H in [(4, 4608, 24), (1, 4608, 24), (1, 15106, 24)]:
    D = 128
    qkv = torch.randn(B, S, 3, H, D, device=dev).bfloat16()
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    v16 = _C.value_to_f16(v)
    o = torch.empty(B, S, H, D, device=dev, dtype=torch.bfloat16)
    for half in (True, False):
        vv = v16 if half else v
        res = [[], []]
        for rep in range(3):
            for i in range(2):
                setv(vals[i])
                for _ in range(5):
                    _C.flash_attn_fwd(q, k, vv, o, None, D ** -0.5, False, half)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    _C.flash_attn_fwd(q, k, vv, o, None, D ** -0.5, False, half)
                e1.record()
                torch.cuda.synchronize()
                res[i].append(e0.elapsed_time(e1) * 1e3 / 20)
        f = 4.0 * B * H * S * S * D / 1e6
        print(f"attention B={B} S={S} H={H} {'half' if half else 'bf16'} P.V: {var}={vals[0]} " +
              " ".join(f"{u:7.1f}" for u in res[0]) + f" us ({f / min(res[0]):6.0f} TF/s)   {var}={vals[1]} " +
              " ".join(f"{u:7.1f}" for u in res[1]) + f" us ({f / min(res[1]):6.0f} TF/s)   ratio "
              f"{min(res[1]) / min(res[0]):.3f}", flush=True)
sys.exit(1 if bad else 0)
