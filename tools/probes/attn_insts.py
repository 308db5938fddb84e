"""Instructions per wave of the attention kernel inside and outside its key loop, from counters:
the Flux-shaped launch with 1, 2 and 4 key tiles, with and without the fused query transform;
per wave  insts = E + tiles * L.

    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv \
        -d gpurun_out/attn_insts -- python3 tools/probes/attn_insts.py run
    python3 tools/probes/attn_insts.py parse gpurun_out/attn_insts
"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CASES = [(f, sk) for f in (False, True) for sk in (64, 128, 256)]


def run():
    sys.path.insert(0, os.path.join(ROOT, "sglang-diffusion_amd"))
    import torch
    from sglang_diffusion_amd import _C
    dev = "cuda"
    B, S, H, D = 4, 4608, 24, 128
    q = torch.randn(B, S, H, D, device=dev).bfloat16()
    o = torch.empty(B, S, H, D, device=dev, dtype=torch.bfloat16)
    w0 = torch.rand(D, device=dev) + 0.5
    ang = torch.rand(S, D // 2, device=dev) * 6.28
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    for fused, Sk in CASES:
        k = torch.randn(B, Sk, H, D, device=dev).bfloat16()
        v = _C.value_to_f16(torch.randn(B, Sk, H, D, device=dev).bfloat16())
        kw = dict(q_norm=(w0, w0, 512, 1e-6), q_rope=(cos, sin)) if fused else {}
        _C.flash_attn_fwd(q, k, v, o, None, D ** -0.5, False, True, **kw)
        torch.cuda.synchronize()


def parse(d):
    rows = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "attn_fwd_kernel" not in r["Kernel_Name"]:
                continue
            rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    seq = [v for _, v in sorted(rows.items())]
    assert len(seq) == len(CASES), len(seq)
    waves = 4 * 24 * 18 * 8
    for fi, fused in enumerate((False, True)):
        sel = seq[fi * 3:fi * 3 + 3]
        print("query transform", "fused" if fused else "none")
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"):
            per = [s[c] / waves for s in sel]
            L = (per[2] - per[0]) / 3
            print(f"   {c}: per key tile {L:.1f}, outside the key loop {per[0] - L:.1f}   (1/2/4 tiles: "
                  + ", ".join(f"{x:.0f}" for x in per) + ")")


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else parse(sys.argv[2])
