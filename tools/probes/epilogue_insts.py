"""Instructions per wave outside and inside the K loop of the big-tile GEMMs, from counters:
two launches of the same M x N with 2 and 4 K tiles per output tile; per wave and output tile
    insts = E + ktiles * L      (E: epilogue, seams, set-up per tile; L: one K tile)
Run under the profiler (counters in a pass of their own), then `parse` the csv it wrote:

    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv \
        -d gpurun_out/epi_insts -- python3 tools/probes/epilogue_insts.py run
    python3 tools/probes/epilogue_insts.py parse gpurun_out/epi_insts
"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
M, N = 16384, 9216
KS = {True: (256, 512), False: (128, 256, 384)}     # 2 / 4 K tiles of 128 bytes (bf16: 2, 4, 6)


def run():
    sys.path.insert(0, os.path.join(ROOT, "sglang-diffusion_amd"))
    import torch
    from sglang_diffusion_amd import _C
    dev = "cuda"
    for fp8 in (True, False):
        for K in KS[fp8]:
            if fp8:
                a = torch.randn(M, K, device=dev).clamp(-448, 448).to(torch.float8_e4m3fn)
                w = torch.randn(N, K, device=dev).clamp(-448, 448).to(torch.float8_e4m3fn)
                sa, sb = torch.rand(M, device=dev) * 0.01, torch.rand(N, device=dev) * 0.01
            else:
                a, w = torch.randn(M, K, device=dev).bfloat16(), torch.randn(N, K, device=dev).bfloat16()
                sa = sb = None
            bias = torch.randn(N, device=dev).bfloat16()
            out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            _C.linear_ex(a, w, out, M=M, N=N, K=K, lda=K, ldw=K, ldo=N, sa=sa, sb=sb, bias=bias, fp8=fp8)
            torch.cuda.synchronize()


def parse(d):
    rows = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if "gemm" not in name:
                continue
            rows.setdefault((int(r["Dispatch_Id"]), name), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    tiles = (M // 256) * (N // 256)
    seq = [v | {"name": k[1]} for k, v in sorted(rows.items())]
    for fp8, lo in ((True, 0), (False, 2)):
        sel = seq[lo:lo + len(KS[fp8])]
        print("fp8" if fp8 else "bf16", [s["name"][:60] for s in sel][0])
        per = []
        for s in sel:
            waves = tiles * 8
            per.append({c: s[c] / waves for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS")})
            print("   per wave and tile:", {c: round(v, 1) for c, v in per[-1].items()}, "waves", s["SQ_WAVES"])
        for c in per[0]:
            L = (per[1][c] - per[0][c]) / 2
            print(f"   {c}: per K tile {L:.1f}, outside the K loop {per[0][c] - 2 * L:.1f}")


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else parse(sys.argv[2])
