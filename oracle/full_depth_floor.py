"""The floor under tests/test_full_depth_gpu.py: how far a SECOND faithful evaluation of the
oracle lands from the one in tests/golden/full_depth_{bf16,w8a8}.npz after all 57 blocks.

TEST INFRASTRUCTURE ONLY.  The second evaluation contracts every linear's K dimension in a
fixed pseudo-random order (x[:, perm] @ W[:, perm]^T: the same products, another fp32 summation
order -- what any GEMM kernel other than the host library's does to the oracle's sums) and is
otherwise the oracle itself (tests/test_oracle_sensitivity.py holds the class).  Nothing is
teacher-forced, as in the GPU test.  Prints the relative L2 per stored checkpoint, the figures
the GPU test's bars are read against.

    python oracle/full_depth_floor.py [bf16|w8a8|both] > profiles/rNN_full_depth_floor.txt
(about 45 GB of host memory, 20 minutes on 8 cores).
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sglang-diffusion_amd"))

from oracle.make_golden_full_depth import (CHECKPOINTS, IMAGE_ROWS, TEXT_ROWS, TT,  # noqa: E402
                                           full_depth_inputs, full_depth_state_dict)
from tests.helpers import bf16, load_golden  # noqa: E402
from tests.test_oracle_sensitivity import _PermutedOracle  # noqa: E402


def rel(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / b.norm()).item()


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    torch.set_num_threads(os.cpu_count() or 1)
    sd, cfg = full_depth_state_dict()
    for mode in (("bf16", "w8a8") if which == "both" else (which,)):
        z = load_golden(f"full_depth_{mode}.npz")
        orc = _PermutedOracle(sd, num_layers=cfg.num_layers, num_single_layers=cfg.num_single_layers,
                              num_heads=cfg.num_attention_heads, head_dim=cfg.attention_head_dim,
                              guidance_embeds=True, fp8=(mode == "w8a8"),
                              axes_dims_rope=cfg.axes_dims_rope)
        lat, txt, pooled, t, guid = full_depth_inputs()
        freqs = orc.freqs_cis(TT, 1024, 1024)
        t0 = time.time()
        with torch.no_grad():
            out, inter = orc.forward(lat, txt, pooled, t, guid, freqs, return_intermediates=True)
        parts = []
        for key in CHECKPOINTS:
            enc, hid = inter[key]
            parts.append(f"{key} (txt {rel(enc[0, TEXT_ROWS], bf16(z[f'{key}_text'])):.2e}, "
                         f"img {rel(hid[0, IMAGE_ROWS], bf16(z[f'{key}_image'])):.2e})")
        print(f"full depth {mode}, oracle with permuted GEMM accumulation vs the fixture's oracle "
              f"({time.time() - t0:.0f} s): " + ", ".join(parts)
              + f", model output {rel(out[0], bf16(z['out'])):.2e}", flush=True)
        del orc


if __name__ == "__main__":
    main()
