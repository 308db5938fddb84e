"""HIP kernels INSIDE the multi-rank exchanges: two rank processes share cuda:0 and talk over
gloo; UlyssesAttention with replicated text, USPAttention's replicated prefix, ring attention
(HIP attention + HIP merge) and the CFG-parallel combine are compared with dense HIP attention
on one rank (bit-equal where the arithmetic is the same) and with the CPU oracle.  The ranks are
started by bench.py's launcher, as the benchmark's own ranks are; what stays unmeasured on this
one-GPU box is only the RCCL transport."""
import io
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_hip_kernels_inside_two_rank_exchanges():
    # the launcher runs in a child python so that this (GPU-initialised) test process only
    # starts ONE plain child; that child never touches the GPU before it starts the ranks
    code = ("import sys; sys.path.insert(0, %r); import bench; "
            "sys.exit(bench.launch_ranks(2, [], timeout_s=240, script=%r))"
            % (ROOT, os.path.join(ROOT, "tests", "two_rank_gpu_worker.py")))
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    print(res)
    for k, v in res.items():
        if k.endswith("exact") or k == "exact_on_every_rank":
            assert v is True, (k, res)
    # the kernel's half-precision P.V product against the fp32-softmax oracle
    assert res["ulysses_vs_oracle"] <= 1e-3 and res["usp_prefix_vs_oracle"] <= 1e-3
    # ring: partial results are rounded to bf16 before the merge (one extra rounding)
    assert res["ring_vs_dense_hip"] <= 6e-3 and res["ring_vs_oracle"] <= 6e-3
    assert res["ring_lse_max_abs"] <= 1e-4
