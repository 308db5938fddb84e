"""RCCL on the one GPU a box has: a world of ONE rank with backend "nccl" (see
tests/rccl_one_rank_worker.py for what that does and does not cover).  The rank runs in a child
process so that this test process never holds a process group."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_initialises_and_carries_the_exchange_helpers_on_one_rank():
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_one_rank_worker.py")], env=env,
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    print(res)
    assert res["backend"] == "nccl" and res["world"] == 1
    for k, v in res.items():
        if k not in ("backend", "world"):
            assert v is True, (k, res)
