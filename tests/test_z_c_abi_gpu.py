"""The C ABI from a caller without Python or torch: tests/c_abi_smoke.cpp (built by
`make -C sglang-diffusion_amd/csrc smoke`, i.e. by __graft_entry__.build()) allocates with the HIP
runtime, calls libmi355dit.so through include/mi355_dit.h and checks the results with CPU loops."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "bin", "c_abi_smoke")


def test_c_program_drives_the_library(device):
    if not os.path.exists(BIN):
        pytest.fail(f"{BIN} is missing: run `make -C sglang-diffusion_amd/csrc smoke`")
    p = subprocess.run([BIN], capture_output=True, text=True, timeout=120)
    print(p.stdout, p.stderr)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "C ABI smoke: OK" in p.stdout
