"""The C-ABI library loads and exports every symbol include/mi355_dit.h declares (CPU only:
no compute call is made)."""
import ctypes
import os
import re

from sglang_diffusion_amd import _C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mi355_dit.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mi355_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_expected_entry_points():
    syms = declared_symbols()
    for must in ("mi355_fp8_scaled_mm", "mi355_flash_attn_fwd", "mi355_per_token_quant_fp8",
                 "mi355_linear_ex", "mi355_layernorm_modulate_quant", "mi355_euler_step"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    lib = _C.load_library()
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"declared in the header but not exported: {missing}"


def test_binding_table_matches_header():
    assert sorted(_C.SIGNATURES) == declared_symbols()


def test_error_reporting_without_gpu():
    lib = _C.load_library()
    assert lib.mi355_abi_version() == _C.ABI_VERSION == 12
    # argument validation happens before any HIP call: a null pointer is rejected on the host
    rc = lib.mi355_per_token_quant_fp8(None, 4, 128, 128, None, None, None)
    assert rc == -1
    assert b"null pointer" in lib.mi355_last_error()
    rc = lib.mi355_flash_attn_fwd(1, 1, 1, 1, None, 1, 8, 8, 2, 2, 96, *([8] * 12), 1.0, 0, None)
    assert rc == -1 and b"head size" in lib.mi355_last_error()
    rc = lib.mi355_flash_attn_fwd(16, 16, 16, 16, None, 1, 8, 8, 2, 2, 128, *([8] * 12), 1.0, 1, None)
    assert rc == -3 and b"causal" in lib.mi355_last_error()


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch

    x = torch.zeros(4, 128, dtype=torch.bfloat16)
    q = torch.zeros(4, 128, dtype=torch.float8_e4m3fn)
    s = torch.zeros(4)
    with pytest.raises(RuntimeError, match="HIP device"):
        _C.per_token_quant_fp8(x, q, s)


def test_graft_entry_build_loads_the_library():
    """build() is the driver's "does it build" check: it must run to the end here (make is a
    no-op when the library is current)."""
    import __graft_entry__ as g

    g.build()
