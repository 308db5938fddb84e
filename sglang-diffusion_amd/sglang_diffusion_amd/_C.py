"""ctypes binding of the C-ABI kernel library (include/mi355_dit.h).

The library is the product; this module only loads it and turns torch tensors
into (pointer, size, stride, stream) arguments.  There is no fallback: if the
shared object is missing every op raises.
"""

from __future__ import annotations

import contextlib
import ctypes
import os
import threading
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# MI355_DIT_LIB points at another build of the library (A/B timing of kernel variants)
LIB_PATH = os.environ.get("MI355_DIT_LIB") or os.path.join(_HERE, "libmi355dit.so")

_i64 = ctypes.c_int64
_f32 = ctypes.c_float
_int = ctypes.c_int
_ptr = ctypes.c_void_p


class RowMap(ctypes.Structure):
    """mi355_rowmap: logical row m -> (m // rows_per_batch) * batch_rows + row_offset + m % rows_per_batch."""

    _fields_ = [("rows_per_batch", _i64), ("batch_rows", _i64), ("row_offset", _i64)]


class LinearArgs(ctypes.Structure):
    _fields_ = [
        ("a", _ptr),
        ("w", _ptr),
        ("sa", _ptr),
        ("sb", _ptr),
        ("bias", _ptr),
        ("out", _ptr),
        ("residual", _ptr),
        ("gate", _ptr),
        ("M", _i64),
        ("N", _i64),
        ("K", _i64),
        ("lda", _i64),
        ("ldw", _i64),
        ("ldo", _i64),
        ("ldg", _i64),
        ("gate_rows", _i64),
        ("a_map", RowMap),
        ("o_map", RowMap),
        ("fp8", _int),
        ("act", _int),
        ("f16_begin", _i64),
        ("f16_end", _i64),
        ("block_scale", _int),
        ("sa_sm", _i64),
        ("sa_sk", _i64),
        ("sb_sk", _i64),
        ("sb_sn", _i64),
        ("out_f16", _int),
    ]


class AttnArgs(ctypes.Structure):
    _fields_ = (
        [(n, _ptr) for n in ("q", "k", "v", "out", "lse")]
        + [(n, _i64) for n in ("B", "Sq", "Sk", "H", "Hk", "D", "q_sb", "q_ss", "q_sh", "k_sb", "k_ss",
                               "k_sh", "v_sb", "v_ss", "v_sh", "o_sb", "o_ss", "o_sh")]
        + [("softmax_scale", _f32), ("causal", _int), ("v_is_f16", _int),
           ("q_norm_w0", _ptr), ("q_norm_w1", _ptr), ("q_split", _i64), ("q_eps", _f32),
           ("q_cos", _ptr), ("q_sin", _ptr)]
    )


# every symbol include/mi355_dit.h declares, with its argument types
SIGNATURES = {
    "mi355_last_error": (ctypes.c_char_p, []),
    "mi355_abi_version": (_int, []),
    "mi355_reload_env": (None, []),
    "mi355_per_token_quant_fp8": (_int, [_ptr, _i64, _i64, _i64, _ptr, _ptr, _ptr]),
    "mi355_per_token_group_quant_fp8": (
        _int,
        [_ptr, _i64, _i64, _i64, _ptr, _ptr, _i64, _i64, _int, _f32, _f32, _f32, _int, _ptr],
    ),
    "mi355_per_tensor_quant_fp8": (_int, [_ptr, _i64, _ptr, _ptr, _int, _ptr]),
    "mi355_fp8_scaled_mm": (
        _int,
        [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _i64, _ptr],
    ),
    "mi355_fp8_blockwise_scaled_mm": (
        _int,
        [_ptr, _ptr, _ptr, _ptr, _ptr] + [_i64] * 10 + [_ptr],
    ),
    "mi355_linear_bf16": (_int, [_ptr, _ptr, _ptr, _ptr] + [_i64] * 6 + [_int, _ptr]),
    "mi355_linear_ex": (_int, [ctypes.POINTER(LinearArgs), _ptr]),
    "mi355_flash_attn_fwd": (_int, [_ptr] * 5 + [_i64] * 18 + [_f32, _int, _ptr]),
    "mi355_flash_attn_fwd_f16v": (_int, [_ptr] * 5 + [_i64] * 18 + [_f32, _int, _ptr]),
    "mi355_bf16_to_f16": (_int, [_ptr, _ptr, _i64, _i64, _i64, _i64, _ptr]),
    "mi355_f16_range_status": (_int, [ctypes.POINTER(_i64), ctypes.POINTER(_i64), _int, _ptr]),
    "mi355_f16_range_snapshot": (_int, [_ptr, _int, _ptr]),
    "mi355_rmsnorm": (_int, [_ptr, _ptr, _ptr, _int, _i64, _i64, _i64, _i64, _f32, _ptr]),
    "mi355_fused_add_rmsnorm": (_int, [_ptr, _ptr, _ptr, _int, _i64, _i64, _f32, _ptr]),
    "mi355_qk_norm_rope": (
        _int,
        [_ptr, _ptr] + [_i64] * 7 + [_ptr] * 4 + [_i64, _f32, _ptr, _ptr, _int, _ptr],
    ),
    "mi355_rotary_embedding": (
        _int,
        [_ptr, _ptr, _ptr] + [_i64] * 7 + [_ptr, _int, _int, _ptr],
    ),
    "mi355_layernorm_modulate": (
        _int,
        [_ptr] * 4 + [_i64] * 6 + [_f32, ctypes.POINTER(RowMap), ctypes.POINTER(RowMap), _ptr],
    ),
    "mi355_scale_residual_layernorm_modulate": (
        _int, [_ptr] * 7 + [_i64] * 9 + [_f32, _ptr]),
    "mi355_layernorm_affine": (_int, [_ptr] * 4 + [_i64] * 4 + [_f32, _ptr]),
    "mi355_token_mean": (_int, [_ptr, _ptr] + [_i64] * 5 + [_ptr]),
    "mi355_layernorm_modulate_quant": (
        _int,
        [_ptr] * 5 + [_i64] * 5 + [_f32, ctypes.POINTER(RowMap), _ptr],
    ),
    "mi355_layernorm_modulate_group_quant": (
        _int,
        [_ptr] * 5 + [_i64] * 5 + [_f32, ctypes.POINTER(RowMap), _ptr],
    ),
    "mi355_flash_attn_fwd_ex": (_int, [ctypes.POINTER(AttnArgs), _ptr]),
    "mi355_merge_attn_states": (
        _int,
        [_ptr] * 6 + [_i64] * 4 + [ctypes.POINTER(_i64)] * 6 + [_ptr],
    ),
    "mi355_gate_residual": (
        _int,
        [_ptr] * 4 + [_i64] * 7 + [ctypes.POINTER(RowMap), ctypes.POINTER(RowMap), _ptr],
    ),
    "mi355_activation": (_int, [_ptr, _ptr, _i64, _i64, _i64, _i64, _int, _ptr]),
    "mi355_timestep_embedding": (_int, [_ptr, _ptr, _i64, _i64, _int, _f32, _f32, _f32, _int, _ptr]),
    "mi355_euler_step": (_int, [_ptr, _ptr, _ptr, _f32, _f32, _i64, _ptr]),
    "mi355_cfg_combine": (_int, [_ptr, _ptr, _ptr, _f32, _i64, _ptr]),
    "mi355_cfg_normalize": (_int, [_ptr, _ptr, _f32, _i64, _ptr, _ptr]),
    "mi355_guidance_rescale": (_int, [_ptr, _ptr, _f32, _i64, _i64, _ptr, _ptr]),
    "mi355_guidance_rescale_static": (_int, [_ptr, _ptr, _f32, _i64, _i64, _ptr]),
}

_lib = None


# the ABI revision these prototypes describe (mi355_abi_version() of the library must agree)
ABI_VERSION = 12


def load_library() -> ctypes.CDLL:
    """dlopen the kernel library and attach prototypes; raises if it is missing or stale."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `make -C sglang-diffusion_amd/csrc` "
            "(or __graft_entry__.build()); there is no non-HIP fallback"
        )
    lib = ctypes.CDLL(LIB_PATH)
    # MI355_DIT_ABI_ANY=1 (A/B timing of an OLDER build through entry points it already had):
    # symbols the build lacks are skipped and the revision check is waived
    lenient = os.environ.get("MI355_DIT_ABI_ANY") == "1"
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None and lenient:
            continue
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.mi355_abi_version() != ABI_VERSION and not lenient:
        raise RuntimeError(f"{LIB_PATH} has ABI {lib.mi355_abi_version()}, this binding expects "
                           f"{ABI_VERSION}: rebuild with `make -C sglang-diffusion_amd/csrc`")
    _lib = lib
    return lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load_library().mi355_last_error()
        raise RuntimeError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def _stream(t: torch.Tensor) -> int:
    """The launch stream for an op on `t`: the current stream of t's device.  The library keys
    its per-device state (function attributes, CU counts, symbol addresses) on the CURRENT
    device, so a tensor on another device is refused instead of being launched with the wrong
    device's state."""
    if not t.is_cuda:
        raise RuntimeError("mi355 kernels need tensors on a HIP device (got a CPU tensor)")
    if t.device.index != torch.cuda.current_device():
        raise RuntimeError(f"tensor on cuda:{t.device.index} but the current device is cuda:"
                           f"{torch.cuda.current_device()}: wrap the call in torch.cuda.device(...)")
    return torch.cuda.current_stream(t.device).cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _rowmap(m) -> Optional[RowMap]:
    if m is None:
        return None
    return RowMap(int(m[0]), int(m[1]), int(m[2]))


# --------------------------------------------------------------------- quant --
def per_token_quant_fp8(x: torch.Tensor, q: torch.Tensor, s: torch.Tensor) -> None:
    M, K = x.shape
    _check(
        load_library().mi355_per_token_quant_fp8(
            x.data_ptr(), M, K, x.stride(0), q.data_ptr(), s.data_ptr(), _stream(x)
        ),
        "per_token_quant_fp8",
    )


def per_token_group_quant_fp8(x, q, s, group_size, eps, fp8_min, fp8_max, scale_ue8m0) -> None:
    M, K = x.shape
    _check(
        load_library().mi355_per_token_group_quant_fp8(
            x.data_ptr(), M, K, x.stride(0), q.data_ptr(), s.data_ptr(), s.stride(0), s.stride(1),
            int(group_size), float(eps), float(fp8_min), float(fp8_max), int(bool(scale_ue8m0)),
            _stream(x),
        ),
        "per_token_group_quant_fp8",
    )


def per_tensor_quant_fp8(x, q, s, is_static: bool) -> None:
    _check(
        load_library().mi355_per_tensor_quant_fp8(
            x.data_ptr(), x.numel(), q.data_ptr(), s.data_ptr(), int(bool(is_static)), _stream(x)
        ),
        "per_tensor_quant_fp8",
    )


# ---------------------------------------------------------------------- GEMM --
def fp8_scaled_mm(a, w, sa, sb, bias, out) -> None:
    """a [M,K] fp8, w [N,K] fp8 (K contiguous), sa [M], sb [N] fp32, out [M,N] bf16 -- or
    float16, then with a float16 bias."""
    M, K = a.shape
    N = w.shape[0]
    if out.dtype == torch.float16:
        linear_ex(a, w, out, M=M, N=N, K=K, lda=a.stride(0), ldw=w.stride(0), ldo=out.stride(0),
                  sa=sa, sb=sb, bias=bias, fp8=True, out_f16=True)
        return
    _check(
        load_library().mi355_fp8_scaled_mm(
            a.data_ptr(), w.data_ptr(), sa.data_ptr(), sb.data_ptr(), _p(bias), out.data_ptr(),
            M, N, K, a.stride(0), w.stride(0), out.stride(0), _stream(a),
        ),
        "fp8_scaled_mm",
    )


def fp8_blockwise_scaled_mm(a, w, sa, sb, out, sa_strides, sb_strides) -> None:
    M, K = a.shape
    N = w.shape[0]
    if out.dtype == torch.float16:
        args = _linear_args(a, w, out, M=M, N=N, K=K, lda=a.stride(0), ldw=w.stride(0),
                            ldo=out.stride(0), sa=sa, sb=sb, fp8=True, out_f16=True)
        args.block_scale = 1
        args.sa_sm, args.sa_sk = sa_strides
        args.sb_sk, args.sb_sn = sb_strides
        _check(load_library().mi355_linear_ex(ctypes.byref(args), _stream(a)),
               "fp8_blockwise_scaled_mm")
        return
    _check(
        load_library().mi355_fp8_blockwise_scaled_mm(
            a.data_ptr(), w.data_ptr(), sa.data_ptr(), sb.data_ptr(), out.data_ptr(),
            M, N, K, a.stride(0), w.stride(0), out.stride(0),
            sa_strides[0], sa_strides[1], sb_strides[0], sb_strides[1], _stream(a),
        ),
        "fp8_blockwise_scaled_mm",
    )


def linear_bf16(x, w, bias, out, act: int = 0) -> None:
    M, K = x.shape
    N = w.shape[0]
    _check(
        load_library().mi355_linear_bf16(
            x.data_ptr(), w.data_ptr(), _p(bias), out.data_ptr(), M, N, K,
            x.stride(0), w.stride(0), out.stride(0), int(act), _stream(x),
        ),
        "linear_bf16",
    )


def _linear_args(
    a, w, out, *, M, N, K, lda, ldw, ldo, sa=None, sb=None, bias=None, residual=None, gate=None,
    ldg=0, gate_rows=0, a_map=None, o_map=None, fp8=False, act=0, f16_cols=None,
    block_scale=False, out_f16=False,
) -> LinearArgs:
    args = LinearArgs()
    args.a, args.w, args.out = a.data_ptr(), w.data_ptr(), out.data_ptr()
    args.sa, args.sb, args.bias = _p(sa), _p(sb), _p(bias)
    args.residual, args.gate = _p(residual), _p(gate)
    args.M, args.N, args.K = M, N, K
    args.lda, args.ldw, args.ldo, args.ldg = lda, ldw, ldo, ldg
    args.gate_rows = gate_rows
    args.a_map = _rowmap(a_map) or RowMap(0, 0, 0)
    args.o_map = _rowmap(o_map) or RowMap(0, 0, 0)
    args.fp8 = int(bool(fp8))
    args.act = int(act)
    args.f16_begin, args.f16_end = (f16_cols if f16_cols else (0, 0))
    args.block_scale = 0
    if block_scale:
        args.block_scale = 1
        args.sa_sm, args.sa_sk = sa.stride(0), sa.stride(1)
        args.sb_sn, args.sb_sk = sb.stride(0), sb.stride(1)
    args.out_f16 = int(bool(out_f16))
    return args


def linear_ex(a, w, out, **kw) -> None:
    """mi355_linear_ex; keyword arguments as _linear_args.  block_scale: `sa` is the [rows, K/128]
    group-scale tensor of the activations and `sb` the [N/128, K/128] block-scale tensor of the
    weight (their strides are taken from the tensors)."""
    args = _linear_args(a, w, out, **kw)
    _check(load_library().mi355_linear_ex(ctypes.byref(args), _stream(a)), "linear_ex")


def reload_env() -> None:
    """Make the library read its MI355_* environment switches again (it reads them once, when
    it is loaded); tests use it to compare kernel variants inside one process."""
    load_library().mi355_reload_env()


# ----------------------------------------------------------------- attention --
def flash_attn_fwd(q, k, v, out, lse, softmax_scale: float, causal: bool = False,
                   v_is_f16: bool = False, q_norm=None, q_rope=None) -> None:
    """q [B,Sq,H,D], k/v [B,Sk,Hk,D], out [B,Sq,H,D] (last dim contiguous).  With v_is_f16 the
    16-bit words of `v` hold IEEE half values and P.V runs in half (more accurate).
    q_norm = (w0, w1, split, eps) and q_rope = (cos, sin) make the kernel apply the per-head
    RMSNorm and the interleaved RoPE of qk_norm_rope to each Q row as it loads it (q itself is
    not modified)."""
    B, Sq, H, D = q.shape
    Sk, Hk = k.shape[1], k.shape[2]
    for t in (q, k, v, out):
        if t.stride(-1) != 1:
            raise RuntimeError("flash_attn_fwd: last dimension must be contiguous")
    a = AttnArgs()
    a.q, a.k, a.v, a.out, a.lse = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), _p(lse)
    a.B, a.Sq, a.Sk, a.H, a.Hk, a.D = B, Sq, Sk, H, Hk, D
    a.q_sb, a.q_ss, a.q_sh = q.stride(0), q.stride(1), q.stride(2)
    a.k_sb, a.k_ss, a.k_sh = k.stride(0), k.stride(1), k.stride(2)
    a.v_sb, a.v_ss, a.v_sh = v.stride(0), v.stride(1), v.stride(2)
    a.o_sb, a.o_ss, a.o_sh = out.stride(0), out.stride(1), out.stride(2)
    a.softmax_scale, a.causal, a.v_is_f16 = float(softmax_scale), int(bool(causal)), int(bool(v_is_f16))
    if q_norm is not None:
        w0, w1, split, eps = q_norm
        for w in (w0, w1):
            if w.dtype != torch.float32 or w.numel() != D or not w.is_contiguous():
                raise RuntimeError("flash_attn_fwd: query norm weights must be contiguous fp32 [D]")
        a.q_norm_w0, a.q_norm_w1, a.q_split, a.q_eps = w0.data_ptr(), w1.data_ptr(), int(split), float(eps)
    if q_rope is not None:
        cos, sin = q_rope
        for t in (cos, sin):
            if t.dtype != torch.float32 or tuple(t.shape) != (Sq, D // 2) or not t.is_contiguous():
                raise RuntimeError("flash_attn_fwd: query cos / sin must be contiguous fp32 [Sq, D/2]")
        a.q_cos, a.q_sin = cos.data_ptr(), sin.data_ptr()
    _check(load_library().mi355_flash_attn_fwd_ex(ctypes.byref(a), _stream(q)), "flash_attn_fwd")


def merge_attn_states(out, out_lse, a, a_lse, b, b_lse, lse_layout: str = "bhs") -> None:
    """out/a/b: bf16 [B, S, H, D] (last dimension contiguous); log-sum-exps fp32, [B, H, S] as
    flash_attn_fwd writes them (lse_layout "bhs") or [B, S, H] ("bsh").  out may alias a or b;
    out_lse may be None."""
    B, S, H, D = a.shape
    for t in (out, a, b):
        if t.shape != a.shape or t.stride(-1) != 1 or t.dtype != torch.bfloat16:
            raise RuntimeError("merge_attn_states: bf16 [B,S,H,D] tensors of one shape expected")
    if lse_layout not in ("bhs", "bsh"):
        raise ValueError("lse_layout must be 'bhs' or 'bsh'")

    def st3(t):
        return (_i64 * 3)(t.stride(0), t.stride(1), t.stride(2))

    def lse3(t):
        want = (B, H, S) if lse_layout == "bhs" else (B, S, H)
        if tuple(t.shape) != want or t.dtype != torch.float32:
            raise RuntimeError(f"merge_attn_states: fp32 log-sum-exp of shape {want} expected, "
                               f"got {tuple(t.shape)} {t.dtype}")
        if lse_layout == "bhs":
            return (_i64 * 3)(t.stride(0), t.stride(2), t.stride(1))
        return (_i64 * 3)(t.stride(0), t.stride(1), t.stride(2))

    _check(
        load_library().mi355_merge_attn_states(
            out.data_ptr(), _p(out_lse), a.data_ptr(), a_lse.data_ptr(), b.data_ptr(),
            b_lse.data_ptr(), B, S, H, D, st3(out), st3(a), st3(b),
            lse3(out_lse) if out_lse is not None else None, lse3(a_lse), lse3(b_lse), _stream(a),
        ),
        "merge_attn_states",
    )


def bf16_to_f16(x: torch.Tensor, y: torch.Tensor) -> None:
    """x, y: 2-D views [rows, cols] with contiguous last dimension; y receives half bits."""
    rows, cols = x.shape
    _check(
        load_library().mi355_bf16_to_f16(
            x.data_ptr(), y.data_ptr(), rows, cols, x.stride(0), y.stride(0), _stream(x)
        ),
        "bf16_to_f16",
    )


def value_to_f16(v: torch.Tensor) -> torch.Tensor:
    """bf16 V [B,Sk,Hk,D] (any view with D contiguous) -> a contiguous tensor of the same shape
    whose 16-bit words are IEEE half: the V operand of flash_attn_fwd(..., v_is_f16=True)."""
    B, Sk, Hk, D = v.shape
    if not (v.stride(3) == 1 and v.stride(2) == D and v.stride(0) == Sk * v.stride(1)):
        v = v.contiguous()
    out = torch.empty((B, Sk, Hk, D), dtype=torch.float16, device=v.device)
    if v.numel():
        bf16_to_f16(v.as_strided((B * Sk, Hk * D), (v.stride(1), 1)), out.view(B * Sk, Hk * D))
    return out


# Process-wide DEFAULT of the V operand's format.  True: V is converted to IEEE half and the
# P.V product runs in half (closest to the fp32-softmax result).  False: V stays bf16 and the
# probabilities are rounded to bf16 (the form every flash kernel the reference calls uses).
# A caller that must deviate for a while (the denoise stage re-running ONE request batch whose
# values left half's range) does so with `value_half_as`, which ends with its `with` block.
VALUE_HALF = True
_value_half_local = threading.local()


def value_half() -> bool:
    """The V format in force on this thread: the innermost `value_half_as`, else VALUE_HALF."""
    stack = getattr(_value_half_local, "stack", None)
    return bool(stack[-1]) if stack else bool(VALUE_HALF)


@contextlib.contextmanager
def value_half_as(flag: bool):
    """Run a block with half (True) or bf16 (False) V, whatever the process default says."""
    stack = getattr(_value_half_local, "stack", None)
    if stack is None:
        stack = _value_half_local.stack = []
    stack.append(bool(flag))
    try:
        yield
    finally:
        stack.pop()


def f16_range_status(device=None, reset: bool = True):
    """(overflow_blocks, tiny_blocks) of the bf16 -> half conversions done on `device` since the
    last reset; waits for the device's current stream."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    over, tiny = _i64(0), _i64(0)
    with torch.cuda.device(dev):
        _check(load_library().mi355_f16_range_status(
            ctypes.byref(over), ctypes.byref(tiny), int(bool(reset)),
            torch.cuda.current_stream(dev).cuda_stream), "f16_range_status")
    return int(over.value), int(tiny.value)


def f16_range_reset(device=None) -> None:
    """Zero the device's range counters (queued on its current stream, no synchronisation)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    with torch.cuda.device(dev):
        _check(load_library().mi355_f16_range_status(
            None, None, 1, torch.cuda.current_stream(dev).cuda_stream), "f16_range_status")


def f16_range_snapshot(pinned: torch.Tensor, reset: bool = True, device=None) -> None:
    """Queue a copy of the device's two range counters into `pinned` (int64[2], pinned host
    memory) on its current stream, then their zeroing; no synchronisation, capturable in a HIP
    graph.  Read `pinned` after an event recorded behind this call (or after a graph replay)."""
    if pinned.dtype != torch.int64 or pinned.numel() < 2 or pinned.is_cuda or not pinned.is_pinned():
        raise RuntimeError("f16_range_snapshot: a pinned host int64[2] tensor is expected")
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    with torch.cuda.device(dev):
        _check(load_library().mi355_f16_range_snapshot(
            pinned.data_ptr(), int(bool(reset)), torch.cuda.current_stream(dev).cuda_stream),
            "f16_range_snapshot")


def attention(q, k, v, out, lse, softmax_scale: float) -> None:
    """Dense attention on bf16 q, k, v: half-precision P.V product while value_half() holds,
    else the bf16 form."""
    if value_half():
        flash_attn_fwd(q, k, value_to_f16(v), out, lse, softmax_scale, False, v_is_f16=True)
    else:
        flash_attn_fwd(q, k, v, out, lse, softmax_scale, False, v_is_f16=False)


# --------------------------------------------------------------- norm / rope --
def rmsnorm(out, x, w, eps: float) -> None:
    rows, dim = x.shape
    _check(
        load_library().mi355_rmsnorm(
            out.data_ptr(), x.data_ptr(), w.data_ptr(), int(w.dtype == torch.float32),
            rows, dim, x.stride(0), out.stride(0), float(eps), _stream(x),
        ),
        "rmsnorm",
    )


def fused_add_rmsnorm(x, residual, w, eps: float) -> None:
    rows, dim = x.shape
    _check(
        load_library().mi355_fused_add_rmsnorm(
            x.data_ptr(), residual.data_ptr(), w.data_ptr(), int(w.dtype == torch.float32),
            rows, dim, float(eps), _stream(x),
        ),
        "fused_add_rmsnorm",
    )


def qk_norm_rope(q, k, wq0, wk0, wq1, wk1, s_split, eps, cos, sin, interleaved=True) -> None:
    """In place on q, k [B,S,H,D] (strided views are fine, D contiguous); q may be None (the
    query side is then left to flash_attn_fwd's q_norm / q_rope)."""
    B, S, H, D = k.shape
    if q is not None and (q.stride() != k.stride() or q.stride(-1) != 1):
        raise RuntimeError("qk_norm_rope: q and k must share strides, D contiguous")
    if k.stride(-1) != 1:
        raise RuntimeError("qk_norm_rope: D must be contiguous")
    _check(
        load_library().mi355_qk_norm_rope(
            _p(q), k.data_ptr(), B, S, H, D, k.stride(0), k.stride(1), k.stride(2),
            _p(wq0), _p(wk0), _p(wq1), _p(wk1), int(s_split), float(eps), _p(cos), _p(sin),
            int(bool(interleaved)), _stream(k),
        ),
        "qk_norm_rope",
    )


def rotary_embedding(positions, q, k, head_size, cos_sin_cache, is_neox) -> None:
    tokens = positions.numel()
    Hq = q.shape[-1] // head_size if q.dim() == 2 else q.shape[-2]
    q2 = q.view(tokens, -1)
    if k is not None:
        k2 = k.view(tokens, -1)
        Hk = k2.shape[-1] // head_size
    else:
        k2, Hk = None, 0
    _check(
        load_library().mi355_rotary_embedding(
            positions.data_ptr(), q2.data_ptr(), _p(k2), tokens, Hq, Hk, head_size,
            cos_sin_cache.shape[-1], q2.stride(0), k2.stride(0) if k2 is not None else 0,
            cos_sin_cache.data_ptr(), int(cos_sin_cache.dtype == torch.float32),
            int(bool(is_neox)), _stream(q),
        ),
        "rotary_embedding",
    )


def layernorm_modulate(out, x, shift, scale, B, T, eps, x_map=None, o_map=None) -> None:
    dim = x.shape[-1]
    xm, om = _rowmap(x_map), _rowmap(o_map)
    mod = shift if shift is not None else scale
    _check(
        load_library().mi355_layernorm_modulate(
            out.data_ptr(), x.data_ptr(), _p(shift), _p(scale), B, T, dim,
            x.stride(-2), out.stride(-2), mod.stride(0) if mod is not None else 8, float(eps),
            ctypes.byref(xm) if xm else None, ctypes.byref(om) if om else None, _stream(x),
        ),
        "layernorm_modulate",
    )


def scale_residual_layernorm_modulate(out, res_out, residual, x, gate, shift, scale, B, T,
                                      eps) -> None:
    """res_out = residual + gate[b] * x; out = LayerNorm(res_out) * (1 + scale[b]) + shift[b];
    2-D [B*T, dim] operands (last dim contiguous), gate/shift/scale [B, dim] rows or None."""
    dim = x.shape[-1]
    mod = shift if shift is not None else scale
    _check(
        load_library().mi355_scale_residual_layernorm_modulate(
            out.data_ptr(), res_out.data_ptr(), residual.data_ptr(), x.data_ptr(), _p(gate),
            _p(shift), _p(scale), B, T, dim, x.stride(0), residual.stride(0), out.stride(0),
            res_out.stride(0), gate.stride(0) if gate is not None else 8,
            mod.stride(0) if mod is not None else 8, float(eps), _stream(x),
        ),
        "scale_residual_layernorm_modulate",
    )


def layernorm_affine(out, x, weight, bias, eps) -> None:
    """nn.LayerNorm(dim, eps, elementwise_affine=True) on 2-D [rows, dim] bf16."""
    rows, dim = x.shape
    _check(
        load_library().mi355_layernorm_affine(
            out.data_ptr(), x.data_ptr(), weight.data_ptr(), bias.data_ptr(), rows, dim,
            x.stride(0), out.stride(0), float(eps), _stream(x),
        ),
        "layernorm_affine",
    )


def token_mean(out, x) -> None:
    """out [B, dim] = mean over dim 1 of x [B, T, dim] (last dim contiguous)."""
    B, T, dim = x.shape
    if x.stride(-1) != 1 or not out.is_contiguous():
        raise RuntimeError("token_mean: last dimension must be contiguous")
    _check(
        load_library().mi355_token_mean(out.data_ptr(), x.data_ptr(), B, T, dim, x.stride(1),
                                        x.stride(0), _stream(x)),
        "token_mean",
    )


def layernorm_modulate_quant(q, s, x, shift, scale, B, T, eps, x_map=None) -> None:
    dim = x.shape[-1]
    xm = _rowmap(x_map)
    mod = shift if shift is not None else scale
    _check(
        load_library().mi355_layernorm_modulate_quant(
            q.data_ptr(), s.data_ptr(), x.data_ptr(), _p(shift), _p(scale), B, T, dim,
            x.stride(-2), mod.stride(0) if mod is not None else 8, float(eps),
            ctypes.byref(xm) if xm else None, _stream(x),
        ),
        "layernorm_modulate_quant",
    )


def layernorm_modulate_group_quant(q, s, x, shift, scale, B, T, eps, x_map=None) -> None:
    """As layernorm_modulate_quant with one scale per (row, 128 channels): s is [B*T, dim/128]."""
    dim = x.shape[-1]
    xm = _rowmap(x_map)
    mod = shift if shift is not None else scale
    if not s.is_contiguous() or s.shape[-1] != dim // 128:
        raise RuntimeError("layernorm_modulate_group_quant: scales must be dense [rows, dim/128]")
    _check(
        load_library().mi355_layernorm_modulate_group_quant(
            q.data_ptr(), s.data_ptr(), x.data_ptr(), _p(shift), _p(scale), B, T, dim,
            x.stride(-2), mod.stride(0) if mod is not None else 8, float(eps),
            ctypes.byref(xm) if xm else None, _stream(x),
        ),
        "layernorm_modulate_group_quant",
    )


def gate_residual(out, residual, gate, y, B, T, y_map=None, r_map=None) -> None:
    dim = y.shape[-1]
    ym, rm = _rowmap(y_map), _rowmap(r_map)
    _check(
        load_library().mi355_gate_residual(
            out.data_ptr(), residual.data_ptr(), gate.data_ptr(), y.data_ptr(), B, T, dim,
            y.stride(-2), residual.stride(-2), out.stride(-2), gate.stride(0),
            ctypes.byref(ym) if ym else None, ctypes.byref(rm) if rm else None, _stream(y),
        ),
        "gate_residual",
    )


def activation(out, x, act: int) -> None:
    rows, dim = x.shape
    _check(
        load_library().mi355_activation(
            out.data_ptr(), x.data_ptr(), rows, dim, x.stride(0), out.stride(0), int(act), _stream(x)
        ),
        "activation",
    )


def timestep_embedding(t, out, flip_sin_to_cos, downscale_freq_shift, scale, max_period) -> None:
    B, dim = out.shape
    _check(
        load_library().mi355_timestep_embedding(
            t.data_ptr(), out.data_ptr(), B, dim, int(bool(flip_sin_to_cos)),
            float(downscale_freq_shift), float(scale), float(max_period),
            int(out.dtype == torch.bfloat16), _stream(t),
        ),
        "timestep_embedding",
    )


def euler_step(out, sample, v, sigma: float, sigma_next: float) -> None:
    _check(
        load_library().mi355_euler_step(
            out.data_ptr(), sample.data_ptr(), v.data_ptr(), float(sigma), float(sigma_next),
            sample.numel(), _stream(sample),
        ),
        "euler_step",
    )


def cfg_combine(out, cond, uncond, scale: float) -> None:
    _check(
        load_library().mi355_cfg_combine(
            out.data_ptr(), cond.data_ptr(), uncond.data_ptr(), float(scale), cond.numel(),
            _stream(cond),
        ),
        "cfg_combine",
    )


def _cfg_scratch(noise) -> torch.Tensor:
    return torch.empty(256 * max(1, noise.shape[0]), dtype=torch.float32, device=noise.device)


def cfg_normalize(noise, cond, factor: float, scratch=None) -> None:
    """In place on `noise` (contiguous bf16): cap its global L2 norm at factor * ||cond||."""
    if not (noise.is_contiguous() and cond.is_contiguous()) or noise.shape != cond.shape:
        raise RuntimeError("cfg_normalize: contiguous tensors of one shape expected")
    scratch = _cfg_scratch(noise) if scratch is None else scratch
    if scratch.numel() < 256 or scratch.dtype != torch.float32:
        raise RuntimeError("cfg_normalize: scratch needs 256 fp32 elements")
    _check(
        load_library().mi355_cfg_normalize(
            noise.data_ptr(), cond.data_ptr(), float(factor), noise.numel(), scratch.data_ptr(),
            _stream(noise),
        ),
        "cfg_normalize",
    )


def guidance_rescale(noise, cond, rescale: float, scratch=None) -> None:
    """In place on `noise` [B, ...] (contiguous bf16): rescale_noise_cfg of the reference."""
    if not (noise.is_contiguous() and cond.is_contiguous()) or noise.shape != cond.shape:
        raise RuntimeError("guidance_rescale: contiguous tensors of one shape expected")
    B = noise.shape[0]
    scratch = _cfg_scratch(noise) if scratch is None else scratch
    if scratch.numel() < 256 * B:
        raise RuntimeError("guidance_rescale: scratch needs 256 floats per sample")
    _check(
        load_library().mi355_guidance_rescale(
            noise.data_ptr(), cond.data_ptr(), float(rescale), B, noise.numel() // max(B, 1),
            scratch.data_ptr(), _stream(noise),
        ),
        "guidance_rescale",
    )


def guidance_rescale_static(noise, cond, rescale: float) -> None:
    """guidance_rescale through the entry point that allocates nothing (the library's own
    64-KiB scratch; at most 64 samples per call)."""
    if not (noise.is_contiguous() and cond.is_contiguous()) or noise.shape != cond.shape:
        raise RuntimeError("guidance_rescale: contiguous tensors of one shape expected")
    B = noise.shape[0]
    with torch.cuda.device(noise.device):
        _check(
            load_library().mi355_guidance_rescale_static(
                noise.data_ptr(), cond.data_ptr(), float(rescale), B, noise.numel() // max(B, 1),
                _stream(noise),
            ),
            "guidance_rescale_static",
        )
