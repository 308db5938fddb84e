"""Per-forward context, as python/sglang/multimodal_gen/runtime/managers/forward_context.py:31-120.

The attention layers read `attn_metadata` from here (layers/attention/layer.py:373-378 in the
reference); the denoise loop sets it around each DiT call.
"""
from __future__ import annotations

from contextlib import contextmanager
from dataclasses import dataclass
from typing import Any, Optional


@dataclass
class ForwardContext:
    current_timestep: int = 0
    attn_metadata: Any = None
    forward_batch: Any = None


_forward_context: Optional[ForwardContext] = None


def get_forward_context() -> ForwardContext:
    """Outside a `set_forward_context` block an empty context is returned (the reference
    asserts; the DiT here is also usable stand-alone)."""
    return _forward_context if _forward_context is not None else ForwardContext()


@contextmanager
def set_forward_context(current_timestep: int, attn_metadata: Any, forward_batch: Any = None):
    global _forward_context
    prev = _forward_context
    _forward_context = ForwardContext(current_timestep, attn_metadata, forward_batch)
    try:
        yield
    finally:
        _forward_context = prev
