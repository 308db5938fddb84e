"""`flash_attn_varlen_func` with the reference's signature
(sgl-kernel/python/sgl_kernel/flash_attn.py:228-311), issuing `torch.ops.sgl_kernel.fwd`."""
from __future__ import annotations

import torch

from . import _ops  # noqa: F401


def is_fa3_supported(device=None) -> bool:
    """The MI355X kernel serves the same call; kept for callers that probe it."""
    return torch.cuda.is_available()


def flash_attn_varlen_func(q, k, v, cu_seqlens_q=None, cu_seqlens_k=None, max_seqlen_q=None,
                           max_seqlen_k=None, seqused_q=None, seqused_k=None, page_table=None,
                           softmax_scale=None, causal=False, qv=None, q_descale=None, k_descale=None,
                           v_descale=None, window_size=(-1, -1), attention_chunk=0, softcap=0.0,
                           num_splits=1, pack_gqa=None, sm_margin=0, return_softmax_lse=False,
                           sinks=None, score_mod=None, aux_tensors=None, ver=3):
    if max_seqlen_q is None or max_seqlen_k is None:
        raise ValueError("max_seqlen_q and max_seqlen_k are required for FA3")
    if score_mod is not None or aux_tensors is not None:
        raise NotImplementedError("score_mod / aux_tensors are not implemented")
    if softmax_scale is None:
        softmax_scale = (q.shape[-1] + (qv.shape[-1] if qv is not None else 0)) ** (-0.5)
    attention_chunk = 0 if attention_chunk is None else int(attention_chunk)
    out, softmax_lse, *rest = torch.ops.sgl_kernel.fwd.default(
        q, k, v, None, None, qv, None, cu_seqlens_q, cu_seqlens_k, None, seqused_q, seqused_k,
        max_seqlen_q, max_seqlen_k, page_table, None, None, None, None, None,
        q_descale, k_descale, v_descale, softmax_scale, causal, window_size[0], window_size[1],
        attention_chunk, softcap, is_rotary_interleaved=False, scheduler_metadata=None,
        num_splits=num_splits, pack_gqa=pack_gqa, sm_margin=sm_margin, sinks=sinks)
    return (out, softmax_lse, *rest) if return_softmax_lse else out
