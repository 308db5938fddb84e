"""Attention backend plugin API and the MI355X backend.

Mirrors python/sglang/multimodal_gen/runtime/layers/attention/backends/attention_backend.py:18-170
(AttentionBackend / AttentionImpl / AttentionMetadata / AttentionMetadataBuilder),
platforms/interface.py:26-53 (AttentionBackendEnum), selector.py:89-159 (get_attn_backend)
and layer.py:221-412 (LocalAttention / USPAttention), with ONE backend: the hand-written
gfx950 flash-attention kernel.  There is no multi-backend dispatch: asking for another
backend raises.
"""
from __future__ import annotations

import contextlib
import enum
from abc import ABC, abstractmethod
from dataclasses import dataclass, fields
from functools import lru_cache
from typing import Any, Generic, Optional, TypeVar

import torch
import torch.nn as nn

from .. import _C
from .forward_context import get_forward_context


class AttentionBackendEnum(enum.Enum):
    """Same members as the reference enum (platforms/interface.py:26-40); only AITER's slot
    (the ROCm default, server_args.py:1234-1243) is served here, by the MI355X kernel."""

    FA2 = enum.auto()
    FA = enum.auto()
    SLIDING_TILE_ATTN = enum.auto()
    TORCH_SDPA = enum.auto()
    SAGE_ATTN = enum.auto()
    SAGE_ATTN_3 = enum.auto()
    VIDEO_SPARSE_ATTN = enum.auto()
    SPARSE_VIDEO_GEN_2_ATTN = enum.auto()
    VMOBA_ATTN = enum.auto()
    AITER = enum.auto()
    SLA_ATTN = enum.auto()
    SAGE_SLA_ATTN = enum.auto()
    NO_ATTENTION = enum.auto()

    def __str__(self) -> str:
        return self.name.lower()


class AttentionBackend(ABC):
    accept_output_buffer: bool = False

    @staticmethod
    @abstractmethod
    def get_enum() -> AttentionBackendEnum:
        raise NotImplementedError

    @staticmethod
    @abstractmethod
    def get_impl_cls() -> type["AttentionImpl"]:
        raise NotImplementedError

    @staticmethod
    @abstractmethod
    def get_metadata_cls() -> type["AttentionMetadata"]:
        raise NotImplementedError

    @staticmethod
    @abstractmethod
    def get_builder_cls() -> type["AttentionMetadataBuilder"]:
        return None


@dataclass
class AttentionMetadata:
    current_timestep: int

    def asdict_zerocopy(self, skip_fields: set[str] | None = None) -> dict[str, Any]:
        skip_fields = skip_fields or set()
        return {f.name: getattr(self, f.name) for f in fields(self) if f.name not in skip_fields}


T = TypeVar("T", bound=AttentionMetadata)


class AttentionMetadataBuilder(ABC, Generic[T]):
    @abstractmethod
    def __init__(self) -> None:
        raise NotImplementedError

    @abstractmethod
    def prepare(self) -> None:
        raise NotImplementedError

    @abstractmethod
    def build(self, **kwargs: dict[str, Any]) -> AttentionMetadata:
        raise NotImplementedError


class AttentionImpl(ABC, Generic[T]):
    @abstractmethod
    def __init__(self, num_heads: int, head_size: int, softmax_scale: float, causal: bool = False,
                 num_kv_heads: int | None = None, prefix: str = "", **extra_impl_args) -> None:
        raise NotImplementedError

    def preprocess_qkv(self, qkv: torch.Tensor, attn_metadata: T) -> torch.Tensor:
        return qkv

    def postprocess_output(self, output: torch.Tensor, attn_metadata: T) -> torch.Tensor:
        return output

    @abstractmethod
    def forward(self, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                attn_metadata: T) -> torch.Tensor:
        raise NotImplementedError


# ------------------------------------------------------------ MI355X backend --
@dataclass
class Mi355AttentionMetadata(AttentionMetadata):
    """Dense attention needs no per-step metadata beyond the timestep (the FA backend's
    cu_seqlens / max_seqlen fields, backends/flash_attn.py, only matter for varlen)."""


class Mi355AttentionMetadataBuilder(AttentionMetadataBuilder):
    def __init__(self) -> None:
        pass

    def prepare(self) -> None:
        pass

    def build(self, current_timestep: int = 0, **kwargs) -> Mi355AttentionMetadata:
        return Mi355AttentionMetadata(current_timestep=current_timestep)


class Mi355AttentionImpl(AttentionImpl):
    """forward(q,k,v,attn_metadata) on [B,S,H,D] bf16, as FlashAttentionImpl / AITerImpl /
    SDPAImpl (backends/flash_attn.py:436-565, aiter.py:63-93, sdpa.py:55-78)."""

    def __init__(self, num_heads: int, head_size: int, softmax_scale: float, causal: bool = False,
                 num_kv_heads: int | None = None, prefix: str = "", **extra_impl_args) -> None:
        if causal:
            raise NotImplementedError("the MI355X DiT attention kernel is non-causal")
        if extra_impl_args.get("dropout_p", 0.0):
            raise ValueError("dropout is not supported (inference only)")
        if head_size not in (64, 128):
            raise ValueError(f"head_size {head_size} not supported (64 or 128)")
        self.num_heads = num_heads
        self.num_kv_heads = num_kv_heads or num_heads
        self.head_size = head_size
        self.softmax_scale = softmax_scale
        self.causal = causal

    def forward(self, query, key, value, attn_metadata=None, *, return_softmax_lse: bool = False,
                out: Optional[torch.Tensor] = None, value_is_half: bool = False):
        """value_is_half: the 16-bit words of `value` already hold IEEE half (the producing
        GEMM wrote them that way, mi355_linear_args.f16_begin / f16_end): the conversion pass
        of the default path is skipped, the arithmetic is the same."""
        if query.dtype != torch.bfloat16:
            raise TypeError(f"attention expects bf16, got {query.dtype}")
        B, S, H, D = query.shape
        if out is None:
            out = torch.empty((B, S, H, D), dtype=query.dtype, device=query.device)
        lse = (torch.empty((B, H, S), dtype=torch.float32, device=query.device)
               if return_softmax_lse else None)
        if value_is_half:
            _C.flash_attn_fwd(query, key, value, out, lse, self.softmax_scale, False, v_is_f16=True)
        else:
            _C.attention(query, key, value, out, lse, self.softmax_scale)
        return (out, lse) if return_softmax_lse else out


class Mi355AttentionBackend(AttentionBackend):
    accept_output_buffer: bool = True

    @staticmethod
    def get_supported_head_sizes() -> list[int]:
        return [64, 128]

    @staticmethod
    def get_enum() -> AttentionBackendEnum:
        return AttentionBackendEnum.AITER

    @staticmethod
    def get_impl_cls() -> type[Mi355AttentionImpl]:
        return Mi355AttentionImpl

    @staticmethod
    def get_metadata_cls() -> type[Mi355AttentionMetadata]:
        return Mi355AttentionMetadata

    @staticmethod
    def get_builder_cls() -> type[Mi355AttentionMetadataBuilder]:
        return Mi355AttentionMetadataBuilder


def get_attn_backend_cls_str(selected_backend: Optional[AttentionBackendEnum], head_size: int,
                             dtype: torch.dtype) -> str:
    """Platform hook, as RocmPlatform.get_attn_backend_cls_str (platforms/rocm.py:95-169)."""
    if selected_backend not in (None, AttentionBackendEnum.AITER, AttentionBackendEnum.FA):
        raise ValueError(f"attention backend {selected_backend} is not available on this build; "
                         "only the MI355X flash-attention kernel is")
    if dtype != torch.bfloat16:
        raise ValueError(f"dtype {dtype} not supported by the MI355X attention backend")
    return f"{__name__}.Mi355AttentionBackend"


@lru_cache(maxsize=None)
def _cached_get_attn_backend(head_size: int, dtype: torch.dtype,
                             supported: Optional[frozenset]) -> type[AttentionBackend]:
    if supported is not None and AttentionBackendEnum.AITER not in supported \
            and AttentionBackendEnum.FA not in supported:
        raise ValueError(f"none of {sorted(map(str, supported))} is available on this build")
    get_attn_backend_cls_str(None, head_size, dtype)
    return Mi355AttentionBackend


def get_attn_backend(head_size: int, dtype: torch.dtype,
                     supported_attention_backends=None) -> type[AttentionBackend]:
    """selector.py:89-159: cached on (head_size, dtype, supported set)."""
    sup = frozenset(supported_attention_backends) if supported_attention_backends else None
    return _cached_get_attn_backend(head_size, dtype, sup)


class LocalAttention(nn.Module):
    """layer.py:221-285."""

    def __init__(self, num_heads: int, head_size: int, num_kv_heads: int | None = None,
                 softmax_scale: float | None = None, causal: bool = False,
                 supported_attention_backends=None, **extra_impl_args) -> None:
        super().__init__()
        self.softmax_scale = head_size ** -0.5 if softmax_scale is None else softmax_scale
        num_kv_heads = num_kv_heads or num_heads
        backend = get_attn_backend(head_size, torch.bfloat16, supported_attention_backends)
        self.attn_impl = backend.get_impl_cls()(num_heads=num_heads, head_size=head_size,
                                               softmax_scale=self.softmax_scale,
                                               num_kv_heads=num_kv_heads, causal=causal,
                                               **extra_impl_args)
        self.num_heads, self.head_size, self.num_kv_heads = num_heads, head_size, num_kv_heads
        self.backend = backend.get_enum()

    def forward(self, q, k, v):
        assert q.dim() == 4 and k.dim() == 4 and v.dim() == 4, "Expected 4D tensors"
        return self.attn_impl.forward(q, k, v, attn_metadata=get_forward_context().attn_metadata)


class USPAttention(nn.Module):
    """layer.py:288-470.  With no sequence-parallel group (the data-parallel Flux path) this
    is local attention.  With a Ulysses group it does the head-scatter / sequence-gather
    all-to-all around the local kernel (layers/usp.py:49-158); a leading run of tokens that is
    REPLICATED on every rank (the text tokens of Flux's joint attention) stays out of the
    exchange and is head-sliced locally instead (layer.py:414-470).  With a ring group the
    exchanged tensors are attended ring-wise (layers/usp.py:161-252).

    `attn_impl` may be given for tests without a GPU; the default is the HIP backend."""

    def __init__(self, num_heads: int, head_size: int, num_kv_heads: int | None = None,
                 softmax_scale: float | None = None, causal: bool = False,
                 supported_attention_backends=None, prefix: str = "", dropout_rate: float = 0.0,
                 skip_sequence_parallel: bool = False, ulysses_group=None, ring_group=None,
                 attn_impl=None, **extra_impl_args):
        super().__init__()
        self.softmax_scale = head_size ** -0.5 if softmax_scale is None else softmax_scale
        num_kv_heads = num_kv_heads or num_heads
        if attn_impl is None:
            backend = get_attn_backend(head_size, torch.bfloat16, supported_attention_backends)
            attn_impl = backend.get_impl_cls()(num_heads=num_heads, head_size=head_size,
                                               causal=causal, softmax_scale=self.softmax_scale,
                                               num_kv_heads=num_kv_heads, prefix=f"{prefix}.impl",
                                               **extra_impl_args)
            self.backend = backend.get_enum()
        self.attn_impl = attn_impl
        self.num_heads, self.head_size, self.num_kv_heads = num_heads, head_size, num_kv_heads
        self.causal = causal
        self.skip_sequence_parallel = skip_sequence_parallel
        self.ulysses_group = ulysses_group
        self.ring_group = ring_group

    @staticmethod
    def _size(group) -> int:
        import torch.distributed as dist

        return dist.get_world_size(group) if (group is not None and dist.is_initialized()) else 1

    def forward(self, q, k, v, num_replicated_prefix: int = 0):
        md = get_forward_context().attn_metadata
        up, rp = self._size(self.ulysses_group), self._size(self.ring_group)
        if self.skip_sequence_parallel or (up == 1 and rp == 1):
            return self.attn_impl.forward(q, k, v, md)
        if up > 1 and num_replicated_prefix > 0:
            if rp > 1:
                raise NotImplementedError("a replicated prefix under ring attention is not built")
            return self._forward_with_replicated_prefix(q, k, v, md, num_replicated_prefix)
        from ..runtime.parallel import ring_attention, ulysses_all_to_all

        if up > 1:
            qkv = ulysses_all_to_all(torch.cat([q, k, v], dim=0), self.ulysses_group,
                                     scatter_dim=2, gather_dim=1)
            q, k, v = qkv.chunk(3, dim=0)
        if rp > 1:
            if self.causal:
                raise NotImplementedError("ring attention: causal masking is not built")
            impl = self.attn_impl

            def attn_fn(q_, k_, v_, scale):
                return impl.forward(q_, k_, v_, md, return_softmax_lse=True)

            merge = getattr(impl, "merge_states", None)
            out, _ = ring_attention(q, k, v, self.ring_group, self.softmax_scale, attn_fn=attn_fn,
                                    merge_fn=merge)
        else:
            out = self.attn_impl.forward(q, k, v, md)
        if up > 1:
            out = ulysses_all_to_all(out.contiguous(), self.ulysses_group, scatter_dim=1, gather_dim=2)
        return out

    def _forward_with_replicated_prefix(self, q, k, v, md, num_rep: int):
        """layer.py:414-470: only the sharded suffix goes through the all-to-all (q, k, v stacked
        in one exchange); the replicated prefix is sliced to this rank's heads, attention runs
        on [prefix | gathered suffix], the suffix of the result is exchanged back and the
        prefix's heads are all-gathered."""
        import torch.distributed as dist

        from ..runtime.parallel import ulysses_all_to_all

        group = self.ulysses_group
        P, rank = dist.get_world_size(group), dist.get_rank(group)
        H = q.shape[2]
        if H % P:
            raise ValueError(f"{H} heads do not divide over {P} ranks")
        shard = ulysses_all_to_all(torch.cat([q[:, num_rep:], k[:, num_rep:], v[:, num_rep:]], dim=0),
                                   group, scatter_dim=2, gather_dim=1)
        qs, ks, vs = shard.chunk(3, dim=0)
        hs = slice(rank * (H // P), (rank + 1) * (H // P))
        out = self.attn_impl.forward(torch.cat([q[:, :num_rep, hs], qs], dim=1),
                                     torch.cat([k[:, :num_rep, hs], ks], dim=1),
                                     torch.cat([v[:, :num_rep, hs], vs], dim=1), md)
        out_shard = ulysses_all_to_all(out[:, num_rep:].contiguous(), group, scatter_dim=1,
                                       gather_dim=2)
        from ..runtime.parallel import all_gather_tensors

        parts = all_gather_tensors(out[:, :num_rep], group)
        return torch.cat([torch.cat(parts, dim=2), out_shard], dim=1)


# Optional per-phase timing of the sequence-parallel wrappers (benchmarks only): a callable
# name -> context manager; None (the default) costs one attribute read per phase.
PHASE_TIMER = None


def _phase(name: str):
    return PHASE_TIMER(name) if PHASE_TIMER is not None else contextlib.nullcontext()


def _append_tokens(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """[B, Sa, H, D] followed by [B, Sb, H, D] along the token dimension.  When `b` already
    sits right behind `a` in one buffer with the same strides (the two streams' slices of a
    joint q|k|v projection) the result is a view of that buffer; otherwise a concatenation."""
    if (a.dtype == b.dtype and a.device == b.device and a.stride() == b.stride()
            and a.shape[0] == b.shape[0] and a.shape[2:] == b.shape[2:]
            and b.data_ptr() == a.data_ptr() + a.shape[1] * a.stride(1) * a.element_size()
            and (a.shape[0] == 1 or a.stride(0) >= (a.shape[1] + b.shape[1]) * a.stride(1))):
        return torch.as_strided(a, (a.shape[0], a.shape[1] + b.shape[1]) + tuple(a.shape[2:]),
                                a.stride(), a.storage_offset())
    return torch.cat([a, b], dim=1)


class UlyssesAttention(nn.Module):
    """layer.py:38-154: sequence-parallel attention with an optional REPLICATED token set.

    q, k, v hold this rank's slice of the main (image / video) sequence with all heads; the
    exchange turns them into the whole sequence with H / P heads.  The replicated tokens (text)
    are present on every rank with all heads: each rank takes its own H / P heads of them
    locally, appends them BEHIND the main sequence, runs the kernel once, and the text part of
    the result is all-gathered over heads.  q, k and v travel stacked in ONE all-to-all
    (three tensors, one message per peer) instead of the reference's cat along the batch.

    `attn_impl` may be given for tests without a GPU; the default is the HIP backend."""

    def __init__(self, num_heads: int, head_size: int, num_kv_heads: int | None = None,
                 softmax_scale: float | None = None, causal: bool = False,
                 supported_attention_backends=None, prefix: str = "", group=None,
                 attn_impl=None, **extra_impl_args) -> None:
        super().__init__()
        self.softmax_scale = head_size ** -0.5 if softmax_scale is None else softmax_scale
        num_kv_heads = num_kv_heads or num_heads
        if attn_impl is None:
            backend = get_attn_backend(head_size, torch.bfloat16, supported_attention_backends)
            attn_impl = backend.get_impl_cls()(num_heads=num_heads, head_size=head_size,
                                               causal=causal, softmax_scale=self.softmax_scale,
                                               num_kv_heads=num_kv_heads, prefix=f"{prefix}.impl",
                                               **extra_impl_args)
            self.backend = backend.get_enum()
        self.attn_impl = attn_impl
        self.num_heads, self.head_size, self.num_kv_heads = num_heads, head_size, num_kv_heads
        self.group = group

    def forward(self, q, k, v, replicated_q=None, replicated_k=None, replicated_v=None, *,
                value_is_half: bool = False, out: Optional[torch.Tensor] = None):
        """value_is_half: see Mi355AttentionImpl.forward.  out: with ONE rank, a [B, S + S_rep, H,
        D] buffer (any row stride) the kernel writes the joint result into -- the two returned
        tensors are then views of it; ignored when the result comes out of an exchange."""
        assert q.dim() == 4 and k.dim() == 4 and v.dim() == 4, "Expected 4D tensors"
        import torch.distributed as dist

        from ..runtime.parallel import ulysses_all_to_all

        B, S, H, D = q.shape
        group = self.group
        P = dist.get_world_size(group) if (group is not None and dist.is_initialized()) else 1
        rank = dist.get_rank(group) if P > 1 else 0
        md = get_forward_context().attn_metadata
        if P > 1:
            if H % P:
                raise ValueError(f"{H} heads do not divide over {P} ranks")
            with _phase("all_to_all"):
                qkv = ulysses_all_to_all(torch.cat([q, k, v], dim=0), group, scatter_dim=2,
                                         gather_dim=1)
            q, k, v = qkv.chunk(3, dim=0)
        n_main = q.shape[1]
        if replicated_q is not None:
            assert replicated_k is not None and replicated_v is not None
            hp = H // P
            hs = slice(rank * hp, (rank + 1) * hp)
            q = _append_tokens(q, replicated_q[:, :, hs])
            k = _append_tokens(k, replicated_k[:, :, hs])
            v = _append_tokens(v, replicated_v[:, :, hs])
        kw = {"value_is_half": True} if value_is_half else {}
        if out is not None and P == 1 and tuple(out.shape) == tuple(q.shape):
            kw["out"] = out
        with _phase("attention"):
            out = self.attn_impl.forward(q, k, v, md, **kw)
        rep_out = None
        if replicated_q is not None:
            rep_out = out[:, n_main:]
            out = out[:, :n_main]
            if P > 1:
                from ..runtime.parallel import all_gather_tensors

                with _phase("all_gather"):
                    rep_out = torch.cat(all_gather_tensors(rep_out, group), dim=2)
        if P > 1:
            with _phase("all_to_all"):
                out = ulysses_all_to_all(out.contiguous(), group, scatter_dim=1, gather_dim=2)
        return out, rep_out
