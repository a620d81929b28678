"""Attention backend of the prefill-only (encode-only, no KV cache) workflow.

Mirrors light_vllm/prefill_only/backends/attention/backends/abstract.py:13-118 -- the backend /
metadata / builder / impl quadruple the reference's selector hands to `Attention` -- with one
implementation, the HIP varlen kernel (`lvllm_varlen_attention`): dense q/k/v cut by `seq_start_loc`,
causal for AttentionType.DECODER, bidirectional for AttentionType.ENCODER.  The reference offers
flash-attn / xformers / torch SDPA / torch-naive here (selector.py:15-120); tests compare this backend
with golden outputs of its torch-naive backend.
"""
import enum
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Type

import torch

from .. import _custom_ops as ops


class AttentionType(enum.Enum):
    """light_vllm/backends/attention/abstract.py: DECODER (causal), ENCODER (bidirectional)."""
    DECODER = enum.auto()
    ENCODER = enum.auto()
    ENCODER_DECODER = enum.auto()

    @staticmethod
    def attn_type_name_to_enum(attn_type: str) -> "AttentionType":
        assert attn_type is not None
        members = AttentionType.__members__
        if attn_type not in members:
            raise ValueError(f"Invalid attn_type '{attn_type}'. Available backends: {', '.join(members)} "
                             "(case-sensitive).")
        return AttentionType[attn_type]


@dataclass
class PrefillOnlyAttentionMetadata:
    max_seq_len: int
    seq_lens: List[int]
    # (batch_size + 1,) cumulative sequence lengths, e.g. [4, 6] -> [0, 4, 10]
    seq_start_loc: Optional[torch.Tensor]

    def to(self, device, non_blocking=True):
        for k, v in self.__dict__.items():
            if isinstance(v, torch.Tensor):
                self.__dict__[k] = v.to(device, non_blocking=non_blocking)
        return self


class PrefillOnlyAttentionMetadataBuilder:

    def __call__(self, seq_lens: List[int]) -> PrefillOnlyAttentionMetadata:
        pin = torch.cuda.is_available()
        seq_lens_tensor = torch.tensor(seq_lens, dtype=torch.long, pin_memory=pin, device="cpu")
        seq_start_loc = torch.zeros(seq_lens_tensor.shape[0] + 1, dtype=torch.int32, device="cpu")
        torch.cumsum(seq_lens_tensor, dim=0, dtype=seq_start_loc.dtype, out=seq_start_loc[1:])
        return PrefillOnlyAttentionMetadata(seq_lens=seq_lens, max_seq_len=max(seq_lens),
                                            seq_start_loc=seq_start_loc)


class PrefillOnlyHIPVarlenImpl:
    """forward(query [T, H*D], key [T, KVH*D], value, kv_cache=None, attn_metadata) -> [T, H*D]"""

    def __init__(self, num_heads: int, head_size: int, scale: float, num_kv_heads: Optional[int] = None,
                 alibi_slopes: Optional[List[float]] = None, sliding_window: Optional[int] = None,
                 kv_cache_dtype: str = "auto", blocksparse_params: Optional[Dict[str, Any]] = None,
                 logits_soft_cap: Optional[float] = None) -> None:
        if blocksparse_params is not None:
            raise ValueError("HIP varlen attention does not support block-sparse attention.")
        if kv_cache_dtype != "auto":
            raise NotImplementedError("HIP varlen attention has no KV cache: kv_cache_dtype must be 'auto'.")
        self.num_heads = num_heads
        self.head_size = head_size
        self.scale = float(scale)
        self.num_kv_heads = num_heads if num_kv_heads is None else num_kv_heads
        self.alibi_slopes = (torch.tensor(alibi_slopes, dtype=torch.float32)
                             if alibi_slopes is not None else None)
        self.sliding_window = sliding_window
        self.logits_soft_cap = float(logits_soft_cap or 0.0)
        assert self.num_heads % self.num_kv_heads == 0
        self.num_queries_per_kv = self.num_heads // self.num_kv_heads

    def forward(self, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                kv_cache: Optional[torch.Tensor], attn_metadata: PrefillOnlyAttentionMetadata,
                k_scale: float = 1.0, v_scale: float = 1.0,
                attn_type: AttentionType = AttentionType.DECODER) -> torch.Tensor:
        assert kv_cache is None
        assert k_scale == 1.0 and v_scale == 1.0, "key/v_scale is not supported in HIP varlen attention."
        if attn_type == AttentionType.ENCODER:
            causal = False
        elif attn_type == AttentionType.DECODER:
            causal = True
        else:
            raise NotImplementedError("Encoder/decoder cross-attention are not implemented for "
                                      "PrefillOnlyHIPVarlenImpl")
        num_tokens, hidden_size = query.shape
        q = query.view(-1, self.num_heads, self.head_size)
        k = key.view(-1, self.num_kv_heads, self.head_size)
        v = value.view(-1, self.num_kv_heads, self.head_size)
        alibi = self.alibi_slopes
        if alibi is not None and alibi.device != q.device:
            alibi = self.alibi_slopes = alibi.to(q.device)
        out = torch.empty_like(q)
        ops.varlen_attention(out, q, k, v, attn_metadata.seq_start_loc, attn_metadata.max_seq_len, self.scale,
                             causal, alibi, self.sliding_window or 0, self.logits_soft_cap)
        return out.view(num_tokens, hidden_size)


class PrefillOnlyHIPVarlenBackend:
    """`PrefillOnlyAttentionBackend` (abstract.py:13-50)."""

    def __init__(self, attn_type: AttentionType = AttentionType.DECODER):
        if attn_type == AttentionType.ENCODER_DECODER:
            raise NotImplementedError("Encoder/decoder cross-attention are not implemented for "
                                      "PrefillOnlyAttentionBackend")
        self._attn_type = attn_type

    @property
    def attn_type(self) -> AttentionType:
        return self._attn_type

    @staticmethod
    def get_name() -> str:
        return "hip_varlen"

    @staticmethod
    def get_impl_cls() -> Type[PrefillOnlyHIPVarlenImpl]:
        return PrefillOnlyHIPVarlenImpl

    @staticmethod
    def get_metadata_cls() -> Type[PrefillOnlyAttentionMetadata]:
        return PrefillOnlyAttentionMetadata

    @classmethod
    def make_metadata(cls, *args, **kwargs) -> PrefillOnlyAttentionMetadata:
        return cls.get_metadata_cls()(*args, **kwargs)

    @staticmethod
    def get_builder_cls() -> Type[PrefillOnlyAttentionMetadataBuilder]:
        return PrefillOnlyAttentionMetadataBuilder

    @classmethod
    def make_metadata_builder(cls, *args, **kwargs) -> PrefillOnlyAttentionMetadataBuilder:
        return cls.get_builder_cls()(*args, **kwargs)
