"""Paged KV-cache layout helpers and the decode-attention entry point.

Mirror of light_vllm/decoding/backends/attention/ops/paged_attn.py:34-248 (class
PagedAttention: get_kv_cache_shape, split_kv_cache, write_to_paged_cache, forward_decode,
swap_blocks, copy_blocks) on top of the gfx950 operators.  The V1/V2 choice is this build's
own (the reference's heuristic, paged_attn.py:118-128, is tuned for one workgroup per query
head; here one workgroup serves a whole GQA group):
  V1 (one workgroup per (kv head, sequence)) when that grid alone fills the 256 CUs or the
  context is a single partition; V2 (512-token partitions) otherwise.
"""
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch

from . import _custom_ops as ops

_PARTITION_SIZE = 512  # csrc/attention/attention_kernels.cu:850
_NUM_CUS = 256


@dataclass
class PagedAttentionMetadata:
    """Metadata for PagedAttention (paged_attn.py:17-31 of the reference)."""
    # (batch_size,). The length of sequences (entire tokens seen so far) per sequence.
    seq_lens_tensor: Optional[torch.Tensor]
    # Maximum sequence length in the batch. 0 if it is prefill-only batch.
    max_decode_seq_len: int
    # (batch_size, max_blocks_per_seq). Block addresses per sequence.
    block_tables: Optional[torch.Tensor]


class PagedAttention:

    @staticmethod
    def get_supported_head_sizes() -> List[int]:
        return [64, 80, 96, 112, 120, 128, 192, 256]

    @staticmethod
    def get_kv_cache_shape(num_blocks: int, block_size: int, num_kv_heads: int,
                           head_size: int) -> Tuple[int, ...]:
        return (2, num_blocks, block_size * num_kv_heads * head_size)

    @staticmethod
    def split_kv_cache(kv_cache: torch.Tensor, num_kv_heads: int,
                       head_size: int) -> Tuple[torch.Tensor, torch.Tensor]:
        x = 16 // kv_cache.element_size()
        num_blocks = kv_cache.shape[1]
        key_cache = kv_cache[0].view(num_blocks, num_kv_heads, head_size // x, -1, x)
        value_cache = kv_cache[1].view(num_blocks, num_kv_heads, head_size, -1)
        return key_cache, value_cache

    @staticmethod
    def write_to_paged_cache(key: torch.Tensor, value: torch.Tensor, key_cache: torch.Tensor,
                             value_cache: torch.Tensor, slot_mapping: torch.Tensor,
                             kv_cache_dtype: str, k_scale: float, v_scale: float) -> None:
        ops.reshape_and_cache(key, value, key_cache, value_cache, slot_mapping.flatten(),
                              kv_cache_dtype, k_scale, v_scale)

    @staticmethod
    def use_v1(num_seqs: int, num_kv_heads: int, num_heads: int, max_seq_len: int) -> bool:
        max_num_partitions = (max_seq_len + _PARTITION_SIZE - 1) // _PARTITION_SIZE
        head_groups = (num_heads // num_kv_heads + 15) // 16
        workgroups_v1 = num_seqs * num_kv_heads * head_groups
        return max_num_partitions == 1 or workgroups_v1 >= _NUM_CUS

    @staticmethod
    def forward_decode(
        query: torch.Tensor,
        key_cache: torch.Tensor,
        value_cache: torch.Tensor,
        block_tables: torch.Tensor,
        seq_lens: torch.Tensor,
        max_seq_len: int,
        kv_cache_dtype: str,
        num_kv_heads: int,
        scale: float,
        alibi_slopes: Optional[torch.Tensor],
        k_scale: float,
        v_scale: float,
        tp_rank: int = 0,
        blocksparse_local_blocks: int = 0,
        blocksparse_vert_stride: int = 0,
        blocksparse_block_size: int = 64,
        blocksparse_head_sliding_step: int = 0,
        force_version: Optional[str] = None,
        scratch: Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = None,
        output: Optional[torch.Tensor] = None,
    ) -> torch.Tensor:
        if output is None:
            output = torch.empty_like(query)
        block_size = value_cache.shape[3]
        num_seqs, num_heads, head_size = query.shape
        max_num_partitions = (max_seq_len + _PARTITION_SIZE - 1) // _PARTITION_SIZE
        if force_version is None:
            v1 = PagedAttention.use_v1(num_seqs, num_kv_heads, num_heads, max_seq_len)
        else:
            v1 = force_version == "v1"
        if v1:
            ops.paged_attention_v1(output, query, key_cache, value_cache, num_kv_heads, scale,
                                   block_tables, seq_lens, block_size, max_seq_len, alibi_slopes,
                                   kv_cache_dtype, k_scale, v_scale, tp_rank,
                                   blocksparse_local_blocks, blocksparse_vert_stride,
                                   blocksparse_block_size, blocksparse_head_sliding_step)
        else:
            if scratch is None:
                # scratch shapes as paged_attn.py:156-166 of the reference
                tmp_output = torch.empty(size=(num_seqs, num_heads, max_num_partitions, head_size),
                                         dtype=output.dtype, device=output.device)
                exp_sums = torch.empty(size=(num_seqs, num_heads, max_num_partitions),
                                       dtype=torch.float32, device=output.device)
                max_logits = torch.empty_like(exp_sums)
            else:
                exp_sums, max_logits, tmp_output = scratch
            ops.paged_attention_v2(output, exp_sums, max_logits, tmp_output, query, key_cache,
                                   value_cache, num_kv_heads, scale, block_tables, seq_lens,
                                   block_size, max_seq_len, alibi_slopes, kv_cache_dtype, k_scale,
                                   v_scale, tp_rank, blocksparse_local_blocks,
                                   blocksparse_vert_stride, blocksparse_block_size,
                                   blocksparse_head_sliding_step)
        return output

    @staticmethod
    def forward_prefix(query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                       key_cache: torch.Tensor, value_cache: torch.Tensor, block_tables: torch.Tensor,
                       query_start_loc: torch.Tensor, seq_lens_tensor: torch.Tensor,
                       context_lens: Optional[torch.Tensor], max_query_len: int,
                       alibi_slopes: Optional[torch.Tensor], sliding_window: Optional[int],
                       scale: Optional[float] = None, softcap: float = 0.0,
                       kv_cache_dtype: str = "auto", output: Optional[torch.Tensor] = None,
                       k_scale: float = 1.0, v_scale: float = 1.0) -> torch.Tensor:
        """Prompt chunks against the paged cache (paged_attn.py:194-225 of the reference, which
        runs the Triton context_attention_fwd; the live backend calls flash_attn_varlen_func with
        block_table, flash_attn.py:538-555).  query [T, H, D]; `key`/`value` of the chunk must
        already be in the cache (write_to_paged_cache runs first in both reference backends), so
        they are not read here; `context_lens` is implied by seq_lens - query lengths."""
        del key, value, context_lens
        if output is None:
            output = torch.empty_like(query)
        num_kv_heads = key_cache.shape[1]
        if scale is None:
            scale = float(query.shape[-1]) ** -0.5  # context_attention_fwd's own default
        ops.paged_prefill_attention(output, query, key_cache, value_cache, num_kv_heads, scale,
                                    block_tables, seq_lens_tensor, query_start_loc, max_query_len,
                                    value_cache.shape[3], alibi_slopes, sliding_window or 0, softcap,
                                    kv_cache_dtype, True, k_scale, v_scale)
        return output

    @staticmethod
    def swap_blocks(src_kv_cache: torch.Tensor, dst_kv_cache: torch.Tensor,
                    src_to_dst: torch.Tensor) -> None:
        ops.swap_blocks(src_kv_cache[0], dst_kv_cache[0], src_to_dst)
        ops.swap_blocks(src_kv_cache[1], dst_kv_cache[1], src_to_dst)

    @staticmethod
    def copy_blocks(kv_caches: List[torch.Tensor], src_to_dists: torch.Tensor) -> None:
        key_caches = [kv_cache[0] for kv_cache in kv_caches]
        value_caches = [kv_cache[1] for kv_cache in kv_caches]
        ops.copy_blocks(key_caches, value_caches, src_to_dists)
