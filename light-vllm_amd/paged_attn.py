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
_HIP_ATTR_MULTIPROCESSOR_COUNT = 63  # hipDeviceAttributeMultiprocessorCount (hip_runtime_api.h)
_cu_count_of = {}


def num_compute_units(device=None) -> int:
    """CUs of the device a launch goes to, asked once per device through the reference's own utility op
    (`_C_cuda_utils.get_device_attribute`, torch_bindings.cpp:271-274) -- the C side sizes its grids from the same
    attribute; 256 on an MI355X."""
    idx = torch.device(device).index if device is not None else None
    if idx is None:
        idx = torch.cuda.current_device()
    n = _cu_count_of.get(idx)
    if n is None:
        n = _cu_count_of[idx] = int(ops.get_device_attribute(_HIP_ATTR_MULTIPROCESSOR_COUNT, idx))
        if n <= 0:
            raise RuntimeError(f"device {idx} reports {n} compute units")
    return n


@dataclass
class PagedAttentionMetadata:
    """What a decode launch needs besides the tensors (paged_attn.py:17-31 of the reference):
    per-sequence lengths [batch], the longest of them (0 for a prompt-only batch), and the block
    table [batch, max blocks per sequence]."""
    seq_lens_tensor: Optional[torch.Tensor]
    max_decode_seq_len: int
    block_tables: Optional[torch.Tensor]


def _ceil_div(a: int, b: int) -> int:
    return -(-a // b)


class PagedAttention:
    """Static helpers only, as in the reference: the class is a namespace."""

    @staticmethod
    def get_supported_head_sizes() -> List[int]:
        return [64, 80, 96, 112, 120, 128, 192, 256]

    @staticmethod
    def get_kv_cache_shape(num_blocks: int, block_size: int, num_kv_heads: int,
                           head_size: int) -> Tuple[int, ...]:
        # one tensor per layer: plane 0 = keys, plane 1 = values, a block's elements flattened
        return (2, num_blocks, block_size * num_kv_heads * head_size)

    @staticmethod
    def split_kv_cache(kv_cache: torch.Tensor, num_kv_heads: int,
                       head_size: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """The two planes as the kernels address them: K [NB, KVH, D/x, BS, x] with x = 16 bytes
        worth of elements (8 for 16-bit types, 16 for an fp8 cache), V [NB, KVH, D, BS]."""
        keys, values = kv_cache[0], kv_cache[1]
        nb = kv_cache.shape[1]
        x = 16 // kv_cache.element_size()
        return keys.view(nb, num_kv_heads, head_size // x, -1, x), values.view(nb, num_kv_heads, head_size, -1)

    @staticmethod
    def write_to_paged_cache(key: torch.Tensor, value: torch.Tensor, key_cache: torch.Tensor,
                             value_cache: torch.Tensor, slot_mapping: torch.Tensor,
                             kv_cache_dtype: str, k_scale: float, v_scale: float) -> None:
        ops.reshape_and_cache(key, value, key_cache, value_cache, slot_mapping.flatten(), kv_cache_dtype,
                              k_scale, v_scale)

    @staticmethod
    def use_v1(num_seqs: int, num_kv_heads: int, num_heads: int, max_seq_len: int,
               num_cus: Optional[int] = None) -> bool:
        """One pass (v1) when the (sequence, kv head, head group) grid alone fills the CUs or the
        context fits a single partition; otherwise v2 cuts the contexts.  `num_cus`: the CUs of the device the
        launch goes to (default: asked of the current device)."""
        if _ceil_div(max_seq_len, _PARTITION_SIZE) == 1:
            return True
        groups_per_kv_head = _ceil_div(num_heads // num_kv_heads, 16)
        if num_cus is None:
            num_cus = num_compute_units()
        return num_seqs * num_kv_heads * groups_per_kv_head >= num_cus

    @staticmethod
    def forward_decode(query: torch.Tensor, key_cache: torch.Tensor, value_cache: torch.Tensor,
                       block_tables: torch.Tensor, seq_lens: torch.Tensor, max_seq_len: int,
                       kv_cache_dtype: str, num_kv_heads: int, scale: float,
                       alibi_slopes: Optional[torch.Tensor], k_scale: float, v_scale: float,
                       tp_rank: int = 0, blocksparse_local_blocks: int = 0, blocksparse_vert_stride: int = 0,
                       blocksparse_block_size: int = 64, blocksparse_head_sliding_step: int = 0,
                       force_version: Optional[str] = None,
                       scratch: Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = None,
                       output: Optional[torch.Tensor] = None) -> torch.Tensor:
        """query [num_seqs, H, D] -> output of the same shape.  `force_version` ("v1" | "v2") overrides
        the heuristic; `scratch` = (exp_sums, max_logits, tmp_out) lets a caller reuse the v2 buffers
        (shapes of paged_attn.py:156-166 of the reference)."""
        out = torch.empty_like(query) if output is None else output
        num_seqs, num_heads, head_size = query.shape
        tail = (kv_cache_dtype, k_scale, v_scale, tp_rank, blocksparse_local_blocks, blocksparse_vert_stride,
                blocksparse_block_size, blocksparse_head_sliding_step)
        common = (query, key_cache, value_cache, num_kv_heads, scale, block_tables, seq_lens,
                  value_cache.shape[3], max_seq_len, alibi_slopes)
        single_pass = (force_version == "v1") if force_version is not None else \
            PagedAttention.use_v1(num_seqs, num_kv_heads, num_heads, max_seq_len, num_compute_units(query.device))
        if single_pass:
            ops.paged_attention_v1(out, *common, *tail)
            return out
        if scratch is None:
            parts = _ceil_div(max_seq_len, _PARTITION_SIZE)
            exp_sums = torch.empty((num_seqs, num_heads, parts), dtype=torch.float32, device=out.device)
            scratch = (exp_sums, torch.empty_like(exp_sums),
                       torch.empty((num_seqs, num_heads, parts, head_size), dtype=out.dtype, device=out.device))
        ops.paged_attention_v2(out, scratch[0], scratch[1], scratch[2], *common, *tail)
        return out

    @staticmethod
    def forward_prefix(query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                       key_cache: torch.Tensor, value_cache: torch.Tensor, block_tables: torch.Tensor,
                       query_start_loc: torch.Tensor, seq_lens_tensor: torch.Tensor,
                       context_lens: Optional[torch.Tensor], max_query_len: int,
                       alibi_slopes: Optional[torch.Tensor], sliding_window: Optional[int],
                       scale: Optional[float] = None, softcap: float = 0.0,
                       kv_cache_dtype: str = "auto", output: Optional[torch.Tensor] = None,
                       k_scale: float = 1.0, v_scale: float = 1.0, max_seq_len: int = 0) -> torch.Tensor:
        """Prompt chunks against the paged cache (paged_attn.py:194-225 of the reference, which
        runs the Triton context_attention_fwd; the live backend calls flash_attn_varlen_func with
        block_table, flash_attn.py:538-555).  query [T, H, D]; `key`/`value` of the chunk must
        already be in the cache (write_to_paged_cache runs first in both reference backends), so
        they are not read here; `context_lens` is implied by seq_lens - query lengths.  `max_seq_len` (the reference
        passes max_seqlen_k, flash_attn.py:547): a bound on seq_lens valid for every run of this call -- launches of
        short chunks cut long key walks with it; 0 (captured steps, whose lengths change under the graph) = never."""
        del key, value, context_lens
        if output is None:
            output = torch.empty_like(query)
        num_kv_heads = key_cache.shape[1]
        if scale is None:
            scale = float(query.shape[-1]) ** -0.5  # context_attention_fwd's own default
        ops.paged_prefill_attention(output, query, key_cache, value_cache, num_kv_heads, scale,
                                    block_tables, seq_lens_tensor, query_start_loc, max_query_len,
                                    value_cache.shape[3], alibi_slopes, sliding_window or 0, softcap,
                                    kv_cache_dtype, True, k_scale, v_scale, max_seq_len)
        return output

    @staticmethod
    def swap_blocks(src_kv_cache: torch.Tensor, dst_kv_cache: torch.Tensor,
                    src_to_dst: torch.Tensor) -> None:
        for plane in (0, 1):  # keys, then values
            ops.swap_blocks(src_kv_cache[plane], dst_kv_cache[plane], src_to_dst)

    @staticmethod
    def copy_blocks(kv_caches: List[torch.Tensor], src_to_dists: torch.Tensor) -> None:
        ops.copy_blocks([c[0] for c in kv_caches], [c[1] for c in kv_caches], src_to_dists)
