"""Operator wrappers of the paged-attention decode path.

Same names, argument order and defaults as the reference's
light_vllm/backends/_custom_ops.py (:46-47 silu_and_mul, :71-130 paged_attention_v1/v2,
:134-143 rotary_embedding, :157-165 rms_norm/fused_add_rms_norm, :424-463 cache ops,
:473-480 device attributes); each one calls the `torch.ops._C*` operator that
light-vllm_amd/csrc/torch_bindings.cpp registers with the reference's schema and
that runs a hand-written gfx950 kernel.  Importing this module loads the native
libraries and raises if they are missing -- there is no other implementation.
"""
from typing import List, Optional

import torch

from . import _native

_native.load_torch_ops()


def is_custom_op_supported(op_name: str) -> bool:
    op, overloads = torch._C._jit_get_operation(op_name)
    return op is not None


_C = torch.ops._C
_CACHE = torch.ops._C_cache_ops
_UTILS = torch.ops._C_cuda_utils

# Optional trailing arguments of the two attention operators and their defaults
# (_custom_ops.py:86-90,117-121 of the reference).
_BS_DEFAULTS = dict(tp_rank=0, blocksparse_local_blocks=0, blocksparse_vert_stride=0,
                    blocksparse_block_size=64, blocksparse_head_sliding_step=0)


def silu_and_mul(out: torch.Tensor, x: torch.Tensor) -> None:
    _C.silu_and_mul(out, x)


def paged_attention_v1(out: torch.Tensor, query: torch.Tensor, key_cache: torch.Tensor,
                       value_cache: torch.Tensor, num_kv_heads: int, scale: float,
                       block_tables: torch.Tensor, seq_lens: torch.Tensor, block_size: int,
                       max_seq_len: int, alibi_slopes: Optional[torch.Tensor], kv_cache_dtype: str,
                       k_scale: float, v_scale: float, tp_rank: int = 0,
                       blocksparse_local_blocks: int = 0, blocksparse_vert_stride: int = 0,
                       blocksparse_block_size: int = 64, blocksparse_head_sliding_step: int = 0) -> None:
    _C.paged_attention_v1(out, query, key_cache, value_cache, num_kv_heads, scale, block_tables, seq_lens,
                          block_size, max_seq_len, alibi_slopes, kv_cache_dtype, k_scale, v_scale, tp_rank,
                          blocksparse_local_blocks, blocksparse_vert_stride, blocksparse_block_size,
                          blocksparse_head_sliding_step)


def paged_attention_v2(out: torch.Tensor, exp_sum: torch.Tensor, max_logits: torch.Tensor,
                       tmp_out: torch.Tensor, query: torch.Tensor, key_cache: torch.Tensor,
                       value_cache: torch.Tensor, num_kv_heads: int, scale: float,
                       block_tables: torch.Tensor, seq_lens: torch.Tensor, block_size: int,
                       max_seq_len: int, alibi_slopes: Optional[torch.Tensor], kv_cache_dtype: str,
                       k_scale: float, v_scale: float, tp_rank: int = 0,
                       blocksparse_local_blocks: int = 0, blocksparse_vert_stride: int = 0,
                       blocksparse_block_size: int = 64, blocksparse_head_sliding_step: int = 0) -> None:
    _C.paged_attention_v2(out, exp_sum, max_logits, tmp_out, query, key_cache, value_cache, num_kv_heads,
                          scale, block_tables, seq_lens, block_size, max_seq_len, alibi_slopes,
                          kv_cache_dtype, k_scale, v_scale, tp_rank, blocksparse_local_blocks,
                          blocksparse_vert_stride, blocksparse_block_size, blocksparse_head_sliding_step)


def paged_prefill_attention(out: torch.Tensor, query: torch.Tensor, key_cache: torch.Tensor,
                            value_cache: torch.Tensor, num_kv_heads: int, scale: float,
                            block_tables: torch.Tensor, seq_lens: torch.Tensor,
                            query_start_loc: torch.Tensor, max_query_len: int, block_size: int,
                            alibi_slopes: Optional[torch.Tensor] = None, sliding_window: int = 0,
                            softcap: float = 0.0, kv_cache_dtype: str = "auto", causal: bool = True,
                            k_scale: float = 1.0, v_scale: float = 1.0, max_seq_len: int = 0) -> None:
    """Causal varlen attention of prompt chunks over the paged cache: the job of
    flash_attn_varlen_func(..., block_table=...) at flash_attn.py:538-555 of the reference.
    max_seq_len: a bound on seq_lens when the caller has one (the reference passes max_seqlen_k); launches of short
    chunks use it to cut long key walks across workgroups (csrc/prefill_chunk.h)."""
    torch.ops._C_amd.paged_prefill_attention(out, query, key_cache, value_cache, num_kv_heads, scale,
                                             block_tables, seq_lens, query_start_loc, max_query_len,
                                             block_size, alibi_slopes, sliding_window, softcap,
                                             kv_cache_dtype, causal, k_scale, v_scale, max_seq_len)


_VARLEN_WS = {}


def varlen_attention(out: torch.Tensor, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                     cu_seqlens: torch.Tensor, max_seq_len: int, scale: float, causal: bool,
                     alibi_slopes: Optional[torch.Tensor] = None, sliding_window: int = 0,
                     softcap: float = 0.0, workspace: Optional[torch.Tensor] = None) -> None:
    """Dense varlen attention without a KV cache: the prefill-only backends' job
    (flash_attn_varlen_func(q, k, v, cu_seqlens, causal=...), torch_naive.py:65-149).
    query [T, H, D], key/value [T, KVH, D], cu_seqlens int32 [num_seqs + 1] on the device."""
    need = torch.ops._C_amd.varlen_attention_workspace_bytes(query.shape[0], cu_seqlens.numel() - 1,
                                                             max_seq_len, key.shape[1], query.shape[2])
    if workspace is None:
        # one scratch buffer per (device, stream): steps in flight on different streams must not share it
        ws_key = (query.device, torch.cuda.current_stream(query.device).cuda_stream)
        workspace = _VARLEN_WS.get(ws_key)
        if workspace is None or workspace.numel() < need:
            workspace = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=query.device)
            _VARLEN_WS[ws_key] = workspace
    torch.ops._C_amd.varlen_attention(out, query, key, value, cu_seqlens, max_seq_len, scale, causal,
                                      alibi_slopes, sliding_window, softcap, workspace)


def scaled_fp8_quant(input: torch.Tensor, scale: Optional[torch.Tensor] = None,
                     num_token_padding: Optional[int] = None, scale_ub: Optional[torch.Tensor] = None,
                     use_per_token_if_dynamic: bool = False):
    """Quantize `input` [tokens, hidden] to fp8 (e4m3fn) and return (quantized, scale): static when a
    scale is given, else dynamic per tensor or per token (_custom_ops.py:313-365 of the reference)."""
    assert input.ndim == 2
    shape = input.shape
    if num_token_padding:
        shape = (max(num_token_padding, input.shape[0]), shape[1])
    output = torch.empty(shape, device=input.device, dtype=torch.float8_e4m3fn)
    if scale is None:
        if use_per_token_if_dynamic:
            scale = torch.empty((shape[0], 1), device=input.device, dtype=torch.float32)
            _C.dynamic_per_token_scaled_fp8_quant(output, input, scale, scale_ub)
        else:
            scale = torch.zeros(1, device=input.device, dtype=torch.float32)
            _C.dynamic_scaled_fp8_quant(output, input, scale)
    else:
        assert scale.numel() == 1 or num_token_padding is None
        _C.static_scaled_fp8_quant(output, input, scale)
    return output, scale


def advance_step(num_seqs: int, num_queries: int, block_size: int, input_tokens: torch.Tensor,
                 sampled_token_ids: torch.Tensor, input_positions: torch.Tensor, seq_lens: torch.Tensor,
                 slot_mapping: torch.Tensor, block_tables: torch.Tensor) -> None:
    """Advance a decode batch's input tensors one token on the device (_custom_ops.py:167-178)."""
    return _C.advance_step(num_seqs, num_queries, block_size, input_tokens, sampled_token_ids,
                           input_positions, seq_lens, slot_mapping, block_tables)


def rotary_embedding(positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor, head_size: int,
                     cos_sin_cache: torch.Tensor, is_neox: bool) -> None:
    _C.rotary_embedding(positions, query, key, head_size, cos_sin_cache, is_neox)


def rms_norm(out: torch.Tensor, input: torch.Tensor, weight: torch.Tensor, epsilon: float) -> None:
    _C.rms_norm(out, input, weight, epsilon)


def fused_add_rms_norm(input: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor,
                       epsilon: float) -> None:
    _C.fused_add_rms_norm(input, residual, weight, epsilon)


def reshape_and_cache(key: torch.Tensor, value: torch.Tensor, key_cache: torch.Tensor,
                      value_cache: torch.Tensor, slot_mapping: torch.Tensor, kv_cache_dtype: str,
                      k_scale: float, v_scale: float) -> None:
    _CACHE.reshape_and_cache(key, value, key_cache, value_cache, slot_mapping, kv_cache_dtype, k_scale, v_scale)


def reshape_and_cache_flash(key: torch.Tensor, value: torch.Tensor, key_cache: torch.Tensor,
                            value_cache: torch.Tensor, slot_mapping: torch.Tensor, kv_cache_dtype: str,
                            k_scale: float, v_scale: float) -> None:
    _CACHE.reshape_and_cache_flash(key, value, key_cache, value_cache, slot_mapping, kv_cache_dtype, k_scale,
                                   v_scale)


def copy_blocks(key_caches: List[torch.Tensor], value_caches: List[torch.Tensor],
                block_mapping: torch.Tensor) -> None:
    _CACHE.copy_blocks(key_caches, value_caches, block_mapping)


def swap_blocks(src: torch.Tensor, dst: torch.Tensor, block_mapping: torch.Tensor) -> None:
    _CACHE.swap_blocks(src, dst, block_mapping)


def convert_fp8(output: torch.Tensor, input: torch.Tensor, scale: float = 1.0, kv_dtype: str = "fp8") -> None:
    """fp8 <-> float/half/bfloat16 elementwise (_custom_ops.py:466-470 of the reference)."""
    torch.ops._C_cache_ops.convert_fp8(output, input, scale, kv_dtype)


def get_device_attribute(attribute: int, device: int) -> int:
    return _UTILS.get_device_attribute(attribute, device)


def get_max_shared_memory_per_block_device_attribute(device: int) -> int:
    return _UTILS.get_max_shared_memory_per_block_device_attribute(device)
