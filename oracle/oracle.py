"""ctypes wrapper over oracle/_build/liboracle.so (paged_ops_oracle.c).

TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Tensors are torch CPU tensors; outputs are written in place
into caller-allocated tensors, mirroring the operator signatures of
light_vllm/backends/_custom_ops.py.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None

_DT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            from . import build_oracle
            build_oracle.build_port()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_num_threads.restype = ctypes.c_int
        _lib.oracle_f32_to_f16.restype = ctypes.c_uint16
        _lib.oracle_f32_to_f16.argtypes = [ctypes.c_float]
        _lib.oracle_f32_to_bf16.restype = ctypes.c_uint16
        _lib.oracle_f32_to_bf16.argtypes = [ctypes.c_float]
        _lib.oracle_f16_to_f32.restype = ctypes.c_float
        _lib.oracle_f16_to_f32.argtypes = [ctypes.c_uint16]
        _lib.oracle_f32_to_e4m3.restype = ctypes.c_uint8
        _lib.oracle_f32_to_e4m3.argtypes = [ctypes.c_float]
        _lib.oracle_e4m3_to_f32.restype = ctypes.c_float
        _lib.oracle_e4m3_to_f32.argtypes = [ctypes.c_uint8]
    return _lib


def num_threads() -> int:
    return lib().oracle_num_threads()


def _p(t):
    if t is None:
        return ctypes.c_void_p(0)
    assert t.device.type == "cpu", "the oracle works on CPU tensors"
    return ctypes.c_void_p(t.data_ptr())


def _i(v):
    return ctypes.c_int(int(v))


def _l(v):
    return ctypes.c_int64(int(v))


def _f(v):
    return ctypes.c_float(float(v))


def paged_attention_v1(out, query, key_cache, value_cache, num_kv_heads, scale, block_tables,
                       seq_lens, block_size, max_seq_len, alibi_slopes=None):
    assert out.is_contiguous() and block_tables.dtype == torch.int32 and seq_lens.dtype == torch.int32
    lib().oracle_paged_attention_v1(
        _p(out), _p(query), _p(key_cache), _p(value_cache), _i(query.size(0)), _i(query.size(1)),
        _i(query.size(2)), _i(num_kv_heads), _f(scale), _p(block_tables), _p(seq_lens),
        _i(block_size), _i(block_tables.size(1)), _p(alibi_slopes), _l(query.stride(0)),
        _l(key_cache.stride(0)), _l(key_cache.stride(1)), _i(_DT[query.dtype]))


def paged_attention_v2(out, exp_sums, max_logits, tmp_out, query, key_cache, value_cache,
                       num_kv_heads, scale, block_tables, seq_lens, block_size, max_seq_len,
                       alibi_slopes=None):
    assert out.is_contiguous() and tmp_out.is_contiguous()
    lib().oracle_paged_attention_v2(
        _p(out), _p(exp_sums), _p(max_logits), _p(tmp_out), _p(query), _p(key_cache),
        _p(value_cache), _i(query.size(0)), _i(query.size(1)), _i(query.size(2)),
        _i(num_kv_heads), _f(scale), _p(block_tables), _p(seq_lens), _i(block_size),
        _i(block_tables.size(1)), _i(exp_sums.size(-1)), _p(alibi_slopes), _l(query.stride(0)),
        _l(key_cache.stride(0)), _l(key_cache.stride(1)), _i(_DT[query.dtype]))


def paged_prefill_attention(out, query, key_cache, value_cache, num_kv_heads, scale, block_tables,
                            seq_lens, query_start_loc, block_size, alibi_slopes=None,
                            sliding_window=0, softcap=0.0):
    """query/out [T, H, D]; see oracle_paged_prefill_attention (flash_attn.py:538-555)."""
    assert block_tables.dtype == torch.int32 and seq_lens.dtype == torch.int32
    assert query_start_loc.dtype == torch.int32 and out.stride(1) == out.size(2)
    lib().oracle_paged_prefill_attention(
        _p(out), _p(query), _p(key_cache), _p(value_cache), _i(seq_lens.numel()), _i(query.size(1)),
        _i(query.size(2)), _i(num_kv_heads), _f(scale), _p(block_tables), _p(seq_lens),
        _p(query_start_loc), _i(block_size), _i(block_tables.size(1)), _p(alibi_slopes),
        _i(sliding_window), _f(softcap), _l(query.stride(0)), _l(out.stride(0)),
        _l(key_cache.stride(0)), _l(key_cache.stride(1)), _i(_DT[query.dtype]))


def varlen_attention(out, query, key, value, cu_seqlens, scale, causal):
    """query/out [T, H, D], key/value [T, KVH, D], cu_seqlens int32 [num_seqs + 1]
    (torch_naive.py:65-149 of the reference's prefill_only backends)."""
    assert cu_seqlens.dtype == torch.int32 and out.stride(1) == out.size(2)
    lib().oracle_varlen_attention(
        _p(out), _p(query), _p(key), _p(value), _p(cu_seqlens), _i(cu_seqlens.numel() - 1),
        _i(query.size(1)), _i(key.size(1)), _i(query.size(2)), _f(scale), _i(1 if causal else 0),
        _l(query.stride(0)), _l(key.stride(0)), _l(value.stride(0)), _l(out.stride(0)),
        _i(_DT[query.dtype]))


def set_blocksparse(vert_stride=0, local_blocks=0, block_size=64, head_sliding_step=0, tp_rank=0):
    """Block-sparse arguments of the attention oracles (off when vert_stride <= 1)."""
    lib().oracle_set_blocksparse(_i(vert_stride), _i(local_blocks), _i(block_size), _i(head_sliding_step),
                                 _i(tp_rank))


def set_kv_cache_fp8(on, k_scale=1.0, v_scale=1.0):
    """kv_cache_dtype of the attention oracles: on = the caches hold OCP e4m3fn bytes."""
    lib().oracle_set_kv_cache_fp8(_i(1 if on else 0), _f(k_scale), _f(v_scale))


def reshape_and_cache_fp8(key, value, key_cache, value_cache, slot_mapping, k_scale=1.0, v_scale=1.0):
    """key_cache uint8 [NB, KVH, D/16, BS, 16], value_cache uint8 [NB, KVH, D, BS]."""
    assert key_cache.dtype == torch.uint8 and value_cache.dtype == torch.uint8 and key_cache.size(4) == 16
    lib().oracle_reshape_and_cache_fp8(
        _p(key), _p(value), _p(key_cache), _p(value_cache), _p(slot_mapping), _i(key.size(0)),
        _i(key.size(1)), _i(key.size(2)), _i(key_cache.size(3)), _l(key.stride(0)), _l(value.stride(0)),
        _i(_DT[key.dtype]), _f(k_scale), _f(v_scale))


def f32_to_e4m3(x: float) -> int:
    return int(lib().oracle_f32_to_e4m3(_f(x)))


def e4m3_to_f32(v: int) -> float:
    return float(lib().oracle_e4m3_to_f32(ctypes.c_uint8(v)))


def advance_step(num_queries, block_size, input_tokens, sampled_token_ids, input_positions, seq_lens,
                 slot_mapping, block_tables):
    assert input_tokens.dtype == torch.int64 and seq_lens.dtype == torch.int32 and block_tables.dtype == torch.int32
    lib().oracle_advance_step(_i(num_queries), _i(block_size), _p(input_tokens), _p(sampled_token_ids),
                              _p(input_positions), _p(seq_lens), _p(slot_mapping), _p(block_tables),
                              _l(block_tables.stride(0)))


def static_scaled_fp8_quant(out, input, scale):
    assert out.dtype == torch.uint8 and scale.dtype == torch.float32
    lib().oracle_static_scaled_fp8_quant(_p(out), _p(input), _p(scale), _l(input.numel()), _i(_DT[input.dtype]))


def dynamic_scaled_fp8_quant(out, input, scale):
    assert out.dtype == torch.uint8 and scale.dtype == torch.float32
    lib().oracle_dynamic_scaled_fp8_quant(_p(out), _p(input), _p(scale), _l(input.numel()), _i(_DT[input.dtype]))


def dynamic_per_token_scaled_fp8_quant(out, scales, input, scale_ub=None):
    assert out.dtype == torch.uint8 and scales.dtype == torch.float32
    lib().oracle_dynamic_per_token_scaled_fp8_quant(_p(out), _p(scales), _p(input), _p(scale_ub),
                                                    _i(input.numel() // input.size(-1)), _i(input.size(-1)),
                                                    _i(_DT[input.dtype]))


def reshape_and_cache(key, value, key_cache, value_cache, slot_mapping):
    assert slot_mapping.dtype == torch.int64
    lib().oracle_reshape_and_cache(
        _p(key), _p(value), _p(key_cache), _p(value_cache), _p(slot_mapping), _i(key.size(0)),
        _i(key.size(1)), _i(key.size(2)), _i(key_cache.size(3)), _i(key_cache.size(4)),
        _l(key.stride(0)), _l(value.stride(0)), _i(_DT[key.dtype]))


def reshape_and_cache_flash(key, value, key_cache, value_cache, slot_mapping):
    lib().oracle_reshape_and_cache_flash(
        _p(key), _p(value), _p(key_cache), _p(value_cache), _p(slot_mapping), _i(key.size(0)),
        _i(key.size(1)), _i(key.size(2)), _i(key_cache.size(1)), _l(key_cache.stride(0)),
        _l(key.stride(0)), _l(value.stride(0)), _i(_DT[key.dtype]))


def copy_blocks(key_caches, value_caches, block_mapping):
    bm = block_mapping.to(torch.int64).contiguous()
    for kc, vc in zip(key_caches, value_caches):
        lib().oracle_copy_blocks(_p(kc), _p(vc), _p(bm), _i(bm.size(0)),
                                 _l(kc[0].numel() * kc.element_size()))


def swap_blocks(src, dst, block_mapping):
    bm = block_mapping.to(torch.int64).contiguous()
    lib().oracle_swap_blocks(_p(src), _p(dst), _p(bm), _i(bm.size(0)),
                             _l(src[0].numel() * src.element_size()))


def rms_norm(out, input, weight, epsilon):
    hidden = input.size(-1)
    lib().oracle_rms_norm(_p(out), _p(input), _p(weight), _f(epsilon),
                          _i(input.numel() // hidden), _i(hidden), _i(_DT[input.dtype]))


def fused_add_rms_norm(input, residual, weight, epsilon):
    hidden = input.size(-1)
    lib().oracle_fused_add_rms_norm(_p(input), _p(residual), _p(weight), _f(epsilon),
                                    _i(input.numel() // hidden), _i(hidden), _i(_DT[input.dtype]))


def rotary_embedding(positions, query, key, head_size, cos_sin_cache, is_neox):
    num_tokens = query.numel() // query.size(-1)
    lib().oracle_rotary_embedding(
        _p(positions), _p(query), _p(key), _i(num_tokens), _i(query.size(-1) // head_size),
        _i(key.size(-1) // head_size), _i(head_size), _i(cos_sin_cache.size(1)),
        _l(query.stride(-2)), _l(key.stride(-2)), _p(cos_sin_cache), _i(1 if is_neox else 0),
        _i(_DT[query.dtype]))


def silu_and_mul(out, input):
    d = input.size(-1) // 2
    lib().oracle_silu_and_mul(_p(out), _p(input), _l(input.numel() // input.size(-1)), _i(d),
                              _i(_DT[input.dtype]))
