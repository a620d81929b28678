"""The C-ABI library loads and exports every symbol include/lvllm_hip.h declares, the torch
ops are registered with the reference's schemas, and argument checks fail loudly.  No compute
call is made (no GPU needed)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    with open(os.path.join(ROOT, "include", "lvllm_hip.h")) as f:
        text = f.read()
    return sorted(set(re.findall(r"\b(lvllm_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from light_vllm_amd import _native
    lib = _native.load_hip_library()
    syms = declared_symbols()
    assert len(syms) >= 14
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/lvllm_hip.h but not exported"
    assert lib.lvllm_version().startswith(b"lvllm_hip gfx950")


def test_torch_ops_registered_with_reference_schemas(ops):
    expected = {
        "_C": ["paged_attention_v1", "paged_attention_v2", "silu_and_mul", "rms_norm", "fused_add_rms_norm",
               "rotary_embedding"],
        "_C_cache_ops": ["swap_blocks", "copy_blocks", "reshape_and_cache", "reshape_and_cache_flash"],
        "_C_cuda_utils": ["get_device_attribute", "get_max_shared_memory_per_block_device_attribute"],
    }
    for ns, names in expected.items():
        for n in names:
            assert ops.is_custom_op_supported(f"{ns}::{n}"), f"{ns}::{n}"
    schema = str(torch.ops._C.paged_attention_v2.default._schema)
    for frag in ("! -> ) out", "Tensor exp_sums", "Tensor tmp_out", "int num_kv_heads", "float scale",
                 "Tensor? alibi_slopes", "str kv_cache_dtype", "int blocksparse_head_sliding_step"):
        assert frag in schema, (frag, schema)
    assert "! -> ) key_cache" in str(torch.ops._C_cache_ops.reshape_and_cache.default._schema)


def test_argument_errors_are_reported_through_the_abi():
    from light_vllm_amd import _native
    lib = _native.load_hip_library()
    lib.lvllm_rotary_embedding.restype = ctypes.c_int
    # rot_dim larger than head_size is rejected before any launch
    rc = lib.lvllm_rotary_embedding(None, None, None, ctypes.c_int(1), ctypes.c_int(1), ctypes.c_int(1),
                                    ctypes.c_int(64), ctypes.c_int(128), ctypes.c_int64(64), ctypes.c_int64(64),
                                    None, ctypes.c_int(1), ctypes.c_int(2), None)
    assert rc != 0
    assert b"rot_dim" in lib.lvllm_last_error()
    lib.lvllm_swap_blocks.restype = ctypes.c_int
    rc = lib.lvllm_swap_blocks(None, None, None, ctypes.c_int(0), ctypes.c_int64(16), ctypes.c_int(0),
                               ctypes.c_int(0), None)
    assert rc != 0 and b"Invalid device combination" in lib.lvllm_last_error()


def test_cpu_tensors_are_rejected_not_silently_computed(ops):
    """The product path has no CPU fallback: a CPU tensor is an error."""
    x = torch.randn(2, 64)
    with pytest.raises((RuntimeError, NotImplementedError)):
        ops.rms_norm(torch.empty_like(x), x, torch.ones(64), 1e-6)


def test_tuning_knobs_round_trip_through_the_abi():
    """lvllm_set_tuning / lvllm_get_tuning: every documented key reads back what was set, the shipped defaults are
    the documented ones, unknown keys and out-of-range values are errors (no GPU needed: host state only)."""
    from light_vllm_amd import _native
    lib = _native.load_hip_library()
    lib.lvllm_set_tuning.restype = ctypes.c_int
    lib.lvllm_get_tuning.restype = ctypes.c_int

    def get(key):
        v = ctypes.c_int(-12345)
        assert lib.lvllm_get_tuning(key, ctypes.byref(v)) == 0, lib.lvllm_last_error()
        return v.value

    defaults = {b"gemm_workgroups": 256, b"attn_waves": 8, b"attn_splits": 0, b"swap_kernel_min_runs": 3,
                b"cache_tile_min_tokens": 384, b"prefill_lds": 1, b"prefill_mfma32_min_query": 64,
                b"gemm_partials_ksplit": 0, b"gemm_balance": 1, b"varlen_dense": 1, b"varlen_dense_waves": 0}
    for key, want in defaults.items():
        if os.environ.get("LVLLM_" + key.decode().upper()) is None:
            assert get(key) == want, key
    for key, val in ((b"prefill_mfma32_min_query", 512), (b"cache_tile_min_tokens", 64), (b"attn_splits", -1),
                     (b"gemm_workgroups", 128)):
        old = get(key)
        assert lib.lvllm_set_tuning(key, ctypes.c_int(val)) == 0
        assert get(key) == val
        assert lib.lvllm_set_tuning(key, ctypes.c_int(old)) == 0
    assert lib.lvllm_set_tuning(b"no_such_knob", ctypes.c_int(1)) != 0 and b"unknown tuning key" in lib.lvllm_last_error()
    assert lib.lvllm_get_tuning(b"no_such_knob", ctypes.byref(ctypes.c_int())) != 0
    assert lib.lvllm_set_tuning(b"attn_waves", ctypes.c_int(5)) != 0
    assert lib.lvllm_set_tuning(b"prefill_mfma32_min_query", ctypes.c_int(-1)) != 0
    assert lib.lvllm_get_tuning(b"attn_waves", None) != 0
