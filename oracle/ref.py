"""Loader of oracle/_ref/_ref_C.so: the reference's own csrc/cpu backend, built unmodified
by oracle/build_oracle.py.  TEST INFRASTRUCTURE ONLY.

After load(), the reference's CPU operators are torch.ops._ref_C.* (paged_attention_v1/v2,
rms_norm, fused_add_rms_norm, rotary_embedding, silu_and_mul) and
torch.ops._ref_C_cache_ops.* (reshape_and_cache, copy_blocks).  Limits of that backend
(csrc/cpu): fp32 and bf16 only, block_size 16 only, no swap_blocks, no fp8.
"""
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = os.path.join(_HERE, "_ref", "_ref_C.so")
_loaded = False


def _cpu_has_avx512() -> bool:
    try:
        with open("/proc/cpuinfo") as f:
            flags = f.read()
    except OSError:
        return False
    return all(f" {x}" in flags for x in ("avx512f", "avx512bw", "avx512dq", "avx512vl"))


def available() -> bool:
    return os.path.exists(REF_SO) and _cpu_has_avx512()


def load() -> bool:
    """Returns True when torch.ops._ref_C is usable."""
    global _loaded
    if _loaded:
        return True
    if not available():
        return False
    torch.ops.load_library(REF_SO)
    _loaded = True
    return True
