"""MI355X-native paged-attention decode path for light-vllm.

Layout (only what the hot path needs):
  csrc/            hand-written gfx950 HIP kernels + the C-ABI (include/lvllm_hip.h)
                   + torch op registrations with the reference's schemas
  _native.py       loads lib/liblvllm_hip.so and lib/_C.so (fails loudly if absent)
  _custom_ops.py   mirror of light_vllm/backends/_custom_ops.py for the path's operators
  paged_attn.py    mirror of light_vllm/decoding/backends/attention/ops/paged_attn.py
  block_manager/   KV-cache block managers v1/v2 (block tables, CoW, swap, prefix cache)
  attention/       paged DecodeOnlyAttentionBackend (plugin boundary B)
  engine/          decode loop: scheduler glue, input builder, cache engine, executor
"""
__version__ = "0.1.0"
