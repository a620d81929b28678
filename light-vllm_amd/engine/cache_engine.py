"""KV-cache storage of one engine replica: per-layer paged cache tensors on the GPU and in
pinned host memory, plus the swap / copy dispatch the scheduler's block lists drive.

Behaviour of light_vllm/decoding/worker/cache_engine.py:15-103: L zero-initialised tensors of
`attn_backend.get_kv_cache_shape(...)`; `block bytes = 2 * L * block_size * kv_heads * head_size
* sizeof(dtype)`; swap_in/swap_out go layer by layer through the backend, copy is one fused call.
"""
from typing import List

import torch

from .config import CacheConfig, ModelConfig


class CacheEngine:

    def __init__(self, cache_config: CacheConfig, model_config: ModelConfig, attn_backend,
                 device="cuda:0") -> None:
        self.cache_config = cache_config
        self.model_config = model_config
        self.attn_backend = attn_backend
        self.device = torch.device(device)
        self.head_size = model_config.head_dim
        self.num_attention_layers = model_config.num_hidden_layers
        self.num_kv_heads = model_config.num_key_value_heads
        self.block_size = cache_config.block_size
        self.num_gpu_blocks = cache_config.num_gpu_blocks
        self.num_cpu_blocks = cache_config.num_cpu_blocks or 0
        # cache_engine.py:33-36 of the reference: "auto" = the model's dtype, fp8 = one byte per
        # element (OCP e4m3fn on gfx950; the byte tensor is what the kernels take)
        self.dtype = self.kv_cache_torch_dtype(cache_config.cache_dtype, model_config.dtype)
        self.gpu_cache = self._allocate_kv_cache(self.num_gpu_blocks, self.device)
        self.cpu_cache = self._allocate_kv_cache(self.num_cpu_blocks, torch.device("cpu"))

    def _allocate_kv_cache(self, num_blocks: int, device: torch.device) -> List[torch.Tensor]:
        shape = self.attn_backend.get_kv_cache_shape(num_blocks, self.block_size, self.num_kv_heads,
                                                     self.head_size)
        pin = device.type == "cpu" and torch.cuda.is_available() and num_blocks > 0
        return [torch.zeros(shape, dtype=self.dtype, device=device, pin_memory=pin)
                for _ in range(self.num_attention_layers)]

    def swap_in(self, src_to_dst: torch.Tensor) -> None:
        for i in range(self.num_attention_layers):
            self.attn_backend.swap_blocks(self.cpu_cache[i], self.gpu_cache[i], src_to_dst)

    def swap_out(self, src_to_dst: torch.Tensor) -> None:
        for i in range(self.num_attention_layers):
            self.attn_backend.swap_blocks(self.gpu_cache[i], self.cpu_cache[i], src_to_dst)

    def copy(self, src_to_dsts: torch.Tensor) -> None:
        self.attn_backend.copy_blocks(self.gpu_cache, src_to_dsts)

    @staticmethod
    def kv_cache_torch_dtype(cache_dtype: str, model_dtype: torch.dtype) -> torch.dtype:
        if cache_dtype == "auto":
            return model_dtype
        if cache_dtype in ("fp8", "fp8_e4m3"):
            return torch.uint8
        raise ValueError(f"Unsupported data type of kv cache: {cache_dtype}")

    @staticmethod
    def get_cache_block_size(cache_config: CacheConfig, model_config: ModelConfig) -> int:
        per_layer = cache_config.block_size * model_config.num_key_value_heads * model_config.head_dim
        total = model_config.num_hidden_layers * 2 * per_layer
        dtype = CacheEngine.kv_cache_torch_dtype(cache_config.cache_dtype, model_config.dtype)
        return total * torch.tensor([], dtype=dtype).element_size()
