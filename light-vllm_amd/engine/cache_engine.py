"""KV-cache storage of one engine replica: one paged tensor per layer on the GPU, a pinned twin per
layer on the host for swapped-out blocks, and the swap / copy dispatch that the scheduler's block
lists drive.

Behaviour of light_vllm/decoding/worker/cache_engine.py:15-103: zero-initialised tensors of
`attn_backend.get_kv_cache_shape(...)`, "auto" = the model's dtype and fp8 = one byte per element
(:33-36), `block bytes = 2 * L * block_size * kv_heads * head_size * sizeof(element)`, swaps layer
by layer through the backend, copy-on-write copies in one fused call.
"""
from typing import List

import torch

from .config import CacheConfig, ModelConfig

_FP8_NAMES = ("fp8", "fp8_e4m3")


class CacheEngine:

    def __init__(self, cache_config: CacheConfig, model_config: ModelConfig, attn_backend,
                 device="cuda:0") -> None:
        self.cache_config, self.model_config, self.attn_backend = cache_config, model_config, attn_backend
        self.device = torch.device(device)
        self.num_attention_layers = model_config.num_hidden_layers
        self.num_kv_heads, self.head_size = model_config.num_key_value_heads, model_config.head_dim
        self.block_size = cache_config.block_size
        self.num_gpu_blocks, self.num_cpu_blocks = cache_config.num_gpu_blocks, cache_config.num_cpu_blocks or 0
        self.dtype = self.kv_cache_torch_dtype(cache_config.cache_dtype, model_config.dtype)
        self.gpu_cache = self._allocate(self.num_gpu_blocks, self.device)
        self.cpu_cache = self._allocate(self.num_cpu_blocks, torch.device("cpu"))

    def _allocate(self, num_blocks: int, where: torch.device) -> List[torch.Tensor]:
        """One [2, num_blocks, block elements] tensor per layer (the reference's shape, cache_engine.py:69-86), each a
        view of a buffer whose rows are `block_pad_bytes` longer: the ops address blocks by the tensors' strides, the
        CPU side pads alike so that runs of blocks swap as single copies."""
        shape = self.attn_backend.get_kv_cache_shape(num_blocks, self.block_size, self.num_kv_heads, self.head_size)
        on_host = where.type == "cpu"
        pinned = on_host and num_blocks > 0 and torch.cuda.is_available()  # swaps are async DMA
        pad = self.block_pad_bytes(self.cache_config, self.model_config) // torch.empty((), dtype=self.dtype).element_size()
        out = []
        for _ in range(self.num_attention_layers):
            if pad == 0:
                out.append(torch.zeros(shape, dtype=self.dtype, device=where, pin_memory=pinned))
            else:
                buf = torch.zeros(shape[0], shape[1], shape[2] + pad, dtype=self.dtype, device=where, pin_memory=pinned)
                out.append(buf[:, :, :shape[2]])
        return out

    @staticmethod
    def block_pad_bytes(cache_config: CacheConfig, model_config: ModelConfig) -> int:
        """Bytes between a block and the next in a plane (CacheConfig.block_pad_bytes; None: 1/32 of the block)."""
        pad = getattr(cache_config, "block_pad_bytes", 0)
        if pad is None:
            dtype = CacheEngine.kv_cache_torch_dtype(cache_config.cache_dtype, model_config.dtype)
            plane = (cache_config.block_size * model_config.num_key_value_heads * model_config.head_dim *
                     torch.empty((), dtype=dtype).element_size())
            pad = ((plane // 32 + 255) // 256) * 256
        if pad % 16 != 0 or pad < 0:
            raise ValueError("block_pad_bytes must be a non-negative multiple of 16")
        return pad

    # ---- block movement: [n, 2] (source block, destination block) pairs ----
    def _swap(self, src: List[torch.Tensor], dst: List[torch.Tensor], pairs: torch.Tensor) -> None:
        for layer_src, layer_dst in zip(src, dst):
            self.attn_backend.swap_blocks(layer_src, layer_dst, pairs)

    def swap_in(self, src_to_dst: torch.Tensor) -> None:
        self._swap(self.cpu_cache, self.gpu_cache, src_to_dst)

    def swap_out(self, src_to_dst: torch.Tensor) -> None:
        self._swap(self.gpu_cache, self.cpu_cache, src_to_dst)

    def copy(self, src_to_dsts: torch.Tensor) -> None:
        self.attn_backend.copy_blocks(self.gpu_cache, src_to_dsts)

    # ---- sizing ----
    @staticmethod
    def kv_cache_torch_dtype(cache_dtype: str, model_dtype: torch.dtype) -> torch.dtype:
        if cache_dtype == "auto":
            return model_dtype
        if cache_dtype in _FP8_NAMES:  # OCP e4m3fn bytes; the kernels take the byte tensor
            return torch.uint8
        raise ValueError(f"Unsupported data type of kv cache: {cache_dtype}")

    @staticmethod
    def get_cache_block_size(cache_config: CacheConfig, model_config: ModelConfig) -> int:
        """Bytes one block takes across all layers (keys and values)."""
        elements = (2 * model_config.num_hidden_layers * cache_config.block_size *
                    model_config.num_key_value_heads * model_config.head_dim)
        dtype = CacheEngine.kv_cache_torch_dtype(cache_config.cache_dtype, model_config.dtype)
        return elements * torch.empty((), dtype=dtype).element_size()

    @staticmethod
    def get_cache_block_footprint(cache_config: CacheConfig, model_config: ModelConfig) -> int:
        """Bytes of memory a block costs: the reference's figure above + the padding behind the block in both planes of
        every layer (CacheConfig.block_pad_bytes).  What the worker divides the free memory by."""
        pad = 2 * model_config.num_hidden_layers * CacheEngine.block_pad_bytes(cache_config, model_config)
        return CacheEngine.get_cache_block_size(cache_config, model_config) + pad
