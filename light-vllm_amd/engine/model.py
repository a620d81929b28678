"""Layer sequence of the decode-only model (the caller of the hot-path operators).

Follows light_vllm/decode_only/modelzoo/qwen2.py:144-292: per layer
  fused_add_rms_norm -> qkv GEMM -> rotary_embedding -> attention (reshape_and_cache +
  paged attention) -> o_proj GEMM -> fused_add_rms_norm -> gate_up GEMM -> silu_and_mul ->
  down GEMM; then the final norm and the lm_head GEMM.
Dense projections are plain torch GEMMs (hipBLASLt / MFMA); everything else is a gfx950
kernel of this package.  Weights are random-initialised: no checkpoint is available offline.
"""
from typing import List, Optional

import torch
import torch.nn.functional as F

from .. import _custom_ops as ops
from .config import ModelConfig


def build_cos_sin_cache(head_dim: int, max_pos: int, base: float, dtype, device) -> torch.Tensor:
    """[max_pos, rot_dim] = [cos | sin] in the model dtype
    (light_vllm/backends/rotary_embedding.py:94-114)."""
    inv_freq = 1.0 / (base ** (torch.arange(0, head_dim, 2, dtype=torch.float) / head_dim))
    t = torch.arange(max_pos, dtype=torch.float)
    freqs = torch.einsum("i,j -> ij", t, inv_freq)
    return torch.cat((freqs.cos(), freqs.sin()), dim=-1).to(dtype).to(device)


class DecoderLayerWeights:
    def __init__(self, cfg: ModelConfig, device, gen: torch.Generator):
        H, KVH, D, hid, inter = (cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim,
                                 cfg.hidden_size, cfg.intermediate_size)

        def w(*shape, std=0.02):
            return (torch.randn(*shape, generator=gen, device=device, dtype=torch.float32) * std).to(cfg.dtype)

        self.input_norm = (1.0 + 0.05 * torch.randn(hid, generator=gen, device=device)).to(cfg.dtype)
        self.post_norm = (1.0 + 0.05 * torch.randn(hid, generator=gen, device=device)).to(cfg.dtype)
        self.qkv = w((H + 2 * KVH) * D, hid)
        self.qkv_bias = w((H + 2 * KVH) * D) if cfg.qkv_bias else None
        self.o = w(hid, H * D)
        self.gate_up = w(2 * inter, hid)
        self.down = w(hid, inter)


class DecoderModel:
    """forward(input_ids, positions, kv_caches, attn_metadata) -> hidden states [T, hidden]."""

    def __init__(self, cfg: ModelConfig, attn_impl, device="cuda:0", seed: int = 0):
        self.cfg = cfg
        self.device = torch.device(device)
        gen = torch.Generator(device=self.device).manual_seed(seed)
        self.embed = (torch.randn(cfg.vocab_size, cfg.hidden_size, generator=gen, device=self.device) * 0.02).to(cfg.dtype)
        self.layers: List[DecoderLayerWeights] = [DecoderLayerWeights(cfg, self.device, gen)
                                                  for _ in range(cfg.num_hidden_layers)]
        self.final_norm = (1.0 + 0.05 * torch.randn(cfg.hidden_size, generator=gen, device=self.device)).to(cfg.dtype)
        self.lm_head = (torch.randn(cfg.vocab_size, cfg.hidden_size, generator=gen, device=self.device) * 0.02).to(cfg.dtype)
        self.cos_sin_cache = build_cos_sin_cache(cfg.head_dim, cfg.max_position_embeddings,
                                                 cfg.rope_theta, cfg.dtype, self.device)
        self.attn = attn_impl  # DecodeOnlyAttentionImpl-like: forward(q, k, v, kv_cache, metadata)
        self.q_size = cfg.num_attention_heads * cfg.head_dim
        self.kv_size = cfg.num_key_value_heads * cfg.head_dim

    def weight_bytes(self) -> int:
        n = self.lm_head.numel()
        for l in self.layers:
            n += l.qkv.numel() + l.o.numel() + l.gate_up.numel() + l.down.numel()
        return n * self.lm_head.element_size()

    def forward(self, input_ids: torch.Tensor, positions: torch.Tensor,
                kv_caches: Optional[List[torch.Tensor]], attn_metadata) -> torch.Tensor:
        cfg = self.cfg
        hidden = F.embedding(input_ids, self.embed)
        residual = None
        for i, lw in enumerate(self.layers):
            if residual is None:  # qwen2.py:203-208
                residual = hidden
                normed = torch.empty_like(hidden)
                ops.rms_norm(normed, hidden, lw.input_norm, cfg.rms_norm_eps)
                hidden = normed
            else:
                ops.fused_add_rms_norm(hidden, residual, lw.input_norm, cfg.rms_norm_eps)
            qkv = F.linear(hidden, lw.qkv, lw.qkv_bias)
            q, k, v = qkv.split([self.q_size, self.kv_size, self.kv_size], dim=-1)  # strided views
            ops.rotary_embedding(positions, q, k, cfg.head_dim, self.cos_sin_cache, True)
            attn_out = self.attn.forward(q, k, v, kv_caches[i] if kv_caches is not None else None,
                                         attn_metadata)
            hidden = F.linear(attn_out, lw.o)
            ops.fused_add_rms_norm(hidden, residual, lw.post_norm, cfg.rms_norm_eps)
            gate_up = F.linear(hidden, lw.gate_up)
            act = torch.empty(gate_up.shape[0], cfg.intermediate_size, dtype=gate_up.dtype, device=gate_up.device)
            ops.silu_and_mul(act, gate_up)
            hidden = F.linear(act, lw.down)
        ops.fused_add_rms_norm(hidden, residual, self.final_norm, cfg.rms_norm_eps)
        return hidden

    def compute_logits(self, hidden: torch.Tensor) -> torch.Tensor:
        return F.linear(hidden, self.lm_head)
