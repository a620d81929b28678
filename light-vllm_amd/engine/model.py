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


class Weight:
    """A projection weight [N, K] kept in two orders: row-major for the library GEMM (prompt
    batches) and, for decode batches, packed into MFMA-fragment order so that the
    weight-streaming kernel reads one contiguous KiB per wave load (csrc/skinny_gemm.hip).
    Twice the bytes; sized for 288 GB of HBM."""

    __slots__ = ("w", "packed", "N", "K")

    def __init__(self, w: torch.Tensor, pack: bool = True):
        self.w = w
        self.N, self.K = w.shape
        ok = pack and w.is_cuda and self.N % 16 == 0 and self.K % 32 == 0 and self.N * self.K * 2 < (1 << 32) - 16
        self.packed = torch.ops._C_amd.pack_weight(w) if ok else None

    def numel(self) -> int:
        return self.w.numel()


def linear(x: torch.Tensor, w: "Weight", bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Dense projection.  Decode batches (<= 64 rows) stream the weights through the gfx950
    weight-streaming kernel; larger batches use the library GEMM (hipBLASLt)."""
    if x.shape[0] <= 64 and x.is_cuda:
        if w.packed is not None:
            return torch.ops._C_amd.skinny_linear_packed(x, w.packed, bias, w.N, w.K)
    return F.linear(x, w.w, bias)


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
        self.qkv = Weight(w((H + 2 * KVH) * D, hid), cfg.pack_weights)
        self.qkv_bias = w((H + 2 * KVH) * D) if cfg.qkv_bias else None
        self.o = Weight(w(hid, H * D), cfg.pack_weights)
        self.gate_up = Weight(w(2 * inter, hid), cfg.pack_weights)
        self.down = Weight(w(hid, inter), cfg.pack_weights)


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
        self.lm_head = Weight((torch.randn(cfg.vocab_size, cfg.hidden_size, generator=gen, device=self.device) * 0.02).to(cfg.dtype),
                              cfg.pack_weights)
        self.cos_sin_cache = build_cos_sin_cache(cfg.head_dim, cfg.max_position_embeddings,
                                                 cfg.rope_theta, cfg.dtype, self.device)
        self.attn = attn_impl  # DecodeOnlyAttentionImpl-like: forward(q, k, v, kv_cache, metadata)
        self.q_size = cfg.num_attention_heads * cfg.head_dim
        self.kv_size = cfg.num_key_value_heads * cfg.head_dim

    def weight_bytes(self) -> int:
        n = self.lm_head.numel()
        for l in self.layers:
            n += l.qkv.numel() + l.o.numel() + l.gate_up.numel() + l.down.numel()
        return n * self.lm_head.w.element_size()

    def _add_norm(self, x, residual: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
        """residual += x; return norm(residual) * weight.  `x` is either the projection output
        [T, hidden] or the fp32 split-K partials [S, T, hidden] of a fused down projection."""
        if x.dim() == 3:
            out = torch.empty_like(residual)
            torch.ops._C_amd.fused_add_rms_norm_splitk(out, residual, x, weight, self.cfg.rms_norm_eps)
            return out
        ops.fused_add_rms_norm(x, residual, weight, self.cfg.rms_norm_eps)
        return x

    def forward(self, input_ids: torch.Tensor, positions: torch.Tensor,
                kv_caches: Optional[List[torch.Tensor]], attn_metadata) -> torch.Tensor:
        cfg = self.cfg
        hidden = F.embedding(input_ids, self.embed)
        T = hidden.shape[0]
        # decode-only steps take the fused launches: rope + cache write in one kernel, split-K
        # partials of the down projection summed inside the next add+norm
        decode_only = (cfg.fuse_decode_ops and kv_caches is not None and T <= 64 and
                       attn_metadata.num_prefill_tokens == 0 and hasattr(self.attn, "decode_attention"))
        residual = None
        for i, lw in enumerate(self.layers):
            if residual is None:  # qwen2.py:203-208
                residual = hidden
                normed = torch.empty_like(hidden)
                ops.rms_norm(normed, hidden, lw.input_norm, cfg.rms_norm_eps)
                hidden = normed
            else:
                hidden = self._add_norm(hidden, residual, lw.input_norm)
            qkv = linear(hidden, lw.qkv, lw.qkv_bias)
            q, k, v = qkv.split([self.q_size, self.kv_size, self.kv_size], dim=-1)  # strided views
            fused = False
            if decode_only:
                key_cache, value_cache = self.attn.split_kv_cache(kv_caches[i])
            if decode_only and key_cache.dtype == q.dtype:  # (an fp8 cache takes the two separate ops)
                fused = torch.ops._C_amd.rotary_embedding_and_cache(
                    positions, q, k, v, cfg.head_dim, self.cos_sin_cache, True, key_cache, value_cache,
                    attn_metadata.slot_mapping)
            if fused:
                attn_out = self.attn.decode_attention(q, key_cache, value_cache, attn_metadata)
            else:
                ops.rotary_embedding(positions, q, k, cfg.head_dim, self.cos_sin_cache, True)
                attn_out = self.attn.forward(q, k, v, kv_caches[i] if kv_caches is not None else None,
                                             attn_metadata)
            hidden = linear(attn_out, lw.o)
            hidden = self._add_norm(hidden, residual, lw.post_norm)
            gate_up = linear(hidden, lw.gate_up)
            act = torch.empty(T, cfg.intermediate_size, dtype=gate_up.dtype, device=gate_up.device)
            ops.silu_and_mul(act, gate_up)
            if decode_only and lw.down.packed is not None:
                # [S, T, hidden] fp32 split-K partial sums; the next add+norm adds them up
                hidden = torch.ops._C_amd.skinny_linear_packed_partials(act, lw.down.packed, lw.down.N,
                                                                        lw.down.K, False)
            else:
                hidden = linear(act, lw.down)
        return self._add_norm(hidden, residual, self.final_norm)

    def compute_logits(self, hidden: torch.Tensor) -> torch.Tensor:
        return linear(hidden, self.lm_head)
