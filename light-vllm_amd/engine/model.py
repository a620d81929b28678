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

    __slots__ = ("w", "packed", "N", "K", "w8_t", "w8_packed", "w_scale", "x_scale", "x_absmax")

    def __init__(self, w: torch.Tensor, pack: bool = True):
        self.w = w
        self.N, self.K = w.shape
        ok = pack and w.is_cuda and self.N % 16 == 0 and self.K % 32 == 0 and self.N * self.K * 2 < (1 << 32) - 16
        self.packed = torch.ops._C_amd.pack_weight(w) if ok else None
        self.w8_t = self.w8_packed = self.w_scale = self.x_scale = None
        self.x_absmax = None  # calibration: running max |x| of this projection's input

    def numel(self) -> int:
        return self.N * self.K

    def quantize_fp8(self) -> None:
        """Per-tensor e4m3 weights + the calibrated static activation scale; the 16-bit copies are
        dropped (fp8.py:196-239 of the reference quantises at load time the same way)."""
        from ..quantization import pack_fp8_weight
        q, self.w_scale = ops.scaled_fp8_quant(self.w)          # [N, K] fp8, scale [1]
        self.w8_t = q.t()                                        # torch._scaled_mm wants [K, N] column-major
        ok = self.N % 16 == 0 and self.K % 64 == 0 and self.N * self.K < (1 << 32) - 16
        self.w8_packed = pack_fp8_weight(q) if ok else None
        absmax = self.x_absmax if self.x_absmax is not None else torch.ones((), device=self.w.device)
        self.x_scale = (absmax.float() / 448.0).clamp_min(1e-6).reshape(1)
        self.w = self.packed = None


def linear(x: torch.Tensor, w: "Weight", bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Dense projection.  Decode batches (<= 64 rows) stream the weights through the gfx950
    weight-streaming kernel; larger batches use the library GEMM (hipBLASLt)."""
    if x.shape[0] == 0:  # a step of unfinished prompt chunks samples nothing
        return x.new_empty((0, w.N))
    if w.w8_t is not None:  # W8A8
        if x.shape[0] <= 64 and w.w8_packed is not None:
            return torch.ops._C_amd.skinny_linear_w8a8(x, w.w8_packed, w.w_scale, w.x_scale, w.N, w.K, bias)
        from ..quantization import apply_fp8_linear
        return apply_fp8_linear(x, w.w8_t, w.w_scale, input_scale=w.x_scale, bias=bias)
    if _CALIBRATING:
        m = x.detach().abs().amax()
        w.x_absmax = m if w.x_absmax is None else torch.maximum(w.x_absmax, m)
    if x.shape[0] <= 64 and x.is_cuda:
        if w.packed is not None:
            return torch.ops._C_amd.skinny_linear_packed(x, w.packed, bias, w.N, w.K)
    elif (x.shape[0] <= _STREAM_GEMM_MAX_ROWS and x.is_cuda and w.packed is not None and x.stride(1) == 1
          and (w.K >= 8192 or (x.shape[0] <= 128 and w.N <= 8192))):
        # 65..256 rows: X through LDS, one pass over the packed weights -- where it beats the library
        # (tools/bench_stream_gemm.py, M = 128: down 37 vs 76 us, o 21 vs 24, qkv 26 vs 27; not gate_up: 59 vs 55)
        return torch.ops._C_amd.stream_linear_packed(x, w.packed, bias, w.N, w.K)
    return F.linear(x, w.w, bias)


_CALIBRATING = False
_STREAM_GEMM_MAX_ROWS = 256  # set from ModelConfig.stream_gemm_max_rows by DecoderModel


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
        global _STREAM_GEMM_MAX_ROWS
        _STREAM_GEMM_MAX_ROWS = cfg.stream_gemm_max_rows
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
        if hasattr(attn_impl, "fuse_rope_over_fp8_cache"):
            attn_impl.fuse_rope_over_fp8_cache = cfg.rope_in_attention_fp8
        self.q_size = cfg.num_attention_heads * cfg.head_dim
        self.kv_size = cfg.num_key_value_heads * cfg.head_dim
        if cfg.quantization is not None:
            if cfg.quantization != "fp8":
                raise ValueError(f"unsupported quantization {cfg.quantization!r}")
            self._quantize_fp8(gen)

    def all_weights(self) -> List["Weight"]:
        ws = [self.lm_head]
        for l in self.layers:
            ws += [l.qkv, l.o, l.gate_up, l.down]
        return ws

    @torch.inference_mode()
    def _quantize_fp8(self, gen: torch.Generator) -> None:
        """Static activation scales from one 16-bit forward of a random 64-token prompt (dense
        causal attention, no KV cache), then every projection becomes fp8."""
        global _CALIBRATING
        cfg = self.cfg
        T = 64
        ids = torch.randint(0, cfg.vocab_size, (T,), generator=gen, device=self.device)
        pos = torch.arange(T, device=self.device)

        class _DenseAttn:  # the calibration pass needs attention outputs of realistic scale only
            def forward(_, q, k, v, kv_cache, md):
                H, KVH, D = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
                qh = q.view(T, H, D).transpose(0, 1)
                kh = k.view(T, KVH, D).transpose(0, 1).repeat_interleave(H // KVH, dim=0)
                vh = v.view(T, KVH, D).transpose(0, 1).repeat_interleave(H // KVH, dim=0)
                o = F.scaled_dot_product_attention(qh[None], kh[None], vh[None], is_causal=True)[0]
                return o.transpose(0, 1).reshape(T, H * D)

        real_attn, self.attn = self.attn, _DenseAttn()
        _CALIBRATING = True
        try:
            md = type("MD", (), {"num_prefill_tokens": T})()
            hidden = self.forward(ids, pos, None, md)
            self.compute_logits(hidden)
        finally:
            _CALIBRATING = False
            self.attn = real_attn
        for w in self.all_weights():
            w.quantize_fp8()
        torch.cuda.empty_cache()

    def weight_bytes(self) -> int:
        n = sum(w.numel() for w in self.all_weights())
        return n * (1 if self.cfg.quantization == "fp8" else self.embed.element_size())

    def _add_norm(self, x, residual: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
        """residual += x; return norm(residual) * weight.  `x` is either the projection output
        [T, hidden] or the fp32 split-K partials [S, T, hidden] of a fused down projection."""
        if isinstance(x, tuple):  # (raw fp32 partials of a W8A8 projection, x_scale, w_scale)
            out = torch.empty_like(residual)
            torch.ops._C_amd.fused_add_rms_norm_splitk_scaled(out, residual, x[0], weight, self.cfg.rms_norm_eps,
                                                              x[1], x[2])
            return out
        if x.dim() == 3:
            out = torch.empty_like(residual)
            torch.ops._C_amd.fused_add_rms_norm_splitk(out, residual, x, weight, self.cfg.rms_norm_eps)
            return out
        ops.fused_add_rms_norm(x, residual, weight, self.cfg.rms_norm_eps)
        return x

    def _norm_fp8(self, x, residual: Optional[torch.Tensor], weight: torch.Tensor, q_scale: torch.Tensor) -> torch.Tensor:
        """The add + norm in front of a W8A8 projection with its result leaving as fp8 ONLY, quantised with the
        projection's static activation scale (bit-identical to the norm followed by static_scaled_fp8_quant): the
        projection then takes its activations as they are (skinny_linear_w8a8_q) instead of re-quantising all of X in
        every one of its workgroups.  x: the layer's first hidden states (residual None), a projection output, or the
        raw split-K partials + scales of a W8A8 down projection."""
        eps = self.cfg.rms_norm_eps
        if residual is None:
            return torch.ops._C_amd.rms_norm_fp8(x, weight, eps, q_scale)
        if isinstance(x, tuple):
            return torch.ops._C_amd.fused_add_rms_norm_splitk_fp8(residual, x[0], weight, eps, x[1], x[2], q_scale)
        if x.dim() == 3:
            return torch.ops._C_amd.fused_add_rms_norm_splitk_fp8(residual, x, weight, eps, None, None, q_scale)
        return torch.ops._C_amd.fused_add_rms_norm_fp8(x, residual, weight, eps, q_scale)

    def forward(self, input_ids: torch.Tensor, positions: torch.Tensor,
                kv_caches: Optional[List[torch.Tensor]], attn_metadata, unified=None) -> torch.Tensor:
        """`unified` = (block_tables, seq_lens, query_start_loc, max_query_len, slot_mapping): the step is
        a mix of prompt chunks and decode tokens with all of its metadata on the device (a captured
        mixed step): rope + cache write fused, attention by the prefill kernel over every sequence."""
        cfg = self.cfg
        hidden = F.embedding(input_ids, self.embed)
        T = hidden.shape[0]
        # decode-only steps take the fused launches: rope + cache write in one kernel, split-K
        # partials of the down projection summed inside the next add+norm
        decode_only = unified is not None or (
            cfg.fuse_decode_ops and kv_caches is not None and T <= 64 and
            attn_metadata.num_prefill_tokens == 0 and hasattr(self.attn, "decode_attention"))
        slot_mapping = unified[4] if unified is not None else getattr(attn_metadata, "slot_mapping", None)
        # W8A8 decode steps of <= 32 rows: activations are quantised ONCE, by the kernel that produces them (the norm
        # launches, the SwiGLU epilogue), not again in every workgroup of the projection that consumes them
        q_once = (decode_only and cfg.fp8_activations_once and 0 < T <= 32 and
                  all(w.w8_packed is not None for w in (self.layers[0].qkv, self.layers[0].gate_up, self.layers[0].down))
                  and self.layers[0].gate_up.N % 32 == 0 and self.layers[0].gate_up.K <= 4096)
        residual = None
        unified_out = None
        for i, lw in enumerate(self.layers):
            qkv_roped = False
            if q_once:
                x8 = self._norm_fp8(hidden, residual, lw.input_norm, lw.qkv.x_scale)
                if residual is None:
                    residual = hidden
                qkv = torch.ops._C_amd.skinny_linear_w8a8_q(x8, lw.qkv.w8_packed, lw.qkv.w_scale, lw.qkv.x_scale,
                                                            lw.qkv.N, lw.qkv.K, lw.qkv_bias, cfg.dtype)
            else:
                if residual is None:  # qwen2.py:203-208
                    residual = hidden
                    normed = torch.empty_like(hidden)
                    ops.rms_norm(normed, hidden, lw.input_norm, cfg.rms_norm_eps)
                    hidden = normed
                else:
                    hidden = self._add_norm(hidden, residual, lw.input_norm)
                qkv = None
                if (unified is not None and cfg.qkv_reduce_in_rope and 32 < T <= 64 and lw.qkv.packed is not None
                        and lw.qkv.w8_packed is None and not _CALIBRATING):
                    # a mixed step of 33..64 rows: the projection splits K over workgroups; its fp32 slabs are summed
                    # by the rope + cache-write launch instead of a reduce launch of their own (bit-identical)
                    part = torch.ops._C_amd.skinny_linear_packed_partials(hidden, lw.qkv.packed, lw.qkv.N, lw.qkv.K, False)
                    if part.shape[0] > 1:
                        key_cache, value_cache = self.attn.split_kv_cache(kv_caches[i])
                        qkv = torch.empty(T, lw.qkv.N, dtype=hidden.dtype, device=hidden.device)
                        if not torch.ops._C_amd.rotary_embedding_and_cache_splitk(
                                positions, qkv, part, lw.qkv_bias, cfg.num_attention_heads, cfg.num_key_value_heads,
                                cfg.head_dim, self.cos_sin_cache, True, key_cache, value_cache, slot_mapping,
                                self.attn.kv_cache_dtype, 1.0, 1.0):
                            qkv = None
                        else:
                            qkv_roped = True
                if qkv is None:
                    qkv = linear(hidden, lw.qkv, lw.qkv_bias)
            q, k, v = qkv.split([self.q_size, self.kv_size, self.kv_size], dim=-1)  # strided views
            fused = False
            attn_out = None
            attn8 = None  # fp8 twin of the attention result for a W8A8 output projection (q_once)
            twin = lw.o.x_scale if (q_once and lw.o.w8_packed is not None and unified is None) else None
            if decode_only:
                key_cache, value_cache = self.attn.split_kv_cache(kv_caches[i])
            if decode_only and unified is None and cfg.rope_in_attention and hasattr(self.attn, "rope_cache_decode_attention"):
                # rope + cache write + attention in one launch (bit-identical to the three)
                attn_out = self.attn.rope_cache_decode_attention(positions, q, k, v, self.cos_sin_cache, key_cache,
                                                                 value_cache, attn_metadata, **({"fp8_twin_scale": twin} if twin is not None else {}))
                if isinstance(attn_out, tuple):
                    attn_out, attn8 = attn_out
            if qkv_roped:
                fused = True  # (rotated and written to the caches by the launch that summed the projection's slabs)
            elif decode_only and attn_out is None:
                fused = torch.ops._C_amd.rotary_embedding_and_cache(
                    positions, q, k, v, cfg.head_dim, self.cos_sin_cache, True, key_cache, value_cache,
                    slot_mapping, self.attn.kv_cache_dtype, 1.0, 1.0)
            if attn_out is not None:
                pass
            elif unified is not None:
                assert fused, "a captured mixed step needs the fused rope + cache write"
                if unified_out is None:  # one zeroed buffer per step (its padding rows stay zero), not one per layer
                    unified_out = torch.zeros(T, self.q_size, dtype=q.dtype, device=q.device)
                attn_out = self.attn.unified_attention(q, key_cache, value_cache, unified[0], unified[1],
                                                       unified[2], unified[3], output=unified_out)
            elif fused and twin is not None:
                attn_out, attn8 = self.attn.decode_attention(q, key_cache, value_cache, attn_metadata, fp8_twin_scale=twin)
            elif fused:
                attn_out = self.attn.decode_attention(q, key_cache, value_cache, attn_metadata)
            else:
                ops.rotary_embedding(positions, q, k, cfg.head_dim, self.cos_sin_cache, True)
                attn_out = self.attn.forward(q, k, v, kv_caches[i] if kv_caches is not None else None,
                                             attn_metadata)
            if decode_only and cfg.o_proj_partials_min_rows <= T <= 64 and lw.o.packed is not None and lw.o.w8_t is None:
                # K is split over workgroups at this many rows: the add+norm launch sums the fp32 partials
                # itself (as for the down projection), one reduce launch less
                hidden = torch.ops._C_amd.skinny_linear_packed_partials(attn_out, lw.o.packed, lw.o.N, lw.o.K, False)
            elif attn8 is not None:
                hidden = torch.ops._C_amd.skinny_linear_w8a8_q(attn8, lw.o.w8_packed, lw.o.w_scale, lw.o.x_scale, lw.o.N,
                                                               lw.o.K, None, cfg.dtype)
            else:
                hidden = linear(attn_out, lw.o)
            if q_once:
                g, d = lw.gate_up, lw.down
                h8 = self._norm_fp8(hidden, residual, lw.post_norm, g.x_scale)
                act8 = torch.ops._C_amd.skinny_linear_w8a8_q_swiglu_fp8(h8, g.w8_packed, g.w_scale, g.x_scale, g.N, g.K,
                                                                        d.x_scale, cfg.dtype)
                part = torch.ops._C_amd.skinny_linear_w8a8_q_partials(act8, d.w8_packed, d.w_scale, d.x_scale, d.N, d.K)
                if part.numel() > 0:  # the next add + norm sums the raw partials and applies the scales
                    hidden = (part, d.x_scale, d.w_scale)
                else:
                    hidden = torch.ops._C_amd.skinny_linear_w8a8_q(act8, d.w8_packed, d.w_scale, d.x_scale, d.N, d.K,
                                                                   None, cfg.dtype)
                continue
            hidden = self._add_norm(hidden, residual, lw.post_norm)
            if (decode_only and cfg.swiglu_epilogue and lw.gate_up.w8_packed is not None and lw.gate_up.N % 32 == 0
                    and T <= 64):
                g = lw.gate_up
                act = torch.ops._C_amd.skinny_linear_w8a8_swiglu(hidden, g.w8_packed, g.w_scale, g.x_scale, g.N, g.K, None)
            elif (decode_only and cfg.swiglu_epilogue and lw.gate_up.packed is not None and lw.gate_up.N % 32 == 0
                    and not _CALIBRATING):
                # gate_up projection with silu_and_mul in its epilogue: one launch, no [T, 2 inter] round trip
                act = torch.ops._C_amd.skinny_linear_packed_swiglu(hidden, lw.gate_up.packed, None,
                                                                   lw.gate_up.N, lw.gate_up.K)
            else:
                gate_up = linear(hidden, lw.gate_up)
                act = torch.empty(T, cfg.intermediate_size, dtype=gate_up.dtype, device=gate_up.device)
                ops.silu_and_mul(act, gate_up)
            hidden = None
            if decode_only and lw.down.packed is not None and lw.down.w8_t is None:
                # [S, T, hidden] fp32 split-K partial sums; the next add+norm adds them up
                hidden = torch.ops._C_amd.skinny_linear_packed_partials(act, lw.down.packed, lw.down.N,
                                                                        lw.down.K, False)
            elif decode_only and lw.down.w8_packed is not None and T <= 64:
                # W8A8: the raw partials and the scales the reduce pass would have applied (empty: K not split here)
                d = lw.down
                part = torch.ops._C_amd.skinny_linear_w8a8_partials(act, d.w8_packed, d.w_scale, d.x_scale, d.N, d.K)
                if part.numel() > 0:
                    hidden = (part, d.x_scale, d.w_scale)
            if hidden is None:
                hidden = linear(act, lw.down)
        return self._add_norm(hidden, residual, self.final_norm)

    def compute_logits(self, hidden: torch.Tensor) -> torch.Tensor:
        return linear(hidden, self.lm_head)

    def greedy_tokens(self, hidden: torch.Tensor) -> torch.Tensor:
        """argmax of the logits, [T] int64.  Decode batches take it from the lm_head projection's
        epilogue (no [T, vocab] tensor, no separate arg-max launch; same tokens as the two ops)."""
        w = self.lm_head
        if self.cfg.argmax_epilogue and 0 < hidden.shape[0] <= 64 and hidden.is_cuda and w.packed is not None:
            return torch.ops._C_amd.skinny_linear_packed_argmax(hidden, w.packed, w.N, w.K)
        if self.cfg.argmax_epilogue and 0 < hidden.shape[0] <= 64 and hidden.is_cuda and w.w8_packed is not None:
            return torch.ops._C_amd.skinny_linear_w8a8_argmax(hidden, w.w8_packed, w.w_scale, w.x_scale, w.N, w.K)
        return torch.argmax(self.compute_logits(hidden), dim=-1)
