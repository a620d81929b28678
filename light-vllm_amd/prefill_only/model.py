"""Encoder of the encode-only workflow: the XLM-RoBERTa / BERT layer sequence of
light_vllm/encode_only/modelzoo/xlm_roberta.py (bge-m3's backbone, BASELINE config 4): embeddings
(word + position + token type, LayerNorm), then per layer fused QKV projection -> bidirectional
attention -> output projection -> add & LayerNorm -> GELU MLP -> add & LayerNorm.

Attention is this package's HIP varlen kernel (AttentionType.ENCODER); the residual add and the LayerNorm
behind it are one launch (`_C_amd.add_layer_norm`; the reference leaves both to torch),
exact GELU is an in-place kernel of this package too; the projections are library GEMMs (prompt batches are hundreds to thousands of rows).
Weights are random-initialised: no checkpoint is available offline."""
from dataclasses import dataclass
from typing import List

import torch
import torch.nn.functional as F

from ..attention.prefill_only import AttentionType, PrefillOnlyHIPVarlenBackend



@dataclass
class EncoderConfig:
    hidden_size: int = 1024
    intermediate_size: int = 4096
    num_hidden_layers: int = 24
    num_attention_heads: int = 16
    vocab_size: int = 250002
    max_position_embeddings: int = 8194
    layer_norm_eps: float = 1e-5
    pad_token_id: int = 1
    dtype: torch.dtype = torch.bfloat16
    fused_layer_norm: bool = True  # residual add + LayerNorm in one launch (lvllm_add_layer_norm); off = A/B runs
    # > 0: the MLP runs over row blocks of this many tokens (fc1 -> GELU -> fc2 per block), so that the block's
    # intermediate [rows, intermediate] is still in the L2s / the Infinity Cache when GELU and fc2 read it instead of
    # making two round trips through HBM (134 MB each way at 16 384 tokens); 0 = one pass over all tokens
    mlp_block_tokens: int = 0

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @staticmethod
    def bge_m3(dtype=torch.bfloat16) -> "EncoderConfig":
        return EncoderConfig(dtype=dtype)

    @staticmethod
    def tiny(dtype=torch.bfloat16) -> "EncoderConfig":
        return EncoderConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                             vocab_size=512, max_position_embeddings=1024, dtype=dtype)


class EncoderLayerWeights:
    def __init__(self, cfg: EncoderConfig, device, gen: torch.Generator):
        hid, inter = cfg.hidden_size, cfg.intermediate_size

        def w(*shape, std=0.02):
            return (torch.randn(*shape, generator=gen, device=device, dtype=torch.float32) * std).to(cfg.dtype)

        def ln():
            return ((1.0 + 0.05 * torch.randn(hid, generator=gen, device=device)).to(cfg.dtype),
                    (0.02 * torch.randn(hid, generator=gen, device=device)).to(cfg.dtype))

        self.qkv_w, self.qkv_b = w(3 * hid, hid), w(3 * hid)
        self.out_w, self.out_b = w(hid, hid), w(hid)
        self.attn_ln = ln()
        self.fc1_w, self.fc1_b = w(inter, hid), w(inter)
        self.fc2_w, self.fc2_b = w(hid, inter), w(hid)
        self.out_ln = ln()


class EncoderModel:
    """forward(input_ids [T], positions [T], attn_metadata) -> hidden states [T, hidden]."""

    def __init__(self, cfg: EncoderConfig, device="cuda:0", seed: int = 0):
        self.cfg = cfg
        self.device = torch.device(device)
        gen = torch.Generator(device=self.device).manual_seed(seed)

        def emb(n):
            return (torch.randn(n, cfg.hidden_size, generator=gen, device=self.device) * 0.02).to(cfg.dtype)

        self.word_emb, self.pos_emb, self.type_emb = emb(cfg.vocab_size), emb(cfg.max_position_embeddings), emb(1)
        self.emb_ln = ((1.0 + 0.05 * torch.randn(cfg.hidden_size, generator=gen, device=self.device)).to(cfg.dtype),
                       (0.02 * torch.randn(cfg.hidden_size, generator=gen, device=self.device)).to(cfg.dtype))
        self.layers: List[EncoderLayerWeights] = [EncoderLayerWeights(cfg, self.device, gen)
                                                  for _ in range(cfg.num_hidden_layers)]
        self.backend = PrefillOnlyHIPVarlenBackend(AttentionType.ENCODER)
        self.attn = self.backend.get_impl_cls()(cfg.num_attention_heads, cfg.head_dim, cfg.head_dim ** -0.5,
                                                cfg.num_attention_heads, None, None, "auto")

    def forward(self, input_ids: torch.Tensor, positions: torch.Tensor, attn_metadata) -> torch.Tensor:
        cfg = self.cfg
        eps = cfg.layer_norm_eps
        hid = cfg.hidden_size
        # xlm_roberta.py: position ids start after the padding index
        x = (F.embedding(input_ids, self.word_emb) + F.embedding(positions + cfg.pad_token_id + 1, self.pos_emb)
             + self.type_emb[0])
        fused = x.is_cuda and hid % 8 == 0 and x.dtype in (torch.bfloat16, torch.float16) and self.cfg.fused_layer_norm

        def add_ln(a, b, ln):  # LayerNorm(a + b), b optional
            if fused:
                out = torch.empty_like(a)
                torch.ops._C_amd.add_layer_norm(out, a, b, ln[0], ln[1], eps)
                return out
            return F.layer_norm(a if b is None else a + b, (hid,), ln[0], ln[1], eps)

        x = add_ln(x, None, self.emb_ln)
        for lw in self.layers:
            qkv = F.linear(x, lw.qkv_w, lw.qkv_b)
            q, k, v = qkv.split([hid, hid, hid], dim=-1)  # strided views of the fused projection
            a = self.attn.forward(q, k, v, None, attn_metadata, attn_type=AttentionType.ENCODER)
            x = add_ln(x, F.linear(a, lw.out_w, lw.out_b), lw.attn_ln)
            blk = cfg.mlp_block_tokens
            if blk > 0 and x.shape[0] > blk:
                y = torch.empty_like(x)
                for r0 in range(0, x.shape[0], blk):
                    hb = F.linear(x[r0:r0 + blk], lw.fc1_w, lw.fc1_b)
                    if fused and hb.numel() % 8 == 0:
                        torch.ops._C_amd.gelu(hb, hb)
                    else:
                        hb = F.gelu(hb)
                    torch.addmm(lw.fc2_b, hb, lw.fc2_w.t(), out=y[r0:r0 + blk])
            else:
                h = F.linear(x, lw.fc1_w, lw.fc1_b)
                if fused and h.numel() % 8 == 0:
                    torch.ops._C_amd.gelu(h, h)  # exact GELU in place
                else:
                    h = F.gelu(h)
                y = F.linear(h, lw.fc2_w, lw.fc2_b)
            x = add_ln(x, y, lw.out_ln)
        return x
