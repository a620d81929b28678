"""Kernel-only micro-benchmark of paged_prefill_attention (prefill_mfma.h): HIP events on the
launch stream.  FLOPs counted = 4 * D * (visible query-key pairs) * H (QK^T and PV, causal).
Optionally times torch SDPA on the equivalent dense problem for scale (--sdpa)."""
import argparse
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import light_vllm_amd  # noqa
from light_vllm_amd import _custom_ops as ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seqs", type=int, default=1)
    ap.add_argument("--ctx", type=int, default=0, help="tokens already in the cache before the chunk")
    ap.add_argument("--qlen", type=int, default=4096)
    ap.add_argument("--heads", type=int, default=32)
    ap.add_argument("--kv-heads", type=int, default=8)
    ap.add_argument("--head-size", type=int, default=128)
    ap.add_argument("--block-size", type=int, default=16)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--sdpa", action="store_true")
    ap.add_argument("--mfma32-min-query", type=int, default=None,
                    help="tuning prefill_mfma32_min_query: chunks at least this long take the 32x32-MFMA body (0: never)")
    ap.add_argument("--dense", action="store_true", help="lvllm_varlen_attention on dense q/k/v (no cache)")
    ap.add_argument("--encoder", action="store_true", help="with --dense: bidirectional attention")
    a = ap.parse_args()
    dev = "cuda:0"
    if a.mfma32_min_query is not None:
        torch.ops._C_amd.set_tuning("prefill_mfma32_min_query", a.mfma32_min_query)
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[a.dtype]
    B, H, KVH, D, BS = a.seqs, a.heads, a.kv_heads, a.head_size, a.block_size
    S = a.ctx + a.qlen
    nblk = (S + BS - 1) // BS
    NB = B * nblk + 7
    torch.manual_seed(0)
    kc = (torch.randn(NB, KVH, D // 8, BS, 8, device=dev) * 0.5).to(dt)
    vc = (torch.randn(NB, KVH, D, BS, device=dev) * 0.5).to(dt)
    bt = torch.randperm(NB, device=dev)[: B * nblk].view(B, nblk).to(torch.int32)
    q = (torch.randn(B * a.qlen, H, D, device=dev) * 0.5).to(dt)
    out = torch.zeros_like(q)
    seq_lens = torch.full((B,), S, dtype=torch.int32, device=dev)
    qsl = (torch.arange(B + 1, device=dev) * a.qlen).to(torch.int32)
    scale = 1 / math.sqrt(D)
    pairs = B * (a.qlen * a.ctx + a.qlen * (a.qlen + 1) // 2)
    flops = 4.0 * D * pairs * H

    def hip():
        ops.paged_prefill_attention(out, q, kc, vc, KVH, scale, bt, seq_lens, qsl, a.qlen, BS, None, 0, 0.0, "auto")

    def time(fn, n):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for s, e in evs:
            s.record()
            fn()
            e.record()
        torch.cuda.synchronize()
        ts = sorted(s.elapsed_time(e) * 1e3 for s, e in evs)
        return ts[len(ts) // 2], ts[0]

    if a.dense:
        assert a.ctx == 0
        kd = (torch.randn(B * a.qlen, KVH, D, device=dev) * 0.5).to(dt)
        vd = (torch.randn(B * a.qlen, KVH, D, device=dev) * 0.5).to(dt)
        if a.encoder:
            flops = 4.0 * D * B * a.qlen * a.qlen * H

        def hip():  # noqa: F811
            ops.varlen_attention(out, q, kd, vd, qsl, a.qlen, scale, not a.encoder)

    med, mn = time(hip, a.iters)
    print(f"hip prefill: seqs {B} ctx {a.ctx} qlen {a.qlen} H {H} KVH {KVH} D {D}: median {med:.1f} us min {mn:.1f} us "
          f"-> {flops / med / 1e6:.1f} TFLOP/s ({flops / med / 1e6 / 2500 * 100:.1f}% of 2.5 PFLOP/s bf16 dense)")
    if a.sdpa:
        qd = q.view(B, a.qlen, H, D).transpose(1, 2)
        kd = (torch.randn(B, H, S, D, device=dev) * 0.5).to(dt)
        vd = (torch.randn(B, H, S, D, device=dev) * 0.5).to(dt)
        if a.ctx == 0:
            fn = lambda: torch.nn.functional.scaled_dot_product_attention(qd, kd, vd, is_causal=not a.encoder,
                                                                           scale=scale)
        else:
            mask = torch.ones(a.qlen, S, dtype=torch.bool, device=dev).tril(diagonal=a.ctx)
            fn = lambda: torch.nn.functional.scaled_dot_product_attention(qd, kd, vd, attn_mask=mask, scale=scale)
        med, mn = time(fn, a.iters)
        print(f"torch sdpa (dense K/V already gathered and GQA-expanded): median {med:.1f} us min {mn:.1f} us "
              f"-> {flops / med / 1e6:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
