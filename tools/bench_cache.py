"""reshape_and_cache at prompt sizes: the LDS-tiled kernel against the per-chunk kernel.
HIP events around trains of launches; a ring of caches much larger than the 256 MiB Infinity Cache so that every
launch writes HBM.  Algorithmic bytes: 4 * T * KVH * D * sizeof + 8 * T (K and V read once, written once).

  python tools/bench_cache.py [--tokens 8192] [--kv-heads 8] [--head-size 128] [--block-size 16]
Under rocprofv3 (`--kernel-trace --stats`) the two kernels show up by name: reshape_and_cache_tile_kernel and
reshape_and_cache_kernel<unsigned short, 8>."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import light_vllm_amd  # noqa
from light_vllm_amd import _custom_ops as ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=8192)
    ap.add_argument("--kv-heads", type=int, default=8)
    ap.add_argument("--head-size", type=int, default=128)
    ap.add_argument("--block-size", type=int, default=16)
    ap.add_argument("--seqs", type=int, default=8, help="the tokens are this many prompts laid out back to back")
    ap.add_argument("--ncaches", type=int, default=24)
    ap.add_argument("--iters", type=int, default=96)
    a = ap.parse_args()
    dev = "cuda:0"
    T, H, D, BS = a.tokens, a.kv_heads, a.head_size, a.block_size
    dt = torch.bfloat16
    torch.manual_seed(0)
    per = T // a.seqs
    nblk_seq = (per + BS - 1) // BS
    NB = a.seqs * nblk_seq + 17
    qkv = torch.randn(T, 3 * H * D, device=dev).to(dt)  # k, v as strided views of a fused row (qwen2.py:151-152)
    key = qkv[:, H * D:2 * H * D].view(T, H, D)
    value = qkv[:, 2 * H * D:].view(T, H, D)
    caches = []
    for _ in range(a.ncaches):
        kc = torch.zeros(NB, H, D // 8, BS, 8, dtype=dt, device=dev)
        vc = torch.zeros(NB, H, D, BS, dtype=dt, device=dev)
        table = torch.randperm(NB, device=dev)[: a.seqs * nblk_seq].view(a.seqs, nblk_seq)
        pos = torch.arange(per, device=dev)
        slots = torch.cat([table[s][pos // BS] * BS + pos % BS for s in range(a.seqs)])
        if slots.numel() < T:
            slots = torch.cat([slots, torch.full((T - slots.numel(),), -1, device=dev, dtype=slots.dtype)])
        caches.append((kc, vc, slots.to(torch.int64)))
    algo = 4 * T * H * D * 2 + 8 * T
    for name, min_tok in (("tiled (LDS)", 64), ("per-chunk", 1 << 30)):
        torch.ops._C_amd.set_tuning("cache_tile_min_tokens", min_tok)
        for i in range(a.ncaches):
            kc, vc, sl = caches[i]
            ops.reshape_and_cache(key, value, kc, vc, sl, "auto", 1.0, 1.0)
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(4)]
        n = a.iters // 4
        k = 0
        for s, e in evs:
            s.record()
            for _ in range(n):
                kc, vc, sl = caches[k % a.ncaches]
                ops.reshape_and_cache(key, value, kc, vc, sl, "auto", 1.0, 1.0)
                k += 1
            e.record()
        torch.cuda.synchronize()
        us = min(s.elapsed_time(e) for s, e in evs) * 1e3 / n
        print(f"{name:12s} T={T} KVH={H} D={D} BS={BS}: {us:7.2f} us/launch  {algo / us / 1e6:6.2f} TB/s of "
              f"{algo / 1e6:.1f} MB algorithmic")
    torch.ops._C_amd.set_tuning("cache_tile_min_tokens", 384)  # the default (csrc/common.h)


if __name__ == "__main__":
    main()
