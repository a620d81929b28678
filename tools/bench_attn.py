"""Kernel-only micro-benchmark of paged_attention_v1/v2 at the BASELINE shape:
HIP events on the launch stream, caches rotated so that reads come from HBM, not the
256 MiB Infinity Cache."""
import argparse
import math
import sys, os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import light_vllm_amd  # noqa
from light_vllm_amd import _custom_ops as ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", type=int, default=32)
    ap.add_argument("--seq", type=int, default=1024)
    ap.add_argument("--heads", type=int, default=32)
    ap.add_argument("--kv-heads", type=int, default=8)
    ap.add_argument("--head-size", type=int, default=128)
    ap.add_argument("--block-size", type=int, default=16)
    ap.add_argument("--ncaches", type=int, default=10)
    ap.add_argument("--contiguous", action="store_true",
                    help="sequence i owns blocks i * nblk .. (what a block manager hands out at prefill) instead of a random permutation")
    ap.add_argument("--block-pad", type=int, default=0,
                    help="bytes added to the block stride of the caches (a strided view; the kernels take the stride)")
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--kv", default="auto", choices=["auto", "fp8"])
    ap.add_argument("--rope", action="store_true",
                    help="also time the ROPE instantiation (rotary_embedding + reshape_and_cache + attention in one launch: "
                         "what a decode step of the engine launches); the new token is the context's last")
    a = ap.parse_args()
    dev = "cuda:0"
    dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[a.dtype]
    esz = 4 if a.dtype == "f32" else 2
    x = 16 // esz
    B, H, KVH, D, BS, L = a.bs, a.heads, a.kv_heads, a.head_size, a.block_size, a.seq
    nblk = (L + BS - 1) // BS
    NB = B * nblk + 7
    torch.manual_seed(0)
    caches = []
    for i in range(a.ncaches):
        if a.kv == "fp8":
            kc = (torch.randn(NB, KVH, D // 16, BS, 16, device=dev) * 0.5).to(torch.float8_e4m3fn).view(torch.uint8)
            vc = (torch.randn(NB, KVH, D, BS, device=dev) * 0.5).to(torch.float8_e4m3fn).view(torch.uint8)
        else:
            kc = (torch.randn(NB, KVH, D // x, BS, x, device=dev) * 0.5).to(dt)
            vc = (torch.randn(NB, KVH, D, BS, device=dev) * 0.5).to(dt)
        if a.block_pad:
            def padded(t):
                per = t[0].numel()
                pad = a.block_pad // t.element_size()
                buf = torch.empty(NB, per + pad, dtype=t.dtype, device=dev)
                buf[:, :per] = t.reshape(NB, per)
                return buf[:, :per].view(t.shape)  # stride(0) = per + pad
            kc, vc = padded(kc), padded(vc)
        if a.contiguous:
            bt = torch.arange(B * nblk, device=dev).view(B, nblk).to(torch.int32)
        else:
            bt = torch.randperm(NB, device=dev)[: B * nblk].view(B, nblk).to(torch.int32)
        caches.append((kc, vc, bt))
    q = (torch.randn(B, H, D, device=dev) * 0.5).to(dt)
    seq_lens = torch.full((B,), L, dtype=torch.int32, device=dev)
    out = torch.zeros_like(q)
    P = (L + 511) // 512
    tmp = torch.zeros(B, H, P, D, dtype=dt, device=dev)
    es = torch.zeros(B, H, P, dtype=torch.float32, device=dev)
    ml = torch.zeros_like(es)
    scale = 1 / math.sqrt(D)
    kvb = 1 if a.kv == "fp8" else esz
    algo_bytes = 2 * B * L * KVH * D * kvb + 2 * B * H * D * esz + B * nblk * 4 + B * 4

    def v1(i):
        kc, vc, bt = caches[i % a.ncaches]
        ops.paged_attention_v1(out, q, kc, vc, KVH, scale, bt, seq_lens, BS, L, None, a.kv, 1.0, 1.0)

    def v2(i):
        kc, vc, bt = caches[i % a.ncaches]
        ops.paged_attention_v2(out, es, ml, tmp, q, kc, vc, KVH, scale, bt, seq_lens, BS, L, None, a.kv, 1.0, 1.0)

    legs = [("v1", v1), ("v2", v2)]
    if a.rope:
        qkv = (torch.randn(B, (H + 2 * KVH) * D, device=dev) * 0.5).to(dt)
        qv, kv_, vv = qkv.split([H * D, KVH * D, KVH * D], dim=-1)
        pos = torch.full((B,), L - 1, dtype=torch.int64, device=dev)
        inv = 1.0 / (500000.0 ** (torch.arange(0, D, 2, dtype=torch.float) / D))
        fr = torch.einsum("i,j -> ij", torch.arange(L + 8, dtype=torch.float), inv)
        cos_sin = torch.cat((fr.cos(), fr.sin()), dim=-1).to(dt).to(dev)
        slots = [(bt[:, (L - 1) // BS].long() * BS + (L - 1) % BS) for _, _, bt in caches]
        out2 = torch.zeros(B, H, D, dtype=dt, device=dev)

        def rope(i):
            kc, vc, bt = caches[i % a.ncaches]
            ok = torch.ops._C_amd.rope_cache_paged_attention(out2, es, ml, tmp, pos, qv, kv_, vv, D, cos_sin, True, kc, vc,
                                                             slots[i % a.ncaches], KVH, scale, bt, seq_lens, BS, L, a.kv,
                                                             1.0, 1.0)
            assert ok
        legs.append(("v2+rope+cache", rope))
    for name, fn in legs:
        for i in range(20):
            fn(i)
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.iters)]
        for i, (s, e) in enumerate(evs):
            s.record()
            fn(i)
            e.record()
        torch.cuda.synchronize()
        ts = sorted(s.elapsed_time(e) * 1e3 for s, e in evs)
        med, mn = ts[len(ts) // 2], ts[0]
        # trains of 32 back-to-back launches between one event pair (what bench.py's roofline object times):
        # the per-launch figure carries the kernel and its launch gap, not the cost of two events per kernel
        tr = []
        k = 0
        for _ in range(max(2, a.iters // 32)):
            s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_.record()
            for _ in range(32):
                fn(k)
                k += 1
            e_.record()
            torch.cuda.synchronize()
            tr.append(s_.elapsed_time(e_) * 1e3 / 32)
        tavg = sum(tr) / len(tr)
        print(f"{name}: median {med:.1f} us  min {mn:.1f} us  -> {algo_bytes / med / 1e6:.2f} TB/s median "
              f"({algo_bytes / med / 1e6 / 8 * 100:.1f}% of 8 TB/s); trains of 32: {tavg:.2f} us per launch "
              f"(min {min(tr):.2f}) -> {algo_bytes / tavg / 1e6:.2f} TB/s = {algo_bytes / tavg / 1e6 / 8:.3f} of 8 TB/s; "
              f"algorithmic bytes {algo_bytes}")


if __name__ == "__main__":
    main()
