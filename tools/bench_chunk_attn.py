"""Mixed steps and short chunks over a cached context: the walk of csrc/prefill_chunk.h against the prefill bodies: HIP events
around trains of launches on the launch stream.  Bytes counted = every K and V byte of every (sequence, kv head) once.
Shapes: VERDICT r02 item 4's two synthetic launches and a mixed step of BASELINE config 3 (decode rows + one chunk)."""
import argparse
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import light_vllm_amd  # noqa
from light_vllm_amd import _custom_ops as ops


def bench(seq_lens, qlens, H=32, KVH=8, D=128, BS=16, iters=200, dt=torch.bfloat16, dev="cuda:0", only_shipped=False):
    B = len(seq_lens)
    nblk = (max(seq_lens) + BS - 1) // BS
    NB = B * nblk + 7
    torch.manual_seed(0)
    kc = (torch.randn(NB, KVH, D // 8, BS, 8, device=dev) * 0.5).to(dt)
    vc = (torch.randn(NB, KVH, D, BS, device=dev) * 0.5).to(dt)
    bt = torch.randperm(NB, device=dev)[: B * nblk].view(B, nblk).to(torch.int32)
    T = sum(qlens)
    q = (torch.randn(T, H, D, device=dev) * 0.5).to(dt)
    out = torch.zeros_like(q)
    sl = torch.tensor(seq_lens, dtype=torch.int32, device=dev)
    qsl = torch.tensor([0] + list(torch.tensor(qlens).cumsum(0)), dtype=torch.int32, device=dev)
    scale = 1 / math.sqrt(D)
    nbytes = sum(seq_lens) * KVH * D * 2 * 2

    def run(max_seq_len):
        ops.paged_prefill_attention(out, q, kc, vc, KVH, scale, bt, sl, qsl, max(qlens), BS, None, 0, 0.0, "auto", True,
                                    1.0, 1.0, max_seq_len)

    def time(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / iters * 1e3

    res = {}
    tune = torch.ops._C_amd.set_tuning
    if only_shipped:
        return {"as shipped": time(lambda: run(max(seq_lens)))}, nbytes
    tune("prefill_chunk_max_avg_x8", 1 << 20)  # the walk of prefill_chunk.h whatever the token count
    res["mixed-step walk (forced), single pass"] = time(lambda: run(0))
    res["mixed-step walk (forced), partitions allowed"] = time(lambda: run(max(seq_lens)))
    tune("prefill_chunk_max_avg_x8", 16)
    tune("prefill_chunk_max_query", 0)
    res["prefill bodies, single pass (round 2)"] = time(lambda: run(0))
    res["prefill bodies, partitions allowed"] = time(lambda: run(max(seq_lens)))
    tune("prefill_chunk_max_query", 64)
    res["as shipped"] = time(lambda: run(max(seq_lens)))
    return res, nbytes


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--only-shipped", action="store_true")
    a = ap.parse_args()
    shapes = {"16 seqs x (16 tokens over 1 024)": ([1040] * 16, [16] * 16),
              "8 seqs x (32 tokens over 2 048)": ([2080] * 8, [32] * 8),
              "mixed step: 32 decode rows at 1 024 + one chunk of 32 over 480": ([1024] * 32 + [512], [1] * 32 + [32]),
              "mixed step: 48 decode rows at 600 + one chunk of 16 over 200": ([600] * 48 + [216], [1] * 48 + [16]),
              "32 decode rows at 1 024 (one-token chunks)": ([1024] * 32, [1] * 32),
              "1 seq x (32 tokens over 4 096)": ([4128], [32])}
    for name, (sl, ql) in shapes.items():
        res, nbytes = bench(sl, ql, iters=a.iters, only_shipped=a.only_shipped)
        print(name, f"({nbytes / 1e6:.1f} MB of K/V)")
        for k, us in res.items():
            print(f"    {k:46s} {us:8.1f} us   {nbytes / us / 1e6:6.2f} TB/s   {nbytes / us / 1e6 / 8:5.3f} of 8 TB/s")
