"""swap_blocks and copy_blocks.

swap: the reference's own benchmark shape (benchmarks/benchmark_swap_blocks.py:66-80: 1024 blocks of [16, 32, 32]
fp16 = 32 KiB each, identity mapping, 100 iterations, host clock around op + synchronize) in both directions, plus
a scattered mapping (every block its own DMA) and the engine's layout (one layer's K cache of the Llama-3-8B
shape, 64 blocks of 32 KiB = one 1024-token sequence).  PCIe Gen5 x16: 63 GB/s per direction (spec).
copy: copy_blocks over 32 layers x K and V, pairs of 32 KiB blocks, device-resident pair list; algorithmic bytes
4 * L * pairs * block bytes (read + write, K and V)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import light_vllm_amd  # noqa
from light_vllm_amd import _custom_ops as ops

dev = "cuda:0"


def timed(fn, n=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def swap(shape, mapping_name, mapping):
    nbytes = len(mapping) * torch.empty(shape[1:], dtype=torch.float16).numel() * 2
    bm = torch.tensor(mapping, dtype=torch.int64).view(-1, 2)
    host = torch.randn(shape, dtype=torch.float16).pin_memory()
    gpu = torch.zeros(shape, dtype=torch.float16, device=dev)
    t_in = timed(lambda: ops.swap_blocks(host, gpu, bm))
    assert torch.equal(gpu.cpu()[bm[:, 1]], host[bm[:, 0]])
    host2 = torch.zeros(shape, dtype=torch.float16).pin_memory()
    t_out = timed(lambda: ops.swap_blocks(gpu, host2, bm))
    print(f"swap_blocks {mapping_name:28s} {len(mapping):5d} blocks {nbytes / 2**20:6.1f} MiB: CPU->GPU {t_in * 1e3:7.3f} ms "
          f"({nbytes / t_in / 1e9:5.1f} GB/s)   GPU->CPU {t_out * 1e3:7.3f} ms ({nbytes / t_out / 1e9:5.1f} GB/s)")


def main():
    n = 1024
    shape = (n, 16, 32, 32)
    swap(shape, "identity (reference shape)", [(i, i) for i in range(n)])
    g = torch.Generator().manual_seed(0)
    perm = torch.randperm(n, generator=g).tolist()
    swap(shape, "scattered (no two adjacent)", [(i, perm[i]) for i in range(n)])
    swap((4096, 8, 16, 16, 8), "one sequence, 64 blocks", [(100 + i, perm[i]) for i in range(64)])
    # copy_blocks
    L, NB, KVH, D, BS = 32, 512, 8, 128, 16
    kcs = [torch.randn(NB, KVH, D // 8, BS, 8, device=dev).to(torch.bfloat16) for _ in range(L)]
    vcs = [torch.randn(NB, KVH, D, BS, device=dev).to(torch.bfloat16) for _ in range(L)]
    for pairs in (1, 8, 64):
        src = torch.randperm(NB, generator=g)[:2 * pairs]
        bm = torch.stack([src[:pairs], src[pairs:]], 1).to(torch.int64).to(dev)
        fn = lambda: ops.copy_blocks(kcs, vcs, bm)
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            fn()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / 50
        nbytes = 4 * L * pairs * KVH * D * BS * 2
        print(f"copy_blocks {pairs:3d} pairs x {L} layers: {us:7.2f} us  {nbytes / us / 1e6:6.2f} TB/s of {nbytes / 2**20:.1f} MiB")


if __name__ == "__main__":
    main()
