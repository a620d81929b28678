"""Two streams running decode GEMMs side by side (what two steps in flight do): aggregate weight
bandwidth for a given `gemm_workgroups` setting."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import light_vllm_amd  # noqa
from light_vllm_amd import _custom_ops  # noqa: F401  (loads the libraries)

dev = "cuda:0"
wgs = int(sys.argv[1]) if len(sys.argv) > 1 else 128
nstreams = int(sys.argv[2]) if len(sys.argv) > 2 else 2
torch.ops._C_amd.set_tuning("gemm_workgroups", wgs)
M = 32
for name, N, K in [("qkv", 6144, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336)]:
    nw = 6
    ws = [[torch.ops._C_amd.pack_weight((torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)) for _ in range(nw)]
          for _ in range(nstreams)]
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    iters = 60

    def run():
        for i in range(iters):
            for s, st in enumerate(streams):
                with torch.cuda.stream(st):
                    torch.ops._C_amd.skinny_linear_packed(x, ws[s][i % nw], None, N, K)
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    total = iters * nstreams * N * K * 2
    print(f"{name:8s} wgs={wgs:3d} streams={nstreams}: {dt / iters * 1e6:7.1f} us per round, aggregate {total / dt / 1e12:.2f} TB/s")
