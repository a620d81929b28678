"""Decode-GEMM micro-benchmark: weight-streaming kernel vs torch (hipBLASLt) on the projection
shapes of Llama-3-8B at M = 32; weights rotated so every launch streams from HBM."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import light_vllm_amd  # noqa
from light_vllm_amd import _custom_ops  # noqa

dev = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for name, N, K in [("qkv", 6144, 4096), ("o", 4096, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336), ("lm_head", 128256, 4096)]:
    nw = max(2, min(12, int(1.2e9 // (N * K * 2))))
    ws = [(torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16) for _ in range(nw)]
    x = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    res = {}
    for label, fn in (("skinny", lambda w: torch.ops._C_amd.skinny_linear(x, w, None)),
                      ("packed", lambda w: torch.ops._C_amd.skinny_linear_packed(x, w, None, N, K)),
                      ("torch", lambda w: F.linear(x, w))):
        for i in range(5):
            fn(ws[i % nw])
        torch.cuda.synchronize()
        evs = []
        for i in range(60):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(ws[i % nw]); b.record(); evs.append((a, b))
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
        res[label] = ts[len(ts) // 2]
    by = N * K * 2
    print(f"{name:8s} N={N:6d} K={K:5d}  skinny {res['skinny']:7.1f} us ({by / res['skinny'] / 1e6:5.2f} TB/s)   packed {res['packed']:7.1f} us ({by / res['packed'] / 1e6:5.2f} TB/s)   torch {res['torch']:7.1f} us ({by / res['torch'] / 1e6:5.2f} TB/s)")
