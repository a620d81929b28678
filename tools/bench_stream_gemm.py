"""65..256-row weight-streaming GEMM vs torch (hipBLASLt) on the projection shapes of Llama-3-8B:
graph trains of 16 launches, weights rotated through HBM."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import light_vllm_amd  # noqa
from light_vllm_amd import _custom_ops  # noqa

dev = "cuda:0"
TRAIN = 16


def timed_graph(fns):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for f in fns[:3]:
            f()
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for f in fns:
                f()
        for _ in range(2):
            g.replay()
        s.synchronize()
        ts = []
        for _ in range(10):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(s); g.replay(); b.record(s); s.synchronize()
            ts.append(a.elapsed_time(b) * 1e3 / len(fns))
    ts.sort()
    return ts[len(ts) // 2]


for M in [int(a) for a in sys.argv[1:]] or [64, 128, 256]:
    print(f"M = {M}")
    for name, N, K in [("qkv", 6144, 4096), ("o", 4096, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336)]:
        nw = max(2, min(8, int(2.0e9 // (N * K * 2))))
        ws = [(torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16) for _ in range(nw)]
        wp = [torch.ops._C_amd.pack_weight(w) for w in ws]
        x = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
        t_s = timed_graph([(lambda w=wp[i % nw]: torch.ops._C_amd.stream_linear_packed(x, w, None, N, K)) for i in range(TRAIN)])
        t_t = timed_graph([(lambda w=ws[i % nw]: F.linear(x, w)) for i in range(TRAIN)])
        t_k = None
        if M <= 64:
            t_k = timed_graph([(lambda w=wp[i % nw]: torch.ops._C_amd.skinny_linear_packed(x, w, None, N, K)) for i in range(TRAIN)])
        by = N * K * 2
        print(f"  {name:8s} stream {t_s:7.1f} us ({by / t_s / 1e6:4.2f} TB/s)   torch {t_t:7.1f} us ({by / t_t / 1e6:4.2f} TB/s)"
              + (f"   register kernel {t_k:7.1f} us" if t_k else ""), flush=True)
        del ws, wp
        torch.cuda.empty_cache()
