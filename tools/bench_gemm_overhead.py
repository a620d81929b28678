"""Where a decode GEMM's fixed cost goes: trains of 32 launches inside one HIP graph (no host launch
cost), weights rotated through HBM vs the same weights every launch (resident in L2 / MALL where
they fit), against the bytes/bandwidth floor.  M = 32, bf16, Llama-3-8B projection shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import light_vllm_amd  # noqa
from light_vllm_amd import _custom_ops  # noqa

dev = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32
TRAIN = 32


def timed_graph(fns):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for f in fns[:4]:
            f()
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for f in fns:
                f()
        for _ in range(3):
            g.replay()
        s.synchronize()
        ts = []
        for _ in range(20):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(s); g.replay(); b.record(s); s.synchronize()
            ts.append(a.elapsed_time(b) * 1e3 / len(fns))
    ts.sort()
    return ts[len(ts) // 2]


empty = torch.zeros(64, device=dev)
print(f"tiny elementwise kernel in a graph train: {timed_graph([lambda: empty.add_(1)] * TRAIN):.2f} us per launch")
for name, N, K in [("qkv", 6144, 4096), ("o", 4096, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336)]:
    nw = max(2, min(12, int(2.4e9 // (N * K * 2))))
    ws = [torch.ops._C_amd.pack_weight((torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)) for _ in range(nw)]
    x = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    rot = timed_graph([(lambda w=ws[i % nw]: torch.ops._C_amd.skinny_linear_packed(x, w, None, N, K)) for i in range(TRAIN)])
    same = timed_graph([(lambda: torch.ops._C_amd.skinny_linear_packed(x, ws[0], None, N, K)) for i in range(TRAIN)])
    by = N * K * 2
    print(f"{name:8s} {by / 1e6:6.1f} MB  rotated {rot:6.2f} us ({by / rot / 1e6:4.2f} TB/s)   resident {same:6.2f} us   "
          f"floor at 6.3 TB/s {by / 6.3e6:5.2f} us", flush=True)
