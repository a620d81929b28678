"""A/B: workgroups per decode-GEMM launch against how evenly the n-tiles divide among them.  M = 32 rows of bf16, the four
projection shapes of an 8B model through torch.ops._C_amd (skinny_linear_packed; gate_up with its SwiGLU epilogue);
trains of 32 launches over rotating packed weights (>> the Infinity Cache), alternating grids in ONE process.
usage: python tools/ab_gemm_grid.py [grid ...]     (default: 256 224 192 128; 0 = the library's own choice)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import light_vllm_amd  # noqa
from light_vllm_amd import _custom_ops  # noqa

dev = "cuda:0"
M, TRAIN, TRAINS = 32, 32, 6
grids = [int(a) for a in sys.argv[1:]] or [256, 224, 192, 128]
SHAPES = [("qkv", 6144, 4096, False), ("o", 4096, 4096, False), ("gate_up", 28672, 4096, True), ("down", 4096, 14336, False)]
res = {}
for name, N, K, glu in SHAPES:
    nw = max(4, min(10, int(1.0e9 // (N * K * 2))))
    ws = [torch.ops._C_amd.pack_weight((torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)) for _ in range(nw)]
    x = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)

    def call(w):
        if glu:
            torch.ops._C_amd.skinny_linear_packed_swiglu(x, w, None, N, K)
        else:
            torch.ops._C_amd.skinny_linear_packed(x, w, None, N, K)
    for rnd in range(2):
        for g in grids:
            torch.ops._C_amd.set_tuning("gemm_workgroups", g if g > 0 else 256)
            torch.ops._C_amd.set_tuning("gemm_balance", 1 if g == 0 else 0)
            for i in range(8):
                call(ws[i % nw])
            torch.cuda.synchronize()
            ts = []
            k = 0
            for _ in range(TRAINS):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(TRAIN):
                    call(ws[k % nw])
                    k += 1
                b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b) / TRAIN * 1e3)
            res.setdefault((name, g), []).append(min(ts))
    del ws
    torch.cuda.empty_cache()
torch.ops._C_amd.set_tuning("gemm_workgroups", 256)
torch.ops._C_amd.set_tuning("gemm_balance", 1)
print("M = 32 bf16; us per launch (min of %d trains of %d), two rounds; TB/s of the weights at the better round" % (TRAINS, TRAIN))
for name, N, K, glu in SHAPES:
    row = []
    for g in grids:
        t = res[(name, g)]
        row.append(f"{'auto' if g == 0 else g}: {t[0]:.2f}/{t[1]:.2f} ({N * K * 2 / min(t) / 1e6:.2f})")
    print(f"{name:8s} " + "   ".join(row))
