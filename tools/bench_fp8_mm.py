"""torch._scaled_mm (hipBLASLt fp8) at decode batch sizes on the Llama-3-8B projection shapes:
the library baseline for an fp8 weight-streaming kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import light_vllm_amd  # noqa
from light_vllm_amd.quantization import pack_fp8_weight, skinny_fp8_linear

dev = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32
one = torch.ones(1, device=dev)
for name, N, K in [("qkv", 6144, 4096), ("o", 4096, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336),
                   ("lm_head", 128256, 4096)]:
    ws = [(torch.randn(N, K, device=dev) * 0.05).to(torch.float8_e4m3fn).t() for _ in range(6 if N < 100000 else 2)]
    x = torch.randn(M, K, device=dev).to(torch.float8_e4m3fn)
    f = lambda i: torch._scaled_mm(x, ws[i % len(ws)], out_dtype=torch.bfloat16, scale_a=one, scale_b=one)
    for i in range(5):
        f(i)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
    for i, (a, b) in enumerate(evs):
        a.record(); f(i); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    med = ts[len(ts) // 2]
    # the W8A8 weight-streaming kernel (bf16 activations quantised in the kernel)
    wps = [pack_fp8_weight(w.t().contiguous()) for w in ws]
    xb = torch.randn(M, K, device=dev).to(torch.bfloat16)
    xs = torch.tensor([0.01], device=dev)
    f2 = lambda i: skinny_fp8_linear(xb, wps[i % len(wps)], one, xs, N, K)
    for i in range(5):
        f2(i)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
    for i, (a, b) in enumerate(evs):
        a.record(); f2(i); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    med2 = ts[len(ts) // 2]
    print(f"{name:8s} N={N:6d} K={K:5d}  _scaled_mm fp8 {med:7.1f} us ({N * K / med / 1e6:.2f} TB/s)   "
          f"skinny w8a8 {med2:7.1f} us ({N * K / med2 / 1e6:.2f} TB/s of weight bytes)")
