"""Phase timestamps of the weight-streaming GEMM (diagnosis builds made by
tools/build_trace_variants.sh; the variant's liblvllm_hip.so is copied over lib/ on the GPU box).
Per wave: entry, first operands arrived, stream done, stores done — 100 MHz clock."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import light_vllm_amd  # noqa
from light_vllm_amd import _custom_ops  # noqa

lib = ctypes.CDLL(os.path.join(os.path.dirname(light_vllm_amd.__file__), "lib", "liblvllm_hip.so"))
dev = "cuda:0"
M = 32
shapes = [("qkv", 6144, 4096), ("o", 4096, 4096), ("gate_up", 28672, 4096)]
for name, N, K in shapes:
    nw = 6
    ws = [torch.ops._C_amd.pack_weight((torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)) for _ in range(nw)]
    x = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    for i in range(8):
        torch.ops._C_amd.skinny_linear_packed(x, ws[i % nw], None, N, K)
    torch.cuda.synchronize()
    spans = []
    for rep in range(6):
        # back to back with a predecessor, as inside a step
        torch.ops._C_amd.skinny_linear_packed(x, ws[(rep + 1) % nw], None, N, K)
        torch.ops._C_amd.skinny_linear_packed(x, ws[rep % nw], None, N, K)
        torch.cuda.synchronize()
        buf = np.zeros(8 * 4096, dtype=np.uint64)
        assert lib.lvllm_gemm_trace_read(buf.ctypes.data_as(ctypes.c_void_p), buf.size) == 0
        t = buf.reshape(1024, 8, 4)[:256].astype(np.int64)
        t = (t - t[:, :, 0].min()) / 100.0  # us
        spans.append(t[:, :, 3].max())
    q = lambda v: "min %5.2f  med %5.2f  max %5.2f" % (v.min(), np.median(v), v.max())
    print(f"== {name}: N={N} K={K}   span of the last 6 launches: " + " ".join("%.1f" % s for s in spans))
    print("  wave entry                         ", q(t[:, :, 0]))
    print("  last wave of a workgroup enters    ", q(t[:, :, 0].max(1) - t[:, :, 0].min(1)), "(after its first)")
    print("  operands arrived - entry (wave)    ", q(t[:, :, 1] - t[:, :, 0]))
    print("  operands arrived (absolute, wave)  ", q(t[:, :, 1]))
    print("  stream done - operands arrived     ", q(t[:, :, 2] - t[:, :, 1]))
    print("  stores done - stream done          ", q(t[:, :, 3] - t[:, :, 2]))
    print("  exit                               ", q(t[:, :, 3]), flush=True)
