"""Workload for the HBM-traffic counters of the decode projections: each of the four Llama-3-8B
projection shapes launched 16 times at M = 32 over 8 rotating packed weight matrices.  Run under
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- python tools/pmc_gemm.py
and again with --pmc WRITE_SIZE (separate passes); tools/pmc_gemm_summary.py turns the two
counter_collection.csv files into profiles/r01_pmc_gemm.json."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import light_vllm_amd  # noqa
from light_vllm_amd import _custom_ops  # noqa

dev = "cuda:0"
M = 32
W8 = "--w8" in sys.argv  # the W8A8 launches of a decode step with activations quantised once (round 3)
for name, N, K in [("qkv", 6144, 4096), ("o", 4096, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336)]:
    nw = 8 if N * K < 1e8 else 4
    if W8:
        from light_vllm_amd.quantization import pack_fp8_weight
        ws = [pack_fp8_weight((torch.randn(N, K, device=dev) * 0.5).clamp(-448, 448).to(torch.float8_e4m3fn)) for _ in range(nw)]
        x8 = torch.randint(0, 120, (M, K), device=dev, dtype=torch.uint8)
        one = torch.ones(1, device=dev)
    else:
        ws = [torch.ops._C_amd.pack_weight((torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)) for _ in range(nw)]
        x = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    torch.cuda.synchronize()
    for i in range(16):
        if not W8:
            torch.ops._C_amd.skinny_linear_packed(x, ws[i % nw], None, N, K)
        elif name == "gate_up":
            torch.ops._C_amd.skinny_linear_w8a8_q_swiglu_fp8(x8, ws[i % nw], one, one, N, K, one, torch.bfloat16)
        elif name == "down":
            torch.ops._C_amd.skinny_linear_w8a8_q_partials(x8, ws[i % nw], one, one, N, K)
        else:
            torch.ops._C_amd.skinny_linear_w8a8_q(x8, ws[i % nw], one, one, N, K, None, torch.bfloat16)
    torch.cuda.synchronize()
    del ws
    torch.cuda.empty_cache()
