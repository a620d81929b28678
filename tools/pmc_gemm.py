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
for name, N, K in [("qkv", 6144, 4096), ("o", 4096, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336)]:
    nw = 8 if N * K < 1e8 else 4
    ws = [torch.ops._C_amd.pack_weight((torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)) for _ in range(nw)]
    x = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    torch.cuda.synchronize()
    for i in range(16):
        torch.ops._C_amd.skinny_linear_packed(x, ws[i % nw], None, N, K)
    torch.cuda.synchronize()
    del ws
    torch.cuda.empty_cache()
