"""Timeline of decode steps across streams from the instrumented build (tools/build_trace.sh):
runs the engine as bench.py does, then dumps the per-workgroup (start, end, kernel, grid, block)
records of the GEMM / attention / norm / rope kernels to gpurun_out/trace_step.npz."""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--on-the-fly", type=int, default=2)
ap.add_argument("--steps", type=int, default=24)
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "trace_step.npz"))
a = ap.parse_args()

import light_vllm_amd  # noqa
from light_vllm_amd.engine.config import CacheConfig, ModelConfig, SchedulerConfig
from light_vllm_amd.engine.llm_engine import LLMEngine

dev = "cuda:0"
B, ctx, fly = 32, 1024, a.on_the_fly
cfg = ModelConfig.llama3_8b()
total = a.steps + 10
max_len = ctx + total // fly + 8
blocks = B * fly * ((max_len + 15) // 16 + 1) + 64
engine = LLMEngine(cfg, CacheConfig(block_size=16, num_gpu_blocks=blocks, num_cpu_blocks=0),
                   SchedulerConfig(max_num_batched_tokens=8192, max_num_seqs=B, max_model_len=(max_len + 511) // 512 * 512,
                                   scheduling="async" if fly > 1 else "sync", max_num_on_the_fly=fly), device=dev)
engine.step_returns_outputs = False
g = torch.Generator().manual_seed(1234)
for i in range(B * fly):
    engine.add_request(str(i), torch.randint(0, cfg.vocab_size, (ctx,), generator=g).tolist(), max_tokens=total + 100)
engine.prefill_synthetic(seed=0)
step = (lambda i, n: engine.async_step(schedule_more=i < n - (fly - 1))) if fly > 1 else (lambda i, n: engine.step())
for i in range(10):
    step(i, 10)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for i in range(a.steps):
    step(i, a.steps)
torch.cuda.synchronize()
print(f"{a.steps} steps, {(time.perf_counter() - t0) / a.steps * 1e3:.3f} ms/step (instrumented build)")

lib = ctypes.CDLL(os.path.join(os.path.dirname(light_vllm_amd.__file__), "lib", "liblvllm_hip.so"))
out = {}
NREC = 1 << 20
for name in ("gemm", "attn", "norm", "rope"):
    buf = np.zeros(3 * NREC, dtype=np.uint64)
    head = ctypes.c_uint(0)
    rc = getattr(lib, "lvllm_trace_read_" + name)(buf.ctypes.data_as(ctypes.c_void_p), ctypes.byref(head))
    assert rc == 0, (name, rc)
    n = min(head.value, NREC)
    rec = buf.reshape(NREC, 3)
    rec = rec[:n] if head.value <= NREC else np.roll(rec, -(head.value % NREC), axis=0)
    out[name] = rec[-min(n, 400000):].copy()
    print(name, head.value, "records")
np.savez_compressed(a.out, **out)
