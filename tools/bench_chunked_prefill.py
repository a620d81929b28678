"""BASELINE config 3: the reference's benchmarks/benchmark_chunked_prefill_throughput.py workload on
this engine -- prompts of `--input-len` random tokens, `--output-len` generated tokens each, chunked
prefill with max_num_batched_tokens = max_num_seqs = 64 (benchmark_chunked_prefill_throughput.py:
176-201), Llama-3-8B shapes, random weights.  Every step is a mixed batch of <= 64 tokens: decode
tokens of the running sequences plus prompt chunks that attend to their earlier chunks through the
paged cache (HIP prefill kernel + paged decode kernel in the same step).
Prints requests/s and tokens/s like the reference script."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import light_vllm_amd  # noqa
from light_vllm_amd.engine.config import CacheConfig, ModelConfig, SchedulerConfig
from light_vllm_amd.engine.llm_engine import LLMEngine


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-prompts", type=int, default=128)
    ap.add_argument("--input-len", type=int, default=512)
    ap.add_argument("--output-len", type=int, default=512)
    ap.add_argument("--budget", type=int, default=64)
    ap.add_argument("--scheduling", default="async", choices=["sync", "async"])
    ap.add_argument("--tiny", action="store_true")
    ap.add_argument("--no-qkv-reduce-in-rope", action="store_true",
                    help="A/B: the QKV projection of a mixed step keeps its own reduce launch (round 3's path)")
    ap.add_argument("--gemm-workgroups", type=int, default=None,
                    help="A/B: SchedulerConfig.gemm_workgroups (default: 128 at two steps in flight)")
    a = ap.parse_args()
    dev = "cuda:0"
    cfg = ModelConfig.tiny() if a.tiny else ModelConfig.llama3_8b()
    cfg.qkv_reduce_in_rope = not a.no_qkv_reduce_in_rope
    max_len = a.input_len + a.output_len + 16
    bs = 16
    blocks = (a.budget + 8) * ((max_len + bs - 1) // bs + 1) + 64
    eng = LLMEngine(cfg, CacheConfig(block_size=bs, num_gpu_blocks=blocks, num_cpu_blocks=0),
                    SchedulerConfig(max_num_batched_tokens=a.budget, max_num_seqs=a.budget,
                                    max_model_len=(max_len + 511) // 512 * 512, scheduling=a.scheduling,
                                    max_num_on_the_fly=2, chunked_prefill_enabled=True,
                                    gemm_workgroups=a.gemm_workgroups),
                    device=dev, use_hip_graph=True)
    eng.step_returns_outputs = False
    g = torch.Generator().manual_seed(0)
    for i in range(a.num_prompts):
        eng.add_request(str(i), torch.randint(0, cfg.vocab_size, (a.input_len,), generator=g).tolist(),
                        max_tokens=a.output_len)
    step = eng.async_step if a.scheduling == "async" else eng.step
    done = 0
    steps = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while eng.has_unfinished_requests() or eng.num_on_the_fly > 0:
        for out in step():
            done += out.finished
        steps += 1
        if steps % 2000 == 0:
            print(f"  step {steps}: {done}/{a.num_prompts} finished, {time.perf_counter() - t0:.1f} s", flush=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    eng.shutdown()
    total = a.num_prompts * (a.input_len + a.output_len)
    print(f"Throughput: {a.num_prompts / dt:.2f} requests/s, {total / dt:.2f} tokens/s "
          f"({steps} steps, {dt / steps * 1e3:.2f} ms/step, {done} finished)")


if __name__ == "__main__":
    main()
