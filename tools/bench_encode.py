"""BASELINE config 4: bge-m3-shaped (XLM-RoBERTa-large: 24 layers, hidden 1024, 16 heads of 64)
encode-only throughput through the prefill-only engine: random prompts of `--len` tokens, dense (CLS)
embeddings out.  Prints sequences/s and tokens/s for the sync and the async (two steps in flight)
engine."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import light_vllm_amd  # noqa
from light_vllm_amd.prefill_only import PrefillOnlySchedulerConfig
from light_vllm_amd.prefill_only.engine import PrefillOnlyEngine
from light_vllm_amd.prefill_only.model import EncoderConfig


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-prompts", type=int, default=512)
    ap.add_argument("--len", type=int, default=512)
    ap.add_argument("--max-num-seqs", type=int, default=32)
    ap.add_argument("--tiny", action="store_true")
    ap.add_argument("--mlp-block", type=int, default=0, help="EncoderConfig.mlp_block_tokens (A/B)")
    ap.add_argument("--gpus", type=int, default=0, help="N > 0: N worker processes behind one scheduler (queue-sharing DP)")
    ap.add_argument("--single-device", action="store_true", help="with --gpus: every worker on cuda:0")
    ap.add_argument("--ragged", action="store_true", help="prompt lengths uniform in [len/8, len] instead of all equal")
    a = ap.parse_args()
    cfg = EncoderConfig.tiny() if a.tiny else EncoderConfig.bge_m3()
    cfg.mlp_block_tokens = a.mlp_block
    g = torch.Generator().manual_seed(0)
    lens = [int(torch.randint(max(1, a.len // 8), a.len + 1, (1,), generator=g)) if a.ragged else a.len
            for _ in range(a.num_prompts)]
    prompts = [torch.randint(2, cfg.vocab_size, (n,), generator=g).tolist() for n in lens]
    ntok = sum(lens)
    if a.gpus > 0:
        # the front end touches no GPU: the workers are spawned first and own the devices
        from light_vllm_amd.prefill_only.dp_executor import DataParallelEncodeEngine
        eng = DataParallelEncodeEngine(cfg, PrefillOnlySchedulerConfig(max_model_len=max(a.len, 8), max_num_seqs=a.max_num_seqs,
                                                                       scheduling="async"),
                                       data_parallel_size=a.gpus, devices=[0] * a.gpus if a.single_device else None)
        eng.encode(prompts[: 2 * a.max_num_seqs * a.gpus])  # warm up every worker
        eng.steps_by_rank.clear()
        t0 = time.perf_counter()
        res = eng.encode(prompts)
        dt = time.perf_counter() - t0
        eng.shutdown()
        assert len(res) == a.num_prompts
        print(f"dp{a.gpus}{' (one device)' if a.single_device else ''}: {a.num_prompts / dt:8.1f} sequences/s  "
              f"{ntok / dt:10.0f} tokens/s  ({a.num_prompts} prompts, {ntok} tokens, {a.max_num_seqs} per step, "
              f"{dt * 1e3:.0f} ms; steps per worker {dict(sorted(eng.steps_by_rank.items()))})")
        return
    for sched in ("sync", "async"):
        eng = PrefillOnlyEngine(cfg, PrefillOnlySchedulerConfig(max_model_len=max(a.len, 8), max_num_seqs=a.max_num_seqs,
                                                                scheduling=sched), device="cuda:0")
        eng.encode(prompts[: 2 * a.max_num_seqs])  # warm up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = eng.encode(prompts)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        eng.shutdown()
        assert len(res) == a.num_prompts
        print(f"{sched:5s}: {a.num_prompts / dt:8.1f} sequences/s  {ntok / dt:10.0f} tokens/s  "
              f"({a.num_prompts} x {a.len} tokens, {a.max_num_seqs} per step, {dt * 1e3:.0f} ms)")


if __name__ == "__main__":
    main()
