"""Host-side cost of one decode step without a GPU: scheduler + block manager + staging arrays + output
processing for 2 x 32 running sequences of ~1k tokens (the bench.py workload), tokens faked.
Runs anywhere; the engine thread's time per step is what one stream spends alone on the GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from light_vllm_amd.engine.config import CacheConfig, SchedulerConfig
from light_vllm_amd.engine.input_builder import DecodeStepArrays
from light_vllm_amd.engine.llm_engine import LLMEngine, ExecuteOutput
from light_vllm_amd.engine.scheduler import DecodingScheduler
from light_vllm_amd.engine.sequence import Sequence, SequenceGroup, SequenceStatus

B, ctx, steps = 32, 1024, 400
cc = CacheConfig(block_size=16, num_gpu_blocks=2 * B * 80, num_cpu_blocks=0)
sc = SchedulerConfig(max_num_batched_tokens=8192, max_num_seqs=B, max_model_len=2048, scheduling="async", max_num_on_the_fly=2)
eng = object.__new__(LLMEngine)
eng.scheduler_config, eng.cache_config = sc, cc
eng.scheduler = DecodingScheduler(sc, cc)
eng.eos_token_id = None
eng.step_returns_outputs = False
eng.groups, eng.seq_to_group = {}, {}
for i in range(2 * B):
    seq = Sequence(i, list(range(ctx)), 16, None)
    g = SequenceGroup(str(i), [seq], time.time(), max_tokens=10000)
    eng.groups[str(i)] = g
    eng.scheduler.add_request(g)
# admit everything as if prefilled (what prefill_synthetic does, without the GPU part)
sched = eng.scheduler
while sched.waiting:
    g = sched.waiting.popleft()
    sched.block_manager.allocate(g)
    for s in g.seqs:
        s.status = SequenceStatus.RUNNING
        s.data.update_num_computed_tokens(ctx)
        s.append_token_id(1, 0.0)
    sched.running.append(g)
arrays = [DecodeStepArrays(B, 160, 16) for _ in range(2)]
acc = {"schedule": 0.0, "stage": 0.0, "process": 0.0}
inflight = []
t_all = time.perf_counter()
for it in range(steps):
    while len(inflight) < 2:
        t = time.perf_counter(); out = sched.schedule(); acc["schedule"] += time.perf_counter() - t
        if out is None or out.is_empty():
            break
        t = time.perf_counter(); ids = arrays[len(inflight)].fill(out.seq_group_metadata_list); acc["stage"] += time.perf_counter() - t
        inflight.append((out, ids))
    out, ids = inflight.pop(0)
    toks = torch.full((len(ids),), 7, dtype=torch.long)
    t = time.perf_counter(); eng._process(out, ExecuteOutput(toks, ids)); acc["process"] += time.perf_counter() - t
el = time.perf_counter() - t_all
print(f"{steps} steps: {el / steps * 1e6:.0f} us per step on this CPU   " + "  ".join(f"{k} {v / steps * 1e6:.0f}" for k, v in acc.items()))
