"""Host-side cost of one decode step in the async engine: wall time of schedule / input build /
launch (graph load + replay) / output processing, measured by wrapping the engine's own methods."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import light_vllm_amd  # noqa
from light_vllm_amd.engine.config import CacheConfig, ModelConfig, SchedulerConfig
from light_vllm_amd.engine.llm_engine import LLMEngine

fly = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = 100
dev = "cuda:0"
B, ctx = 32, 1024
cfg = ModelConfig.llama3_8b()
total = steps + 10
max_len = ctx + total // fly + 8
blocks = B * fly * ((max_len + 15) // 16 + 1) + 64
engine = LLMEngine(cfg, CacheConfig(block_size=16, num_gpu_blocks=blocks, num_cpu_blocks=0),
                   SchedulerConfig(max_num_batched_tokens=8192, max_num_seqs=B, max_model_len=(max_len + 511) // 512 * 512,
                                   scheduling="async", max_num_on_the_fly=fly), device=dev)
engine.step_returns_outputs = False
g = torch.Generator().manual_seed(1234)
for i in range(B * fly):
    engine.add_request(str(i), torch.randint(0, cfg.vocab_size, (ctx,), generator=g).tolist(), max_tokens=total + 100)
engine.prefill_synthetic(seed=0)

acc = {}
def wrap(obj, name, label):
    f = getattr(obj, name)
    def w(*a, **k):
        t = time.perf_counter()
        r = f(*a, **k)
        acc.setdefault(label, []).append(time.perf_counter() - t)
        return r
    setattr(obj, name, w)
from light_vllm_amd.engine.graph_runner import DecodeGraph
from light_vllm_amd.engine.input_builder import DecodeStepArrays
for cls, name, label in ((DecodeStepArrays, "fill", "staging fill"), (DecodeGraph, "load_staged", "staged H2D copy"),
                         (DecodeGraph, "replay", "graph replay (launch)")):
    def mk(f, label):
        def w(self, *a, **k):
            t = time.perf_counter()
            r = f(self, *a, **k)
            acc.setdefault(label, []).append(time.perf_counter() - t)
            return r
        return w
    setattr(cls, name, mk(getattr(cls, name), label))
wrap(engine.scheduler, "schedule", "schedule")
wrap(engine, "_process", "process")
wrap(engine.worker, "execute", "launch (load + replay + d2h enqueue)")
ib = engine.input_builder
class IB:
    def __call__(self, s):
        t = time.perf_counter(); r = ib(s); acc.setdefault("input build", []).append(time.perf_counter() - t); return r
engine.input_builder = IB()
oget = engine.executor_out.get
def get(*a, **k):
    t = time.perf_counter(); r = oget(*a, **k); acc.setdefault("wait for a result", []).append(time.perf_counter() - t); return r
engine.executor_out.get = get

step = lambda i, n: engine.async_step(schedule_more=i < n - (fly - 1))
for i in range(10):
    step(i, 10)
torch.cuda.synchronize()
for v in acc.values():
    v.clear()
t0 = time.perf_counter()
for i in range(steps):
    step(i, steps)
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(f"{fly} in flight: {el / steps * 1e3:.3f} ms/step")
for k, v in acc.items():
    v = sorted(v)
    print(f"  {k:40s} n={len(v):4d}  mean {sum(v) / len(v) * 1e6:7.1f} us  med {v[len(v) // 2] * 1e6:7.1f}  max {v[-1] * 1e6:7.1f}")

if os.environ.get("LVLLM_HOST_PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile()
    n = 60
    pr.enable()
    for i in range(n):
        step(i, n)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(45)
