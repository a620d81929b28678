"""The device-side sampler launch (csrc/sampler.hip) at the model's vocabulary: 32 rows x 128 256 bf16 logits, every row
with a state slot, by sampling configuration; HIP events around trains of launches.  Beside it the torch statement of
the same stages (light_vllm_amd.sampling: what round 2's engine ran per sampled step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import light_vllm_amd  # noqa
from light_vllm_amd import sampling
from light_vllm_amd.device_sampler import DeviceSampler
from light_vllm_amd.engine.sampling_params import SamplingParams

dev = "cuda:0"
B, V = 32, 128256
g = torch.Generator(device=dev).manual_seed(0)
logits = (torch.randn(B, V, generator=g, device=dev) * 3).to(torch.bfloat16)
cases = {"greedy rows (no state)": None,
         "greedy rows with a state slot (first pass only)": dict(temperature=0.0),
         "greedy + penalties": dict(temperature=0.0, repetition_penalty=1.2, frequency_penalty=0.3),
         "temperature only": dict(temperature=0.8),
         "top-k 50": dict(temperature=0.8, top_k=50),
         "top-p 0.9": dict(temperature=0.8, top_p=0.9),
         "top-k 50 + top-p 0.9 + min-p 0.05": dict(temperature=0.8, top_k=50, top_p=0.9, min_p=0.05),
         "penalties + top-k + top-p": dict(temperature=0.8, top_k=50, top_p=0.9, repetition_penalty=1.2, frequency_penalty=0.3)}
for name, kw in cases.items():
    ds = DeviceSampler(V, dev, num_slots=B, seed=0)
    if kw is None:
        slots = torch.full((B,), -1, dtype=torch.int32, device=dev)
    else:
        slots = torch.tensor([ds.ensure(i, SamplingParams(seed=i, **kw), list(range(100)), list(range(50)), None) for i in range(B)],
                             dtype=torch.int32, device=dev)
    scratch = ds.new_scratch(B)
    out = torch.empty(B, dtype=torch.long, device=dev)
    for _ in range(5):
        ds.sample(logits, slots, tokens_out=out, scratch=scratch, update_state=False)
    torch.cuda.synchronize()
    # a captured train of 20 launches, replayed 5 times: the launch's own duration, not the host's dispatch cost
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            for _ in range(20):
                ds.sample(logits, slots, tokens_out=out, scratch=scratch, update_state=False)
    graph.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        graph.replay()
    b.record()
    torch.cuda.synchronize()
    t_kernel = a.elapsed_time(b) / 100 * 1e3
    t_torch = float("nan")
    if kw is not None:
        x = logits.float()
        T = torch.full((B,), kw["temperature"] or 1.0, device=dev)
        tk = torch.full((B,), kw.get("top_k", V), device=dev, dtype=torch.long)
        tp = torch.full((B,), kw.get("top_p", 1.0), device=dev)
        mp = torch.full((B,), kw.get("min_p", 0.0), device=dev) if "min_p" in kw else None
        fn = lambda: sampling.sample(x, T, tp if ("top_p" in kw or "top_k" in kw) else None, tk if ("top_p" in kw or "top_k" in kw) else None, mp)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a.record()
        for _ in range(10):
            fn()
        b.record()
        torch.cuda.synchronize()
        t_torch = a.elapsed_time(b) / 10 * 1e3
    print(f"{name:40s} kernel {t_kernel:8.1f} us   torch statement {t_torch:9.1f} us")
