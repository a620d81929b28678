"""KV-cache sizing pinned to the reference.  tests/golden/kv_sizing.json holds the outputs of the
reference's own `Worker.determine_num_available_blocks` (decoding/worker/gpu_worker.py:95-144) on scripted
memory readings, `CacheEngine.get_cache_block_size` (cache_engine.py:85-103) and the prompt lengths of
`GPUModelRunner.profile_run` (runner/model_runner.py:111-145); generator: oracle/make_golden.py kv_sizing.
Integer results: exact."""
import json
import os

import pytest
import torch

from light_vllm_amd.engine.cache_engine import CacheEngine
from light_vllm_amd.engine.config import CacheConfig, ModelConfig, SchedulerConfig
from light_vllm_amd.engine.llm_engine import Worker

with open(os.path.join(os.path.dirname(__file__), "golden", "kv_sizing.json")) as f:
    GOLD = json.load(f)


def test_block_bytes_and_block_counts_equal_the_reference():
    for c in GOLD["cases"]:
        mc = ModelConfig(hidden_size=c["head_size"] * 4, num_attention_heads=4, num_key_value_heads=c["num_kv_heads"],
                         num_hidden_layers=c["num_layers"], dtype=getattr(torch, c["model_dtype"]))
        assert mc.head_dim == c["head_size"]
        cc = CacheConfig(block_size=c["block_size"], cache_dtype=c["cache_dtype"],
                         gpu_memory_utilization=c["gpu_memory_utilization"], swap_space_bytes=c["swap_space_bytes"])
        block_bytes = CacheEngine.get_cache_block_size(cc, mc)
        assert block_bytes == c["block_bytes"]
        got = Worker.kv_blocks_from_profile(c["total"], c["init_free"], c["free_after_load"], c["free_after_profile"],
                                            c["gpu_memory_utilization"], c["scheduling"], block_bytes,
                                            c["swap_space_bytes"])
        assert got == (c["num_gpu_blocks"], c["num_cpu_blocks"]), c


def test_profile_prompt_lengths_equal_the_reference():
    for p in GOLD["profile_runs"]:
        assert Worker.profile_seq_lens(p["max_num_batched_tokens"], p["max_num_seqs"]) == p["seq_lens"]


def test_graph_reserve_only_shrinks_the_cache():
    c = GOLD["cases"][0]
    a = Worker.kv_blocks_from_profile(c["total"], c["init_free"], c["free_after_load"], c["free_after_profile"],
                                      c["gpu_memory_utilization"], c["scheduling"], c["block_bytes"],
                                      c["swap_space_bytes"])
    b = Worker.kv_blocks_from_profile(c["total"], c["init_free"], c["free_after_load"], c["free_after_profile"],
                                      c["gpu_memory_utilization"], c["scheduling"], c["block_bytes"],
                                      c["swap_space_bytes"], reserve_bytes=10 * c["block_bytes"])
    assert b[0] == max(a[0] - 10, 0) and b[1] == a[1]


@pytest.mark.gpu
def test_profile_run_sizes_the_cache_on_the_gpu():
    """The profile forward really runs (activations show up in the memory readings), the async modes count
    them twice, and an engine sized this way decodes."""
    from light_vllm_amd.attention.backend import PagedAttnBackend
    from light_vllm_amd.engine.llm_engine import LLMEngine
    mc = ModelConfig.tiny()
    cc = CacheConfig(block_size=16, gpu_memory_utilization=0.05, swap_space_bytes=1 << 20)
    sc = SchedulerConfig(max_num_batched_tokens=2048, max_num_seqs=16, max_model_len=2048)
    w = Worker(mc, cc, PagedAttnBackend(), "cuda:0", use_hip_graph=False)
    assert w.init_gpu_memory > 0
    n_sync, n_cpu = w.determine_num_available_blocks(sc)
    prof = w.profile
    runtime = prof["free_after_load"] - prof["free_after_profile"]
    assert runtime > 2048 * mc.hidden_size * 2, "the profile forward left no trace in the memory readings"
    assert prof["init_free"] - prof["free_after_load"] > 0  # the weights
    assert n_cpu == (1 << 20) // prof["block_bytes"]
    sc_async = SchedulerConfig(max_num_batched_tokens=2048, max_num_seqs=16, max_model_len=2048, scheduling="async")
    # (eager worker: no graph reserve; the device sampler's state and working rows are reserved -- it is created with
    # the first request that is not plain greedy, after the cache has been sized: ADVICE r03)
    reserve = w.sampler_reserve_bytes(sc)
    assert reserve >= w.sampler_state_slots(sc) * mc.vocab_size * 4
    exp_sync = Worker.kv_blocks_from_profile(prof["total"], prof["init_free"], prof["free_after_load"],
                                             prof["free_after_profile"], 0.05, "sync", prof["block_bytes"], 1 << 20,
                                             reserve)
    exp_async = Worker.kv_blocks_from_profile(prof["total"], prof["init_free"], prof["free_after_load"],
                                              prof["free_after_profile"], 0.05, "async", prof["block_bytes"], 1 << 20,
                                              reserve)
    assert n_sync == exp_sync[0]
    assert exp_sync[0] - exp_async[0] in (runtime // prof["block_bytes"], runtime // prof["block_bytes"] + 1)
    del w
    torch.cuda.empty_cache()
    eng = LLMEngine(mc, CacheConfig(block_size=16, gpu_memory_utilization=0.05, swap_space_bytes=1 << 20), sc_async)
    assert eng.cache_config.num_gpu_blocks > 100
    eng.add_request("a", list(range(1, 40)), max_tokens=8)
    outs = {}
    while eng.has_unfinished_requests():
        for r in eng.async_step():
            outs[r.request_id] = r
    assert outs["a"].finished and len(outs["a"].token_ids) == 8


@pytest.mark.gpu
def test_rocm_platform_answers_on_the_gpu():
    """the capability the reference indexes (`[0] >= 8` for bf16, gpu_worker.py:227-238) exists."""
    from light_vllm_amd.platforms import RocmPlatform
    cap = RocmPlatform.get_device_capability()
    assert cap[0] >= 8 and cap >= (9, 0) and RocmPlatform.has_device_capability(80)
    assert RocmPlatform.has_device_capability((8, 0)) and not RocmPlatform.has_device_capability((99, 0))
    assert isinstance(RocmPlatform.get_device_name(), str) and RocmPlatform.get_device_name()
    assert RocmPlatform.get_device_total_memory() > 100 << 30


def test_block_padding_is_sized_and_validated():
    """CacheConfig.block_pad_bytes: None = 1/32 of a block's bytes in one plane (multiples of 256), 0 = the reference's
    dense layout; the worker sizes the cache with the padded footprint while get_cache_block_size stays the reference's
    figure (the golden above)."""
    mc = ModelConfig(hidden_size=4096, num_attention_heads=32, num_key_value_heads=8, num_hidden_layers=32,
                     dtype=torch.bfloat16)
    dense = CacheEngine.get_cache_block_size(CacheConfig(block_size=16, block_pad_bytes=0), mc)
    assert dense == 2 * 32 * 16 * 8 * 128 * 2
    assert CacheEngine.get_cache_block_footprint(CacheConfig(block_size=16, block_pad_bytes=0), mc) == dense
    auto = CacheConfig(block_size=16)
    assert CacheEngine.block_pad_bytes(auto, mc) == 1024  # 32 KiB blocks
    assert CacheEngine.get_cache_block_size(auto, mc) == dense
    assert CacheEngine.get_cache_block_footprint(auto, mc) == dense + 2 * 32 * 1024
    assert CacheEngine.block_pad_bytes(CacheConfig(block_size=16, cache_dtype="fp8"), mc) == 512  # 16 KiB blocks
    assert CacheEngine.block_pad_bytes(CacheConfig(block_size=16, block_pad_bytes=4096), mc) == 4096
    with pytest.raises(ValueError):
        CacheEngine.block_pad_bytes(CacheConfig(block_size=16, block_pad_bytes=100), mc)
