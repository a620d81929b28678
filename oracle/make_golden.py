"""Regenerates the golden fixtures under tests/golden/ from the REAL reference.
TEST INFRASTRUCTURE ONLY; runs in the dev container only (needs /root/reference and
oracle/_ref).  The fixtures are data: inputs and the reference's outputs.

  block_manager_<name>.json   traces of the reference's BlockSpaceManagerV1/V2 under the seeded
                              programs of tests/bm_driver.py (every block table, verdict, CoW and
                              swap pair after every operation)
  ops_<name>.npz              seeded inputs and outputs of the reference's csrc/cpu operators
                              (paged_attention_v1/v2, reshape_and_cache, copy_blocks, rms_norm,
                              fused_add_rms_norm, rotary_embedding, silu_and_mul)

  input_builder.json          per-step input arrays (token ids, positions, slot mapping, block tables,
                              sequence lengths, start offsets) the reference's ModelInputForGPUBuilder +
                              flash-attn metadata builder produce for the seeded scenarios of
                              tests/ib_driver.py

  kv_sizing.json              num_gpu_blocks / num_cpu_blocks of the reference's
                              Worker.determine_num_available_blocks on scripted memory readings, block
                              bytes of CacheEngine.get_cache_block_size, prompt lengths of profile_run

  scheduler_<name>.json       what the reference's DecodingScheduler returns step by step for the seeded programs of
                              tests/sched_driver.py

usage: python oracle/make_golden.py [block_manager] [ops] [prefill_only] [sampler] [input_builder] [kv_sizing] [scheduler]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _recording_free_hook(bm, op_idx, _recorded):
    """Logs the order in which the reference releases blocks during one operation."""
    log = []
    pools = []
    for dev, name in ((0, "gpu_allocator"), (1, "cpu_allocator")):
        alloc = getattr(bm, name, None)
        if alloc is None:
            continue
        orig = alloc.free

        def wrapped(block, _orig=orig, _dev=dev):
            log.append([_dev, block.block_number])
            return _orig(block)

        alloc.free = wrapped
        pools.append((alloc, orig))

    def finish():
        for alloc, orig in pools:
            alloc.free = orig
        return log

    return finish


def make_block_manager_traces():
    import bm_driver
    from oracle import ref_block_manager
    ns = ref_block_manager.load()
    adapter = bm_driver.ReferenceAdapter(ns)

    def make_manager(cfg):
        cls = ns.BlockSpaceManagerV1 if cfg["version"] == "v1" else ns.BlockSpaceManagerV2
        return cls(block_size=cfg["block_size"], num_gpu_blocks=cfg["num_gpu_blocks"],
                   num_cpu_blocks=cfg["num_cpu_blocks"], watermark=cfg["watermark"],
                   sliding_window=cfg["sliding_window"], enable_caching=cfg["enable_caching"])

    only = os.environ.get("GOLDEN_ONLY")  # e.g. GOLDEN_ONLY=lookahead: regenerate just the matching traces
    for name, cfg, seed, num_ops in bm_driver.DEFAULT_CONFIGS + getattr(bm_driver, "V2_CONFIGS", []):
        if only and only not in name:
            continue
        hook = _recording_free_hook if cfg["version"] == "v1" else None
        trace = bm_driver.run_program(make_manager, adapter, cfg, seed, num_ops, free_hook=hook)
        for op in trace:  # only table frees are order-sensitive; drop the rest of the logs
            if op.get("op") != "free":
                op.pop("free_order", None)
        path = os.path.join(GOLDEN, f"block_manager_{name}.json")
        with open(path, "w") as f:
            json.dump({"config": cfg, "seed": seed, "num_ops": num_ops, "trace": trace}, f,
                      separators=(",", ":"))
        kinds = {}
        for op in trace:
            kinds[op["op"]] = kinds.get(op["op"], 0) + 1
        print(f"{path}: {len(trace)} ops {kinds} ({os.path.getsize(path) / 1024:.0f} KiB)")


def make_op_vectors():
    import numpy as np
    import torch
    from helpers import make_paged_inputs, v2_scratch
    from oracle import ref
    assert ref.load(), "oracle/_ref/_ref_C.so missing: run oracle/build_oracle.py"
    R, RC = torch.ops._ref_C, torch.ops._ref_C_cache_ops

    def npy(t):  # bf16 travels as uint16 bit patterns
        if t.dtype == torch.bfloat16:
            return t.contiguous().view(torch.int16).numpy().view(np.uint16)
        return t.contiguous().numpy()

    # ---- attention: bf16 and fp32, ragged lengths, GQA; block_size 16 (csrc/cpu limit) ----
    for name, dtype, (S, H, KVH, D, lens) in [
        ("attn_bf16_gqa4_d128", torch.bfloat16, (4, 4, 1, 128, [530, 100, 17, 1])),
        ("attn_f32_gqa2_d64", torch.float32, (3, 4, 2, 64, [520, 33, 16])),
        ("attn_bf16_mha_d80", torch.bfloat16, (3, 2, 2, 80, [300, 47, 16])),
    ]:
        inp = make_paged_inputs(S, H, KVH, D, 16, lens, dtype=dtype, seed=len(name))
        q = inp["query"]
        o1 = torch.zeros_like(q)
        R.paged_attention_v1(o1, q, inp["key_cache"], inp["value_cache"], KVH, inp["scale"], inp["block_tables"],
                             inp["seq_lens"], 16, inp["max_seq_len"], None, "auto", 1.0, 1.0, 0, 0, 0, 64, 0)
        es, ml, tmp = v2_scratch(S, H, D, inp["max_seq_len"], dtype)
        o2 = torch.zeros_like(q)
        R.paged_attention_v2(o2, es, ml, tmp, q, inp["key_cache"], inp["value_cache"], KVH, inp["scale"],
                             inp["block_tables"], inp["seq_lens"], 16, inp["max_seq_len"], None, "auto", 1.0, 1.0,
                             0, 0, 0, 64, 0)
        np.savez_compressed(os.path.join(GOLDEN, f"ops_{name}.npz"), query=npy(q), key_cache=npy(inp["key_cache"]),
                            value_cache=npy(inp["value_cache"]), block_tables=inp["block_tables"].numpy(),
                            seq_lens=inp["seq_lens"].numpy(), scale=np.float32(inp["scale"]),
                            num_kv_heads=np.int32(KVH), out_v1=npy(o1), out_v2=npy(o2),
                            dtype=str(dtype).split(".")[-1])
        print(f"ops_{name}.npz")

    # ---- reshape_and_cache + copy_blocks (bit-exact) ----
    g = torch.Generator().manual_seed(7)
    T, KVH, D, BS, NB = 37, 2, 128, 16, 9
    key = torch.randn(T, KVH, D, generator=g).to(torch.bfloat16)
    value = torch.randn(T, KVH, D, generator=g).to(torch.bfloat16)
    slots = torch.randperm(NB * BS, generator=g)[:T].to(torch.int64)
    slots[5] = -1
    kc = torch.randn(NB, KVH, D // 8, BS, 8, generator=g).to(torch.bfloat16)
    vc = torch.randn(NB, KVH, D, BS, generator=g).to(torch.bfloat16)
    kc0, vc0 = kc.clone(), vc.clone()
    RC.reshape_and_cache(key, value, kc, vc, slots, "auto", 1.0, 1.0)
    mapping = torch.tensor([[0, 7], [3, 8], [7, 2]], dtype=torch.int64)
    kc2, vc2 = kc.clone(), vc.clone()
    RC.copy_blocks([kc2], [vc2], mapping)
    np.savez_compressed(os.path.join(GOLDEN, "ops_cache_bf16.npz"), key=npy(key), value=npy(value), slots=slots.numpy(),
                        key_cache_in=npy(kc0), value_cache_in=npy(vc0), key_cache_out=npy(kc), value_cache_out=npy(vc),
                        copy_mapping=mapping.numpy(), key_cache_copied=npy(kc2), value_cache_copied=npy(vc2))
    print("ops_cache_bf16.npz")

    # ---- rms_norm / fused_add_rms_norm / rotary / silu (bf16 + fp32) ----
    out = {}
    for tag, dtype in (("bf16", torch.bfloat16), ("f32", torch.float32)):
        g = torch.Generator().manual_seed(11)
        x = torch.randn(6, 512, generator=g).to(dtype)
        w = (1 + 0.1 * torch.randn(512, generator=g)).to(dtype)
        res = torch.randn(6, 512, generator=g).to(dtype)
        o = torch.empty_like(x)
        R.rms_norm(o, x, w, 1e-6)
        x2, r2 = x.clone(), res.clone()
        R.fused_add_rms_norm(x2, r2, w, 1e-6)
        out.update({f"norm_x_{tag}": npy(x), f"norm_w_{tag}": npy(w), f"norm_res_{tag}": npy(res),
                    f"rms_out_{tag}": npy(o), f"fused_out_{tag}": npy(x2), f"fused_res_{tag}": npy(r2)})
        H, KVH, D, rot = 4, 2, 64, 64
        inv = 1.0 / (10000 ** (torch.arange(0, rot, 2).float() / rot))
        fr = torch.outer(torch.arange(256).float(), inv)
        cache = torch.cat([fr.cos(), fr.sin()], -1).to(dtype)
        pos = torch.randint(0, 256, (6,), generator=g, dtype=torch.int64)
        qk = torch.randn(6, (H + KVH) * D, generator=g).to(dtype)
        for neox in (True, False):
            t = qk.clone()
            R.rotary_embedding(pos, t[:, :H * D], t[:, H * D:], D, cache, neox)
            out[f"rope_out_{'neox' if neox else 'gptj'}_{tag}"] = npy(t)
        out.update({f"rope_in_{tag}": npy(qk), f"rope_cache_{tag}": npy(cache), "rope_pos": pos.numpy()})
        gu = (torch.randn(6, 2 * 320, generator=g) * 2).to(dtype)
        so = torch.empty(6, 320, dtype=dtype)
        R.silu_and_mul(so, gu)
        out.update({f"silu_in_{tag}": npy(gu), f"silu_out_{tag}": npy(so)})
    np.savez_compressed(os.path.join(GOLDEN, "ops_elementwise.npz"), **out)
    print("ops_elementwise.npz")


def make_prefill_only_vectors():
    """Golden vectors of the prefill-only (no KV cache) attention path, produced by the reference's
    own in-tree backend PrefillOnlyTorchNaiveBackendImpl.forward (prefill_only/backends/attention/
    backends/torch_naive.py:64-124) on the shapes of its test
    (tests/prefill_only/attention/test_basic_correctness.py:25-57): SEQ_LENS primes, D = 64,
    H in {8, 16}, KVH in {1, 2, 4, 8}, DECODER (causal) and ENCODER.  Inputs are torch.rand values
    rounded to bf16 (exactly representable in every dtype under test); the reference runs in fp32."""
    import numpy as np
    import torch
    from oracle import ref_block_manager
    ref_block_manager.load()  # registers the light_vllm namespace + stubs
    from light_vllm.backends.attention.abstract import AttentionType
    from light_vllm.prefill_only.backends.attention.backends.abstract import \
        PrefillOnlyAttentionMetadata
    from light_vllm.prefill_only.backends.attention.backends.torch_naive import \
        PrefillOnlyTorchNaiveBackendImpl
    SEQ_LENS = [1, 2, 3, 5, 7, 11, 13, 17, 19, 23, 29]
    out = {}
    cases = []
    g = torch.Generator().manual_seed(2024)
    for H, KVH, n_seqs in [(8, 1, 8), (8, 2, 7), (8, 8, 3), (16, 4, 10), (16, 8, 5)]:
        D = 64
        seq_lens = SEQ_LENS[:n_seqs]
        T = sum(seq_lens)
        q = torch.rand(T, H * D, generator=g).to(torch.bfloat16).float()
        k = torch.rand(T, KVH * D, generator=g).to(torch.bfloat16).float()
        v = torch.rand(T, KVH * D, generator=g).to(torch.bfloat16).float()
        impl = PrefillOnlyTorchNaiveBackendImpl(H, D, D ** -0.5, KVH, None, None, "auto")
        # (the reference's metadata builder pins memory, which needs a GPU; forward reads seq_lens only)
        md = PrefillOnlyAttentionMetadata(max_seq_len=max(seq_lens), seq_lens=seq_lens, seq_start_loc=None)
        tag = f"h{H}_kvh{KVH}_n{n_seqs}"
        bits = lambda t: t.to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)  # exact: values are bf16
        out[f"{tag}_q"], out[f"{tag}_k"], out[f"{tag}_v"] = bits(q), bits(k), bits(v)
        out[f"{tag}_seq_lens"] = np.array(seq_lens, dtype=np.int32)
        for name, at in (("decoder", AttentionType.DECODER), ("encoder", AttentionType.ENCODER)):
            o = impl.forward(q, k, v, None, md, attn_type=at)
            out[f"{tag}_{name}"] = o.numpy()
        cases.append(tag)
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(GOLDEN, "prefill_only_attn.npz"), **out)
    print("prefill_only_attn.npz", len(cases), "cases")


def make_sampler_vectors():
    """Golden outputs of the reference's sampler front half (decoding/backends/sampler.py:
    _apply_penalties :281-301, _apply_top_k_top_p :304-330, _apply_min_p :333-347) on random logits
    and token histories (fp32, CPU)."""
    import numpy as np
    import torch
    from oracle import ref_block_manager
    ref_block_manager.load()
    from light_vllm.decoding.backends import sampler as S
    g = torch.Generator().manual_seed(77)
    N, V = 6, 97
    logits = torch.randn(N, V, generator=g) * 3
    prompt = torch.randint(0, V, (N, 12), generator=g)
    output = torch.randint(0, V, (N, 9), generator=g)
    prompt[0, 8:] = V  # padding id
    output[1, 3:] = V
    pres = torch.tensor([0.0, 0.5, 1.0, 0.0, 0.3, 2.0])
    freq = torch.tensor([0.0, 0.2, 0.0, 0.7, 0.1, 1.0])
    rep = torch.tensor([1.0, 1.3, 1.0, 2.0, 0.8, 1.1])
    out = {"logits": logits.numpy(), "prompt": prompt.numpy(), "output": output.numpy(), "pres": pres.numpy(),
           "freq": freq.numpy(), "rep": rep.numpy()}
    out["penalised"] = S._apply_penalties(logits.clone(), prompt, output, pres.clone(), freq.clone(), rep.clone()).numpy()
    p = torch.tensor([1.0, 0.9, 0.5, 0.1, 0.99, 0.3])
    k = torch.tensor([V, 10, V, 3, 1, 50])
    out["top_p"], out["top_k"] = p.numpy(), k.numpy()
    out["filtered"] = S._apply_top_k_top_p(logits.clone(), p, k).numpy()
    mp = torch.tensor([0.0, 0.05, 0.2, 0.5, 0.01, 0.9])
    out["min_p"] = mp.numpy()
    out["min_p_out"] = S._apply_min_p(logits.clone(), mp.clone()).numpy()
    np.savez_compressed(os.path.join(GOLDEN, "sampler_front_half.npz"), **out)
    print("sampler_front_half.npz")


def make_input_builder_vectors():
    """what the reference's own input builder (model_input_builder.py:105-378) and the metadata
    builder of its wired attention backend (flash_attn.py:208-365, slot mapping backends/utils.py:31-75)
    produce for every scenario of tests/ib_driver.py."""
    import ib_driver
    from oracle import ref_block_manager
    ns = ref_block_manager.load_input_builder()
    scenarios = ib_driver.make_scenarios()
    n_tok = 0
    for sc in scenarios:
        mi = ref_block_manager.ref_build_model_input(ns, sc)
        sc["expect"] = ib_driver.record(mi)
        n_tok += len(sc["expect"]["input_tokens"])
    path = os.path.join(GOLDEN, "input_builder.json")
    with open(path, "w") as f:
        json.dump({"seed": 1234, "scenarios": scenarios}, f, separators=(",", ":"))
    print(f"{path}: {len(scenarios)} steps, {n_tok} tokens ({os.path.getsize(path) / 1024:.0f} KiB)")


def make_kv_sizing_vectors():
    """KV-cache sizing of the reference, run here on made-up memory readings.
    `Worker.determine_num_available_blocks` (decoding/worker/gpu_worker.py:95-144) is arithmetic around
    torch.cuda.mem_get_info(); it is called UNMODIFIED on a stand-in `self` while torch.cuda.mem_get_info /
    empty_cache / synchronize are replaced by a scripted sequence of readings (no GPU in this container).
    Also recorded: CacheEngine.get_cache_block_size (cache_engine.py:85-103) and the prompt lengths of
    GPUModelRunner.profile_run (decoding/runner/model_runner.py:111-145)."""
    import random
    import types
    import torch
    from oracle import ref_block_manager
    ref_block_manager.load_input_builder()
    from light_vllm.decoding.runner import model_runner
    from light_vllm.decoding.worker import cache_engine, gpu_worker
    rng = random.Random(99)
    GiB = 1 << 30
    cases = []
    real = (torch.cuda.mem_get_info, torch.cuda.empty_cache, torch.cuda.synchronize)
    try:
        torch.cuda.empty_cache = lambda: None
        torch.cuda.synchronize = lambda *a, **k: None
        for i in range(40):
            total = rng.choice([288, 192, 80, 24]) * GiB - rng.randrange(0, GiB)
            init_free = total - rng.randrange(0, 2 * GiB)
            weights = rng.randrange(GiB // 4, 40 * GiB)
            runtime = rng.randrange(GiB // 64, 6 * GiB)
            util = rng.choice([0.9, 0.95, 0.5, 0.3])
            scheduling = rng.choice(["sync", "simple_async", "async", "double_buffer"])
            swap = rng.choice([4 * GiB, 0, GiB + 12345])
            L, bs, kvh, d = rng.choice([(32, 16, 8, 128), (28, 16, 4, 128), (2, 32, 2, 64), (80, 16, 8, 128)])
            cache_dtype, model_dtype = rng.choice([("auto", torch.bfloat16), ("auto", torch.float32),
                                                   ("fp8", torch.bfloat16), ("fp8_e4m3", torch.float16)])
            mc = types.SimpleNamespace(get_head_size=lambda d=d: d, get_num_kv_heads=lambda k=kvh: k,
                                       get_num_attention_layers=lambda L=L: L, dtype=model_dtype)
            cc = types.SimpleNamespace(block_size=bs, cache_dtype=cache_dtype, gpu_memory_utilization=util,
                                       swap_space_bytes=swap)
            block_bytes = cache_engine.CacheEngine.get_cache_block_size(cc, mc)
            readings = iter([(init_free - weights, total), (init_free - weights - runtime, total)])
            torch.cuda.mem_get_info = lambda *a, **k: next(readings)
            me = types.SimpleNamespace(
                init_gpu_memory=init_free, cache_config=cc, model_config=mc,
                scheduler_config=types.SimpleNamespace(scheduling=scheduling),
                model_runner=types.SimpleNamespace(profile_run=lambda: None),
                get_cache_block_size_bytes=lambda b=block_bytes: b)
            fn = gpu_worker.Worker.determine_num_available_blocks
            num_gpu, num_cpu = fn(me)
            cases.append(dict(total=total, init_free=init_free, free_after_load=init_free - weights,
                              free_after_profile=init_free - weights - runtime, gpu_memory_utilization=util,
                              scheduling=scheduling, swap_space_bytes=swap, num_layers=L, block_size=bs,
                              num_kv_heads=kvh, head_size=d, cache_dtype=cache_dtype,
                              model_dtype=str(model_dtype).split(".")[-1], block_bytes=block_bytes,
                              num_gpu_blocks=num_gpu, num_cpu_blocks=num_cpu))
    finally:
        torch.cuda.mem_get_info, torch.cuda.empty_cache, torch.cuda.synchronize = real
    # prompt lengths of the profile run
    profiles = []
    for tokens, seqs in [(8192, 256), (64, 64), (2048, 7), (100, 33), (512, 512), (4097, 256)]:
        seen = {}
        me = types.SimpleNamespace(
            scheduler_config=types.SimpleNamespace(max_num_batched_tokens=tokens, max_num_seqs=seqs),
            model_config=types.SimpleNamespace(get_num_layers=lambda: 3), vocab_size=1000,
            prepare_model_input=lambda s, seen=seen: seen.setdefault("seqs", s) and types.SimpleNamespace(to=lambda d: None),
            execute_model=lambda mi, kv, seen=seen: seen.setdefault("kv", kv))
        real_sync = torch.cuda.synchronize
        torch.cuda.synchronize = lambda *a, **k: None
        try:
            model_runner.GPUModelRunner.profile_run(me)
        finally:
            torch.cuda.synchronize = real_sync
        lens = [list(m.seq_data.values())[0].get_len() for m in seen["seqs"]]
        assert all(m.is_prompt and m.block_tables is None for m in seen["seqs"]) and seen["kv"] == [None] * 3
        profiles.append(dict(max_num_batched_tokens=tokens, max_num_seqs=seqs, seq_lens=lens))
    path = os.path.join(GOLDEN, "kv_sizing.json")
    with open(path, "w") as f:
        json.dump({"cases": cases, "profile_runs": profiles}, f, separators=(",", ":"))
    print(f"{path}: {len(cases)} sizing cases, {len(profiles)} profile shapes")


def make_scheduler_traces():
    """What the reference's DecodingScheduler (decoding/scheduler.py:235-1132) returns, step by step, for the seeded
    programs of tests/sched_driver.py (default and chunked-prefill policies, v1 / v2 managers, recompute and swap
    preemption, prefix caching, lookahead slots), stepping synchronously as core/llm_engine.py:119-130 does."""
    import contextlib
    import io
    import types
    import bm_driver
    import sched_driver
    from oracle import ref_block_manager
    ns = ref_block_manager.load_input_builder()
    from light_vllm.decoding.scheduler import DecodingScheduler
    adapter = bm_driver.ReferenceAdapter(ns)

    def make(cfg):
        sc = types.SimpleNamespace(max_num_batched_tokens=cfg["max_num_batched_tokens"], max_num_seqs=cfg["max_num_seqs"],
                                   max_model_len=cfg["max_model_len"], use_v2_block_manager=cfg["version"] == "v2",
                                   num_lookahead_slots=cfg.get("lookahead", 0), delay_factor=0.0,
                                   chunked_prefill_enabled=cfg["chunked"], preemption_mode=cfg["preemption_mode"])
        cc = types.SimpleNamespace(block_size=cfg["block_size"], num_gpu_blocks=cfg["num_gpu_blocks"],
                                   num_cpu_blocks=cfg["num_cpu_blocks"], sliding_window=None,
                                   enable_prefix_caching=cfg["enable_caching"])
        return DecodingScheduler(sc, cc, None)

    def recording_hook(bm, _recorded):
        """Only the releases made inside _free_block_table are order-sensitive (set() order); swap_in / swap_out
        release block by block in table order on both sides."""
        inner = _recording_free_hook(bm, 0, None)
        orig_table_free = bm._free_block_table
        log, depth = [], [0]
        for alloc in (bm.gpu_allocator, bm.cpu_allocator):
            logged = alloc.free

            def gated(block, _logged=logged, _dev=(0 if alloc is bm.gpu_allocator else 1)):
                if depth[0]:
                    log.append([_dev, block.block_number])
                return _logged(block)
            alloc.free = gated

        def table_free(table):
            depth[0] += 1
            try:
                return orig_table_free(table)
            finally:
                depth[0] -= 1
        bm._free_block_table = table_free

        def finish():
            del bm._free_block_table  # back to the class's method
            inner()
            return log
        return finish

    only = os.environ.get("GOLDEN_ONLY")  # e.g. GOLDEN_ONLY=lookahead_prefix: record just the matching programs
    for name, cfg in sched_driver.CONFIGS:
        if only and only not in name:
            continue
        hook = recording_hook if cfg["version"] == "v1" else None
        with contextlib.redirect_stdout(io.StringIO()):  # the reference's _schedule_prefills prints its token counts
            trace = sched_driver.run_program(make, adapter, cfg, free_hook=hook)
        path = os.path.join(GOLDEN, f"scheduler_{name}.json")
        with open(path, "w") as f:
            json.dump({"config": cfg, "trace": trace}, f, separators=(",", ":"))
        steps = [t for t in trace if "groups" in t]
        print(f"{path}: {len(steps)} steps, {sum(len(t['groups']) for t in steps)} scheduled groups, "
              f"{steps[-1]['cumulative_preemption']} preemptions, {sum(len(t['swap_out']) for t in steps)} blocks swapped out, "
              f"{sum(len(t['ignored']) for t in steps)} ignored ({os.path.getsize(path) / 1024:.0f} KiB)")


if __name__ == "__main__":
    os.makedirs(GOLDEN, exist_ok=True)
    what = sys.argv[1:] or ["block_manager", "ops", "prefill_only", "sampler", "input_builder", "kv_sizing", "scheduler"]
    if "block_manager" in what:
        make_block_manager_traces()
    if "ops" in what:
        make_op_vectors()
    if "prefill_only" in what:
        make_prefill_only_vectors()
    if "sampler" in what:
        make_sampler_vectors()
    if "input_builder" in what:
        make_input_builder_vectors()
    if "kv_sizing" in what:
        make_kv_sizing_vectors()
    if "scheduler" in what:
        make_scheduler_traces()
