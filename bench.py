#!/usr/bin/env python3
"""Headline benchmark: decode tokens/s at Llama-3-8B shapes, bs=32, context ~1k, bf16, through
the engine's async decoding scheduler (BASELINE.json configs[1]), plus the HBM roofline of the
dominant hot-path kernel (paged_attention_v2's pass over the paged KV cache) and CPU baselines.

  python bench.py --gpus N --steps K --warmup W
      N > 1: launched by the driver through torch.distributed.run, or -- started as one plain process --
      launching its own N ranks that way (one rank per GPU, independent replicas)

A "step" is one MODEL step: 32 decoding sequences go through the 32-layer forward, lm_head and greedy
arg-max (a captured HIP graph) and get one token each.  The scheduler admits one batch of 32 per engine step
(`max_num_seqs`), `max_num_on_the_fly = 2` engine steps are in flight (the reference's default for async
scheduling and the setting BASELINE.md section 4 quotes the headline on; each on its own stream as in the
reference's async_execute_loop, core/executor.py:62-93), and an engine step is a burst of `--num-scheduler-steps`
model steps chained on the device by advance_step over lookahead slots of the block manager.  The timed region
holds exactly K model steps with an empty pipeline on both sides.  Every rank is an independent replica with its
own weights, KV cache and scheduler (replicas only, no collective on the data path); `value`
is the sum over ranks.

Output: ONE JSON line on rank 0 (contract in the task description; roofline / cpu_baseline / ops_baseline /
other_settings objects described in DESIGN.md section 5).
"""
import argparse
import ctypes
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# multi-process GPU work on this platform needs dmabuf IPC (RCCL fails with hipIpcGetMemHandle: invalid argument
# otherwise); the boxes export it already -- kept for a shell that does not
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak (no sparsity), same guide


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--batch-size", type=int, default=32)
    ap.add_argument("--context", type=int, default=1024)
    ap.add_argument("--scheduling", default="async", choices=["sync", "simple_async", "async", "double_buffer"],
                    help="the reference's execute loops (core/executor.py:37-185): double_buffer = its third loop, whose "
                         "default is three steps in flight (decoding/config.py:149-155) -- here the async pipeline at "
                         "--on-the-fly 3 unless given; the headline (BASELINE config 2) is async")
    ap.add_argument("--kv-cache-dtype", default="auto", choices=["auto", "fp8"],
                    help="non-default runs only: fp8 = OCP e4m3fn KV cache (the headline is bf16)")
    ap.add_argument("--kv-block-pad-bytes", type=int, default=None,
                    help="non-default runs only: CacheConfig.block_pad_bytes (default: the engine's, 1024; 0 = the "
                         "reference's dense block layout)")
    ap.add_argument("--quantization", default=None, choices=["fp8"],
                    help="non-default runs only: W8A8 projections (BASELINE config 5; the headline is bf16)")
    ap.add_argument("--on-the-fly", type=int, default=2,
                    help="engine steps in flight with async scheduling = SchedulerConfig.max_num_on_the_fly of the "
                         "reference (decoding/config.py:149-155: 2 by default -- BASELINE.md section 4 step 4 quotes the "
                         "headline on that -- 3 for its double_buffer mode)")
    ap.add_argument("--also-on-the-fly", type=int, default=3,
                    help="after the headline region: B more sequences per extra step in flight are admitted and the same "
                         "number of model steps is timed again at this many steps in flight, reported as "
                         "`other_settings` (never as `value`); 0 = skip")
    ap.add_argument("--num-scheduler-steps", type=int, default=8,
                    help="model steps per engine step (multi-step decode: advance_step on the device between them); "
                         "1 = one host round trip per model step.  The timed region holds exactly --steps MODEL steps: "
                         "bursts use the largest divisor of --steps (and of --warmup) that is <= this")
    ap.add_argument("--attn-version", default="v2", choices=["v1", "v2", "auto"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--library-gemm", action="store_true", help="dense projections through hipBLASLt (F.linear)")
    ap.add_argument("--no-fusion", action="store_true", help="reference op sequence (no fused decode launches)")
    ap.add_argument("--no-rope-in-attention", action="store_true",
                    help="rope + cache write as their own launch in front of attention (A/B of the fused kernel)")
    ap.add_argument("--no-fp8-activations-once", action="store_true",
                    help="A/B (W8A8): every projection quantises its own activations again (round 2's path)")
    ap.add_argument("--no-rope-in-attention-fp8", dest="rope_in_attention_fp8", action="store_false",
                    help="A/B: over an fp8 KV cache rope, cache write and attention as three launches (round 2's path)")
    ap.add_argument("--o-proj-partials-min-rows", type=int, default=None,
                    help="A/B: decode steps of at least this many rows split o_proj's K over workgroups (default 33)")
    ap.add_argument("--gemm-partials-ksplit", type=int, default=None,
                    help="A/B: projections that leave split-K partials split K at least this many ways")
    ap.add_argument("--gemm-workgroups", type=int, default=None,
                    help="workgroups per decode GEMM launch (default: 256 for one stream, 128 for several)")
    ap.add_argument("--skip-cpu-baseline", action="store_true")
    ap.add_argument("--skip-ops-baseline", action="store_true", help="no per-op GPU / CPU timings of the small operators")
    ap.add_argument("--skip-other-configs", action="store_true",
                    help="no other_settings entries for BASELINE configs 3 (chunked prefill), 4 (encode-only) and 5 (fp8 "
                         "weights + fp8 KV cache); each builds its own engine after the headline region")
    ap.add_argument("--skip-prefill-roofline", action="store_true",
                    help="no roofline_prefill object (one 16k-token prompt through the prefill kernel)")
    ap.add_argument("--repeats", type=int, default=0,
                    help="the timed region (EXACTLY --steps model steps, barrier + synchronize on both sides) is run this "
                         "many times back to back and `value` is the MEDIAN region; 0 = auto: min(5, 100 // steps), i.e. "
                         "one region at the default 64 steps, five when a caller passes --steps 20 (a 68 ms region moves "
                         "+-1 % from run to run)")
    ap.add_argument("--no-pin", action="store_true", help="N > 1: do not pin the rank to its GPU's NUMA cores")
    ap.add_argument("--kernel-iters", type=int, default=224)  # 20 warm-up + 200 timed launches
    ap.add_argument("--tiny", action="store_true", help="tiny model (plumbing check)")
    ap.add_argument("--replica-backend", default=None, choices=["nccl", "gloo"],
                    help="process-group backend of the barrier / clock between replicas (default: nccl = RCCL)")
    ap.add_argument("--single-device", action="store_true",
                    help="plumbing check of the N > 1 launch on a one-GPU box: every rank uses cuda:0 (needs "
                         "--replica-backend gloo: RCCL refuses two ranks on one device)")
    return ap.parse_args()


def kernel_leg(engine, B, iters, seq_len=None):
    """Per-launch time of paged_attention_v2's partition pass (the dominant hot-path kernel) on
    the engine's own KV caches: HIP events on the launch stream, one layer's cache per launch so
    that consecutive launches read 32 different 134 MB caches (>> the 256 MiB Infinity Cache).
    `seq_len`: attend to exactly that many tokens of every sequence (the metric's seq = 1024; the
    sequences have grown past it during the timed steps), None = their current lengths."""
    from light_vllm_amd import _native
    from light_vllm_amd.paged_attn import PagedAttention
    lib = _native.load_hip_library()
    fn = lib.lvllm_paged_attention_v2_phases
    fn.restype = ctypes.c_int
    cfg = engine.model_config
    dev = engine.device
    H, KVH, D, BS = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim, engine.cache_config.block_size
    bm = engine.scheduler.block_manager
    groups = list(engine.scheduler.running)[:B]
    seqs = [g.seqs[0] for g in groups]
    lens = [s.get_len() for s in seqs]
    if seq_len is not None:
        assert min(lens) >= seq_len
        lens = [seq_len] * len(seqs)
    tables = [bm.get_block_table(s) for s in seqs]
    width = max(len(t) for t in tables)
    bt = torch.zeros(len(seqs), width, dtype=torch.int32)
    for i, t in enumerate(tables):
        bt[i, :len(t)] = torch.tensor(t, dtype=torch.int32)
    bt = bt.to(dev)
    sl = torch.tensor(lens, dtype=torch.int32, device=dev)
    max_len = max(lens)
    P = (max_len + 511) // 512
    q = (torch.randn(len(seqs), H, D, device=dev) * 0.5).to(cfg.dtype)
    out = torch.empty_like(q)
    tmp = torch.empty(len(seqs), H, P, D, dtype=cfg.dtype, device=dev)
    es = torch.empty(len(seqs), H, P, dtype=torch.float32, device=dev)
    ml = torch.empty_like(es)
    caches = [PagedAttention.split_kv_cache(kv, KVH, D) for kv in engine.worker.cache_engine.gpu_cache]
    dt = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}[cfg.dtype]
    kv_fp8 = engine.cache_config.cache_dtype != "auto"  # strides below are in cache elements
    assert caches[0][0].element_size() == (1 if kv_fp8 else q.element_size())
    vp = ctypes.c_void_p

    def launch(i, phases):
        kc, vc = caches[i % len(caches)]
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = fn(vp(out.data_ptr()), vp(es.data_ptr()), vp(ml.data_ptr()), vp(tmp.data_ptr()), vp(q.data_ptr()),
                vp(kc.data_ptr()), vp(vc.data_ptr()), ctypes.c_int(len(seqs)), ctypes.c_int(H), ctypes.c_int(D),
                ctypes.c_int(KVH), ctypes.c_float(D ** -0.5), vp(bt.data_ptr()), vp(sl.data_ptr()),
                ctypes.c_int(BS), ctypes.c_int(max_len), ctypes.c_int(width), ctypes.c_int(P), vp(0),
                ctypes.c_int64(q.stride(0)), ctypes.c_int64(kc.stride(0)), ctypes.c_int64(kc.stride(1)),
                ctypes.c_int(dt), ctypes.c_int(1 if kv_fp8 else 0), ctypes.c_float(1.0), ctypes.c_float(1.0),
                ctypes.c_int(0),
                ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(64), ctypes.c_int(0),
                ctypes.c_int64(kc.size(0) * kc.stride(0) * kc.element_size()), ctypes.c_int(phases), vp(stream))
        assert rc == 0, lib.lvllm_last_error()

    for i in range(32):  # one untimed train
        launch(i, 1)
    torch.cuda.synchronize(dev)
    # HIP events on the launch stream around TRAINS of back-to-back launches, one layer's cache per
    # launch: the per-launch average then carries the kernel and its launch gap, not the cost of
    # recording two events per kernel (which added ~3 us to a 24 us kernel and made the figure
    # disagree with rocprof's duration of the same kernel).  A train is a captured graph of its launches, as in the
    # in-step leg: issued one by one from Python (a ctypes call of 35 arguments each) the train is only as fast as
    # the host on a busy box -- one run of round 4 read 27.3 us average against 23.4 minimum for that reason.
    train = len(caches)
    ntrains = max(2, iters // train)
    side = torch.cuda.Stream(dev)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            for k in range(train):
                launch(k, 1)
    torch.cuda.synchronize(dev)
    graph.replay()
    torch.cuda.synchronize(dev)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(ntrains)]
    for a, b in evs:
        a.record()
        graph.replay()
        b.record()
    torch.cuda.synchronize(dev)
    ts = [a.elapsed_time(b) * 1e-3 / train for a, b in evs]  # seconds per launch
    avg = sum(ts) / len(ts)
    esz = 2
    algo_bytes = (2 * sum(lens) * KVH * D * (1 if kv_fp8 else esz) + 2 * len(seqs) * H * D * esz +
                  len(seqs) * ((max_len + BS - 1) // BS) * 4 + len(seqs) * 4)
    # shares per context the library picks for this launch (attention.hip: want = ceil(2048 / (8 * pairs)),
    # pairs = seqs * kv heads * ceil(G / 16)), for the label of the roofline object
    pairs = len(seqs) * KVH * ((H // KVH + 15) // 16)
    nsplit = max(1, min(P, (2048 + 8 * pairs - 1) // (8 * pairs), ((max_len + 15) // 16) // 4))
    return dict(avg_s=avg, min_s=min(ts), algo_bytes=algo_bytes, lens=lens, partitions=P, nsplit=nsplit)


def in_step_attention_leg(engine, B, seq_len, trains=6):
    """The launch a decode step of the engine really makes: rotary_embedding + reshape_and_cache + paged_attention_v2
    as ONE kernel (the ROPE instantiation of paged_attn_mfma_kernel, `_C_amd.rope_cache_paged_attention`), timed like
    `kernel_leg` -- trains of one launch per layer cache between a HIP event pair on the launch stream -- with the new
    token at position seq_len - 1 of every sequence (it is rotated, written to the caches and attended to).  Each train
    is a captured graph of the 32 launches, so the host's dispatch cost is not in the figure.  None when the engine's
    configuration is outside the fused launch's envelope."""
    from light_vllm_amd.paged_attn import PagedAttention
    cfg = engine.model_config
    dev = engine.device
    H, KVH, D, BS = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim, engine.cache_config.block_size
    kv = engine.cache_config.cache_dtype
    model = engine.worker.model
    if not hasattr(torch.ops._C_amd, "rope_cache_paged_attention") or getattr(model, "cos_sin_cache", None) is None:
        return None
    bm = engine.scheduler.block_manager
    seqs = [g.seqs[0] for g in list(engine.scheduler.running)[:B]]
    if not seqs or min(s.get_len() for s in seqs) < seq_len:
        return None
    tables = [bm.get_block_table(s) for s in seqs]
    width = max(len(t) for t in tables)
    bt = torch.zeros(len(seqs), width, dtype=torch.int32)
    for i, t in enumerate(tables):
        bt[i, :len(t)] = torch.tensor(t, dtype=torch.int32)
    slots = (bt[:, (seq_len - 1) // BS].long() * BS + (seq_len - 1) % BS).to(dev)
    bt = bt.to(dev)
    sl = torch.full((len(seqs),), seq_len, dtype=torch.int32, device=dev)
    pos = torch.full((len(seqs),), seq_len - 1, dtype=torch.int64, device=dev)
    qkv = (torch.randn(len(seqs), (H + 2 * KVH) * D, device=dev) * 0.5).to(cfg.dtype)
    qv, kv_, vv = qkv.split([H * D, KVH * D, KVH * D], dim=-1)
    out = torch.empty(len(seqs), H, D, dtype=cfg.dtype, device=dev)
    P = (seq_len + 511) // 512
    tmp = torch.empty(len(seqs), H, P, D, dtype=cfg.dtype, device=dev)
    es = torch.empty(len(seqs), H, P, dtype=torch.float32, device=dev)
    ml = torch.empty_like(es)
    caches = [PagedAttention.split_kv_cache(c, KVH, D) for c in engine.worker.cache_engine.gpu_cache]
    cos_sin = model.cos_sin_cache

    def launch(i):
        kc, vc = caches[i % len(caches)]
        return torch.ops._C_amd.rope_cache_paged_attention(out, es, ml, tmp, pos, qv, kv_, vv, D, cos_sin, True, kc, vc,
                                                           slots, KVH, D ** -0.5, bt, sl, BS, seq_len, kv, 1.0, 1.0)
    if not launch(0):
        return None
    for i in range(len(caches)):
        launch(i)
    torch.cuda.synchronize(dev)
    side = torch.cuda.Stream(dev)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            for i in range(len(caches)):
                launch(i)
    torch.cuda.synchronize(dev)
    graph.replay()
    torch.cuda.synchronize(dev)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(trains)]
    for a_, b_ in evs:
        a_.record()
        graph.replay()
        b_.record()
    torch.cuda.synchronize(dev)
    ts = [a_.elapsed_time(b_) * 1e-3 / len(caches) for a_, b_ in evs]
    avg = sum(ts) / len(ts)
    kv_fp8 = kv != "auto"
    n = len(seqs)
    plain = (2 * n * seq_len * KVH * D * (1 if kv_fp8 else 2) + 2 * n * H * D * 2 + n * ((seq_len + BS - 1) // BS) * 4 + n * 4)
    # what the fused launch moves besides: the new key / value rows in, their cache rows out, one cos / sin row, the
    # position and the slot of every sequence
    own = plain + 2 * n * KVH * D * 2 + 2 * n * KVH * D * (1 if kv_fp8 else 2) + n * D * 2 + n * 16
    return {"kernel": "paged_attn_mfma_kernel<..., ROPE = true> (rotary_embedding + reshape_and_cache + "
                      "paged_attention_v2 in one launch: the attention launch of the engine's decode step)",
            "avg_launch_us": round(avg * 1e6, 2), "min_launch_us": round(min(ts) * 1e6, 2),
            "algorithmic_bytes_per_launch": own, "achieved": round(own / avg / 1e9, 1), "unit": "GB/s",
            "frac": round(own / avg / 1e9 / HBM_PEAK_GBS, 4),
            "frac_on_plain_bytes": round(plain / avg / 1e9 / HBM_PEAK_GBS, 4),
            "timing": f"HIP events around {trains} replays of a captured train of {len(caches)} launches (one per layer "
                      f"cache), seq_lens = {seq_len}"}


def gemm_leg(engine, B):
    """The other HBM stream of a decode step: the four weight-streaming projections of a layer
    (csrc/skinny_gemm.hip), timed on the engine's own packed weights -- trains of one launch per layer
    (32 different weight matrices per train, >> the Infinity Cache) with HIP events on the launch
    stream, at the workgroup count a lone step would use (256).  W8A8 engines (`--quantization fp8`) time
    lvllm_skinny_gemm_w8a8 on the packed fp8 weights (one byte per weight).  Returns None when the engine's
    weights are not packed (library GEMM)."""
    model = engine.worker.model
    layers = model.layers
    w8 = all(l.qkv.w8_packed is not None for l in layers)
    if not w8 and any(l.qkv.packed is None for l in layers):
        return None
    dev = engine.device
    cfg = engine.model_config
    torch.ops._C_amd.set_tuning("gemm_workgroups", 256)
    try:
        per_shape, tot_bytes, tot_s = {}, 0, 0.0
        for name in ("qkv", "o", "gate_up", "down"):
            ws = [getattr(l, name) for l in layers]
            N, K = ws[0].N, ws[0].K
            x = (torch.randn(B, K, device=dev) * 0.5).to(cfg.dtype)

            once = w8 and cfg.fp8_activations_once and B <= 32
            glu = (not w8) and name == "gate_up" and cfg.swiglu_epilogue and N % 32 == 0
            x8 = torch.randint(0, 120, (B, K), device=dev, dtype=torch.uint8) if once else None
            down_scale = layers[0].down.x_scale

            def call(w):
                if once and name in ("qkv", "o"):  # the launches the engine's step makes (activations already fp8)
                    torch.ops._C_amd.skinny_linear_w8a8_q(x8, w.w8_packed, w.w_scale, w.x_scale, N, K, None, cfg.dtype)
                elif once and name == "gate_up":
                    torch.ops._C_amd.skinny_linear_w8a8_q_swiglu_fp8(x8, w.w8_packed, w.w_scale, w.x_scale, N, K, down_scale,
                                                                     cfg.dtype)
                elif once:
                    torch.ops._C_amd.skinny_linear_w8a8_q_partials(x8, w.w8_packed, w.w_scale, w.x_scale, N, K)
                elif w8:
                    torch.ops._C_amd.skinny_linear_w8a8(x, w.w8_packed, w.w_scale, w.x_scale, N, K, None)
                elif glu:  # the launch the engine's step makes: gate_up with silu_and_mul in its epilogue
                    torch.ops._C_amd.skinny_linear_packed_swiglu(x, w.packed, None, N, K)
                else:
                    torch.ops._C_amd.skinny_linear_packed(x, w.packed, None, N, K)
            for w in ws[:4]:
                call(w)
            torch.cuda.synchronize(dev)
            # (a train = a captured graph of one launch per layer, as in the attention legs: the host's dispatch of 32
            # operator calls is not in the figure)
            side = torch.cuda.Stream(dev)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side):
                    for w in ws:
                        call(w)
            torch.cuda.synchronize(dev)
            graph.replay()
            torch.cuda.synchronize(dev)
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(4)]
            for a, b in evs:
                a.record()
                graph.replay()
                b.record()
            torch.cuda.synchronize(dev)
            del graph
            t = min(a.elapsed_time(b) for a, b in evs) * 1e-3 / len(ws)
            by = N * K * (1 if w8 else 2) + B * K * (1 if once else 2) + B * N * 2  # weights once + activations in + result out
            if glu:
                by = N * K * 2 + B * K * 2 + B * (N // 2) * 2  # SwiGLU result out
            if once and name == "gate_up":
                by = N * K + B * K + B * (N // 2)          # fp8 in, SwiGLU result out as fp8
            elif once and name == "down":
                by = N * K + B * K + 4 * B * N * ((K + 4095) // 4096)  # fp8 in, fp32 split-K slabs out
            per_shape[name] = {"us": round(t * 1e6, 2), "GB/s": round(by / t / 1e9, 1)}
            tot_bytes += by
            tot_s += t
        return dict(per_shape=per_shape, bytes_per_layer=tot_bytes, s_per_layer=tot_s, w8=w8)
    finally:
        torch.ops._C_amd.set_tuning("gemm_workgroups", engine.gemm_workgroups)


def cpu_baseline_leg(engine, B, budget_s=20.0):
    """paged_attention_v2 of the same shapes on the host cores: the reference's own csrc/cpu
    backend (oracle/_ref, kind "reference") when it is present and the CPU has AVX512, else
    our C restatement (oracle/, kind "port").  Bounded sample: one layer's attention for the
    batch, repeated until ~budget_s; scaled to decode tokens/s by the 32 layers of the model
    would be misleading (the CPU would also run the GEMMs), so it is reported per kernel call."""
    from oracle import oracle as port
    from oracle import ref
    from helpers import make_paged_inputs, v2_scratch
    cfg = engine.model_config
    H, KVH, D, BS = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim, engine.cache_config.block_size
    L = 1024
    inp = make_paged_inputs(B, H, KVH, D, BS, [L] * B, dtype=torch.bfloat16, seed=0)
    q = inp["query"]
    es, ml, tmp = v2_scratch(B, H, D, L, q.dtype)
    out = torch.zeros_like(q)
    use_ref = ref.load() and BS == 16
    cores = torch.get_num_threads()
    if use_ref:
        def call():
            torch.ops._ref_C.paged_attention_v2(out, es, ml, tmp, q, inp["key_cache"], inp["value_cache"], KVH,
                                                inp["scale"], inp["block_tables"], inp["seq_lens"], BS, L, None,
                                                "auto", 1.0, 1.0, 0, 0, 0, 64, 0)
        kind = "reference"
    else:
        def call():
            port.paged_attention_v2(out, es, ml, tmp, q, inp["key_cache"], inp["value_cache"], KVH, inp["scale"],
                                    inp["block_tables"], inp["seq_lens"], BS, L)
        kind = "port"
        cores = port.num_threads()
    call()
    t0 = time.perf_counter()
    n = 0
    while True:
        call()
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 200:
            break
    per_call = el / n
    algo_bytes = 2 * B * L * KVH * D * 2 + 2 * B * H * D * 2 + B * (L // BS) * 4 + B * 4
    return dict(value=round(algo_bytes / per_call / 1e9, 3), unit="GB/s", cores=int(cores), kind=kind,
                sample=f"{n} calls of paged_attention_v2 (one layer, bs={B}, seq={L}, H={H}, KVH={KVH}, D={D}, bf16) "
                       f"in {el:.1f} s; {per_call * 1e3:.2f} ms/call")


def ops_baseline_leg(engine, B, iters=200, cpu_budget_s=1.5):
    """The small operators of the path at the decode shapes (T = B tokens):
    reshape_and_cache, rms_norm, fused_add_rms_norm, rotary_embedding, silu_and_mul -- GPU microseconds per launch
    (HIP events around trains of back-to-back launches) and GB/s of the algorithmic bytes, next to the reference's
    own csrc/cpu operator (oracle/_ref, kind "reference") or our C restatement (kind "port") on the host cores.
    At T = 32 these kernels are launch-latency bound on the GPU (about 1 MiB of traffic each): the GB/s column
    says how far from a bandwidth regime they are, not how good the kernel is."""
    from oracle import oracle as port
    from oracle import ref
    from light_vllm_amd import _custom_ops as ops
    cfg = engine.model_config
    dev = engine.device
    T, hid, inter = B, cfg.hidden_size, cfg.intermediate_size
    H, KVH, D, BS = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim, engine.cache_config.block_size
    dt = cfg.dtype
    es = 2
    use_ref = ref.load() and BS == 16 and dt == torch.bfloat16
    g = torch.Generator().manual_seed(0)
    x = torch.randn(T, hid, generator=g).to(dt)
    res = torch.randn(T, hid, generator=g).to(dt)
    w = (1 + 0.1 * torch.randn(hid, generator=g)).to(dt)
    qkv = torch.randn(T, (H + 2 * KVH) * D, generator=g).to(dt)
    pos = torch.randint(0, 1024, (T,), generator=g, dtype=torch.int64)
    cache = engine.worker.model.cos_sin_cache.cpu()
    gu = torch.randn(T, 2 * inter, generator=g).to(dt)
    NB = 64
    kc = torch.zeros(NB, KVH, D // 8, BS, 8, dtype=dt)
    vc = torch.zeros(NB, KVH, D, BS, dtype=dt)
    slots = torch.randperm(NB * BS, generator=g)[:T].to(torch.int64)

    def on(device):
        d = {}
        d["x"], d["res"], d["w"], d["qkv"], d["pos"], d["cache"], d["gu"], d["kc"], d["vc"], d["slots"] = (
            t.to(device) for t in (x, res, w, qkv, pos, cache, gu, kc, vc, slots))
        d["out"] = torch.empty_like(d["x"])
        d["act"] = torch.empty(T, inter, dtype=dt, device=device)
        d["q"] = d["qkv"][:, :H * D]
        d["k"] = d["qkv"][:, H * D:(H + KVH) * D]
        d["v"] = d["qkv"][:, (H + KVH) * D:]
        return d

    def calls(o, cache_ops, d):
        return {
            "reshape_and_cache": (lambda: cache_ops.reshape_and_cache(d["k"].view(T, KVH, D), d["v"].view(T, KVH, D), d["kc"],
                                                                      d["vc"], d["slots"], "auto", 1.0, 1.0),
                                  4 * T * KVH * D * es + 8 * T),
            "rms_norm": (lambda: o.rms_norm(d["out"], d["x"], d["w"], 1e-5), (2 * T * hid + hid) * es),
            "fused_add_rms_norm": (lambda: o.fused_add_rms_norm(d["x"], d["res"], d["w"], 1e-5), (4 * T * hid + hid) * es),
            "rotary_embedding": (lambda: o.rotary_embedding(d["pos"], d["q"], d["k"], D, d["cache"], True),
                                 2 * T * (H + KVH) * D * es + T * (8 + D * es)),
            "silu_and_mul": (lambda: o.silu_and_mul(d["act"], d["gu"]), 3 * T * inter * es),
        }

    gd = on(dev)
    gpu_calls = calls(ops, ops, gd)
    cd = on("cpu")
    if use_ref:
        cpu_calls = calls(torch.ops._ref_C, torch.ops._ref_C_cache_ops, cd)
        kind, cores = "reference", torch.get_num_threads()
    else:
        class _P:  # the port takes the same arguments
            rms_norm = staticmethod(port.rms_norm)
            fused_add_rms_norm = staticmethod(port.fused_add_rms_norm)
            rotary_embedding = staticmethod(port.rotary_embedding)
            silu_and_mul = staticmethod(port.silu_and_mul)
            reshape_and_cache = staticmethod(lambda k_, v_, kc_, vc_, s_, *_: port.reshape_and_cache(k_, v_, kc_, vc_, s_))
        cpu_calls = calls(_P, _P, cd)
        kind, cores = "port", port.num_threads()
    out = {}
    # (the engine's step threads have just gone idle and the code objects of these kernels may not be loaded yet: a
    # pass over every operator and a short pause first, or the first one measured pays for both -- 41 us once)
    for fn, _ in gpu_calls.values():
        for _ in range(30):
            fn()
    torch.cuda.synchronize(dev)
    time.sleep(0.3)
    for name, (fn, nbytes) in gpu_calls.items():
        for _ in range(10):
            fn()
        torch.cuda.synchronize(dev)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize(dev)
        us = a.elapsed_time(b) * 1e3 / iters
        cfn = cpu_calls[name][0]
        cfn()
        t0 = time.perf_counter()
        n = 0
        while True:
            cfn()
            n += 1
            el = time.perf_counter() - t0
            if el > cpu_budget_s or n >= 2000:
                break
        cpu_us = el / n * 1e6
        out[name] = {"algorithmic_bytes": nbytes, "gpu_us": round(us, 2), "gpu_GB/s": round(nbytes / us / 1e3, 1),
                     "cpu_us": round(cpu_us, 1), "cpu_GB/s": round(nbytes / cpu_us / 1e3, 2), "cpu_calls": n}
    return {"tokens": T, "dtype": "bf16", "cpu_kind": kind, "cpu_cores": int(cores),
            "note": "eager launches back to back, launch gap included; T = batch rows of one decode step", "ops": out}


def prefill_leg(engine, qlen=16384, iters=12):
    """Row f-1 of the scope table next to the headline: causal attention of ONE prompt of `qlen` tokens over the
    paged cache (the call that replaces flash_attn_varlen_func(..., block_table=...), flash_attn.py:538-555) at the
    model's head shapes, timed with HIP events on the launch stream.  MFMA-bound: FLOPs = 4 D (visible query-key
    pairs) H against the dense bf16 peak."""
    from light_vllm_amd import _custom_ops as ops
    cfg = engine.model_config
    dev = engine.device
    H, KVH, D, BS = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim, engine.cache_config.block_size
    if cfg.dtype != torch.bfloat16 or BS not in (16, 32):
        return None
    nblk = (qlen + BS - 1) // BS
    g = torch.Generator(device="cpu").manual_seed(5)
    kc = (torch.randn(nblk + 3, KVH, D // 8, BS, 8, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    vc = (torch.randn(nblk + 3, KVH, D, BS, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    bt = torch.randperm(nblk + 3, generator=g)[:nblk].view(1, nblk).to(torch.int32).to(dev)
    q = (torch.randn(qlen, H, D, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    out = torch.empty_like(q)
    seq_lens = torch.tensor([qlen], dtype=torch.int32, device=dev)
    qsl = torch.tensor([0, qlen], dtype=torch.int32, device=dev)

    def run():
        ops.paged_prefill_attention(out, q, kc, vc, KVH, 1 / math.sqrt(D), bt, seq_lens, qsl, qlen, BS, None, 0, 0.0, "auto")

    for _ in range(6):  # (the clock needs a few launches of this kind of load to settle)
        run()
    torch.cuda.synchronize(dev)
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        run()
        b.record()
        torch.cuda.synchronize(dev)
        ts.append(a.elapsed_time(b) * 1e-3)
    avg = sum(ts) / len(ts)
    flops = 4.0 * D * (qlen * (qlen + 1) // 2) * H
    return {"bound": "mfma",
            "kernel": f"paged_prefill_mfma32_kernel (one causal prompt of {qlen} tokens over the paged cache, H {H} KVH {KVH} "
                      f"D {D}, block {BS}; SURVEY 8f-1)",
            "achieved": round(flops / avg / 1e12, 1), "peak": MFMA_PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(flops / avg / 1e12 / MFMA_PEAK_BF16_TFLOPS, 4), "flops_per_launch": flops,
            "avg_launch_us": round(avg * 1e6, 1), "min_launch_us": round(min(ts) * 1e6, 1)}


def decode_region(engine, B, steps, burst, in_flight, sync=False):
    """`steps` model steps of the decode engine as bursts of `burst`, `in_flight` engine steps in flight, the
    pipeline empty on both sides; returns (tokens appended -- counted by the engine, not inferred --, seconds)."""
    n = steps // burst
    in_flight = max(1, min(in_flight, n))
    engine.scheduler_config.num_scheduler_steps = burst
    engine.scheduler_config.max_num_on_the_fly = in_flight
    tok0, st0 = engine.stat_tokens_appended, engine.stat_model_steps
    torch.cuda.synchronize(engine.device)
    t0 = time.perf_counter()
    for i in range(n):
        if sync:
            engine.step()
        else:
            engine.async_step(schedule_more=i < n - (in_flight - 1))
    torch.cuda.synchronize(engine.device)
    el = time.perf_counter() - t0
    assert engine.num_on_the_fly == 0
    assert engine.stat_model_steps - st0 == steps, (engine.stat_model_steps - st0, steps)
    return engine.stat_tokens_appended - tok0, el


def fp8_config_leg(a, dev, B, ctx, steps=32):
    """BASELINE config 5 beside the headline: the same decode workload with fp8 (e4m3) W8A8 projections
    (csrc/quantization fp8 -> CDNA4 fp8 MFMA) AND an fp8 KV cache, two engine steps in flight, bursts of 8.  Its own
    engine (own weights); returns the other_settings entry with tokens/s, ms/step, algorithmic bytes / time against
    8 TB/s, and the rooflines of its two HBM streams (fp8 attention launch, W8A8 projections)."""
    from light_vllm_amd.engine.config import CacheConfig, ModelConfig, SchedulerConfig
    from light_vllm_amd.engine.llm_engine import LLMEngine
    cfg = ModelConfig.tiny() if a.tiny else ModelConfig.llama3_8b()
    cfg.quantization = "fp8"
    k = largest_divisor_at_most(steps, 8)
    fly = 2
    max_len = ctx + steps // fly + 6 * k + 8
    bs = 16
    blocks = B * fly * ((max_len + bs - 1) // bs + 1) + 64
    eng = LLMEngine(cfg, CacheConfig(block_size=bs, num_gpu_blocks=blocks, num_cpu_blocks=0, cache_dtype="fp8"),
                    SchedulerConfig(max_num_batched_tokens=max(8192, B), max_num_seqs=B,
                                    max_model_len=(max_len + 511) // 512 * 512, scheduling="async", max_num_on_the_fly=fly,
                                    use_v2_block_manager=True, num_scheduler_steps=k), device=dev, seed=7)
    eng.step_returns_outputs = False
    g = torch.Generator().manual_seed(99)
    for i in range(B * fly):
        eng.add_request(str(i), torch.randint(0, cfg.vocab_size, (ctx,), generator=g).tolist(), max_tokens=4 * steps + 100)
    eng.prefill_synthetic(seed=7)
    eng.capture_decode_graphs(B)
    decode_region(eng, B, k * fly, k, fly)  # warm-up: one burst per slot
    toks, el = decode_region(eng, B, steps, k, fly)
    assert toks == steps * B, (toks, steps, B)
    kl = kernel_leg(eng, B, a.kernel_iters, seq_len=ctx)
    in_step = in_step_attention_leg(eng, B, ctx)
    gm = gemm_leg(eng, B) if B <= 64 else None
    L = cfg.num_hidden_layers
    kv_step = 2 * B * ctx * cfg.num_key_value_heads * cfg.head_dim * 1 * L
    step_bytes = eng.worker.model.weight_bytes() + kv_step
    eng.shutdown()
    out = {"value": round(toks / el, 1), "unit": "tokens/s", "ms_per_step": round(el / steps * 1e3, 4),
           "steps": steps, "config": f"BASELINE config 5: fp8 e4m3 W8A8 projections + fp8 KV cache, decode bs={B} ctx={ctx}, "
                                     f"{fly} in flight, {k} model steps per engine step",
           "algorithmic_bytes_per_step": step_bytes,
           "hbm": {"achieved": round(step_bytes / (el / steps) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": round(step_bytes / (el / steps) / 1e9 / HBM_PEAK_GBS, 4)},
           "roofline_attention": {"bound": "hbm", "kernel": "paged_attn_mfma_kernel<KV8> (fp8 cache), single pass",
                                  "achieved": round(kl["algo_bytes"] / kl["avg_s"] / 1e9, 1), "peak": HBM_PEAK_GBS,
                                  "unit": "GB/s", "frac": round(kl["algo_bytes"] / kl["avg_s"] / 1e9 / HBM_PEAK_GBS, 4),
                                  "algorithmic_bytes_per_launch": kl["algo_bytes"],
                                  "avg_launch_us": round(kl["avg_s"] * 1e6, 2), "in_step": in_step}}
    try:
        with open(os.path.join(ROOT, "profiles", "r03_pmc_attn_fp8.json")) as f:
            out["roofline_attention"]["traffic"] = int(json.load(f)["traffic_over_algorithmic"] * kl["algo_bytes"])
    except (OSError, KeyError, ValueError):
        out["roofline_attention"]["traffic"] = None
    if gm is not None:
        ach = gm["bytes_per_layer"] / gm["s_per_layer"] / 1e9
        w8_traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r03_pmc_gemm_w8a8.json")) as f:
                pm = json.load(f)
            if B == pm["M"] and not a.tiny and cfg.fp8_activations_once:
                w8_traffic = sum(v["hbm_bytes_per_launch"] for v in pm["shapes"].values())
        except (OSError, KeyError, ValueError):
            pass
        out["roofline_projections"] = {"bound": "hbm", "kernel": f"skinny_gemm_kernel<W8, XQ> (qkv, o, gate_up + SwiGLU, down of one "
                                                                  f"layer, M = {B}, fp8 weights, fp8 activations in)",
                                       "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": w8_traffic,
                                       "algorithmic_bytes_per_layer": gm["bytes_per_layer"],
                                       "us_per_layer": round(gm["s_per_layer"] * 1e6, 2), "per_shape": gm["per_shape"]}
    del eng
    return out


def sampled_decode_leg(a, dev, B, ctx, steps=32):
    """The headline workload with SAMPLED requests (temperature 0.8, top-k 50, top-p 0.95, per-request seeds) instead of
    greedy ones: every step ends with the lm_head's logits and the device-side sampler (csrc/sampler.hip) inside the
    captured graph and the multi-step burst.  Its own engine; same counting as the headline."""
    from light_vllm_amd.engine.config import CacheConfig, ModelConfig, SchedulerConfig
    from light_vllm_amd.engine.llm_engine import LLMEngine
    from light_vllm_amd.engine.sampling_params import SamplingParams
    cfg = ModelConfig.tiny() if a.tiny else ModelConfig.llama3_8b()
    k = largest_divisor_at_most(steps, 8)
    fly = 2
    max_len = ctx + steps // fly + 6 * k + 8
    bs = 16
    blocks = B * fly * ((max_len + bs - 1) // bs + 1) + 64
    eng = LLMEngine(cfg, CacheConfig(block_size=bs, num_gpu_blocks=blocks, num_cpu_blocks=0),
                    SchedulerConfig(max_num_batched_tokens=max(8192, B), max_num_seqs=B,
                                    max_model_len=(max_len + 511) // 512 * 512, scheduling="async", max_num_on_the_fly=fly,
                                    use_v2_block_manager=True, num_scheduler_steps=k), device=dev, seed=11)
    eng.step_returns_outputs = False
    g = torch.Generator().manual_seed(123)
    for i in range(B * fly):
        sp = SamplingParams(temperature=0.8, top_k=50, top_p=0.95, seed=1000 + i, max_tokens=4 * steps + 100, ignore_eos=True)
        eng.add_request(str(i), torch.randint(0, cfg.vocab_size, (ctx,), generator=g).tolist(), sampling_params=sp)
    eng.prefill_synthetic(seed=11)
    eng.capture_decode_graphs(B)
    decode_region(eng, B, k * fly, k, fly)  # warm-up: one burst per slot (captures the sampler flavour of the graph)
    toks, el = decode_region(eng, B, steps, k, fly)
    assert toks == steps * B, (toks, steps, B)
    eng.shutdown()
    del eng
    return {"value": round(toks / el, 1), "unit": "tokens/s", "ms_per_step": round(el / steps * 1e3, 4), "steps": steps,
            "config": f"the headline workload (bf16, bs={B}, ctx={ctx}, {fly} in flight, {k} model steps per engine step) with every "
                      "request sampled: temperature 0.8, top-k 50, top-p 0.95, seeded; lm_head logits + the device-side sampler "
                      "inside the captured step"}


def chunked_prefill_leg(a, dev, num_prompts=1000, input_len=512, output_len=512, budget=64, short_prompts=96):
    """BASELINE config 3 beside the headline: the reference's benchmarks/benchmark_chunked_prefill_throughput.py
    workload (:176-201: 1000 prompts of 512 tokens, 512 output tokens each, chunked prefill, max_num_batched_tokens =
    max_num_seqs = 64) on its own bf16 engine, async, every mixed step a captured graph.  tokens/s counts prompt and
    output tokens like the reference script.  The job has a steady state (32 decodes + a 32-token chunk per step) and a
    drain of 512 decode-only steps with ever fewer sequences; rounds 2 - 4 quoted a 96-prompt run, a third of which is
    drain: that run is still made first, on the same engine, and reported beside the full one (`short_run`).
    Algorithmic bytes of a step: every weight once (<= 64 rows) + the K/V of every sequence in the step up to its
    current length, tallied from the scheduler's metadata."""
    from light_vllm_amd.engine.config import CacheConfig, ModelConfig, SchedulerConfig
    from light_vllm_amd.engine.llm_engine import LLMEngine
    cfg = ModelConfig.tiny() if a.tiny else ModelConfig.llama3_8b()
    if a.tiny:
        num_prompts, short_prompts, input_len, output_len = 12, 6, 96, 24
    max_len = input_len + output_len + 16
    bs = 16
    blocks = (budget + 8) * ((max_len + bs - 1) // bs + 1) + 64
    eng = LLMEngine(cfg, CacheConfig(block_size=bs, num_gpu_blocks=blocks, num_cpu_blocks=0),
                    SchedulerConfig(max_num_batched_tokens=budget, max_num_seqs=budget,
                                    max_model_len=(max_len + 511) // 512 * 512, scheduling="async", max_num_on_the_fly=2,
                                    chunked_prefill_enabled=True), device=dev, seed=3)
    eng.step_returns_outputs = False
    kv_tok = 2 * cfg.num_key_value_heads * cfg.head_dim * 2 * cfg.num_hidden_layers  # bytes of K+V per token, all layers
    tally = {"kv": 0, "steps": 0}
    real = eng._execute

    def counting(sched, slot):
        for m in sched.seq_group_metadata_list:
            for d in m.seq_data.values():
                n = d.get_len() if not m.is_prompt else min(d.get_len(), d.get_num_computed_tokens() + m.token_chunk_size)
                tally["kv"] += n * kv_tok
        tally["steps"] += 1
        return real(sched, slot)
    eng._execute = counting
    g = torch.Generator().manual_seed(0)
    next_id = [0]

    def job(n_prompts):
        tally["kv"] = tally["steps"] = 0
        tok0 = eng.stat_tokens_appended
        for _ in range(n_prompts):
            eng.add_request(str(next_id[0]), torch.randint(0, cfg.vocab_size, (input_len,), generator=g).tolist(),
                            max_tokens=output_len)
            next_id[0] += 1
        done = 0
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        while eng.has_unfinished_requests() or eng.num_on_the_fly > 0:
            for o in eng.async_step():
                done += o.finished
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        assert done == n_prompts and eng.stat_tokens_appended - tok0 == n_prompts * output_len
        by = tally["steps"] * eng.worker.model.weight_bytes() + tally["kv"]
        total = n_prompts * (input_len + output_len)
        return {"value": round(total / el, 1), "unit": "tokens/s", "requests_per_s": round(n_prompts / el, 2),
                "ms_per_step": round(el / tally["steps"] * 1e3, 4), "steps": tally["steps"],
                "config": f"BASELINE config 3: chunked prefill, {n_prompts} prompts x ({input_len} in + {output_len} out), "
                          f"max_num_batched_tokens = max_num_seqs = {budget}, async (2 in flight), bf16; tokens/s counts prompt + output tokens",
                "algorithmic_bytes": by,
                "hbm": {"achieved": round(by / el / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(by / el / 1e9 / HBM_PEAK_GBS, 4)}}

    short = job(short_prompts)  # (also the warm-up of the full job: every mixed-step shape has met its graph)
    out = job(num_prompts)
    out["short_run"] = short
    eng.shutdown()
    del eng
    return out


def encode_only_leg(a, dev, num_prompts=4096, length=512, max_num_seqs=32):
    """BASELINE config 4 beside the headline: bge-m3 shapes (XLM-RoBERTa-large: 24 layers, hidden 1024, 16 heads of
    64) through the prefill-only engine (light_vllm.encode_only workflow: varlen bidirectional attention + the fused
    add + LayerNorm / GELU kernels, library GEMMs), two steps in flight, dense (CLS) embeddings out.  4 096 prompts =
    128 steps = 1.4 s (the reference's benchmarks/benchmark_bge-m3.py:89-100 runs 10 000; rounds 2 - 4 quoted 512 = 16
    steps, a 0.19 s region that moved +-3 % between runs and in which the pipeline's fill and drain weigh 5 %).  MFMA-bound:
    FLOPs = tokens x (2 x layer weights + 4 x length x hidden per layer)."""
    from light_vllm_amd.prefill_only import PrefillOnlySchedulerConfig
    from light_vllm_amd.prefill_only.engine import PrefillOnlyEngine
    from light_vllm_amd.prefill_only.model import EncoderConfig
    cfg = EncoderConfig.tiny() if a.tiny else EncoderConfig.bge_m3()
    if a.tiny:
        num_prompts, length, max_num_seqs = 24, 64, 4
    g = torch.Generator().manual_seed(0)
    prompts = [torch.randint(2, cfg.vocab_size, (length,), generator=g).tolist() for _ in range(num_prompts)]
    eng = PrefillOnlyEngine(cfg, PrefillOnlySchedulerConfig(max_model_len=max(length, 8), max_num_seqs=max_num_seqs,
                                                            scheduling="async"), device=dev)
    eng.encode(prompts[: 2 * max_num_seqs])  # warm up
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    res = eng.encode(prompts)
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    eng.shutdown()
    assert len(res) == num_prompts
    hid, inter, L = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
    ntok = num_prompts * length
    flops = ntok * L * (2.0 * (4 * hid * hid + 2 * hid * inter) + 4.0 * length * hid)
    steps = (num_prompts + max_num_seqs - 1) // max_num_seqs
    by = steps * L * (4 * hid * hid + 2 * hid * inter) * 2 + ntok * hid * 2 * L * 12  # weights per step + ~12 activation passes
    out = {"value": round(num_prompts / el, 1), "unit": "sequences/s", "tokens_per_s": round(ntok / el, 1),
           "ms_per_step": round(el / steps * 1e3, 4), "steps": steps,
           "config": f"BASELINE config 4: encode-only, bge-m3 shapes (L{L} hidden {hid} heads {cfg.num_attention_heads}), "
                     f"{num_prompts} prompts x {length} tokens, {max_num_seqs} per step, async (2 in flight), bf16, CLS pooling",
           "mfma": {"achieved": round(flops / el / 1e12, 1), "peak": MFMA_PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(flops / el / 1e12 / MFMA_PEAK_BF16_TFLOPS, 4)},
           "algorithmic_bytes": by,
           "hbm": {"achieved": round(by / el / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": round(by / el / 1e9 / HBM_PEAK_GBS, 4)}}
    del eng
    return out


def largest_divisor_at_most(n, k):
    """Largest d <= k with n % d == 0 (>= 1)."""
    for d in range(max(1, min(k, n)), 0, -1):
        if n % d == 0:
            return d
    return 1


def spawn_replicas(a):
    """`python bench.py --gpus N` started as ONE plain process: start the N ranks itself, before anything in this
    process touches a GPU -- a child `python -m torch.distributed.run` (the launcher the driver uses), whose exit
    code becomes ours.  (Never an exec: this process stays the parent.)"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def plan_steps(a):
    """(burst for the timed region, burst for the warm-up, steps in flight): bursts divide the step counts so that
    the timed region holds exactly --steps model steps; the pipeline cannot hold more engine steps than there are."""
    k_req = max(1, a.num_scheduler_steps)
    k = largest_divisor_at_most(a.steps, k_req)
    kw = largest_divisor_at_most(a.warmup, k_req) if a.warmup > 0 else 1
    on_the_fly = max(1, a.on_the_fly) if a.scheduling != "sync" else 1
    if a.scheduling == "double_buffer" and "--on-the-fly" not in sys.argv:
        on_the_fly = 3
    on_the_fly = max(1, min(on_the_fly, a.steps // k))
    return k, kw, on_the_fly


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_replicas(a))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # N > 1: before this process touches its GPU it pins itself to its GPU's NUMA cores, cut among the ranks that share
    # the node (a call, not a re-exec): eight Python schedulers on one host are the scaling risk SURVEY 8e names
    from light_vllm_amd.engine.replicas import pin_to_gpu_numa
    pin = {"pinned": False, "reason": "one replica"}
    if a.gpus > 1 and not a.no_pin:
        try:
            pin = pin_to_gpu_numa()
        except Exception as e:  # placement is an optimisation: a host that does not say where its GPUs hang runs unpinned
            pin = {"pinned": False, "reason": f"{type(e).__name__}: {e}"}
    dev = "cuda:0" if a.single_device else f"cuda:{local_rank}"
    torch.cuda.set_device(dev)

    import light_vllm_amd  # noqa: F401
    from light_vllm_amd.engine.config import CacheConfig, ModelConfig, SchedulerConfig
    from light_vllm_amd.engine.llm_engine import LLMEngine
    from light_vllm_amd.engine.replicas import ReplicaGroup

    # one process per GPU; RCCL is used for the barrier and the max-over-ranks clock only
    group = ReplicaGroup(backend=a.replica_backend, device=torch.device(dev))
    rank, world = group.rank, group.world_size
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: one rank per GPU"

    B, ctx = a.batch_size, a.context
    k, kw, on_the_fly = plan_steps(a)
    k_max = max(k, kw)
    extra = a.also_on_the_fly if (a.scheduling != "sync" and a.also_on_the_fly > on_the_fly
                                  and a.steps // k >= a.also_on_the_fly) else 0
    slots = max(on_the_fly, extra)
    n_req = B * on_the_fly
    cfg = ModelConfig.tiny() if a.tiny else ModelConfig.llama3_8b()
    cfg.pack_weights = not a.library_gemm
    cfg.fuse_decode_ops = not a.no_fusion
    cfg.rope_in_attention = not a.no_rope_in_attention
    cfg.rope_in_attention_fp8 = a.rope_in_attention_fp8
    cfg.fp8_activations_once = not a.no_fp8_activations_once
    if a.o_proj_partials_min_rows is not None:
        cfg.o_proj_partials_min_rows = a.o_proj_partials_min_rows
    cfg.quantization = a.quantization
    reps = a.repeats if a.repeats > 0 else max(1, min(5, 100 // max(1, a.steps)))
    total_steps = a.steps * reps + a.warmup
    reps_extra = min(reps, 3)  # regions of the more-in-flight leg (median reported, like the headline's)
    max_len = ctx + total_steps // on_the_fly + ((a.steps + 3 * k_max) * reps_extra + 2 * k) // max(1, slots) + 2 * k_max + 8
    max_model_len = (max_len + 511) // 512 * 512
    bs = 16
    blocks = B * slots * ((max_len + bs - 1) // bs + 1) + 64
    from light_vllm_amd.engine.cache_engine import CacheEngine
    cache_cfg = CacheConfig(block_size=bs, num_gpu_blocks=blocks, num_cpu_blocks=0, cache_dtype=a.kv_cache_dtype)
    if a.kv_block_pad_bytes is not None:
        cache_cfg.block_pad_bytes = a.kv_block_pad_bytes
    engine = LLMEngine(cfg, cache_cfg,
                       SchedulerConfig(max_num_batched_tokens=max(8192, B), max_num_seqs=B,
                                       max_model_len=max_model_len, scheduling=a.scheduling,
                                       max_num_on_the_fly=slots, use_v2_block_manager=k_max > 1,
                                       num_scheduler_steps=k_max, gemm_workgroups=a.gemm_workgroups),
                       device=dev, use_hip_graph=not a.no_graph,
                       decode_version=None if a.attn_version == "auto" else a.attn_version, seed=rank)
    engine.step_returns_outputs = False
    block_pad = CacheEngine.block_pad_bytes(engine.cache_config, engine.model_config)
    if a.gemm_partials_ksplit is not None:
        torch.ops._C_amd.set_tuning("gemm_partials_ksplit", a.gemm_partials_ksplit)
    g = torch.Generator().manual_seed(1234 + rank)
    for i in range(n_req):
        prompt = torch.randint(0, cfg.vocab_size, (ctx,), generator=g).tolist()
        engine.add_request(str(i), prompt, max_tokens=2 * total_steps + 100)
    engine.prefill_synthetic(seed=rank)
    # set-up, not a step: every stream's HIP graph of the step is captured now, so that the timed region
    # replays graphs whatever W is (capture otherwise happens at a slot's first step)
    engine.capture_decode_graphs(B)

    def run(n_model_steps, burst, in_flight=None):
        """n_model_steps model steps as n / burst engine steps (each `burst` model steps chained on the device);
        the last (in_flight - 1) calls only collect, so the pipeline is empty on both sides of the region.
        Returns the tokens produced."""
        n = n_model_steps // burst
        in_flight = max(1, min(in_flight or on_the_fly, n))  # the pipeline cannot hold more engine steps than there are
        engine.scheduler_config.num_scheduler_steps = burst  # lookahead slots stay at k_max - 1
        engine.scheduler_config.max_num_on_the_fly = in_flight
        # tokens are COUNTED, not inferred: what the engine appended to its sequences, cross-checked against the
        # sequences' own output lengths; and every engine step of the region must have been a burst of `burst`
        # model steps (a step that fell back to the general path would otherwise be credited `burst` tokens)
        tok0, steps0 = engine.stat_tokens_appended, engine.stat_model_steps
        len0 = sum(s.get_output_len() for grp in engine.groups.values() for s in grp.seqs)
        for i in range(n):
            if a.scheduling != "sync":
                engine.async_step(schedule_more=i < n - (in_flight - 1))
            else:
                engine.step()
        assert engine.num_on_the_fly == 0
        produced = engine.stat_tokens_appended - tok0
        len1 = sum(s.get_output_len() for grp in engine.groups.values() for s in grp.seqs)
        assert len1 - len0 == produced, (len1 - len0, produced)
        assert engine.stat_model_steps - steps0 == n_model_steps, (engine.stat_model_steps - steps0, n_model_steps, burst)
        return produced

    if a.warmup > 0:
        run(a.warmup, kw)
    # The timed region: EXACTLY --steps model steps between barrier + synchronize on both sides, the clock the
    # slowest rank's.  Run `reps` times back to back (each an empty pipeline on both sides); the line reports the
    # MEDIAN region and lists them all -- a 20-step region lasts 68 ms and moves +-1 % from run to run.
    regions = []
    for _ in range(reps):
        torch.cuda.synchronize(dev)
        group.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        tokens = run(a.steps, k)
        torch.cuda.synchronize(dev)
        group.barrier()
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        assert tokens == a.steps * B, (tokens, a.steps, B)
        el = group.max(el)                          # slowest replica's clock
        regions.append((group.sum(tokens) / el, el))  # whole-job tokens/s of this region
    value, elapsed = sorted(regions)[len(regions) // 2]
    # More steps in flight (a SchedulerConfig setting of the reference; BASELINE.md quotes the headline at 2): B more
    # sequences per extra step are admitted at the same context, one burst per slot warms the new pipeline depth,
    # then the same number of model steps is timed again.  Reported beside the headline, never as `value`.
    other = {}
    if extra:
        for i in range(n_req, B * extra):
            prompt = torch.randint(0, cfg.vocab_size, (ctx,), generator=g).tolist()
            engine.add_request(str(i), prompt, max_tokens=2 * total_steps + 100)
        engine.prefill_synthetic(seed=rank + 1000)
        run(k * extra, k, in_flight=extra)
        # whole rounds of `extra` bursts: a region whose burst count is not a multiple of the pipeline depth ends with
        # bursts that run alone at the single-stream rate (20 steps = 4 bursts of 5 at three in flight read 9.1 k
        # against 9.6 k at two for that reason alone, profiles/r04_tuning.md section 4)
        steps_x = max(1, a.steps // (k * extra)) * (k * extra)
        regions_x = []
        for _ in range(reps_extra):
            torch.cuda.synchronize(dev)
            group.barrier()
            t1 = time.perf_counter()
            tok2 = run(steps_x, k, in_flight=extra)
            torch.cuda.synchronize(dev)
            group.barrier()
            el2 = group.max(time.perf_counter() - t1)
            regions_x.append((group.sum(tok2) / el2, el2))
        v2, el2 = sorted(regions_x)[len(regions_x) // 2]
        other[f"max_num_on_the_fly={extra}"] = {"value": round(v2, 1), "unit": "tokens/s",
                                                "ms_per_step": round(el2 / steps_x * 1e3, 4), "steps": steps_x,
                                                "sequences_resident": B * extra,
                                                "regions_tokens_per_s": [round(v, 1) for v, _ in regions_x],
                                                "note": "regions of whole rounds of bursts (steps = a multiple of burst x "
                                                        "steps in flight): a region that ends with fewer bursts than the "
                                                        "pipeline is deep runs them at the single-stream rate"}
    kl = kernel_leg(engine, B, a.kernel_iters, seq_len=ctx)
    in_step = in_step_attention_leg(engine, B, ctx) if rank == 0 else None
    gm = gemm_leg(engine, B) if B <= 64 else None
    cpu = ops_base = None
    if rank == 0 and world == 1 and not a.skip_cpu_baseline:
        cpu = cpu_baseline_leg(engine, B)
    if rank == 0 and world == 1 and not a.skip_ops_baseline:
        ops_base = ops_baseline_leg(engine, B)
    pf = None
    if rank == 0 and world == 1 and not a.skip_prefill_roofline and not a.tiny:
        pf = prefill_leg(engine)
    ctx_end = sum(s.get_len() for grp in engine.scheduler.running for s in grp.seqs) / max(1, len(engine.scheduler.running))
    # HBM traffic per launch from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE and
    # --pmc WRITE_SIZE in separate runs, FETCH_SIZE doubled as the microarch guide prescribes for gfx950),
    # measured on the same kernel at seq=1024 and scaled by this launch's algorithmic bytes.
    traffic = None
    try:
        pmc_attn = "r03_pmc_attn_fp8.json" if a.kv_cache_dtype != "auto" else "r03_pmc_attn.json"
        with open(os.path.join(ROOT, "profiles", pmc_attn)) as f:
            traffic = int(json.load(f)["traffic_over_algorithmic"] * kl["algo_bytes"])
    except (OSError, KeyError, ValueError):
        pass
    engine.shutdown()
    # BASELINE configs 3, 4, 5 beside the headline (never as `value`): each on its own engine, after this one is gone
    if (rank == 0 and world == 1 and not a.skip_other_configs and a.scheduling == "async"
            and not (a.quantization or a.kv_cache_dtype != "auto")):
        import gc
        engine.worker.graph_pools = None
        del engine
        gc.collect()
        torch.cuda.empty_cache()
        for key, leg in (("config3_chunked_prefill", lambda: chunked_prefill_leg(a, dev)),
                         ("config5_fp8_weights_fp8_kv", lambda: fp8_config_leg(a, dev, B, ctx)),
                         ("config4_encode_only", lambda: encode_only_leg(a, dev)),
                         ("sampled_decode", lambda: sampled_decode_leg(a, dev, B, ctx))):
            other[key] = leg()
            gc.collect()
            torch.cuda.empty_cache()
    if rank == 0:
        achieved = kl["algo_bytes"] / kl["avg_s"] / 1e9
        line = {
            "metric": "decode tokens/sec Llama-3-8B bs=32 seq=1k; paged-attn HBM GB/s vs roofline",
            "value": round(value, 1), "unit": "tokens/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16" if not (a.quantization or a.kv_cache_dtype != "auto") else
                     "bf16" + (" + fp8 (e4m3) W8A8 projections" if a.quantization else "") +
                     (" + fp8 (e4m3) KV cache" if a.kv_cache_dtype != "auto" else ""),
            "data": "synthetic",
            "config": {"workload": "Llama-3-8B shapes (L32 H32 KVH8 D128 hidden4096 inter14336 vocab128256), "
                                   f"decode bs={B} per step, context {ctx} at the first step growing to {int(ctx_end)}, "
                                   f"block_size 16 (blocks {block_pad} bytes apart beyond their "
                                   f"size), {a.scheduling} scheduling ({on_the_fly} batches in flight, one stream "
                                   f"each), {k} model steps per engine step"
                                   + (" (advance_step on the device between them)" if k > 1 else "") +
                                   f", attention {a.attn_version}, HIP graph {'off' if a.no_graph else 'on'}, "
                                   "random-init weights, synthetic KV",
                       "global_batch": B * world, "seq_len": ctx, "parallelism": f"dp{world} (independent replicas)",
                       "num_scheduler_steps": k, "max_num_on_the_fly": on_the_fly},
            "timed_regions": {"count": reps, "value": "median", "tokens_per_s": [round(v, 1) for v, _ in regions],
                              "ms_per_step": [round(e / a.steps * 1e3, 4) for _, e in regions]},
            "roofline": {"bound": "hbm", "kernel": "paged_attention_v2 (paged_attn_mfma_kernel): "
                                                      + ("single pass, nsplit = 1 -- the batch alone fills the GPU, no "
                                                         "reduce launch, scratch untouched" if kl["nsplit"] == 1 else
                                                         f"partition pass, {kl['nsplit']} shares per context") +
                                                      f"; seq_lens = {ctx} for all {B} sequences",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": "profiles/r03_pmc_attn[_fp8].json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in "
                                           "separate passes of the same kernel at seq = 1024: 2*FETCH_SIZE + WRITE_SIZE per "
                                           "launch, scaled by this launch's algorithmic bytes; a recording, not this run)",
                         "algorithmic_bytes_per_launch": kl["algo_bytes"],
                         "avg_launch_us": round(kl["avg_s"] * 1e6, 2), "min_launch_us": round(kl["min_s"] * 1e6, 2),
                         "in_step": in_step},
            "cpu_baseline": cpu,
        }
        if world > 1:
            line["affinity_rank0"] = pin
        if other:
            line["other_settings"] = other
        if ops_base is not None:
            line["ops_baseline"] = ops_base
        if pf is not None:  # the MFMA-bound kernel of the path's next row
            line["roofline_prefill"] = pf
        if gm is not None:  # the second HBM stream of the step: one layer's four projections
            g_ach = gm["bytes_per_layer"] / gm["s_per_layer"] / 1e9
            gm_traffic = None
            try:  # PMC passes of the same kernels at the same shapes (tools/pmc_gemm.py), M = 32 only
                with open(os.path.join(ROOT, "profiles", "r03_pmc_gemm_w8a8.json" if gm["w8"] else "r03_pmc_gemm.json")) as f:
                    pm = json.load(f)
                if B == pm["M"] and not a.tiny and (not gm["w8"] or cfg.fp8_activations_once):
                    gm_traffic = sum(v["hbm_bytes_per_launch"] for v in pm["shapes"].values())
            except (OSError, KeyError, ValueError):
                pass
            line["roofline_projections"] = {
                "bound": "hbm", "kernel": ("skinny_gemm_w8a8_kernel" if gm["w8"] else "skinny_gemm_kernel") +
                                          f" (qkv, o, gate_up with its SwiGLU epilogue, down of one layer, M = {B}, at most 256 workgroups: the fewest that divide the n-tiles evenly)",
                "achieved": round(g_ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(g_ach / HBM_PEAK_GBS, 4),
                "traffic": gm_traffic, "traffic_source": "profiles/r03_pmc_gemm[_w8a8].json (2*FETCH_SIZE + WRITE_SIZE per "
                                                         "launch, the four shapes summed; M = 32; a recording)",
                "algorithmic_bytes_per_layer": gm["bytes_per_layer"],
                "us_per_layer": round(gm["s_per_layer"] * 1e6, 2), "per_shape": gm["per_shape"]}
        print(json.dumps(line), flush=True)
    group.shutdown()


if __name__ == "__main__":
    main()
