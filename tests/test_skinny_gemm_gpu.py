"""The weight-streaming decode GEMM (extension op, csrc/skinny_gemm.hip) against torch: fp32
matmul of the same bf16/f16 operands; tolerance = output rounding + accumulation-order noise."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SHAPES = [  # M, N, K
    (32, 6144, 4096), (32, 4096, 4096), (32, 28672, 4096), (32, 4096, 14336), (32, 128256, 4096),
    (1, 4096, 4096), (7, 1024, 512), (16, 512, 256), (17, 4096, 4096), (33, 2048, 4096), (64, 4096, 14336),
    (5, 48, 64), (32, 16, 32), (3, 4096, 11008), (32, 1536, 8960),
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("use_bias", [False, True])
def test_skinny_linear_matches_fp32_matmul(ops, dtype, M, N, K, use_bias):
    g = torch.Generator(device=DEV).manual_seed(M * 7 + N + K)
    x = (torch.randn(M, K, generator=g, device=DEV) * 0.5).to(dtype)
    w = (torch.randn(N, K, generator=g, device=DEV) * 0.05).to(dtype)
    b = (torch.randn(N, generator=g, device=DEV) * 0.5).to(dtype) if use_bias else None
    y = torch.ops._C_amd.skinny_linear(x, w, b)
    if N % 16 == 0 and K % 32 == 0:
        # the packed-weight path computes the same sums in the same order: bit-identical
        yp = torch.ops._C_amd.skinny_linear_packed(x, torch.ops._C_amd.pack_weight(w), b, N, K)
        assert torch.equal(y.view(torch.int16), yp.view(torch.int16))
    ref = x.float() @ w.float().T
    if b is not None:
        ref = ref + b.float()
    assert y.shape == (M, N) and y.dtype == dtype
    err = (y.float() - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= (2 ** -7 if dtype == torch.bfloat16 else 2 ** -9) * scale + 1e-3, (err, scale)


def test_strided_rows_and_fallback(ops):
    g = torch.Generator(device=DEV).manual_seed(0)
    big = (torch.randn(8, 3 * 512, generator=g, device=DEV) * 0.5).to(torch.bfloat16)
    x = big[:, 512:1024]  # row stride 1536, as a split of a fused projection
    w = (torch.randn(256, 512, generator=g, device=DEV) * 0.05).to(torch.bfloat16)
    y = torch.ops._C_amd.skinny_linear(x, w, None)
    assert torch.allclose(y.float(), x.float() @ w.float().T, atol=2e-2, rtol=2e-2)
    # outside the envelope (M > 64, odd N): falls back to the library GEMM, same result
    x2 = (torch.randn(100, 512, generator=g, device=DEV) * 0.5).to(torch.bfloat16)
    assert torch.allclose(torch.ops._C_amd.skinny_linear(x2, w, None).float(), x2.float() @ w.float().T, atol=2e-2, rtol=2e-2)
    w3 = (torch.randn(50, 512, generator=g, device=DEV) * 0.05).to(torch.bfloat16)
    assert torch.allclose(torch.ops._C_amd.skinny_linear(x, w3, None).float(), x.float() @ w3.float().T, atol=2e-2, rtol=2e-2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_fused_decode_ops_are_bit_identical_to_the_unfused_sequence(ops, dtype):
    """The three fused launches (rope + cache write; split-K sum inside add+norm; optionally SwiGLU
    inside the down projection) keep the rounding points of the operators they replace: results are bit-equal."""
    g = torch.Generator(device=DEV).manual_seed(3)
    T, H, KVH, D, BS, NB, hid, inter = 9, 8, 2, 128, 16, 12, 1024, 14336
    for neox in (True, False):
        qkv = (torch.randn(T, (H + 2 * KVH) * D, generator=g, device=DEV)).to(dtype)
        inv = 1.0 / (10000 ** (torch.arange(0, D, 2).float() / D))
        fr = torch.outer(torch.arange(512).float(), inv)
        cache = torch.cat([fr.cos(), fr.sin()], -1).to(dtype).to(DEV)
        pos = torch.randint(0, 512, (T,), generator=g, device=DEV)
        slots = torch.randperm(NB * BS, generator=g, device=DEV)[:T].to(torch.int64)
        slots[2] = -1
        kc0 = torch.randn(NB, KVH, D // 8, BS, 8, generator=g, device=DEV).to(dtype)
        vc0 = torch.randn(NB, KVH, D, BS, generator=g, device=DEV).to(dtype)
        a, b = qkv.clone(), qkv.clone()
        kc_a, vc_a, kc_b, vc_b = kc0.clone(), vc0.clone(), kc0.clone(), vc0.clone()
        qa, ka, va = a.split([H * D, KVH * D, KVH * D], dim=-1)
        ops.rotary_embedding(pos, qa, ka, D, cache, neox)
        ops.reshape_and_cache(ka.view(T, KVH, D), va.view(T, KVH, D), kc_a, vc_a, slots, "auto", 1.0, 1.0)
        qb, kb, vb = b.split([H * D, KVH * D, KVH * D], dim=-1)
        assert torch.ops._C_amd.rotary_embedding_and_cache(pos, qb, kb, vb, D, cache, neox, kc_b, vc_b, slots)
        for x, y in ((a, b), (kc_a, kc_b), (vc_a, vc_b)):
            assert torch.equal(x.view(torch.int16), y.view(torch.int16))
        # the same into an fp8 cache: rope, then reshape_and_cache(fp8) == the fused launch with kv "fp8"
        kc8 = torch.randint(0, 255, (NB, KVH, D // 16, BS, 16), generator=g, device=DEV, dtype=torch.uint8)
        vc8 = torch.randint(0, 255, (NB, KVH, D, BS), generator=g, device=DEV, dtype=torch.uint8)
        kc8_a, vc8_a, kc8_b, vc8_b = kc8.clone(), vc8.clone(), kc8.clone(), vc8.clone()
        a8, b8 = qkv.clone(), qkv.clone()
        qa, ka, va = a8.split([H * D, KVH * D, KVH * D], dim=-1)
        ops.rotary_embedding(pos, qa, ka, D, cache, neox)
        ops.reshape_and_cache(ka.view(T, KVH, D), va.view(T, KVH, D), kc8_a, vc8_a, slots, "fp8", 0.5, 1.7)
        qb, kb, vb = b8.split([H * D, KVH * D, KVH * D], dim=-1)
        assert torch.ops._C_amd.rotary_embedding_and_cache(pos, qb, kb, vb, D, cache, neox, kc8_b, vc8_b, slots,
                                                           "fp8", 0.5, 1.7)
        assert torch.equal(a8.view(torch.int16), b8.view(torch.int16))
        assert torch.equal(kc8_a, kc8_b) and torch.equal(vc8_a, vc8_b)
    # SwiGLU + down projection + add + norm
    gate_up = (torch.randn(T, 2 * inter, generator=g, device=DEV)).to(dtype)
    w = (torch.randn(hid, inter, generator=g, device=DEV) * 0.02).to(dtype)
    wp = torch.ops._C_amd.pack_weight(w)
    res = torch.randn(T, hid, generator=g, device=DEV).to(dtype)
    nw = (1 + 0.1 * torch.randn(hid, generator=g, device=DEV)).to(dtype)
    act = torch.empty(T, inter, dtype=dtype, device=DEV)
    ops.silu_and_mul(act, gate_up)
    y = torch.ops._C_amd.skinny_linear_packed(act, wp, None, hid, inter)
    res_a = res.clone()
    ops.fused_add_rms_norm(y, res_a, nw, 1e-5)
    for swiglu, x in ((True, gate_up), (False, act)):
        partials = torch.ops._C_amd.skinny_linear_packed_partials(x, wp, hid, inter, swiglu)
        assert partials.shape == (4, T, hid) and partials.dtype == torch.float32
        out_b, res_b = torch.empty_like(res), res.clone()
        torch.ops._C_amd.fused_add_rms_norm_splitk(out_b, res_b, partials, nw, 1e-5)
        assert torch.equal(res_a.view(torch.int16), res_b.view(torch.int16))
        assert torch.equal(y.view(torch.int16), out_b.view(torch.int16))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,inter,K", [(32, 14336, 4096), (1, 512, 256), (17, 1040, 1024), (64, 2048, 2048),
                                       (33, 11008, 4096), (5, 48, 64), (32, 16, 32)])
@pytest.mark.parametrize("wgs", [256, 128, 9])
def test_swiglu_epilogue_is_bit_identical_to_projection_then_silu_and_mul(ops, dtype, M, inter, K, wgs):
    g = torch.Generator(device=DEV).manual_seed(M + inter + K)
    x = (torch.randn(M, K, generator=g, device=DEV) * 0.5).to(dtype)
    w = (torch.randn(2 * inter, K, generator=g, device=DEV) * 0.05).to(dtype)
    b = (torch.randn(2 * inter, generator=g, device=DEV) * 0.5).to(dtype)
    wp = torch.ops._C_amd.pack_weight(w)
    torch.ops._C_amd.set_tuning("gemm_workgroups", wgs)
    try:
        for bias in (None, b):
            gate_up = torch.ops._C_amd.skinny_linear_packed(x, wp, bias, 2 * inter, K)
            ref = torch.empty(M, inter, dtype=dtype, device=DEV)
            ops.silu_and_mul(ref, gate_up)
            out = torch.ops._C_amd.skinny_linear_packed_swiglu(x, wp, bias, 2 * inter, K)
            assert out.shape == (M, inter)
            assert torch.equal(out.view(torch.int16), ref.view(torch.int16))
    finally:
        torch.ops._C_amd.set_tuning("gemm_workgroups", 256)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(128, 6144, 4096), (128, 4096, 14336), (65, 28672, 4096), (96, 4096, 4096),
                                   (129, 4096, 4096), (256, 6144, 4096), (200, 1536, 8960), (1, 4096, 4096),
                                   (77, 48, 64), (128, 16, 32), (130, 272, 1056), (64, 128256, 4096)])
@pytest.mark.parametrize("use_bias", [False, True])
def test_stream_linear_matches_fp32_matmul(ops, dtype, M, N, K, use_bias):
    """The 65..256-row weight-streaming kernel (X through LDS, K split over workgroups for small N)
    against the fp32 product of the same operands; strided rows as the engine passes them."""
    g = torch.Generator(device=DEV).manual_seed(M * 7 + N + K)
    big = (torch.randn(M, K + 64, generator=g, device=DEV) * 0.5).to(dtype)
    x = big[:, 32:32 + K]  # a strided view
    w = (torch.randn(N, K, generator=g, device=DEV) * 0.05).to(dtype)
    b = (torch.randn(N, generator=g, device=DEV) * 0.5).to(dtype) if use_bias else None
    wp = torch.ops._C_amd.pack_weight(w)
    for wgs in (256, 128):
        torch.ops._C_amd.set_tuning("gemm_workgroups", wgs)
        try:
            y = torch.ops._C_amd.stream_linear_packed(x, wp, b, N, K)
        finally:
            torch.ops._C_amd.set_tuning("gemm_workgroups", 256)
        ref = x.float() @ w.float().T
        if b is not None:
            ref = ref + b.float()
        assert y.shape == (M, N) and y.dtype == dtype
        err = (y.float() - ref).abs().max().item()
        scale = ref.abs().max().item()
        assert err <= (2 ** -7 if dtype == torch.bfloat16 else 2 ** -9) * scale + 1e-3, (err, scale, wgs)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(32, 128256, 4096), (1, 512, 256), (17, 32000, 1024), (64, 4096, 2048),
                                   (33, 1040, 8192), (5, 48, 64), (32, 16, 32), (64, 128256, 4096), (40, 48, 8192)])
@pytest.mark.parametrize("wgs", [256, 128, 9])
def test_argmax_epilogue_gives_torch_argmax_of_the_projection(ops, dtype, M, N, K, wgs):
    """tokens = argmax of the rounded logits, ties to the smaller index -- also with many exact ties
    (duplicated weight rows give bit-equal logits in different workgroups, waves and lanes)."""
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    x = (torch.randn(M, K, generator=g, device=DEV) * 0.5).to(dtype)
    w = (torch.randn(N, K, generator=g, device=DEV) * 0.05).to(dtype)
    if N >= 48:  # copies of the strongest rows scattered over the vocabulary: exact ties across tiles
        top = (x.float() @ w.float().T).argmax(-1)
        for j, t in enumerate(top.tolist()[:8]):
            for dst in ((t + 16 * (j + 1)) % N, (t * 7 + 3) % N, N - 1 - j):
                w[dst] = w[t]
    wp = torch.ops._C_amd.pack_weight(w)
    torch.ops._C_amd.set_tuning("gemm_workgroups", wgs)
    try:
        logits = torch.ops._C_amd.skinny_linear_packed(x, wp, None, N, K)
        tokens = torch.ops._C_amd.skinny_linear_packed_argmax(x, wp, N, K)
    finally:
        torch.ops._C_amd.set_tuning("gemm_workgroups", 256)
    want = torch.argmax(logits, dim=-1)
    # torch.argmax documents "first maximal value"; compare values too in case a backend differs
    assert tokens.dtype == torch.int64 and tokens.shape == (M,)
    picked = logits.gather(1, tokens[:, None])[:, 0]
    assert torch.equal(picked, logits.max(dim=-1).values)
    first = (logits == logits.max(dim=-1, keepdim=True).values).float().argmax(-1)
    assert torch.equal(tokens, first)
    assert torch.equal(first, want), "torch.argmax no longer returns the first maximal index on this backend"


# ------------------------------------------------ rope + cache write inside the attention launch
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [
    # S, H, KVH, D, BS, lens (context INCLUDING the step's new token), forced shares
    (5, 32, 8, 128, 16, [1024, 1, 17, 512, 700], 0),
    (4, 8, 2, 128, 32, [300, 33, 64, 1], 0),
    (3, 4, 4, 64, 16, [100, 2050, 16], 0),       # long context: the library cuts shares by itself
    (3, 16, 1, 256, 16, [40, 700, 129], 0),      # GQA 16
    (6, 8, 2, 128, 16, [1500, 16, 17, 1, 0, 999], 3),  # forced shares; an empty padding row (slot -1)
    (2, 14, 2, 128, 16, [77, 513], 2),           # GQA 7
], ids=["metric", "bs32", "d64_long", "d256_gqa16", "forced3_padding", "gqa7"])
def test_rope_cache_attention_in_one_launch_is_bit_identical(ops, dtype, case):
    """rope_cache_paged_attention == rotary_embedding + reshape_and_cache + paged_attention_v2 (qwen2.py:151-154 +
    paged_attn.py:65-85,87-191), bit for bit: attention output, key cache, value cache -- every byte, the
    untouched blocks included."""
    S, H, KVH, D, BS, lens, forced = case
    from helpers import make_paged_inputs, v2_scratch
    inp = make_paged_inputs(S, H, KVH, D, BS, [max(n, 1) for n in lens], dtype=dtype, seed=S * 3 + D)
    g = torch.Generator().manual_seed(17)
    qkv = (torch.randn(S, (H + 2 * KVH) * D, generator=g) * 0.5).to(dtype).to(DEV)
    max_pos = 4096
    inv = 1.0 / (10000 ** (torch.arange(0, D, 2).float() / D))
    fr = torch.outer(torch.arange(max_pos).float(), inv)
    cos_sin = torch.cat([fr.cos(), fr.sin()], -1).to(dtype).to(DEV)
    seq_lens = torch.tensor(lens, dtype=torch.int32, device=DEV)
    positions = (seq_lens.long() - 1).clamp_min(0)
    bt = inp["block_tables"].to(DEV)
    slots = torch.tensor([int(inp["block_tables"][i, (n - 1) // BS]) * BS + (n - 1) % BS if n > 0 else -1
                          for i, n in enumerate(lens)], dtype=torch.int64, device=DEV)
    max_len = max(lens)
    scale = D ** -0.5
    torch.ops._C_amd.set_tuning("attn_splits", forced)
    try:
        results = []
        for fused in (False, True):
            kc, vc = inp["key_cache"].to(DEV).clone(), inp["value_cache"].to(DEV).clone()
            x = qkv.clone()
            q, k, v = x.split([H * D, KVH * D, KVH * D], dim=-1)
            es, ml, tmp = v2_scratch(S, H, D, max_len, dtype, DEV)
            out = torch.zeros(S, H, D, dtype=dtype, device=DEV)
            if fused:
                ok = torch.ops._C_amd.rope_cache_paged_attention(out, es, ml, tmp, positions, q, k, v, D, cos_sin, True,
                                                                 kc, vc, slots, KVH, scale, bt, seq_lens, BS, max_len,
                                                                 "auto")
                assert ok
            else:
                ops.rotary_embedding(positions, q, k, D, cos_sin, True)
                ops.reshape_and_cache(k.view(S, KVH, D), v.view(S, KVH, D), kc, vc, slots, "auto", 1.0, 1.0)
                ops.paged_attention_v2(out, es, ml, tmp, q.view(S, H, D), kc, vc, KVH, scale, bt, seq_lens, BS,
                                       max_len, None, "auto", 1.0, 1.0)
            results.append((out, kc, vc))
    finally:
        torch.ops._C_amd.set_tuning("attn_splits", 0)
    (o0, k0, v0), (o1, k1, v1) = results
    assert torch.equal(k0, k1) and torch.equal(v0, v1)
    assert torch.equal(o0, o1), (o0.float() - o1.float()).abs().max()


def test_rope_cache_attention_declines_outside_its_envelope(ops):
    S, H, KVH, D, BS = 2, 4, 2, 80, 16  # head size 80: the NeoX halves do not fall on k-slice boundaries
    q = torch.zeros(S, H * D, dtype=torch.bfloat16, device=DEV)
    k = torch.zeros(S, KVH * D, dtype=torch.bfloat16, device=DEV)
    kc = torch.zeros(4, KVH, D // 8, BS, 8, dtype=torch.bfloat16, device=DEV)
    vc = torch.zeros(4, KVH, D, BS, dtype=torch.bfloat16, device=DEV)
    es = torch.zeros(S, H, 1, device=DEV)
    ok = torch.ops._C_amd.rope_cache_paged_attention(
        torch.zeros(S, H, D, dtype=torch.bfloat16, device=DEV), es, es.clone(), torch.zeros(S, H, 1, D, dtype=torch.bfloat16, device=DEV),
        torch.zeros(S, dtype=torch.int64, device=DEV), q, k, k.clone(), D, torch.zeros(16, D, dtype=torch.bfloat16, device=DEV), True,
        kc, vc, torch.zeros(S, dtype=torch.int64, device=DEV), KVH, 0.1, torch.zeros(S, 2, dtype=torch.int32, device=DEV),
        torch.ones(S, dtype=torch.int32, device=DEV), BS, 16, "auto")
    assert ok is False


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("scales", [(1.0, 1.0), (0.5, 2.0)])
@pytest.mark.parametrize("case", [
    (5, 32, 8, 128, 16, [1024, 1, 17, 512, 700], 0),
    (3, 8, 2, 128, 32, [300, 33, 64], 0),
    (4, 16, 1, 256, 16, [40, 700, 129, 0], 2),
], ids=["metric", "bs32", "d256_gqa16_forced2_padding"])
def test_rope_cache_attention_in_one_launch_over_an_fp8_cache_is_bit_identical(ops, dtype, scales, case):
    """The same fusion over an fp8 (e4m3fn) KV cache: the rotated key and the value are quantised with their scales on
    the way into the cache exactly as reshape_and_cache("fp8") does, and the new token is attended to through the same
    fp8 bytes: output and both caches equal rotary_embedding + reshape_and_cache + paged_attention_v2 bit for bit."""
    S, H, KVH, D, BS, lens, forced = case
    ks, vs = scales
    g = torch.Generator().manual_seed(S * 5 + D)
    NB = sum((max(n, 1) + BS - 1) // BS for n in lens) + 5
    kc0 = (torch.randn(NB, KVH, D // 16, BS, 16, generator=g) * 0.5 / ks).to(torch.float8_e4m3fn).view(torch.uint8).to(DEV)
    vc0 = (torch.randn(NB, KVH, D, BS, generator=g) * 0.5 / vs).to(torch.float8_e4m3fn).view(torch.uint8).to(DEV)
    perm = torch.randperm(NB, generator=g).tolist()
    W = max((max(n, 1) + BS - 1) // BS for n in lens)
    bt = torch.zeros(S, W, dtype=torch.int32)
    for i, n in enumerate(lens):
        nb = (max(n, 1) + BS - 1) // BS
        bt[i, :nb] = torch.tensor(perm[:nb], dtype=torch.int32)
        perm = perm[nb:]
    qkv = (torch.randn(S, (H + 2 * KVH) * D, generator=g) * 0.5).to(dtype).to(DEV)
    inv = 1.0 / (10000 ** (torch.arange(0, D, 2).float() / D))
    fr = torch.outer(torch.arange(4096).float(), inv)
    cos_sin = torch.cat([fr.cos(), fr.sin()], -1).to(dtype).to(DEV)
    seq_lens = torch.tensor(lens, dtype=torch.int32, device=DEV)
    positions = (seq_lens.long() - 1).clamp_min(0)
    slots = torch.tensor([int(bt[i, (n - 1) // BS]) * BS + (n - 1) % BS if n > 0 else -1 for i, n in enumerate(lens)],
                         dtype=torch.int64, device=DEV)
    bt = bt.to(DEV)
    max_len = max(lens)
    from helpers import v2_scratch
    torch.ops._C_amd.set_tuning("attn_splits", forced)
    try:
        results = []
        for fused in (False, True):
            kc, vc = kc0.clone(), vc0.clone()
            x = qkv.clone()
            q, k, v = x.split([H * D, KVH * D, KVH * D], dim=-1)
            es, ml, tmp = v2_scratch(S, H, D, max_len, dtype, DEV)
            out = torch.zeros(S, H, D, dtype=dtype, device=DEV)
            if fused:
                assert torch.ops._C_amd.rope_cache_paged_attention(out, es, ml, tmp, positions, q, k, v, D, cos_sin, True, kc, vc,
                                                                   slots, KVH, D ** -0.5, bt, seq_lens, BS, max_len, "fp8", ks, vs)
            else:
                ops.rotary_embedding(positions, q, k, D, cos_sin, True)
                ops.reshape_and_cache(k.view(S, KVH, D), v.view(S, KVH, D), kc, vc, slots, "fp8", ks, vs)
                ops.paged_attention_v2(out, es, ml, tmp, q.view(S, H, D), kc, vc, KVH, D ** -0.5, bt, seq_lens, BS, max_len,
                                       None, "fp8", ks, vs)
            results.append((out, kc, vc))
    finally:
        torch.ops._C_amd.set_tuning("attn_splits", 0)
    (o0, k0, v0), (o1, k1, v1) = results
    assert torch.equal(k0, k1) and torch.equal(v0, v1)
    assert torch.equal(o0, o1), (o0.float() - o1.float()).abs().max()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T,H,KVH,D,K,kv", [(64, 32, 8, 128, 4096, "auto"), (33, 8, 2, 128, 4096, "auto"),
                                            (48, 4, 4, 64, 4096, "auto"), (57, 8, 2, 128, 4096, "fp8"),
                                            (40, 6, 2, 256, 8192, "auto")])
@pytest.mark.parametrize("use_bias", [False, True])
def test_qkv_splitk_reduce_inside_the_rope_and_cache_launch_is_bit_identical(ops, dtype, T, H, KVH, D, K, kv, use_bias):
    """Mixed steps of 33..64 rows (round 4): the QKV projection splits K over workgroups and leaves fp32 slabs; the rope +
    cache-write launch sums them (lvllm_rotary_embedding_and_cache_splitk) instead of a reduce launch of their own.  The
    qkv rows (rotated q and k, v) and both caches are bit for bit what projection (with its own reduce pass) ->
    rotary_embedding_and_cache leaves; padding slots (-1) write no cache line but still get their rows."""
    g = torch.Generator(device=DEV).manual_seed(11 + T)
    N = (H + 2 * KVH) * D
    BS, NB = 16, 9
    x = (torch.randn(T, K, generator=g, device=DEV) * 0.5).to(dtype)
    w = (torch.randn(N, K, generator=g, device=DEV) * 0.03).to(dtype)
    bias = (torch.randn(N, generator=g, device=DEV) * 0.1).to(dtype) if use_bias else None
    wp = torch.ops._C_amd.pack_weight(w)
    inv = 1.0 / (10000 ** (torch.arange(0, D, 2).float() / D))
    fr = torch.outer(torch.arange(700).float(), inv)
    cos_sin = torch.cat([fr.cos(), fr.sin()], -1).to(dtype).to(DEV)
    pos = torch.randint(0, 700, (T,), generator=g, device=DEV)
    slots = torch.randperm(NB * BS, generator=g, device=DEV)[:T].to(torch.int64)
    slots[T // 2] = -1
    if kv == "fp8":
        kc0 = torch.randint(0, 255, (NB, KVH, D // 16, BS, 16), generator=g, device=DEV, dtype=torch.uint8)
        vc0 = torch.randint(0, 255, (NB, KVH, D, BS), generator=g, device=DEV, dtype=torch.uint8)
        ks, vs = 0.5, 1.7
    else:
        kc0 = torch.randn(NB, KVH, D // 8, BS, 8, generator=g, device=DEV).to(dtype)
        vc0 = torch.randn(NB, KVH, D, BS, generator=g, device=DEV).to(dtype)
        ks = vs = 1.0
    # the two launches
    qkv_a = torch.ops._C_amd.skinny_linear_packed(x, wp, bias, N, K)
    kc_a, vc_a = kc0.clone(), vc0.clone()
    qa, ka, va = qkv_a.split([H * D, KVH * D, KVH * D], dim=-1)
    assert torch.ops._C_amd.rotary_embedding_and_cache(pos, qa, ka, va, D, cos_sin, True, kc_a, vc_a, slots, kv, ks, vs)
    # the one
    part = torch.ops._C_amd.skinny_linear_packed_partials(x, wp, N, K, False)
    assert part.shape[0] >= 2 and part.dtype == torch.float32  # K is split at these row counts
    qkv_b = torch.full((T, N), float("nan"), dtype=dtype, device=DEV)
    kc_b, vc_b = kc0.clone(), vc0.clone()
    assert torch.ops._C_amd.rotary_embedding_and_cache_splitk(pos, qkv_b, part, bias, H, KVH, D, cos_sin, True, kc_b, vc_b,
                                                              slots, kv, ks, vs)
    torch.cuda.synchronize()
    assert torch.equal(qkv_a.view(torch.int16), qkv_b.view(torch.int16))
    assert torch.equal(kc_a.view(torch.uint8), kc_b.view(torch.uint8)) and torch.equal(vc_a.view(torch.uint8), vc_b.view(torch.uint8))
