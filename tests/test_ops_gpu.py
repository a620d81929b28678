"""Parity of the HIP operators (through torch.ops._C* -> C-ABI -> gfx950 kernels) against the
CPU oracle (oracle/paged_ops_oracle.c) on the same seeded inputs.

Bars: byte movement bit-exact; rms_norm / rope / silu within 1 ulp of the
element type typically and 2 ulp at most; attention max-abs <= 2e-2 * max|out| and cosine
>= 0.999 per (sequence, head), v1 vs v2 of the build within 2 bf16 ulp.
"""
import math

import pytest
import torch

from helpers import dense_attention_fp64, make_paged_inputs, v2_scratch
from oracle import oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def ulp_diff(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """|a-b| in units of the element type's ulp at max(|a|,|b|)."""
    assert a.dtype == b.dtype
    mant = {torch.bfloat16: 8, torch.float16: 11, torch.float32: 24}[a.dtype]
    af, bf = a.double().cpu(), b.double().cpu()
    mag = torch.maximum(af.abs(), bf.abs()).clamp_min(1e-30)
    ulp = torch.pow(2.0, torch.floor(torch.log2(mag)) - (mant - 1))
    return (af - bf).abs() / ulp


def rows_close_in_ulp(a, b, n_ulp):
    """|a-b| <= n_ulp ulps of the element type at the magnitude of the row's largest element
    (rows = last dim): the natural bar for two roundings of the same weighted sum."""
    mant = {torch.bfloat16: 8, torch.float16: 11, torch.float32: 24}[a.dtype]
    af, bf = a.double().cpu(), b.double().cpu()
    mag = torch.maximum(af.abs(), bf.abs()).amax(dim=-1, keepdim=True).clamp_min(1e-30)
    ulp = torch.pow(2.0, torch.floor(torch.log2(mag)) - (mant - 1))
    return bool(((af - bf).abs() <= n_ulp * ulp).all())


def check_attention(out, ref_out, d64=None, tol=2e-2):
    o, r = out.double().cpu(), ref_out.double().cpu()
    scale = r.abs().max().clamp_min(1e-6)
    assert ((o - r).abs().max() / scale).item() <= tol
    cos = torch.nn.functional.cosine_similarity(o.flatten(1), r.flatten(1), dim=1) if o.dim() == 2 else \
        torch.nn.functional.cosine_similarity(o, r, dim=-1)
    nz = r.abs().sum(-1) > 0
    assert (cos[nz] >= 0.999).all(), cos[nz].min()
    if d64 is not None:
        assert ((o - d64).abs().max() / scale).item() <= tol


def to_dev(inp):
    return {k: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for k, v in inp.items()}


# ---------------------------------------------------------------- cache ops (bit-exact)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("block_size", [8, 16, 32])
@pytest.mark.parametrize("num_tokens,num_heads,head_size", [(1, 1, 64), (37, 8, 128), (256, 4, 80), (3, 2, 256)])
def test_reshape_and_cache_bit_exact(ops, dtype, block_size, num_tokens, num_heads, head_size):
    g = torch.Generator().manual_seed(num_tokens * 131 + head_size)
    x = 16 // torch.tensor([], dtype=dtype).element_size()
    num_blocks = (num_tokens + block_size - 1) // block_size + 5
    # k, v as strided views of a fused qkv row, as the model passes them (qwen2.py:151-152)
    qkv = torch.randn(num_tokens, 3 * num_heads * head_size, generator=g).to(dtype)
    key = qkv[:, num_heads * head_size: 2 * num_heads * head_size].view(num_tokens, num_heads, head_size)
    value = qkv[:, 2 * num_heads * head_size:].view(num_tokens, num_heads, head_size)
    slots = torch.randperm(num_blocks * block_size, generator=g)[:num_tokens].to(torch.int64)
    if num_tokens > 2:
        slots[1] = -1  # padding token: skipped (cache_kernels.cu:166-169)
    kc = torch.randn(num_blocks, num_heads, head_size // x, block_size, x, generator=g).to(dtype)
    vc = torch.randn(num_blocks, num_heads, head_size, block_size, generator=g).to(dtype)
    kc_o, vc_o = kc.clone(), vc.clone()
    oracle.reshape_and_cache(key, value, kc_o, vc_o, slots)
    qkv_d = qkv.to(DEV)
    key_d = qkv_d[:, num_heads * head_size: 2 * num_heads * head_size].view(num_tokens, num_heads, head_size)
    value_d = qkv_d[:, 2 * num_heads * head_size:].view(num_tokens, num_heads, head_size)
    kc_d, vc_d = kc.to(DEV), vc.to(DEV)
    ops.reshape_and_cache(key_d, value_d, kc_d, vc_d, slots.to(DEV), "auto", 1.0, 1.0)
    assert torch.equal(kc_d.cpu().view(torch.uint8), kc_o.view(torch.uint8))
    assert torch.equal(vc_d.cpu().view(torch.uint8), vc_o.view(torch.uint8))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_reshape_and_cache_flash_bit_exact(ops, dtype):
    g = torch.Generator().manual_seed(5)
    T, H, D, BS, NB = 45, 4, 128, 16, 9
    key = torch.randn(T, H, D, generator=g).to(dtype)
    value = torch.randn(T, H, D, generator=g).to(dtype)
    slots = torch.randperm(NB * BS, generator=g)[:T].to(torch.int64)
    slots[3] = -1
    kc = torch.randn(NB, BS, H, D, generator=g).to(dtype)
    vc = torch.randn(NB, BS, H, D, generator=g).to(dtype)
    kc_o, vc_o = kc.clone(), vc.clone()
    oracle.reshape_and_cache_flash(key, value, kc_o, vc_o, slots)
    kc_d, vc_d = kc.to(DEV), vc.to(DEV)
    ops.reshape_and_cache_flash(key.to(DEV), value.to(DEV), kc_d, vc_d, slots.to(DEV), "auto", 1.0, 1.0)
    assert torch.equal(kc_d.cpu().view(torch.uint8), kc_o.view(torch.uint8))
    assert torch.equal(vc_d.cpu().view(torch.uint8), vc_o.view(torch.uint8))


@pytest.mark.parametrize("num_layers,num_pairs", [(1, 1), (4, 7), (32, 3)])
def test_copy_blocks_bit_exact(ops, num_layers, num_pairs):
    g = torch.Generator().manual_seed(7)
    NB, KVH, D, BS = 40, 2, 64, 16
    kcs = [torch.randn(NB, KVH, D // 8, BS, 8, generator=g).to(torch.bfloat16) for _ in range(num_layers)]
    vcs = [torch.randn(NB, KVH, D, BS, generator=g).to(torch.bfloat16) for _ in range(num_layers)]
    perm = torch.randperm(NB, generator=g)
    mapping = torch.stack([perm[:num_pairs], perm[num_pairs: 2 * num_pairs]], dim=1).to(torch.int64)
    kcs_o, vcs_o = [t.clone() for t in kcs], [t.clone() for t in vcs]
    oracle.copy_blocks(kcs_o, vcs_o, mapping)
    kcs_d, vcs_d = [t.to(DEV) for t in kcs], [t.to(DEV) for t in vcs]
    ops.copy_blocks(kcs_d, vcs_d, mapping.to(DEV))
    for a, b in zip(kcs_d + vcs_d, kcs_o + vcs_o):
        assert torch.equal(a.cpu().view(torch.int16), b.view(torch.int16))
    # second call with the same cache set reuses the cached pointer table
    mapping2 = torch.tensor([[int(mapping[0, 1]), int(perm[-1])]], dtype=torch.int64)
    oracle.copy_blocks(kcs_o, vcs_o, mapping2)
    ops.copy_blocks(kcs_d, vcs_d, mapping2.to(DEV))
    for a, b in zip(kcs_d + vcs_d, kcs_o + vcs_o):
        assert torch.equal(a.cpu().view(torch.int16), b.view(torch.int16))


def test_swap_blocks_round_trip(ops):
    """swap out GPU->CPU(pinned), swap in CPU->GPU, D2D; block_mapping on CPU
    (cache_kernels.cu:40-43); runs of consecutive blocks are merged into one copy."""
    g = torch.Generator().manual_seed(9)
    NB, KVH, D, BS = 64, 2, 128, 16
    gpu = torch.randn(NB, KVH * D * BS, generator=g).to(torch.bfloat16).to(DEV)
    cpu = torch.zeros(NB, KVH * D * BS, dtype=torch.bfloat16).pin_memory()
    mapping = torch.tensor([[3, 10], [4, 11], [5, 12], [40, 2], [7, 63]], dtype=torch.int64)
    exp = cpu.clone()
    oracle.swap_blocks(gpu.cpu(), exp, mapping)
    ops.swap_blocks(gpu, cpu, mapping)
    torch.cuda.synchronize()
    assert torch.equal(cpu.view(torch.int16), exp.view(torch.int16))
    gpu2 = torch.zeros_like(gpu)
    back = torch.tensor([[10, 0], [11, 1], [12, 2], [2, 5]], dtype=torch.int64)
    exp2 = torch.zeros(NB, KVH * D * BS, dtype=torch.bfloat16)
    oracle.swap_blocks(cpu, exp2, back)
    ops.swap_blocks(cpu, gpu2, back)
    assert torch.equal(gpu2.cpu().view(torch.int16), exp2.view(torch.int16))
    gpu3 = torch.zeros_like(gpu)
    ops.swap_blocks(gpu, gpu3, mapping)
    exp3 = torch.zeros(NB, KVH * D * BS, dtype=torch.bfloat16)
    oracle.swap_blocks(gpu.cpu(), exp3, mapping)
    assert torch.equal(gpu3.cpu().view(torch.int16), exp3.view(torch.int16))
    with pytest.raises(RuntimeError):
        ops.swap_blocks(gpu, cpu, mapping.to(DEV))  # block_mapping must be on CPU


# ---------------------------------------------------------------- norm / rope / activation
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("num_tokens,hidden", [(1, 64), (32, 4096), (7, 8192), (300, 1024), (5, 5120), (3, 100), (2, 16384)])
def test_rms_norm(ops, dtype, num_tokens, hidden):
    g = torch.Generator().manual_seed(hidden + num_tokens)
    x = torch.randn(num_tokens, hidden, generator=g).to(dtype)
    w = (1.0 + 0.1 * torch.randn(hidden, generator=g)).to(dtype)
    exp = torch.empty_like(x)
    oracle.rms_norm(exp, x, w, 1e-6)
    out = torch.empty_like(x, device=DEV)
    ops.rms_norm(out, x.to(DEV), w.to(DEV), 1e-6)
    u = ulp_diff(out.cpu(), exp)
    # fp32: the 4096-term fp32 variance sum is order dependent (~sqrt(n) * 2^-24 relative)
    max_ulp = 2 if dtype != torch.float32 else 256
    assert u.max() <= max_ulp, u.max()
    if dtype != torch.float32:
        assert (u > 1).float().mean() < 1e-3

    res = torch.randn(num_tokens, hidden, generator=g).to(dtype)
    x_o, r_o = x.clone(), res.clone()
    oracle.fused_add_rms_norm(x_o, r_o, w, 1e-6)
    x_d, r_d = x.to(DEV), res.to(DEV)
    ops.fused_add_rms_norm(x_d, r_d, w.to(DEV), 1e-6)
    if dtype != torch.float32:  # the residual add is a single rounded op: bit-exact
        assert torch.equal(r_d.cpu().view(torch.int16), r_o.view(torch.int16))
    else:
        assert torch.equal(r_d.cpu(), r_o)
    u = ulp_diff(x_d.cpu(), x_o)
    assert u.max() <= max_ulp, u.max()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("is_neox", [True, False])
@pytest.mark.parametrize("num_tokens,H,KVH,D,rot", [(32, 32, 8, 128, 128), (5, 4, 4, 64, 32), (3, 28, 4, 128, 128), (17, 8, 1, 80, 80), (2, 2, 2, 96, 20)])
def test_rotary_embedding(ops, dtype, is_neox, num_tokens, H, KVH, D, rot):
    g = torch.Generator().manual_seed(rot + num_tokens)
    max_pos = 4096
    inv_freq = 1.0 / (10000 ** (torch.arange(0, rot, 2).float() / rot))
    freqs = torch.outer(torch.arange(max_pos).float(), inv_freq)
    cache = torch.cat([freqs.cos(), freqs.sin()], dim=-1).to(dtype)  # backends/rotary_embedding.py:105-114
    positions = torch.randint(0, max_pos, (num_tokens,), generator=g, dtype=torch.int64)
    qkv = torch.randn(num_tokens, (H + 2 * KVH) * D, generator=g).to(dtype)
    qkv_o = qkv.clone()
    q_o, k_o = qkv_o[:, : H * D], qkv_o[:, H * D: (H + KVH) * D]
    oracle.rotary_embedding(positions, q_o, k_o, D, cache, is_neox)
    qkv_d = qkv.to(DEV)
    q_d, k_d = qkv_d[:, : H * D], qkv_d[:, H * D: (H + KVH) * D]
    ops.rotary_embedding(positions.to(DEV), q_d, k_d, D, cache.to(DEV), is_neox)
    if dtype == torch.float32:
        assert ulp_diff(qkv_d.cpu(), qkv_o).max() <= 4  # fp32: FMA contraction may differ
    else:
        # every operation is individually rounded to T on both sides: bit-exact
        assert torch.equal(qkv_d.cpu().view(torch.int16), qkv_o.view(torch.int16))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("num_tokens,d", [(32, 14336), (1, 128), (7, 11008), (300, 1000), (3, 13)])
def test_silu_and_mul(ops, dtype, num_tokens, d):
    g = torch.Generator().manual_seed(d)
    x = (torch.randn(num_tokens, 2 * d, generator=g) * 2).to(dtype)
    exp = torch.empty(num_tokens, d, dtype=dtype)
    oracle.silu_and_mul(exp, x)
    out = torch.empty(num_tokens, d, dtype=dtype, device=DEV)
    ops.silu_and_mul(out, x.to(DEV))
    u = ulp_diff(out.cpu(), exp)
    # fp32: device expf and libm expf differ by an ulp or two of fp32
    assert u.max() <= (2 if dtype != torch.float32 else 8), u.max()
    assert (u > 0).float().mean() < (2e-2 if dtype != torch.float32 else 0.2)


# ---------------------------------------------------------------- paged attention
ATTN_CASES = [
    # S, H, KVH, D, BS, seq_lens
    (4, 32, 8, 128, 16, [1024, 1000, 513, 17]),
    (3, 8, 8, 64, 16, [1, 16, 33]),
    (2, 28, 4, 128, 16, [700, 129]),           # Qwen2-7B grouping (7 q heads per kv head)
    (2, 16, 1, 128, 16, [300, 64]),            # MQA, 16 q heads on one kv head
    (2, 40, 2, 128, 16, [250, 31]),            # 20 q heads per kv head -> two head groups
    (2, 8, 2, 80, 16, [100, 15]),
    (2, 8, 2, 96, 32, [100, 65]),
    (2, 4, 2, 112, 16, [47, 512]),
    (2, 4, 2, 120, 16, [200, 3]),
    (2, 4, 1, 192, 16, [90, 511]),
    (2, 4, 2, 256, 32, [600, 2]),
    (3, 8, 2, 128, 8, [77, 8, 530]),           # 8-token blocks -> generic kernel
    (2, 8, 2, 128, 32, [1025, 31]),
    (1, 32, 8, 128, 16, [4099]),               # long context: v1 with 8 waves, v2 with 9 partitions
]


def run_v1(ops, inp, alibi=None):
    q = inp["query"]
    out = torch.zeros(q.shape, dtype=q.dtype, device=q.device)
    ops.paged_attention_v1(out, q, inp["key_cache"], inp["value_cache"], inp["num_kv_heads"], inp["scale"],
                           inp["block_tables"], inp["seq_lens"], inp["block_size"], inp["max_seq_len"], alibi,
                           "auto", 1.0, 1.0)
    return out


def run_v2(ops, inp, alibi=None):
    q = inp["query"]
    S, H, D = q.shape
    es, ml, tmp = v2_scratch(S, H, D, inp["max_seq_len"], q.dtype, q.device)
    out = torch.zeros(q.shape, dtype=q.dtype, device=q.device)
    ops.paged_attention_v2(out, es, ml, tmp, q, inp["key_cache"], inp["value_cache"], inp["num_kv_heads"],
                           inp["scale"], inp["block_tables"], inp["seq_lens"], inp["block_size"],
                           inp["max_seq_len"], alibi, "auto", 1.0, 1.0)
    return out, es, ml, tmp


def oracle_v1(inp, alibi=None):
    q = inp["query"]
    out = torch.zeros(q.shape, dtype=q.dtype)
    oracle.paged_attention_v1(out, q, inp["key_cache"], inp["value_cache"], inp["num_kv_heads"], inp["scale"],
                              inp["block_tables"], inp["seq_lens"], inp["block_size"], inp["max_seq_len"], alibi)
    return out


def oracle_v2(inp, alibi=None):
    q = inp["query"]
    S, H, D = q.shape
    es, ml, tmp = v2_scratch(S, H, D, inp["max_seq_len"], q.dtype)
    out = torch.zeros(q.shape, dtype=q.dtype)
    oracle.paged_attention_v2(out, es, ml, tmp, q, inp["key_cache"], inp["value_cache"], inp["num_kv_heads"],
                              inp["scale"], inp["block_tables"], inp["seq_lens"], inp["block_size"],
                              inp["max_seq_len"], alibi)
    return out, es, ml, tmp


def split_ranges(seq_len, num_splits):
    """Token ranges of the non-empty shares (split_range() of csrc/attention_params.h)."""
    ntiles = (seq_len + 15) // 16
    if ntiles == 0:
        return []
    chunk = (ntiles + num_splits - 1) // num_splits
    out = []
    for s in range(num_splits):
        if s * chunk >= ntiles:
            break
        out.append((s * chunk * 16, min(seq_len, (s + 1) * chunk * 16)))
    return out


@pytest.fixture
def force_splits(monkeypatch):
    def _set(n):
        if n is None:
            monkeypatch.delenv("LVLLM_ATTN_SPLITS", raising=False)
        else:
            monkeypatch.setenv("LVLLM_ATTN_SPLITS", str(n))
    yield _set
    monkeypatch.delenv("LVLLM_ATTN_SPLITS", raising=False)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("case", ATTN_CASES, ids=[f"S{c[0]}H{c[1]}KV{c[2]}D{c[3]}BS{c[4]}" for c in ATTN_CASES])
def test_paged_attention_v1_v2(ops, dtype, case, force_splits):
    S, H, KVH, D, BS, lens = case
    inp = make_paged_inputs(S, H, KVH, D, BS, lens, dtype=dtype, seed=D + H, q_in_qkv=True)
    exp1 = oracle_v1(inp)
    exp2 = oracle_v2(inp)[0]
    d64 = dense_attention_fp64(inp)
    dinp = to_dev(inp)
    tol = 2e-2 if dtype != torch.float32 else 1e-4
    out1 = run_v1(ops, dinp)
    check_attention(out1, exp1, d64, tol)
    P = (max(lens) + 511) // 512
    # v2 with the library's own choice of shares and with every forced share count
    for forced in [None] + sorted({1, 2, P}):
        force_splits(forced)
        out2, es, ml, tmp = run_v2(ops, dinp)
        check_attention(out2, exp2, d64, tol)
        # v1 and v2 of the build agree to rounding
        assert rows_close_in_ulp(out1, out2, 2 if dtype != torch.float32 else 64), forced
        if forced is not None and forced > 1:
            # the scratch holds, per share, the softmax statistics and the normalised partial
            # output (the quantities of attention_kernels.cu:349-357): merging them reproduces out
            n = min(forced, P)
            for s_i, L in enumerate(lens):
                rng = split_ranges(L, n)
                if len(rng) < 2:
                    continue
                k = len(rng)
                m, e = ml[s_i, :, :k].float(), es[s_i, :, :k].float()
                w = e * torch.exp(m - m.max(dim=-1, keepdim=True).values)
                merged = (tmp[s_i, :, :k].float() * (w / w.sum(-1, keepdim=True)).unsqueeze(-1)).sum(1)
                assert torch.allclose(merged.cpu(), out2[s_i].float().cpu(), atol=2e-2 * float(exp2.float().abs().max()),
                                      rtol=2e-2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_paged_attention_alibi(ops, dtype):
    inp = make_paged_inputs(3, 8, 2, 128, 16, [600, 40, 513], dtype=dtype, seed=3)
    slopes = torch.tensor([2.0 ** (-(i + 1)) for i in range(8)], dtype=torch.float32)
    exp1 = oracle_v1(inp, slopes)
    d64 = dense_attention_fp64(inp, slopes)
    dinp = to_dev(inp)
    tol = 2e-2 if dtype != torch.float32 else 1e-4
    check_attention(run_v1(ops, dinp, slopes.to(DEV)), exp1, d64, tol)
    check_attention(run_v2(ops, dinp, slopes.to(DEV))[0], exp1, d64, tol)


def test_paged_attention_edge_cases(ops):
    # empty context -> zeros (attention_kernels.cu: num_seq_blocks = 0, acc stays 0)
    inp = make_paged_inputs(3, 8, 2, 128, 16, [0, 5, 0], dtype=torch.bfloat16, seed=11)
    dinp = to_dev(inp)
    o1 = run_v1(ops, dinp)
    o2 = run_v2(ops, dinp)[0]
    assert (o1[0] == 0).all() and (o1[2] == 0).all() and (o2[0] == 0).all() and (o2[2] == 0).all()
    check_attention(o1[1:2], oracle_v1(inp)[1:2])
    # slots past the context hold NaN: they must not leak (attention_kernels.cu:420-430)
    inp = make_paged_inputs(2, 8, 2, 128, 16, [37, 520], dtype=torch.bfloat16, seed=12)
    for s, n in enumerate([37, 520]):
        blk = inp["block_tables"][s, n // 16].long()
        inp["value_cache"][blk, :, :, n % 16:] = float("nan")
        inp["key_cache"][blk, :, :, n % 16:, :] = float("nan")
    exp = dense_attention_fp64(inp)
    dinp = to_dev(inp)
    for out in (run_v1(ops, dinp), run_v2(ops, dinp)[0]):
        assert torch.isfinite(out).all()
        check_attention(out, exp.to(torch.bfloat16), exp)
    # a spike that forces the running max to jump late in the context (online-softmax rescale)
    inp = make_paged_inputs(1, 4, 1, 128, 16, [800], dtype=torch.bfloat16, seed=13)
    q = inp["query"]
    tok = 700
    blk = inp["block_tables"][0, tok // 16].long()
    kvec = (q[0, 1].float() * 6).to(torch.bfloat16)  # key aligned with head 1's query -> huge logit
    inp["key_cache"][blk, 0, :, tok % 16, :] = kvec.view(16, 8)
    inp["k_dense"][0][tok, 0] = kvec
    exp = dense_attention_fp64(inp)
    dinp = to_dev(inp)
    check_attention(run_v1(ops, dinp), oracle_v1(inp), exp)
    check_attention(run_v2(ops, dinp)[0], oracle_v1(inp), exp)


def test_unsupported_arguments_raise(ops):
    inp = to_dev(make_paged_inputs(1, 4, 2, 128, 16, [20], dtype=torch.bfloat16))
    q = inp["query"]
    out = torch.zeros_like(q)
    with pytest.raises(RuntimeError):  # head size outside the reference's list (attention_kernels.cu:767)
        bad = make_paged_inputs(1, 4, 2, 72, 16, [20], dtype=torch.float32)
        bad = to_dev(bad)
        ops.paged_attention_v1(torch.zeros_like(bad["query"]), bad["query"], bad["key_cache"], bad["value_cache"], 2,
                               1.0, bad["block_tables"], bad["seq_lens"], 16, 20, None, "auto", 1.0, 1.0)
    with pytest.raises(RuntimeError):  # block size (attention_kernels.cu:804)
        ops.paged_attention_v1(out, q, inp["key_cache"], inp["value_cache"], 2, 1.0, inp["block_tables"],
                               inp["seq_lens"], 64, 20, None, "auto", 1.0, 1.0)
    with pytest.raises(RuntimeError):  # kv cache dtype string (quant_utils.cuh:571)
        ops.paged_attention_v1(out, q, inp["key_cache"], inp["value_cache"], 2, 1.0, inp["block_tables"],
                               inp["seq_lens"], 16, 20, None, "int4", 1.0, 1.0)
    torch.cuda.synchronize()


def test_full_size_properties(ops):
    """BASELINE config (B=32,H=32,KVH=8,D=128,BS=16,seq=1024, bf16): size-independent checks.
    1. permuting the physical blocks (and the table with them) leaves the output unchanged bit for bit;
    2. v1 == v2 to rounding; 3. linearity in V: attn(V1+V2) ~= attn(V1)+attn(V2);
    4. a sampled subset of (seq, head) rows against the fp64 dense computation."""
    B, H, KVH, D, BS, L = 32, 32, 8, 128, 16, 1024
    lens = [L] * 28 + [1000, 513, 17, 1]
    inp = make_paged_inputs(B, H, KVH, D, BS, lens, dtype=torch.bfloat16, seed=0)
    dinp = to_dev(inp)
    o1 = run_v1(ops, dinp)
    o2 = run_v2(ops, dinp)[0]
    assert rows_close_in_ulp(o1, o2, 2)
    # 1. permutation invariance
    NB = inp["key_cache"].shape[0]
    perm = torch.randperm(NB, generator=torch.Generator().manual_seed(1))
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(NB)
    pin = dict(dinp)
    pin["key_cache"] = dinp["key_cache"][perm.to(DEV)].contiguous()
    pin["value_cache"] = dinp["value_cache"][perm.to(DEV)].contiguous()
    pin["block_tables"] = inv.to(DEV)[dinp["block_tables"].long()].to(torch.int32)
    assert torch.equal(run_v2(ops, pin)[0], o2)
    assert torch.equal(run_v1(ops, pin), o1)
    # 3. linearity in V
    v2 = (torch.randn(inp["value_cache"].shape, generator=torch.Generator().manual_seed(2)) * 0.5).to(torch.bfloat16).to(DEV)
    a = dict(dinp); a["value_cache"] = v2
    b = dict(dinp); b["value_cache"] = (dinp["value_cache"].float() + v2.float()).to(torch.bfloat16)
    lhs = run_v2(ops, b)[0].float()
    rhs = o2.float() + run_v2(ops, a)[0].float()
    assert (lhs - rhs).abs().max() <= 2e-2
    # 4. sampled rows vs fp64
    sub = [0, 5, 28, 29, 30, 31]
    sinp = dict(inp)
    sinp["query"] = inp["query"][sub]
    sinp["k_dense"] = [inp["k_dense"][i] for i in sub]
    sinp["v_dense"] = [inp["v_dense"][i] for i in sub]
    d64 = dense_attention_fp64(sinp)
    check_attention(o2[sub], d64.to(torch.bfloat16), d64)
    check_attention(o1[sub], d64.to(torch.bfloat16), d64)


# ---------------------------------------------------------------- advance_step (bit-exact)
@pytest.mark.parametrize("num_seqs,num_queries,block_size", [(1, 1, 16), (37, 37, 16), (300, 290, 32), (8, 0, 8)])
def test_advance_step_bit_exact(ops, num_seqs, num_queries, block_size):
    g = torch.Generator().manual_seed(num_seqs)
    width = 12
    seq_lens = torch.randint(1, width * block_size - 1, (num_seqs,), generator=g, dtype=torch.int32)
    block_tables = torch.randint(0, 5000, (num_seqs, width), generator=g, dtype=torch.int32)
    tokens = torch.randint(0, 1000, (num_seqs,), generator=g)
    positions = (seq_lens - 1).long()
    slots = torch.randint(0, 1000, (num_seqs,), generator=g)
    sampled = torch.randint(0, 128000, (num_queries, 1), generator=g)
    want = [t.clone() for t in (tokens, positions, seq_lens, slots)]
    oracle.advance_step(num_queries, block_size, want[0], sampled, want[1], want[2], want[3], block_tables)
    got = [t.clone().to(DEV) for t in (tokens, positions, seq_lens, slots)]
    ops.advance_step(num_seqs, num_queries, block_size, got[0], sampled.to(DEV), got[1], got[2], got[3],
                     block_tables.to(DEV))
    torch.cuda.synchronize()
    for a, b in zip(got, want):
        assert torch.equal(a.cpu(), b)
    # what the host-side input builder would have produced for the same step
    for i in range(num_queries):
        pos = int(seq_lens[i])
        assert int(want[3][i]) == int(block_tables[i, pos // block_size]) * block_size + pos % block_size
    with pytest.raises(RuntimeError, match="is not as expected"):
        ops.advance_step(num_seqs, num_queries, block_size, got[0].int(), sampled.to(DEV), got[1], got[2], got[3],
                         block_tables.to(DEV))


# ---------------------------------------------------------------- block-sparse attention
def _blocksparse_fp64(inp, vert, local, bsz, step, tp_rank=0):
    """Independent statement: a token counts for head h iff its cache block's block-sparse index k
    satisfies (k + offset(h)) % vert == 0 or k > q - local (attention_kernels.cu:209-247)."""
    q = inp["query"]
    S, H, D = q.shape
    KVH, BS = inp["num_kv_heads"], inp["block_size"]
    out = torch.zeros(S, H, D, dtype=torch.float64)
    for s in range(S):
        k, v = inp["k_dense"][s].double(), inp["v_dense"][s].double()
        n = k.shape[0]
        if n == 0:
            continue
        tok = torch.arange(n)
        k_bs = (tok // BS) * BS // bsz
        q_bs = (n - 1) // bsz
        for h in range(H):
            kv = h // (H // KVH)
            off = (tp_rank * H + h) * step + 1 if step >= 0 else (tp_rank * KVH + kv) * (-step) + 1
            att = ((k_bs + off) % vert == 0) | (k_bs > q_bs - local)
            logits = (k[:, kv] @ q[s, h].double()) * inp["scale"]
            logits = logits.masked_fill(~att, float("-inf"))
            out[s, h] = torch.softmax(logits, 0) @ v[:, kv]
    return out


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("version", ["v1", "v2"])
@pytest.mark.parametrize("vert,local,bsz,step", [(4, 2, 64, 1), (8, 1, 32, 0), (3, 4, 16, -1), (2, 1, 64, 2)])
def test_blocksparse_attention(ops, dtype, version, vert, local, bsz, step):
    lens = [700, 130, 64, 1, 17]
    inp = make_paged_inputs(len(lens), 8, 2, 64, 16, lens, dtype=dtype, seed=vert + bsz)
    oracle.set_blocksparse(vert, local, bsz, step, 1)
    try:
        want = torch.zeros_like(inp["query"])
        oracle.paged_attention_v1(want, inp["query"], inp["key_cache"], inp["value_cache"], 2, inp["scale"],
                                  inp["block_tables"], inp["seq_lens"], 16, max(lens))
    finally:
        oracle.set_blocksparse(0)
    d = to_dev(inp)
    out = torch.full_like(d["query"], float("nan"))
    args = (d["query"], d["key_cache"], d["value_cache"], 2, inp["scale"], d["block_tables"], d["seq_lens"], 16,
            max(lens), None, "auto", 1.0, 1.0, 1, local, vert, bsz, step)
    if version == "v1":
        ops.paged_attention_v1(out, *args)
    else:
        es, ml, tmp = v2_scratch(len(lens), 8, 64, max(lens), dtype, DEV)
        ops.paged_attention_v2(out, es, ml, tmp, *args)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    check_attention(out, want, _blocksparse_fp64(inp, vert, local, bsz, step, tp_rank=1))
    # and it differs from dense attention (the mask really bites)
    dense = dense_attention_fp64(inp)
    assert float((out.double().cpu() - dense).abs().max()) > 1e-3


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T,hidden", [(1, 8), (7, 128), (33, 1024), (4, 4096), (3, 16384), (2, 1000)])
@pytest.mark.parametrize("with_y", [True, False])
def test_add_layer_norm_matches_torch(ops, dtype, T, hidden, with_y):
    """LayerNorm(x + y) in one launch against torch's add + layer_norm evaluated in fp32 on the
    same rounded sum; tolerance: output rounding + reduction order."""
    g = torch.Generator(device=DEV).manual_seed(T * 31 + hidden)
    x = (torch.randn(T, hidden, generator=g, device=DEV) * 2).to(dtype)
    y = (torch.randn(T, hidden, generator=g, device=DEV) + 0.5).to(dtype) if with_y else None
    w = (1 + 0.1 * torch.randn(hidden, generator=g, device=DEV)).to(dtype)
    b = (0.1 * torch.randn(hidden, generator=g, device=DEV)).to(dtype)
    out = torch.empty_like(x)
    torch.ops._C_amd.add_layer_norm(out, x, y, w, b, 1e-5)
    z = (x + y) if with_y else x  # the rounded sum, as torch materialises it
    ref = torch.nn.functional.layer_norm(z.float(), (hidden,), w.float(), b.float(), 1e-5)
    err = (out.float() - ref).abs().max().item()
    assert err <= (2 ** -7 if dtype == torch.bfloat16 else 2 ** -10) * max(1.0, ref.abs().max().item()), err
    # in place
    x2 = x.clone()
    torch.ops._C_amd.add_layer_norm(x2, x2, y, w, b, 1e-5)
    assert torch.equal(x2.view(torch.int16), out.view(torch.int16))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_gelu_matches_torch(ops, dtype):
    g = torch.Generator(device=DEV).manual_seed(1)
    x = (torch.randn(37, 4096, generator=g, device=DEV) * 3).to(dtype)
    x[0, :8] = torch.tensor([0.0, -0.0, 1e-4, -1e-4, 20.0, -20.0, 65000.0 if dtype == torch.float16 else 3e38, -10.0],
                            device=DEV).to(dtype)
    want = torch.nn.functional.gelu(x)
    out = torch.empty_like(x)
    torch.ops._C_amd.gelu(out, x)
    d = (out.float() - want.float()).abs()
    ulp = torch.finfo(dtype).eps * want.float().abs().clamp_min(torch.finfo(dtype).tiny)
    assert bool((d <= ulp).all()), float(d.max())  # the same fp32 formula: at most one rounding step apart
    x2 = x.clone()
    torch.ops._C_amd.gelu(x2, x2)
    assert torch.equal(x2.view(torch.int16), out.view(torch.int16))
