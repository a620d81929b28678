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


@pytest.mark.parametrize("pinned", [True, False])
@pytest.mark.parametrize("min_runs", [0, 1 << 20], ids=["kernel", "dma"])
def test_swap_blocks_scattered_both_paths_bit_exact(ops, pinned, min_runs):
    """Scattered mappings move through ONE kernel launch when the host side is pinned (device-mapped), through one
    DMA per run otherwise: same bytes either way, in both directions, repeated calls (the pinned ring of pair lists
    wraps), a pair list changed right after the call, untouched blocks untouched."""
    torch.ops._C_amd.set_tuning("swap_kernel_min_runs", min_runs)
    try:
        g = torch.Generator().manual_seed(21)
        NB, E = 300, 8 * 128 * 16  # one layer's K blocks of the Llama-3-8B shape: 32 KiB each
        gpu = torch.randn(NB, E, generator=g).to(torch.bfloat16).to(DEV)
        host = torch.zeros(NB, E, dtype=torch.bfloat16)
        host = host.pin_memory() if pinned else host
        exp = host.clone()
        gpu_cpu = gpu.cpu()
        for it in range(140):  # more calls than the ring has slots
            n = 1 + (it * 7) % 40
            src = torch.randperm(NB, generator=g)[:n]
            dst = torch.randperm(NB, generator=g)[:n]
            mapping = torch.stack([src, dst], 1).to(torch.int64).contiguous()
            ops.swap_blocks(gpu, host, mapping)
            oracle.swap_blocks(gpu_cpu, exp, mapping)
            mapping.fill_(0)  # the caller's list is free to change once the call has returned
        torch.cuda.synchronize()
        assert torch.equal(host.view(torch.int16), exp.view(torch.int16))
        back = torch.zeros_like(gpu)
        exp_back = torch.zeros(NB, E, dtype=torch.bfloat16)
        src = torch.randperm(NB, generator=g)[:200]
        dst = torch.randperm(NB, generator=g)[:200]
        mapping = torch.stack([src, dst], 1).to(torch.int64).contiguous()
        ops.swap_blocks(host, back, mapping)
        oracle.swap_blocks(exp, exp_back, mapping)
        assert torch.equal(back.cpu().view(torch.int16), exp_back.view(torch.int16))
    finally:
        torch.ops._C_amd.set_tuning("swap_kernel_min_runs", 3)


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
    (3, 8, 2, 128, 8, [77, 8, 530]),           # 8-token blocks: a 16-token tile spans two blocks (fp32: generic kernel)
    (4, 32, 8, 128, 8, [1024, 1000, 9, 1]),    # the metric's grouping on 8-token blocks
    (2, 28, 4, 80, 8, [333, 15]),              # padded head size, 7 q heads per kv head
    (2, 4, 4, 256, 8, [130, 7]),
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
def force_splits(ops):
    """lvllm_set_tuning("attn_splits", n): n shares; None / 0 = the library's choice; -1 = the
    reference's 512-token partitions."""
    def _set(n):
        torch.ops._C_amd.set_tuning("attn_splits", 0 if n is None else int(n))
    yield _set
    torch.ops._C_amd.set_tuning("attn_splits", 0)


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
    # slots past the context hold NaN: they must not leak (attention_kernels.cu:420-430); with 8-token blocks the
    # tile's second block may lie wholly past the context (its table entry is padding): NaN there too
    for BS in (16, 8):
        inp = make_paged_inputs(2, 8, 2, 128, BS, [37, 520], dtype=torch.bfloat16, seed=12)
        for s, n in enumerate([37, 520]):
            if n % BS == 0:
                continue  # the context ends on a block boundary: no partly used block
            blk = inp["block_tables"][s, n // BS].long()
            inp["value_cache"][blk, :, :, n % BS:] = float("nan")
            inp["key_cache"][blk, :, :, n % BS:, :] = float("nan")
        used = {int(b) for s, n in enumerate([37, 520]) for b in inp["block_tables"][s, :(n + BS - 1) // BS]}
        for b in range(inp["key_cache"].shape[0]):  # every block no context reaches (table entries past the
            if b not in used:                        # context point at some of them) holds NaN as well
                inp["value_cache"][b] = float("nan")
                inp["key_cache"][b] = float("nan")
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


# ------------------------------------------------ v2 scratch under the reference's partitioning
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("case", [
    (4, 8, 2, 128, 16, [1024, 1500, 512, 17]),
    (3, 4, 4, 64, 32, [513, 2048, 1]),
    (2, 14, 2, 128, 16, [700, 0]),
    (5, 32, 8, 128, 16, [1024] * 5),
], ids=["gqa4", "mha_bs32", "gqa7_empty", "metric_shape"])
def test_v2_scratch_equals_the_reference_partitions(ops, dtype, case, force_splits):
    """With lvllm_set_tuning("attn_splits", -1) paged_attention_v2 cuts contexts at 512 tokens as
    attention_kernels.cu:850 does, and exp_sums / max_logits / tmp_out hold, slot for slot, what the
    reference's partition kernel stores (attention_kernels.cu:349-357,483-495; oracle_paged_attention_v2
    restates it).  Slots of partitions past a sequence's end are never written by either."""
    S, H, KVH, D, BS, lens = case
    inp = make_paged_inputs(S, H, KVH, D, BS, lens, dtype=dtype, seed=S * 7 + D, q_in_qkv=True)
    exp_out, exp_es, exp_ml, exp_tmp = oracle_v2(inp)
    force_splits(-1)
    out, es, ml, tmp = run_v2(ops, to_dev(inp))
    tol = 2e-2 if dtype != torch.float32 else 1e-4
    check_attention(out, exp_out, dense_attention_fp64(inp), tol)
    es, ml, tmp = es.cpu(), ml.cpu(), tmp.cpu()
    scale = float(exp_out.float().abs().max().clamp_min(1e-6))
    for s_i, L in enumerate(lens):
        for j in range((L + 511) // 512):
            assert torch.allclose(ml[s_i, :, j], exp_ml[s_i, :, j], atol=2e-3, rtol=1e-3), (s_i, j)
            assert torch.allclose(es[s_i, :, j], exp_es[s_i, :, j], rtol=5e-3, atol=1e-4), (s_i, j)
            assert ((tmp[s_i, :, j].float() - exp_tmp[s_i, :, j].float()).abs().max() <= tol * max(
                scale, float(exp_tmp[s_i, :, j].float().abs().max()))), (s_i, j)


def test_v2_forced_shares_on_512_multiples_are_the_reference_partitions(ops, force_splits):
    """Contexts that are multiples of 512 cut into max_seq/512 equal shares: share j IS partition j, so the
    default kernel's scratch equals the oracle's without the reference-partition switch."""
    lens = [2048, 2048, 2048]
    inp = make_paged_inputs(3, 8, 2, 128, 16, lens, dtype=torch.bfloat16, seed=11)
    exp_out, exp_es, exp_ml, exp_tmp = oracle_v2(inp)
    force_splits(4)
    out, es, ml, tmp = run_v2(ops, to_dev(inp))
    check_attention(out, exp_out)
    assert torch.allclose(ml.cpu(), exp_ml, atol=2e-3, rtol=1e-3)
    assert torch.allclose(es.cpu(), exp_es, rtol=5e-3, atol=1e-4)
    assert (tmp.cpu().float() - exp_tmp.float()).abs().max() <= 2e-2 * float(exp_tmp.float().abs().max())


# ------------------------------------------------ stated cache extents (kv_cache_bytes of the C-ABI)
def test_block_numbers_beyond_the_stated_extent_are_clamped_not_followed(ops):
    """The caches are the FRONT of one larger allocation whose tail holds NaN: a block table that points
    past the caches' last block must read inside the stated extent (the torch binding states it), so the
    result stays finite and equals the run with the numbers clamped by hand."""
    inp = make_paged_inputs(2, 8, 2, 128, 16, [300, 77], dtype=torch.bfloat16, seed=2)
    NB = inp["key_cache"].shape[0]
    d = to_dev(inp)
    big_k = torch.full((3 * NB,) + tuple(inp["key_cache"].shape[1:]), float("nan"), dtype=torch.bfloat16, device=DEV)
    big_v = torch.full((3 * NB,) + tuple(inp["value_cache"].shape[1:]), float("nan"), dtype=torch.bfloat16, device=DEV)
    big_k[:NB], big_v[:NB] = d["key_cache"], d["value_cache"]
    bad = d["block_tables"].clone()
    bad[0, 3] = NB + 5       # inside the big allocation, outside the stated caches
    bad[1, 1] = 3 * NB - 1
    clamped = bad.clamp(max=NB - 1)
    outs = []
    for bt in (bad, clamped):
        dd = dict(d, key_cache=big_k[:NB], value_cache=big_v[:NB], block_tables=bt)
        outs.append(run_v1(ops, dd))
        outs.append(run_v2(ops, dd)[0])
    assert all(torch.isfinite(o.float()).all() for o in outs)
    assert torch.equal(outs[0], outs[2]) and torch.equal(outs[1], outs[3])


def test_slots_beyond_the_stated_extent_are_skipped(ops):
    g = torch.Generator().manual_seed(3)
    T, H, D, BS, NB = 70, 2, 128, 16, 6
    key = torch.randn(T, H, D, generator=g).to(torch.bfloat16).to(DEV)
    value = torch.randn(T, H, D, generator=g).to(torch.bfloat16).to(DEV)
    big_k = torch.zeros(2 * NB, H, D // 8, BS, 8, dtype=torch.bfloat16, device=DEV)
    big_v = torch.zeros(2 * NB, H, D, BS, dtype=torch.bfloat16, device=DEV)
    slots = torch.randperm(NB * BS, generator=g)[:T].to(torch.int64)
    slots[5] = NB * BS + 3       # first slot past the stated caches
    slots[40] = 2 * NB * BS - 1
    for min_tok in (64, 1 << 20):  # the tiled kernel and the per-chunk kernel
        torch.ops._C_amd.set_tuning("cache_tile_min_tokens", min_tok)
        big_k.zero_(), big_v.zero_()
        ops.reshape_and_cache(key, value, big_k[:NB], big_v[:NB], slots.to(DEV), "auto", 1.0, 1.0)
        assert not big_k[NB:].any() and not big_v[NB:].any()
        ok = slots.clone()
        ok[5] = ok[40] = -1
        kc_o, vc_o = torch.zeros_like(big_k[:NB]).cpu(), torch.zeros_like(big_v[:NB]).cpu()
        oracle.reshape_and_cache(key.cpu(), value.cpu(), kc_o, vc_o, ok)
        assert torch.equal(big_k[:NB].cpu(), kc_o) and torch.equal(big_v[:NB].cpu(), vc_o)
    torch.ops._C_amd.set_tuning("cache_tile_min_tokens", 384)  # the default (csrc/common.h)


# ------------------------------------------------ reshape_and_cache at prompt sizes (LDS-tiled kernel)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("block_size", [8, 16, 32])
@pytest.mark.parametrize("num_heads,head_size", [(8, 128), (2, 64), (3, 80), (1, 256), (4, 120)])
def test_reshape_and_cache_tiled_bit_exact(ops, dtype, block_size, num_heads, head_size):
    """Prompt-sized batches take reshape_and_cache_tile_kernel: ragged sequences laid out back to back (token
    tiles straddle sequences and blocks, contexts start mid-block as after a prefix hit or a chunk), padding
    slots, a reversed run and fully random slots -- against the oracle, bit for bit, untouched bytes included."""
    g = torch.Generator().manual_seed(block_size * 1000 + head_size + num_heads)
    seqs = [(0, 37), (5, 130), (block_size - 1, 3), (17, 64), (0, 1), (block_size, 200), (3, 77)]
    blocks_needed = sum((c + n + block_size - 1) // block_size for c, n in seqs)
    NB = blocks_needed + 9
    perm = torch.randperm(NB, generator=g).tolist()
    slots = []
    for ctx, n in seqs:
        nb = (ctx + n + block_size - 1) // block_size
        table, perm = perm[:nb], perm[nb:]
        slots += [table[p // block_size] * block_size + p % block_size for p in range(ctx, ctx + n)]
    T0 = len(slots)
    free = [b for b in perm]
    slots += [-1] * 5
    slots += [free[0] * block_size + o for o in reversed(range(block_size))]          # descending run
    rnd = torch.randperm(len(free[1:]) * block_size, generator=g)[:70] + 0
    slots += [free[1 + int(r) // block_size] * block_size + int(r) % block_size for r in rnd]
    slots[10] = -1
    T = len(slots)
    assert T > 64 and T0 > 400
    x = 8
    qkv = torch.randn(T, 3 * num_heads * head_size, generator=g).to(dtype)
    key = qkv[:, num_heads * head_size: 2 * num_heads * head_size].view(T, num_heads, head_size)
    value = qkv[:, 2 * num_heads * head_size:].view(T, num_heads, head_size)
    kc = torch.randn(NB, num_heads, head_size // x, block_size, x, generator=g).to(dtype)
    vc = torch.randn(NB, num_heads, head_size, block_size, generator=g).to(dtype)
    slots_t = torch.tensor(slots, dtype=torch.int64)
    kc_o, vc_o = kc.clone(), vc.clone()
    oracle.reshape_and_cache(key, value, kc_o, vc_o, slots_t)
    qkv_d = qkv.to(DEV)
    key_d = qkv_d[:, num_heads * head_size: 2 * num_heads * head_size].view(T, num_heads, head_size)
    value_d = qkv_d[:, 2 * num_heads * head_size:].view(T, num_heads, head_size)
    kc_d, vc_d = kc.to(DEV), vc.to(DEV)
    ops.reshape_and_cache(key_d, value_d, kc_d, vc_d, slots_t.to(DEV), "auto", 1.0, 1.0)
    assert torch.equal(kc_d.cpu().view(torch.int16), kc_o.view(torch.int16))
    assert torch.equal(vc_d.cpu().view(torch.int16), vc_o.view(torch.int16))


def test_reshape_and_cache_full_size_round_trip(ops):
    """BASELINE prefill size (8192 tokens x 8 kv heads x 128): the tiled kernel and the per-chunk kernel fill
    two caches identically, and gathering the cache back through the slots returns the inputs."""
    g = torch.Generator(device=DEV).manual_seed(0)
    T, H, D, BS = 8192, 8, 128, 16
    NB = T // BS + 50
    key = torch.randn(T, H, D, generator=g, device=DEV).to(torch.bfloat16)
    value = torch.randn(T, H, D, generator=g, device=DEV).to(torch.bfloat16)
    table = torch.randperm(NB, generator=g, device=DEV)[: T // BS]
    pos = torch.arange(T, device=DEV)
    slots = table[pos // BS] * BS + pos % BS
    caches = []
    for min_tok in (64, 1 << 20):
        torch.ops._C_amd.set_tuning("cache_tile_min_tokens", min_tok)
        kc = torch.zeros(NB, H, D // 8, BS, 8, dtype=torch.bfloat16, device=DEV)
        vc = torch.zeros(NB, H, D, BS, dtype=torch.bfloat16, device=DEV)
        ops.reshape_and_cache(key, value, kc, vc, slots, "auto", 1.0, 1.0)
        caches.append((kc, vc))
    torch.ops._C_amd.set_tuning("cache_tile_min_tokens", 384)  # the default (csrc/common.h)
    assert torch.equal(caches[0][0], caches[1][0]) and torch.equal(caches[0][1], caches[1][1])
    kc, vc = caches[0]
    blk, off = slots // BS, slots % BS
    k_back = kc[blk, :, :, off, :].reshape(T, H, D)
    v_back = vc[blk, :, :, off]
    assert torch.equal(k_back, key) and torch.equal(v_back, value)
