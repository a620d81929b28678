"""The CPU oracle (oracle/paged_ops_oracle.c) is pinned here:
  * against the committed golden vectors the REFERENCE's csrc/cpu backend produced
    (tests/golden/ops_*.npz, oracle/make_golden.py) -- runs everywhere;
  * against that backend itself when oracle/_ref/_ref_C.so is present (dev container and,
    as a prebuilt file, the GPU box);
  * against an independent fp64 dense computation.
Byte movement is compared bit-exactly; arithmetic within the tolerance stated per test (the
oracle restates the reference's GPU kernels, which round in T where csrc/cpu keeps fp32)."""
import os

import numpy as np
import pytest
import torch

from helpers import dense_attention_fp64, make_paged_inputs, v2_scratch
from oracle import oracle, ref

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_t(npz, key, dtype=None):
    a = npz[key]
    if a.dtype == np.uint16:
        return torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16)
    return torch.from_numpy(a.copy())


def bf16_ulp(x):
    return 2.0 ** (torch.floor(torch.log2(x.abs().double().clamp_min(1e-30))) - 7)


def test_conversions_match_numpy_and_torch():
    L = oracle.lib()
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.standard_normal(3000).astype(np.float32) * s for s in (1e-8, 1e-5, 1e-3, 1, 100, 7e4)]
                        + [np.array([0, -0.0, np.inf, -np.inf, 65504, 65519.9, 65520, 6e-8, 5.96e-8, 2.98e-8,
                                     2.9802322e-8, 3e-8, 6.1e-5, 6.097555e-05], dtype=np.float32)])
    with np.errstate(over="ignore"):
        want16 = xs.astype(np.float16).view(np.uint16)
    wantb = torch.from_numpy(xs).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    for x, w16, wb in zip(xs.tolist(), want16.tolist(), wantb.tolist()):
        assert L.oracle_f32_to_f16(x) == w16, x
        assert L.oracle_f32_to_bf16(x) == wb, x
    hs = np.arange(0, 65536, dtype=np.uint16)
    want = hs.view(np.float16).astype(np.float32)
    for h in range(0, 65536, 5):
        f = L.oracle_f16_to_f32(h)
        assert f == want[h] or (f != f and want[h] != want[h])


@pytest.mark.parametrize("name", ["attn_bf16_gqa4_d128", "attn_f32_gqa2_d64", "attn_bf16_mha_d80"])
def test_attention_vs_golden(name):
    z = np.load(os.path.join(GOLDEN, f"ops_{name}.npz"))
    q, kc, vc = load_t(z, "query"), load_t(z, "key_cache"), load_t(z, "value_cache")
    bt, sl = torch.from_numpy(z["block_tables"]), torch.from_numpy(z["seq_lens"])
    kvh, scale = int(z["num_kv_heads"]), float(z["scale"])
    S, H, D = q.shape
    o1 = torch.zeros_like(q)
    oracle.paged_attention_v1(o1, q, kc, vc, kvh, scale, bt, sl, 16, int(sl.max()))
    es, ml, tmp = v2_scratch(S, H, D, int(sl.max()), q.dtype)
    o2 = torch.zeros_like(q)
    oracle.paged_attention_v2(o2, es, ml, tmp, q, kc, vc, kvh, scale, bt, sl, 16, int(sl.max()))
    g1, g2 = load_t(z, "out_v1"), load_t(z, "out_v2")
    if q.dtype == torch.bfloat16:
        # both sides round the output to bf16; the GPU-kernel semantics also round P to bf16
        # (attention_kernels.cu:398-400) where csrc/cpu keeps fp32: <= 2 ulp at the row scale
        for o, g in ((o1, g1), (o2, g2)):
            scale_ulp = bf16_ulp(g.float().abs().amax(dim=-1, keepdim=True))
            assert ((o.double() - g.double()).abs() <= 2 * scale_ulp).all()
    else:
        # fp32: only the 1e-6 in the softmax normaliser differs (relative 1e-6 .. 1e-8)
        assert torch.allclose(o1, g1, rtol=5e-6, atol=1e-7)
        assert torch.allclose(o2, g2, rtol=5e-6, atol=1e-7)


def test_cache_ops_vs_golden_bit_exact():
    z = np.load(os.path.join(GOLDEN, "ops_cache_bf16.npz"))
    key, value = load_t(z, "key"), load_t(z, "value")
    kc, vc = load_t(z, "key_cache_in"), load_t(z, "value_cache_in")
    oracle.reshape_and_cache(key, value, kc, vc, torch.from_numpy(z["slots"]))
    assert torch.equal(kc.view(torch.int16), load_t(z, "key_cache_out").view(torch.int16))
    assert torch.equal(vc.view(torch.int16), load_t(z, "value_cache_out").view(torch.int16))
    oracle.copy_blocks([kc], [vc], torch.from_numpy(z["copy_mapping"]))
    assert torch.equal(kc.view(torch.int16), load_t(z, "key_cache_copied").view(torch.int16))
    assert torch.equal(vc.view(torch.int16), load_t(z, "value_cache_copied").view(torch.int16))


@pytest.mark.parametrize("tag,dtype", [("bf16", torch.bfloat16), ("f32", torch.float32)])
def test_elementwise_vs_golden(tag, dtype):
    z = np.load(os.path.join(GOLDEN, "ops_elementwise.npz"))
    tol = dict(rtol=2 ** -6, atol=1e-6) if dtype == torch.bfloat16 else dict(rtol=2e-6, atol=1e-7)
    x, w, res = load_t(z, f"norm_x_{tag}"), load_t(z, f"norm_w_{tag}"), load_t(z, f"norm_res_{tag}")
    o = torch.empty_like(x)
    oracle.rms_norm(o, x, w, 1e-6)
    assert torch.allclose(o.float(), load_t(z, f"rms_out_{tag}").float(), **tol)  # <= 2 ulp (two roundings in T vs one after an fp32 multiply)
    x2, r2 = x.clone(), res.clone()
    oracle.fused_add_rms_norm(x2, r2, w, 1e-6)
    # The reference's CPU backend, built without -mavx512bf16 (g++ < 12.3, cmake/cpu_extension.cmake:59-66),
    # stores fp32 -> bf16 by TRUNCATION (csrc/cpu/cpu_types_x86.hpp:482-488) where its GPU kernels -- and
    # this oracle -- round to nearest: the residual differs by <= 1 ulp, the normed output by <= 4 ulp.
    tol4 = dict(rtol=2 ** -5, atol=1e-6) if dtype == torch.bfloat16 else tol
    assert torch.allclose(x2.float(), load_t(z, f"fused_out_{tag}").float(), **tol4)
    assert torch.allclose(r2.float(), load_t(z, f"fused_res_{tag}").float(),
                          rtol=2 ** -7 if dtype == torch.bfloat16 else 0, atol=0)
    qk, cache, pos = load_t(z, f"rope_in_{tag}"), load_t(z, f"rope_cache_{tag}"), torch.from_numpy(z["rope_pos"])
    H, KVH, D = 4, 2, 64
    for neox, key in ((True, "neox"), (False, "gptj")):
        t = qk.clone()
        oracle.rotary_embedding(pos, t[:, :H * D], t[:, H * D:], D, cache, neox)
        want = load_t(z, f"rope_out_{key}_{tag}").float()
        # GPU semantics round each product to T; csrc/cpu rounds once: <= 2 ulp of the operands' scale
        assert ((t.float() - want).abs() <= (2 ** -6 if dtype == torch.bfloat16 else 1e-6) * qk.float().abs().clamp_min(0.25).max()).all()
    gu = load_t(z, f"silu_in_{tag}")
    so = torch.empty(gu.shape[0], gu.shape[1] // 2, dtype=dtype)
    oracle.silu_and_mul(so, gu)
    assert torch.allclose(so.float(), load_t(z, f"silu_out_{tag}").float(), **tol)


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref/_ref_C.so not built (needs /root/reference) or no AVX512")
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_oracle_vs_reference_backend_live(dtype):
    assert ref.load()
    inp = make_paged_inputs(6, 8, 2, 128, 16, [700, 640, 513, 17, 1, 512], dtype=dtype, seed=21)
    q = inp["query"]
    S, H, D = q.shape
    o_or, o_ref = torch.zeros_like(q), torch.zeros_like(q)
    oracle.paged_attention_v1(o_or, q, inp["key_cache"], inp["value_cache"], 2, inp["scale"], inp["block_tables"],
                              inp["seq_lens"], 16, inp["max_seq_len"])
    torch.ops._ref_C.paged_attention_v1(o_ref, q, inp["key_cache"], inp["value_cache"], 2, inp["scale"],
                                        inp["block_tables"], inp["seq_lens"], 16, inp["max_seq_len"], None, "auto",
                                        1.0, 1.0, 0, 0, 0, 64, 0)
    d64 = dense_attention_fp64(inp)
    tol = 4e-3 if dtype == torch.bfloat16 else 5e-6
    assert (o_or.double() - o_ref.double()).abs().max() <= tol
    assert (o_or.double() - d64).abs().max() <= tol
    # reshape_and_cache: bit-exact against the reference's CPU kernel
    g = torch.Generator().manual_seed(3)
    key = torch.randn(20, 2, 128, generator=g).to(dtype)
    value = torch.randn(20, 2, 128, generator=g).to(dtype)
    slots = torch.randperm(inp["key_cache"].shape[0] * 16, generator=g)[:20].to(torch.int64)
    kc1, vc1 = inp["key_cache"].clone(), inp["value_cache"].clone()
    kc2, vc2 = inp["key_cache"].clone(), inp["value_cache"].clone()
    oracle.reshape_and_cache(key, value, kc1, vc1, slots)
    torch.ops._ref_C_cache_ops.reshape_and_cache(key, value, kc2, vc2, slots, "auto", 1.0, 1.0)
    assert torch.equal(kc1.view(torch.uint8), kc2.view(torch.uint8))
    assert torch.equal(vc1.view(torch.uint8), vc2.view(torch.uint8))


def test_oracle_v2_scratch_semantics():
    """exp_sums / max_logits / tmp_out carry per-partition softmax statistics that merge back
    to the un-partitioned result (attention_kernels.cu:349-357, 577-668)."""
    inp = make_paged_inputs(2, 4, 2, 64, 16, [1300, 512], dtype=torch.float32, seed=5)
    q = inp["query"]
    S, H, D = q.shape
    es, ml, tmp = v2_scratch(S, H, D, inp["max_seq_len"], q.dtype)
    o2, o1 = torch.zeros_like(q), torch.zeros_like(q)
    oracle.paged_attention_v2(o2, es, ml, tmp, q, inp["key_cache"], inp["value_cache"], 2, inp["scale"],
                              inp["block_tables"], inp["seq_lens"], 16, inp["max_seq_len"])
    oracle.paged_attention_v1(o1, q, inp["key_cache"], inp["value_cache"], 2, inp["scale"], inp["block_tables"],
                              inp["seq_lens"], 16, inp["max_seq_len"])
    assert torch.allclose(o1, o2, rtol=1e-5, atol=1e-6)
    w = es[0] * torch.exp(ml[0] - ml[0].max(dim=-1, keepdim=True).values)
    merged = (tmp[0] * (w / w.sum(-1, keepdim=True)).unsqueeze(-1)).sum(1)
    assert torch.allclose(merged, o1[0], rtol=1e-4, atol=1e-6)
    assert torch.equal(o2[1], tmp[1, :, 0])  # single partition: copied through


# ---- prefill (next row §8f-1): the oracle's varlen causal attention over the paged cache ----
def _sdpa_reference(inp, alibi=None, window=0):
    """torch.nn.functional.scaled_dot_product_attention in fp32 on the gathered dense K/V: the
    function the reference's in-tree prefill backends evaluate (torch_naive.py:125-149 restates
    it, torch_sdpa.py calls it), with the bottom-right causal mask of a chunk over its context."""
    from helpers import dense_prefill_fp64  # noqa: F401  (same mask convention)
    q = inp["query"].float()
    T, H, D = q.shape
    KVH = inp["num_kv_heads"]
    out = torch.zeros(T, H, D)
    qsl = inp["query_start_loc"].tolist()
    for s, S in enumerate(inp["seq_lens"].tolist()):
        L = qsl[s + 1] - qsl[s]
        if L == 0:
            continue
        k = inp["k_dense"][s].float().repeat_interleave(H // KVH, dim=1).transpose(0, 1)
        v = inp["v_dense"][s].float().repeat_interleave(H // KVH, dim=1).transpose(0, 1)
        pos = torch.arange(S - L, S)[:, None]
        keys = torch.arange(S)[None, :]
        mask = keys <= pos
        if window:
            mask &= keys > pos - window
        bias = torch.zeros(H, L, S)
        if alibi is not None:
            bias += alibi[:, None, None] * (keys - pos).float()[None]
        bias = bias.masked_fill(~mask[None], float("-inf"))
        o = torch.nn.functional.scaled_dot_product_attention(
            q[qsl[s]:qsl[s + 1]].transpose(0, 1)[None], k[None], v[None], attn_mask=bias[None],
            scale=inp["scale"])[0]
        out[qsl[s]:qsl[s + 1]] = o.transpose(0, 1)
    return out


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", ["prompts", "chunks", "window", "alibi"])
def test_prefill_oracle_vs_torch_sdpa(case, dtype):
    from helpers import make_prefill_inputs
    H, KVH, D, BS = 8, 2, 64, 16
    if case == "prompts":
        seq, ql = [37, 64, 5, 1], [37, 64, 5, 1]
    else:
        seq, ql = [37, 90, 17, 48, 33], [37, 10, 1, 16, 0]
    inp = make_prefill_inputs(H, KVH, D, BS, seq, ql, dtype=dtype, seed=11)
    alibi = torch.tensor([0.5 ** (i + 1) for i in range(H)]) if case == "alibi" else None
    window = 24 if case == "window" else 0
    out = torch.full_like(inp["query"], float("nan"))
    oracle.paged_prefill_attention(out, inp["query"], inp["key_cache"], inp["value_cache"], KVH,
                                   inp["scale"], inp["block_tables"], inp["seq_lens"],
                                   inp["query_start_loc"], BS, alibi_slopes=alibi, sliding_window=window)
    want = _sdpa_reference(inp, alibi, window)
    # T-rounded probabilities and output: 2^-8 relative for bf16, 2^-11 for f16, at row scale
    tol = (2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10) * max(1.0, float(want.abs().max()))
    assert torch.isfinite(out).all()
    assert float((out.float() - want).abs().max()) <= tol


def test_prefill_oracle_softcap_and_decode_equivalence():
    """softcap follows cap*tanh(x/cap); a chunk of one token equals paged_attention_v1 of the
    same sequence up to the 1e-6 in the decode normaliser."""
    from helpers import dense_prefill_fp64, make_prefill_inputs
    H, KVH, D, BS = 4, 4, 64, 16
    inp = make_prefill_inputs(H, KVH, D, BS, [50, 81], [1, 1], dtype=torch.float16, seed=5)
    a = torch.zeros_like(inp["query"])
    oracle.paged_prefill_attention(a, inp["query"], inp["key_cache"], inp["value_cache"], KVH, inp["scale"],
                                   inp["block_tables"], inp["seq_lens"], inp["query_start_loc"], BS)
    b = torch.zeros_like(inp["query"])
    oracle.paged_attention_v1(b, inp["query"], inp["key_cache"], inp["value_cache"], KVH, inp["scale"],
                              inp["block_tables"], inp["seq_lens"], BS, 81)
    assert float((a.float() - b.float()).abs().max()) <= 2e-3
    c = torch.zeros_like(inp["query"])
    oracle.paged_prefill_attention(c, inp["query"], inp["key_cache"], inp["value_cache"], KVH, 4.0,
                                   inp["block_tables"], inp["seq_lens"], inp["query_start_loc"], BS,
                                   softcap=2.0)
    inp["scale"] = 4.0
    want = dense_prefill_fp64(inp, softcap=2.0)
    assert float((c.double() - want).abs().max()) <= 2e-3


# ---- prefill-only (no KV cache) attention: golden vectors from the reference's own backend ----
def _load_prefill_only_cases():
    z = np.load(os.path.join(GOLDEN, "prefill_only_attn.npz"))
    for tag in z["cases"]:
        tag = str(tag)
        H, KVH = int(tag.split("_")[0][1:]), int(tag.split("_")[1][3:])
        bf = lambda a: torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16)
        yield dict(tag=tag, H=H, KVH=KVH, q=bf(z[f"{tag}_q"]), k=bf(z[f"{tag}_k"]), v=bf(z[f"{tag}_v"]),
                   seq_lens=z[f"{tag}_seq_lens"].tolist(), decoder=torch.from_numpy(z[f"{tag}_decoder"]),
                   encoder=torch.from_numpy(z[f"{tag}_encoder"]))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_varlen_oracle_vs_reference_torch_naive_golden(dtype):
    """oracle_varlen_attention against PrefillOnlyTorchNaiveBackendImpl.forward outputs recorded by
    oracle/make_golden.py (fp32 run of the reference on bf16-exact inputs)."""
    n = 0
    for c in _load_prefill_only_cases():
        H, KVH, D = c["H"], c["KVH"], 64
        q = c["q"].to(dtype).view(-1, H, D)
        k = c["k"].to(dtype).view(-1, KVH, D)
        v = c["v"].to(dtype).view(-1, KVH, D)
        cu = torch.tensor([0] + list(np.cumsum(c["seq_lens"])), dtype=torch.int32)
        for name, causal in (("decoder", True), ("encoder", False)):
            out = torch.full_like(q, float("nan"))
            oracle.varlen_attention(out, q, k, v, cu, D ** -0.5, causal)
            want = c[name].view(-1, H, D)
            tol = {torch.bfloat16: 2.0 ** -7, torch.float16: 2.0 ** -10, torch.float32: 1e-5}[dtype]
            assert float((out.float() - want).abs().max()) <= tol * max(1.0, float(want.abs().max())), (c["tag"], name)
            n += 1
    assert n == 10
