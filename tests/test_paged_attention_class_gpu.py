"""SURVEY a9, directly: the `PagedAttention` namespace (light-vllm_amd/paged_attn.py; reference
light_vllm/decoding/backends/attention/ops/paged_attn.py:34-248) without an engine around it -- cache shape and the
two views that alias its bytes, `write_to_paged_cache` into those views, `forward_decode` in its v1 and v2 forms
against each other, the oracle and an independent fp64 statement, `swap_blocks` / `copy_blocks` on the layer tensors,
and the CU count the v1 / v2 choice is made with (asked of the device, not a constant)."""
import pytest
import torch

from helpers import dense_attention_fp64, make_paged_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def attention_close(out, ref, what, tol=2e-2):
    """max-abs <= 2e-2 * max|out| and cosine >= 0.999 per (sequence, head)."""
    o, r = out.double().cpu(), ref.double().cpu()
    scale = r.abs().max().clamp_min(1e-6)
    assert ((o - r).abs().max() / scale).item() <= tol, what
    cos = torch.nn.functional.cosine_similarity(o, r, dim=-1)
    nz = r.abs().sum(-1) > 0
    assert (cos[nz] >= 0.999).all(), (what, cos[nz].min())


def PA():
    import light_vllm_amd  # noqa: F401
    from light_vllm_amd.paged_attn import PagedAttention
    return PagedAttention


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_kv_cache_shape_and_split_views_alias_the_same_bytes(dtype):
    P = PA()
    NB, BS, KVH, D = 11, 16, 4, 128
    shape = P.get_kv_cache_shape(NB, BS, KVH, D)
    assert shape == (2, NB, BS * KVH * D)  # paged_attn.py:40-48
    kv = torch.zeros(shape, dtype=dtype, device=DEV)
    kc, vc = P.split_kv_cache(kv, KVH, D)
    x = 16 // kv.element_size()
    assert kc.shape == (NB, KVH, D // x, BS, x) and vc.shape == (NB, KVH, D, BS)  # paged_attn.py:50-63
    assert kc.data_ptr() == kv[0].data_ptr() and vc.data_ptr() == kv[1].data_ptr()
    # a write through a view lands in the layer tensor at the index formula of cache_kernels.cu:184-192
    kc[3, 2, 5, 7, 1] = 1.5
    vc[4, 1, 9, 6] = -2.0
    flat_k = ((2 * (D // x) + 5) * BS + 7) * x + 1
    flat_v = (1 * D + 9) * BS + 6
    assert kv[0, 3, flat_k].item() == 1.5 and kv[1, 4, flat_v].item() == -2.0
    assert int((kv != 0).sum()) == 2


def test_write_to_paged_cache_fills_the_views_like_the_oracle():
    from oracle import oracle
    P = PA()
    NB, BS, KVH, D, T = 9, 16, 8, 128, 37
    g = torch.Generator().manual_seed(1)
    key = (torch.randn(T, KVH, D, generator=g) * 0.5).to(torch.bfloat16)
    value = (torch.randn(T, KVH, D, generator=g) * 0.5).to(torch.bfloat16)
    slots = torch.randperm(NB * BS, generator=g)[:T].to(torch.int64)
    slots[5] = -1  # padding: skipped
    kv = torch.zeros(P.get_kv_cache_shape(NB, BS, KVH, D), dtype=torch.bfloat16, device=DEV)
    kc, vc = P.split_kv_cache(kv, KVH, D)
    P.write_to_paged_cache(key.to(DEV), value.to(DEV), kc, vc, slots.to(DEV), "auto", 1.0, 1.0)
    ek = torch.zeros(NB, KVH, D // 8, BS, 8, dtype=torch.bfloat16)
    ev = torch.zeros(NB, KVH, D, BS, dtype=torch.bfloat16)
    oracle.reshape_and_cache(key, value, ek, ev, slots)
    torch.cuda.synchronize()
    assert torch.equal(kc.cpu(), ek) and torch.equal(vc.cpu(), ev)
    assert torch.equal(kv.cpu()[0].view(-1), ek.view(-1))  # the layer tensor holds exactly those bytes


@pytest.mark.parametrize("seq_lens", [[1024, 700, 513, 17], [2048, 1, 1500, 600, 2047, 512]])
def test_forward_decode_v1_and_v2_agree_with_each_other_the_oracle_and_fp64(seq_lens):
    from oracle import oracle
    P = PA()
    H, KVH, D, BS = 32, 8, 128, 16
    inp = make_paged_inputs(len(seq_lens), H, KVH, D, BS, seq_lens, dtype=torch.bfloat16, seed=3)
    q = inp["query"]
    exp = torch.zeros_like(q)
    oracle.paged_attention_v1(exp, q, inp["key_cache"], inp["value_cache"], KVH, inp["scale"], inp["block_tables"],
                              inp["seq_lens"], BS, inp["max_seq_len"])
    ref64 = dense_attention_fp64(inp)
    d = {k: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for k, v in inp.items()}
    outs = {}
    for ver in ("v1", "v2", None):
        outs[ver] = P.forward_decode(d["query"], d["key_cache"], d["value_cache"], d["block_tables"], d["seq_lens"],
                                     inp["max_seq_len"], "auto", KVH, inp["scale"], None, 1.0, 1.0,
                                     force_version=ver)
    torch.cuda.synchronize()
    for ver, o in outs.items():
        attention_close(o.cpu(), exp, what=f"forward_decode {ver} vs oracle")
        attention_close(o.cpu(), ref64, what=f"forward_decode {ver} vs fp64")
    # v1 against v2: <= 2 ulp at row scale
    a, b = outs["v1"].double().cpu(), outs["v2"].double().cpu()
    mag = torch.maximum(a.abs(), b.abs()).amax(dim=-1, keepdim=True).clamp_min(1e-30)
    ulp = torch.pow(2.0, torch.floor(torch.log2(mag)) - 7)  # bf16: 8 significant bits
    assert bool(((a - b).abs() <= 2 * ulp).all())
    # a caller-owned scratch of the reference's shapes (paged_attn.py:156-166) and output buffer are used as given
    parts = -(-inp["max_seq_len"] // 512)
    S = len(seq_lens)
    es = torch.empty(S, H, parts, dtype=torch.float32, device=DEV)
    scratch = (es, torch.empty_like(es), torch.empty(S, H, parts, D, dtype=torch.bfloat16, device=DEV))
    out = torch.full_like(d["query"], float("nan"))
    r = P.forward_decode(d["query"], d["key_cache"], d["value_cache"], d["block_tables"], d["seq_lens"],
                         inp["max_seq_len"], "auto", KVH, inp["scale"], None, 1.0, 1.0, force_version="v2",
                         scratch=scratch, output=out)
    torch.cuda.synchronize()
    assert r.data_ptr() == out.data_ptr() and torch.equal(out, outs["v2"])


def test_the_v1_v2_choice_uses_the_devices_cu_count():
    import light_vllm_amd  # noqa: F401
    from light_vllm_amd import paged_attn
    P = paged_attn.PagedAttention
    n = paged_attn.num_compute_units(DEV)
    assert n == torch.cuda.get_device_properties(0).multi_processor_count and n > 0
    assert paged_attn.num_compute_units() == n
    # one partition: always one pass; otherwise one pass iff the (sequence, kv head) grid fills the CUs
    assert P.use_v1(1, 8, 32, 512, n)
    assert P.use_v1(n // 8, 8, 32, 4096, n) and not P.use_v1(n // 8 - 1, 8, 32, 4096, n)
    assert not P.use_v1(4, 8, 32, 4096, 256) and P.use_v1(4, 8, 32, 4096, 32)


def test_swap_and_copy_blocks_on_the_layer_tensors():
    P = PA()
    NB, BS, KVH, D, L = 12, 16, 2, 64, 3
    g = torch.Generator().manual_seed(4)
    shape = P.get_kv_cache_shape(NB, BS, KVH, D)
    gpu = [(torch.randn(shape, generator=g)).to(torch.bfloat16).to(DEV) for _ in range(L)]
    before = [c.clone() for c in gpu]
    pairs = torch.tensor([[0, 5], [0, 7], [3, 9]], dtype=torch.int64, device=DEV)
    P.copy_blocks(gpu, pairs)
    torch.cuda.synchronize()
    for c, b in zip(gpu, before):
        want = b.clone()
        for s_, d_ in pairs.tolist():
            want[:, d_] = b[:, s_]
        assert torch.equal(c, want)
    host = torch.zeros(shape, dtype=torch.bfloat16).pin_memory()
    mapping = torch.tensor([[9, 1], [2, 4]], dtype=torch.int64)
    P.swap_blocks(gpu[0], host, mapping)
    torch.cuda.synchronize()
    for s_, d_ in mapping.tolist():
        assert torch.equal(host[:, d_], gpu[0][:, s_].cpu())
    assert int((host != 0).any(dim=-1).sum()) == 4  # two blocks in each plane, nothing else
