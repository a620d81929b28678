"""Caches whose blocks are padded apart (CacheConfig.block_pad_bytes: blocks a power of two apart alias on the HBM
channels, profiles/r03_tuning.md section 9): every op that addresses blocks takes the stride from the tensors
(key_cache.stride(0)), so a padded cache and the reference's dense one hold and return the same bits -- cache writes
(reshape_and_cache, its tiled form, rotary_embedding_and_cache, the fused rope + cache + attention launch), block
copies and swaps, decode and prompt attention, and the engine end to end."""
import pytest
import torch

from helpers import make_paged_inputs, make_prefill_inputs
from test_ops_gpu import to_dev

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
PAD = 1024  # bytes


def padded_like(t, pad_bytes=PAD, fill=float("nan")):
    """The same blocks inside a buffer whose rows are pad_bytes longer; the padding holds `fill`."""
    nb = t.shape[0]
    per = t[0].numel()
    pad = pad_bytes // t.element_size()
    buf = torch.full((nb, per + pad), fill, dtype=t.dtype, device=t.device) if t.dtype.is_floating_point else \
        torch.full((nb, per + pad), 0x7f, dtype=t.dtype, device=t.device)
    buf[:, :per] = t.reshape(nb, per)
    v = buf[:, :per].view(t.shape)
    assert v.stride(0) == per + pad and v.data_ptr() == buf.data_ptr()
    return v


def dense(t):
    return t.contiguous()


@pytest.mark.parametrize("kv", ["auto", "fp8"])
@pytest.mark.parametrize("tokens", [5, 700])  # 700: the LDS-tiled cache write
def test_cache_writes_into_padded_blocks(ops, kv, tokens):
    KVH, D, BS, NB = 8, 128, 16, 64
    g = torch.Generator(device=DEV).manual_seed(1)
    key = (torch.randn(tokens, KVH, D, device=DEV, generator=g) * 0.5).to(torch.bfloat16)
    value = (torch.randn(tokens, KVH, D, device=DEV, generator=g) * 0.5).to(torch.bfloat16)
    slots = torch.randperm(NB * BS, device=DEV, generator=g)[:tokens].sort().values if tokens > 100 else \
        torch.randperm(NB * BS, device=DEV, generator=g)[:tokens]
    slots = slots.to(torch.long)
    slots[0] = -1  # a padding token
    cdt = torch.bfloat16 if kv == "auto" else torch.uint8
    x = 8 if kv == "auto" else 16
    outs = []
    for pad in (False, True):
        kc = torch.zeros(NB, KVH, D // x, BS, x, dtype=cdt, device=DEV)
        vc = torch.zeros(NB, KVH, D, BS, dtype=cdt, device=DEV)
        if pad:
            kc, vc = padded_like(kc, fill=0.0), padded_like(vc, fill=0.0)
        ops.reshape_and_cache(key, value, kc, vc, slots, kv, 1.0, 1.0)
        torch.cuda.synchronize()
        outs.append((dense(kc), dense(vc)))
    assert torch.equal(outs[0][0].view(torch.uint8), outs[1][0].view(torch.uint8))
    assert torch.equal(outs[0][1].view(torch.uint8), outs[1][1].view(torch.uint8))
    assert outs[0][0].view(torch.uint8).any()


@pytest.mark.parametrize("kv", ["auto", "fp8"])
def test_rope_and_cache_write_into_padded_blocks(ops, kv):
    H, KVH, D, BS, NB, T = 32, 8, 128, 16, 40, 9
    g = torch.Generator(device=DEV).manual_seed(2)
    pos = torch.randint(0, 500, (T,), device=DEV, generator=g)
    cos_sin = torch.randn(2048, D, device=DEV, generator=g).to(torch.bfloat16)
    slots = torch.randperm(NB * BS, device=DEV, generator=g)[:T].to(torch.long)
    cdt = torch.bfloat16 if kv == "auto" else torch.uint8
    x = 8 if kv == "auto" else 16
    q0 = (torch.randn(T, H * D, device=DEV, generator=g) * 0.5).to(torch.bfloat16)
    k0 = (torch.randn(T, KVH * D, device=DEV, generator=g) * 0.5).to(torch.bfloat16)
    v0 = (torch.randn(T, KVH * D, device=DEV, generator=g) * 0.5).to(torch.bfloat16)
    outs = []
    for pad in (False, True):
        kc = torch.zeros(NB, KVH, D // x, BS, x, dtype=cdt, device=DEV)
        vc = torch.zeros(NB, KVH, D, BS, dtype=cdt, device=DEV)
        if pad:
            kc, vc = padded_like(kc, fill=0.0), padded_like(vc, fill=0.0)
        q, k = q0.clone(), k0.clone()
        assert torch.ops._C_amd.rotary_embedding_and_cache(pos, q, k, v0, D, cos_sin, True, kc, vc, slots, kv, 1.0, 1.0)
        torch.cuda.synchronize()
        outs.append((q, k, dense(kc), dense(vc)))
    for a, b in zip(*outs):
        assert torch.equal(a.view(torch.uint8), b.view(torch.uint8))


@pytest.mark.parametrize("kv", ["auto", "fp8"])
def test_attention_over_padded_blocks_is_bit_identical(ops, kv):
    """paged_attention_v1 / v2, the prompt kernels (all three bodies) and the fused rope + cache + attention launch:
    the padding holds NaN (0x7f bytes for fp8), the results are those of the dense caches."""
    inp = make_paged_inputs(6, 32, 8, 128, 16, [1024, 37, 500, 1, 129, 64], dtype=torch.bfloat16, seed=3)
    d = to_dev(inp)
    if kv == "fp8":
        d["key_cache"] = (d["key_cache"].float().reshape(d["key_cache"].shape[0], 8, 16, 16, 8).permute(0, 1, 2, 4, 3)
                          .reshape(-1, 8, 128, 16).view(-1, 8, 8, 16, 16).transpose(3, 4).contiguous()
                          .to(torch.float8_e4m3fn).view(torch.uint8))
        d["value_cache"] = d["value_cache"].to(torch.float8_e4m3fn).view(torch.uint8)
    kcs = (d["key_cache"], padded_like(d["key_cache"]))
    vcs = (d["value_cache"], padded_like(d["value_cache"]))
    res = []
    for kc, vc in zip(kcs, vcs):
        o1 = torch.zeros_like(d["query"])
        ops.paged_attention_v1(o1, d["query"], kc, vc, 8, inp["scale"], d["block_tables"], d["seq_lens"], 16, 1024, None,
                               kv, 1.0, 1.0)
        P = 2
        o2 = torch.zeros_like(d["query"])
        es = torch.zeros(6, 32, P, dtype=torch.float32, device=DEV)
        ml, tmp = torch.zeros_like(es), torch.zeros(6, 32, P, 128, dtype=torch.bfloat16, device=DEV)
        ops.paged_attention_v2(o2, es, ml, tmp, d["query"], kc, vc, 8, inp["scale"], d["block_tables"], d["seq_lens"], 16,
                               1024, None, kv, 1.0, 1.0)
        torch.cuda.synchronize()
        res.append((o1, o2))
    for a, b in zip(*res):
        assert torch.isfinite(b.float()).all()
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))


@pytest.mark.parametrize("shape", [([300, 90, 17], [300, 10, 1]), ([200] * 30 + [300], [1] * 30 + [30]), ([40] * 200, [20] * 200)])
def test_prompt_attention_over_padded_blocks_is_bit_identical(ops, shape):
    seq, ql = shape
    inp = make_prefill_inputs(8, 2, 128, 16, seq, ql, dtype=torch.bfloat16, seed=4)
    d = to_dev(inp)
    outs = []
    for kc, vc in ((d["key_cache"], d["value_cache"]), (padded_like(d["key_cache"]), padded_like(d["value_cache"]))):
        out = torch.full_like(d["query"], float("nan"))
        ops.paged_prefill_attention(out, d["query"], kc, vc, 2, inp["scale"], d["block_tables"], d["seq_lens"],
                                    d["query_start_loc"], inp["max_query_len"], 16, None, 0, 0.0, "auto")
        torch.cuda.synchronize()
        outs.append(out)
    assert torch.isfinite(outs[1].float()).all()
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))


def test_copy_and_swap_of_padded_blocks(ops):
    """copy_blocks and swap_blocks move a block and its padding; both sides pad alike (the engine's CPU cache does)."""
    KVH, D, BS, NB, L = 2, 64, 16, 12, 3
    g = torch.Generator(device=DEV).manual_seed(5)
    per = KVH * D * BS
    dense_k = [torch.randn(NB, per, device=DEV, generator=g).to(torch.bfloat16) for _ in range(L)]
    dense_v = [torch.randn(NB, per, device=DEV, generator=g).to(torch.bfloat16) for _ in range(L)]
    pk, pv = [padded_like(t, fill=0.0) for t in dense_k], [padded_like(t, fill=0.0) for t in dense_v]
    pairs = torch.tensor([[0, 5], [0, 7], [3, 1]], dtype=torch.long)
    ops.copy_blocks(dense_k, dense_v, pairs.to(DEV))
    ops.copy_blocks(pk, pv, pairs.to(DEV))
    torch.cuda.synchronize()
    for a, b in zip(dense_k + dense_v, pk + pv):
        assert torch.equal(a, dense(b))
    # swap out three scattered blocks and a run of four, back into other blocks
    hbuf = torch.zeros(NB, per + PAD // 2, dtype=torch.bfloat16).pin_memory()
    host = hbuf[:, :per]
    out_pairs = torch.tensor([[2, 0], [9, 1], [4, 2], [5, 3], [6, 4], [7, 5]], dtype=torch.long)
    ops.swap_blocks(pk[0], host, out_pairs)
    torch.cuda.synchronize()
    for s, t in out_pairs.tolist():
        assert torch.equal(host[t], dense(pk[0])[s].cpu())
    back = torch.tensor([[0, 11], [1, 10], [2, 8]], dtype=torch.long)
    ops.swap_blocks(host, pk[1], back)
    torch.cuda.synchronize()
    for s, t in back.tolist():
        assert torch.equal(dense(pk[1])[t].cpu(), host[s])


@pytest.mark.parametrize("graph", [False, True])
def test_engine_tokens_do_not_depend_on_the_padding(graph):
    """The engine with block_pad_bytes = 1024 (what None resolves to for an 8B model's bf16 blocks), with the default
    and with 0 (the reference's dense layout): same greedy tokens, with
    prompts, decode steps, chunked prefill and swap-outs in the run (22 blocks for 6 sequences)."""
    import light_vllm_amd  # noqa: F401
    from light_vllm_amd.engine.config import CacheConfig, ModelConfig, SchedulerConfig
    from light_vllm_amd.engine.llm_engine import LLMEngine
    from test_engine_gpu import run_to_completion

    def run(pad):
        eng = LLMEngine(ModelConfig.tiny(),
                        CacheConfig(block_size=16, num_gpu_blocks=22, num_cpu_blocks=32, block_pad_bytes=pad),
                        SchedulerConfig(max_num_batched_tokens=64, max_num_seqs=8, max_model_len=512,
                                        chunked_prefill_enabled=True, preemption_mode="swap", max_num_on_the_fly=2),
                        device=DEV, use_hip_graph=graph, seed=0)
        kv = eng.worker.cache_engine.gpu_cache[0]
        # tiny model: blocks of 2 kv heads x 64 x 16 tokens of bf16 = 4 KiB; None = 1/32 of that, in multiples of 256
        assert kv.stride(1) * kv.element_size() == 2 * 64 * 16 * 2 + (256 if pad is None else pad)
        return run_to_completion(eng, max_tokens=40)

    assert run(1024) == run(0) == run(None)
