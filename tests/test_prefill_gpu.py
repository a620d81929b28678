"""Parity of the HIP prefill kernel (torch.ops._C_amd.paged_prefill_attention -> C-ABI
lvllm_paged_prefill_attention -> prefill_mfma.h) against the CPU oracle and an independent fp64
dense computation: prompts, chunked prefill over a cached context, prefix hits, ragged batches.

Bar (same as paged_attention): max-abs <= 2e-2 * max|out| and cosine >= 0.999 per
(token, head) against the oracle and against fp64.
"""
import math

import pytest
import torch

from helpers import dense_prefill_fp64, make_prefill_inputs
from oracle import oracle
from test_ops_gpu import check_attention, to_dev

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def first_body_everywhere(ops, request):
    """These cases pin prefill_mfma.h (the 16x16x32 body: every head size, fp8 caches, masks and biases, short
    chunks); tests/test_prefill_mfma32_gpu.py runs the same cases through the 32x32-MFMA body, and
    test_prefill_dispatch_* below the shipped choice between the two."""
    default = int(torch.ops._C_amd.get_tuning("prefill_mfma32_min_query"))
    default_chunk = int(torch.ops._C_amd.get_tuning("prefill_chunk_max_query"))
    if "dispatch" not in request.node.name:
        torch.ops._C_amd.set_tuning("prefill_mfma32_min_query", 0)
        torch.ops._C_amd.set_tuning("prefill_chunk_max_query", 0)  # (tests/test_prefill_chunk_gpu.py has that walk)
    yield
    torch.ops._C_amd.set_tuning("prefill_mfma32_min_query", default)
    torch.ops._C_amd.set_tuning("prefill_chunk_max_query", default_chunk)


def run_hip(ops, inp, alibi=None, window=0, softcap=0.0, out=None, causal=True):
    d = to_dev(inp)
    if out is None:
        out = torch.full_like(d["query"], float("nan"))
    ops.paged_prefill_attention(out, d["query"], d["key_cache"], d["value_cache"], inp["num_kv_heads"],
                                inp["scale"], d["block_tables"], d["seq_lens"], d["query_start_loc"],
                                inp["max_query_len"], inp["block_size"],
                                alibi.to(DEV) if alibi is not None else None, window, softcap, "auto", causal)
    torch.cuda.synchronize()
    return out


def run_oracle(inp, alibi=None, window=0, softcap=0.0):
    out = torch.zeros_like(inp["query"])
    oracle.paged_prefill_attention(out, inp["query"], inp["key_cache"], inp["value_cache"],
                                   inp["num_kv_heads"], inp["scale"], inp["block_tables"], inp["seq_lens"],
                                   inp["query_start_loc"], inp["block_size"], alibi_slopes=alibi,
                                   sliding_window=window, softcap=softcap)
    return out


RAGGED = dict(seq=[37, 200, 5, 1, 129, 64, 48], ql=[37, 40, 5, 1, 129, 0, 17])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("block_size", [16, 32])
@pytest.mark.parametrize("H,KVH", [(8, 8), (8, 4), (8, 2), (14, 2), (8, 1), (16, 1), (20, 1)])
def test_prefill_gqa_groups(ops, dtype, block_size, H, KVH):
    inp = make_prefill_inputs(H, KVH, 64, block_size, RAGGED["seq"], RAGGED["ql"], dtype=dtype, seed=H + KVH)
    out = run_hip(ops, inp)
    assert torch.isfinite(out).all()
    check_attention(out, run_oracle(inp), dense_prefill_fp64(inp))


@pytest.mark.parametrize("head_size", [64, 80, 96, 112, 120, 128, 192, 256])
def test_prefill_head_sizes(ops, head_size):
    inp = make_prefill_inputs(8, 2, head_size, 16, [150, 33, 70], [150, 33, 19], dtype=torch.bfloat16, seed=head_size)
    out = run_hip(ops, inp)
    check_attention(out, run_oracle(inp), dense_prefill_fp64(inp))


@pytest.mark.parametrize("case", ["window", "alibi", "softcap", "window+alibi"])
def test_prefill_masks_and_biases(ops, case):
    H = 8
    inp = make_prefill_inputs(H, 2, 128, 16, [300, 90, 17, 64], [300, 10, 1, 48], dtype=torch.float16, seed=3)
    alibi = torch.tensor([0.5 ** (i + 1) for i in range(H)]) if "alibi" in case else None
    window = 50 if "window" in case else 0
    softcap = 1.5 if case == "softcap" else 0.0
    if softcap:
        inp["scale"] = 1.0
    out = run_hip(ops, inp, alibi, window, softcap)
    check_attention(out, run_oracle(inp, alibi, window, softcap), dense_prefill_fp64(inp, alibi, window, softcap))


def test_prefill_garbage_beyond_the_sequence_is_ignored(ops):
    """Cache slots past seq_len (rest of the last block, padding blocks) may hold NaN."""
    inp = make_prefill_inputs(8, 2, 128, 16, [35, 70, 17], [35, 21, 1], dtype=torch.bfloat16, seed=9,
                              garbage=float("nan"))
    out = run_hip(ops, inp)
    assert torch.isfinite(out).all()
    check_attention(out, run_oracle(inp), dense_prefill_fp64(inp))


def test_prefill_late_spike_forces_rescale(ops):
    """A key near the end of the context with a much larger logit: the lazy accumulator
    rescale must fire for the columns that see it and only for those."""
    inp = make_prefill_inputs(4, 1, 128, 16, [400], [64], dtype=torch.bfloat16, seed=21)
    # make key 390 align with every query of head 0
    s = 0
    b = int(inp["block_tables"][s, 390 // 16])
    k = torch.ones(128, dtype=torch.bfloat16) * 1.5
    inp["key_cache"][b, 0, :, 390 % 16, :] = k.view(16, 8)
    inp["k_dense"][s][390, 0] = k
    inp["query"][:, 0] = 1.0
    out = run_hip(ops, inp)
    check_attention(out, run_oracle(inp), dense_prefill_fp64(inp))


def test_prefill_chunks_compose_to_the_whole_prompt(ops):
    """Chunked prefill: computing a 300-token prompt in chunks of 128/128/44 against the same
    cache gives, row for row, the bits of the single-chunk run (each query's key walk is the same
    sequence of tiles whichever chunk it is in)."""
    inp = make_prefill_inputs(8, 2, 128, 16, [300], [300], dtype=torch.bfloat16, seed=4)
    whole = run_hip(ops, inp)
    pieces = []
    done = 0
    for n in (128, 128, 44):
        part = dict(inp)
        part["query"] = inp["query"][done:done + n].contiguous()
        part["seq_lens"] = torch.tensor([done + n], dtype=torch.int32)
        part["query_start_loc"] = torch.tensor([0, n], dtype=torch.int32)
        part["max_query_len"] = n
        pieces.append(run_hip(ops, part))
        done += n
    got = torch.cat(pieces)
    assert torch.equal(got.view(torch.int16), whole.view(torch.int16))


def test_prefill_single_token_chunks_match_decode(ops):
    """query_len 1 everywhere: the kernel computes what paged_attention_v1 computes."""
    inp = make_prefill_inputs(32, 8, 128, 16, [100, 257, 16, 1], [1, 1, 1, 1], dtype=torch.bfloat16, seed=8)
    out = run_hip(ops, inp)
    d = to_dev(inp)
    dec = torch.zeros_like(d["query"])
    ops.paged_attention_v1(dec, d["query"], d["key_cache"], d["value_cache"], 8, inp["scale"], d["block_tables"],
                           d["seq_lens"], 16, 257, None, "auto", 1.0, 1.0)
    check_attention(out, dec.cpu(), None, tol=5e-3)


def test_decode_with_a_logits_soft_cap_goes_through_one_token_chunks(ops):
    """PagedAttnImpl(logits_soft_cap=...) (the reference's live decode passes the cap to flash_attn_with_kvcache,
    flash_attn.py:554; paged_attention_v1/v2 have no such argument): decode tokens run as chunks of one token through
    the prefill kernel -- forward(), decode_attention() and the refusal of the fused rope + attention launch --
    against the oracle and fp64 with the cap, and visibly different from the run without it."""
    from light_vllm_amd.attention.backend import PagedAttnImpl, PagedAttnMetadata
    lens = [100, 257, 16, 1, 513]
    inp = make_prefill_inputs(32, 8, 128, 16, lens, [1] * len(lens), dtype=torch.bfloat16, seed=18)
    inp["query"] = inp["query"] * 4  # logits large enough for the cap to bite
    cap = 1.5
    want_o, want_64 = run_oracle(inp, softcap=cap), dense_prefill_fp64(inp, None, 0, cap)
    d = to_dev(inp)
    n = len(lens)
    impl = PagedAttnImpl(32, 128, inp["scale"], 8, None, None, "auto", None, cap)
    md = PagedAttnMetadata(num_prefills=0, num_prefill_tokens=0, num_decode_tokens=n,
                           slot_mapping=torch.full((n,), -1, dtype=torch.int64, device=DEV), seq_lens=None,
                           seq_lens_tensor=d["seq_lens"], max_query_len=1, max_prefill_seq_len=0, max_decode_seq_len=max(lens),
                           query_start_loc=None, seq_start_loc=None, context_lens_tensor=None, block_tables=d["block_tables"])
    q2 = d["query"].reshape(n, 32 * 128)
    got = impl.decode_attention(q2, d["key_cache"], d["value_cache"], md).view(n, 32, 128)
    torch.cuda.synchronize()
    check_attention(got.cpu(), want_o, want_64)
    assert impl.rope_cache_decode_attention(None, q2, None, None, None, d["key_cache"], d["value_cache"], md) is None
    plain = PagedAttnImpl(32, 128, inp["scale"], 8, None, None, "auto").decode_attention(q2, d["key_cache"], d["value_cache"], md)
    assert float((plain.view(n, 32, 128).cpu().float() - want_o.float()).abs().max()) > 0.05  # the cap changes the result


def test_prefill_strided_query_and_output(ops):
    """query as a view of the fused qkv projection, output into a slice of a wider buffer."""
    H, KVH, D = 8, 2, 128
    inp = make_prefill_inputs(H, KVH, D, 16, [90, 40], [50, 40], dtype=torch.bfloat16, seed=12)
    T = inp["query"].shape[0]
    qkv = torch.zeros(T, (H + 2 * KVH) * D, dtype=torch.bfloat16)
    qkv[:, :H * D] = inp["query"].view(T, H * D)
    d = to_dev(inp)
    q_view = qkv.to(DEV)[:, :H * D].view(T, H, D)
    wide = torch.zeros(T, 2 * H * D, dtype=torch.bfloat16, device=DEV)
    out_view = wide[:, :H * D].view(T, H, D)
    ops.paged_prefill_attention(out_view, q_view, d["key_cache"], d["value_cache"], KVH, inp["scale"],
                                d["block_tables"], d["seq_lens"], d["query_start_loc"], inp["max_query_len"], 16,
                                None, 0, 0.0, "auto")
    torch.cuda.synchronize()
    assert float(wide[:, H * D:].abs().max()) == 0.0
    check_attention(out_view, run_oracle(inp), dense_prefill_fp64(inp))


def test_prefill_argument_errors(ops):
    inp = make_prefill_inputs(8, 2, 128, 16, [20], [20], dtype=torch.bfloat16)
    d = to_dev(inp)
    out = torch.zeros_like(d["query"])
    with pytest.raises(RuntimeError, match="int32"):
        ops.paged_prefill_attention(out, d["query"], d["key_cache"], d["value_cache"], 2, 1.0,
                                    d["block_tables"].long(), d["seq_lens"], d["query_start_loc"], 20, 16,
                                    None, 0, 0.0, "auto")
    with pytest.raises(RuntimeError, match="fp8"):
        ops.paged_prefill_attention(out, d["query"], d["key_cache"], d["value_cache"], 2, 1.0,
                                    d["block_tables"], d["seq_lens"], d["query_start_loc"], 20, 16,
                                    None, 0, 0.0, "fp8")
    with pytest.raises(RuntimeError, match="float16 or bfloat16"):
        ops.paged_prefill_attention(out.float(), d["query"].float(), d["key_cache"].float(), d["value_cache"].float(), 2, 1.0,
                                    d["block_tables"], d["seq_lens"], d["query_start_loc"], 20, 16,
                                    None, 0, 0.0, "auto")


def test_prefill_long_prompt_properties(ops):
    """Config-3 sized (one 4096-token prompt, Llama-3-8B heads): too big for the scalar oracle in
    seconds, so check sampled rows against fp64 and block-permutation invariance bit for bit."""
    H, KVH, D, BS, S = 32, 8, 128, 16, 4096
    inp = make_prefill_inputs(H, KVH, D, BS, [S], [S], dtype=torch.bfloat16, seed=77)
    out = run_hip(ops, inp).cpu()
    # sampled rows vs fp64
    k, v = inp["k_dense"][0].double(), inp["v_dense"][0].double()
    for t in (0, 1, 15, 16, 17, 1023, 2048, 4095):
        for h in (0, 5, 31):
            logits = (k[:t + 1, h // 4] @ inp["query"][t, h].double()) * inp["scale"]
            want = torch.softmax(logits, 0) @ v[:t + 1, h // 4]
            assert float((out[t, h].double() - want).abs().max()) <= 2e-2 * max(float(want.abs().max()), 1e-3)
    # same data under another physical block placement: identical bits
    nb = inp["key_cache"].shape[0]
    perm = torch.randperm(nb, generator=torch.Generator().manual_seed(1))
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(nb)
    inp2 = dict(inp)
    inp2["key_cache"] = inp["key_cache"][perm]
    inp2["value_cache"] = inp["value_cache"][perm]
    inp2["block_tables"] = inv[inp["block_tables"].long()].to(torch.int32)
    out2 = run_hip(ops, inp2).cpu()
    assert torch.equal(out.view(torch.int16), out2.view(torch.int16))


@pytest.mark.parametrize("block_size", [16, 32])
def test_prefill_non_causal_sees_the_whole_context(ops, block_size):
    """causal=False (encoder attention over a paged context): every query attends to all
    seq_len keys, also those after it; NaN garbage past seq_len stays masked."""
    inp = make_prefill_inputs(8, 2, 64, block_size, [37, 200, 5, 129], [37, 40, 5, 129], dtype=torch.bfloat16,
                              seed=6, garbage=float("nan"))
    out = run_hip(ops, inp, causal=False).cpu()
    assert torch.isfinite(out).all()
    q = inp["query"]
    qsl = inp["query_start_loc"].tolist()
    want = torch.zeros(q.shape, dtype=torch.float64)
    for s, S in enumerate(inp["seq_lens"].tolist()):
        k, v = inp["k_dense"][s].double(), inp["v_dense"][s].double()
        for h in range(8):
            logits = (q[qsl[s]:qsl[s + 1], h].double() @ k[:, h // 4].T) * inp["scale"]
            want[qsl[s]:qsl[s + 1], h] = torch.softmax(logits, dim=1) @ v[:, h // 4]
    check_attention(out, want)


def test_prefill_dispatch_default_threshold(ops):
    """As shipped: a launch whose longest chunk has >= 64 query tokens takes the 32x32-MFMA body, and so does one with
    chunks of 16+ tokens whose grid fits the CUs at once; the others the first body (launches that are mostly
    one-token sequences take the walk of prefill_chunk.h: tests/test_prefill_chunk_gpu.py::test_dispatch_rule).  Both
    meet the bar, and a launch is bit-identical to the body it is documented to take."""
    assert int(torch.ops._C_amd.get_tuning("prefill_mfma32_min_query")) == 64
    # (chunk lengths, takes the 32x32 body): >= 64 tokens always; 16..63 when the grid fits the CUs (2 kv heads x 3
    # sequences here: it does); below 16 never; 16..63 in a launch of many sequences (2 x 200 workgroups) not either
    cases = [([300, 90, 17], [300, 10, 1], True), ([300, 90, 17], [63, 10, 1], True), ([64], [64], True),
             ([300, 90, 17], [15, 10, 1], False), ([40] * 200, [20] * 200, False)]
    for seq, ql, new_body in cases:
        inp = make_prefill_inputs(8, 2, 128, 16, seq, ql, dtype=torch.bfloat16, seed=31)
        out = run_hip(ops, inp)
        check_attention(out, run_oracle(inp), dense_prefill_fp64(inp))
        torch.ops._C_amd.set_tuning("prefill_mfma32_min_query", 1 if new_body else 0)
        torch.ops._C_amd.set_tuning("prefill_chunk_max_query", 0)
        same = run_hip(ops, inp)
        torch.ops._C_amd.set_tuning("prefill_mfma32_min_query", 64)
        torch.ops._C_amd.set_tuning("prefill_chunk_max_query", 64)
        assert torch.equal(out.view(torch.int16), same.view(torch.int16))


def test_prefill_block_numbers_beyond_the_stated_extent_are_clamped_not_followed(ops):
    """The caches are the FRONT of a larger allocation whose tail holds NaN: a block table that points past the
    caches' last block reads inside the stated extent (kv_cache_bytes of the C-ABI, which the torch binding states):
    the result is finite and equals the run with the numbers clamped by hand."""
    inp = make_prefill_inputs(8, 2, 128, 16, [300, 150], [300, 70], dtype=torch.bfloat16, seed=41)
    NB = inp["key_cache"].shape[0]
    d = to_dev(inp)
    big_k = torch.full((3 * NB,) + tuple(inp["key_cache"].shape[1:]), float("nan"), dtype=torch.bfloat16, device=DEV)
    big_v = torch.full((3 * NB,) + tuple(inp["value_cache"].shape[1:]), float("nan"), dtype=torch.bfloat16, device=DEV)
    big_k[:NB], big_v[:NB] = d["key_cache"], d["value_cache"]
    bad = d["block_tables"].clone()
    bad[0, 3] = NB + 5
    bad[1, 1] = 3 * NB - 1
    outs = []
    for bt in (bad, bad.clamp(max=NB - 1)):
        out = torch.full_like(d["query"], float("nan"))
        ops.paged_prefill_attention(out, d["query"], big_k[:NB], big_v[:NB], 2, inp["scale"], bt, d["seq_lens"],
                                    d["query_start_loc"], inp["max_query_len"], 16, None, 0, 0.0, "auto")
        torch.cuda.synchronize()
        outs.append(out)
    assert torch.isfinite(outs[0].float()).all()
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
