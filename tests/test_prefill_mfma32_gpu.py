"""Parity of the 32x32-MFMA prefill body (prefill_mfma32.h: long plain chunks, head size 64 / 128) against the
CPU oracle and fp64, through the same op and C-ABI entry as tests/test_prefill_gpu.py; the tuning key
`prefill_mfma32_min_query` is lowered to 1 so that every chunk of these cases takes it.

Bar: max-abs <= 2e-2 * max|out| and cosine >= 0.999 per (token, head), as for paged_attention.
"""
import pytest
import torch

from helpers import dense_prefill_fp64, make_prefill_inputs
from oracle import oracle  # noqa: F401
from test_ops_gpu import check_attention, to_dev
from test_prefill_gpu import RAGGED, run_hip, run_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def mfma32_everywhere(ops):
    default = int(torch.ops._C_amd.get_tuning("prefill_mfma32_min_query"))
    torch.ops._C_amd.set_tuning("prefill_mfma32_min_query", 1)
    yield
    torch.ops._C_amd.set_tuning("prefill_mfma32_min_query", default)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("block_size", [16, 32])
@pytest.mark.parametrize("D", [64, 128])
@pytest.mark.parametrize("H,KVH", [(8, 8), (8, 4), (8, 2), (14, 2), (8, 1), (16, 1), (20, 1), (40, 1)])
def test_mfma32_gqa_groups(ops, dtype, block_size, D, H, KVH):
    inp = make_prefill_inputs(H, KVH, D, block_size, RAGGED["seq"], RAGGED["ql"], dtype=dtype, seed=H + KVH)
    out = run_hip(ops, inp)
    assert torch.isfinite(out).all()
    check_attention(out, run_oracle(inp), dense_prefill_fp64(inp))


@pytest.mark.parametrize("block_size", [16, 32])
def test_mfma32_many_tiles_and_partial_workgroups(ops, block_size):
    """Contexts of several 64-key tiles, chunks that end inside a workgroup's 64 query tokens, a chunk over a
    long cached context, a one-token chunk."""
    inp = make_prefill_inputs(8, 2, 128, block_size, [700, 333, 1025, 64, 65, 1], [700, 100, 65, 64, 1, 1],
                              dtype=torch.bfloat16, seed=5)
    out = run_hip(ops, inp)
    assert torch.isfinite(out).all()
    check_attention(out, run_oracle(inp), dense_prefill_fp64(inp))


@pytest.mark.parametrize("block_size", [16, 32])
def test_mfma32_garbage_beyond_the_sequence_is_ignored(ops, block_size):
    inp = make_prefill_inputs(8, 2, 128, block_size, [35, 70, 17, 130], [35, 21, 1, 130], dtype=torch.bfloat16, seed=9,
                              garbage=float("nan"))
    out = run_hip(ops, inp)
    assert torch.isfinite(out).all()
    check_attention(out, run_oracle(inp), dense_prefill_fp64(inp))


def test_mfma32_late_spike_forces_rescale(ops):
    """A key near the end of the context whose logit exceeds the running maximum by far more than the 2^6 the
    lazy rescale tolerates: the accumulators of the columns that see it must be rescaled."""
    inp = make_prefill_inputs(4, 1, 128, 16, [400], [64], dtype=torch.bfloat16, seed=21)
    b = int(inp["block_tables"][0, 390 // 16])
    k = torch.ones(128, dtype=torch.bfloat16) * 1.5
    inp["key_cache"][b, 0, :, 390 % 16, :] = k.view(16, 8)
    inp["k_dense"][0][390, 0] = k
    inp["query"][:, 0] = 1.0
    out = run_hip(ops, inp)
    check_attention(out, run_oracle(inp), dense_prefill_fp64(inp))


def test_mfma32_moderate_growth_without_rescale_stays_accurate(ops):
    """Logits that keep growing by less than the rescale threshold per tile: probabilities above 1 are summed and
    rounded relative to their own size."""
    inp = make_prefill_inputs(4, 1, 128, 16, [512], [512], dtype=torch.bfloat16, seed=22)
    ramp = torch.linspace(0.0, 2.0, 512).view(512, 1).to(torch.bfloat16)
    for t in range(512):
        b = int(inp["block_tables"][0, t // 16])
        inp["key_cache"][b, 0, :, t % 16, :] = ramp[t]
        inp["k_dense"][0][t, 0] = ramp[t]
    inp["query"][:] = 0.25
    out = run_hip(ops, inp)
    check_attention(out, run_oracle(inp), dense_prefill_fp64(inp))


def test_mfma32_chunks_compose_to_the_whole_prompt(ops):
    """A column's arithmetic depends on its own key walk only (64-key tiles from key 0, its own rescale
    decisions): a prompt computed in chunks gives the bits of the single-chunk run."""
    inp = make_prefill_inputs(8, 2, 128, 16, [300], [300], dtype=torch.bfloat16, seed=4)
    whole = run_hip(ops, inp)
    pieces, done = [], 0
    for n in (128, 100, 72):
        part = dict(inp)
        part["query"] = inp["query"][done:done + n].contiguous()
        part["seq_lens"] = torch.tensor([done + n], dtype=torch.int32)
        part["query_start_loc"] = torch.tensor([0, n], dtype=torch.int32)
        part["max_query_len"] = n
        pieces.append(run_hip(ops, part))
        done += n
    assert torch.equal(torch.cat(pieces).view(torch.int16), whole.view(torch.int16))


def test_mfma32_strided_query_and_output(ops):
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


def test_mfma32_masks_and_biases_still_take_the_other_body(ops):
    """ALiBi / sliding window / soft cap are not this body's cases: the dispatch must leave them to prefill_mfma.h."""
    H = 8
    inp = make_prefill_inputs(H, 2, 128, 16, [300, 90], [300, 10], dtype=torch.float16, seed=3)
    alibi = torch.tensor([0.5 ** (i + 1) for i in range(H)])
    out = run_hip(ops, inp, alibi, 50, 0.0)
    check_attention(out, run_oracle(inp, alibi, 50, 0.0), dense_prefill_fp64(inp, alibi, 50, 0.0))


@pytest.mark.parametrize("block_size", [16, 32])
def test_mfma32_long_prompt_properties(ops, block_size):
    """Config-3 sized: sampled rows against fp64, and identical bits under another physical block placement."""
    H, KVH, D, BS, S = 32, 8, 128, block_size, 4096
    inp = make_prefill_inputs(H, KVH, D, BS, [S], [S], dtype=torch.bfloat16, seed=77)
    out = run_hip(ops, inp).cpu()
    k, v = inp["k_dense"][0].double(), inp["v_dense"][0].double()
    for t in (0, 1, 15, 16, 17, 63, 64, 1023, 2048, 4095):
        for h in (0, 5, 31):
            logits = (k[:t + 1, h // 4] @ inp["query"][t, h].double()) * inp["scale"]
            want = torch.softmax(logits, 0) @ v[:t + 1, h // 4]
            assert float((out[t, h].double() - want).abs().max()) <= 2e-2 * max(float(want.abs().max()), 1e-3)
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
def test_mfma32_non_causal_sees_the_whole_context(ops, block_size):
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


def test_mfma32_dense_varlen_twin(ops):
    """lvllm_varlen_attention (encode-only models, no cache) packs K/V into scratch tiles and runs the same kernel."""
    torch.manual_seed(3)
    H, KVH, D = 8, 4, 64
    lens = [70, 1, 129, 64]
    T = sum(lens)
    q = (torch.randn(T, H, D) * 0.5).to(torch.bfloat16)
    k = (torch.randn(T, KVH, D) * 0.5).to(torch.bfloat16)
    v = (torch.randn(T, KVH, D) * 0.5).to(torch.bfloat16)
    cu = torch.tensor([0] + torch.tensor(lens).cumsum(0).tolist(), dtype=torch.int32)
    for causal in (True, False):
        out = torch.full_like(q, float("nan")).to(DEV)
        ops.varlen_attention(out, q.to(DEV), k.to(DEV), v.to(DEV), cu.to(DEV), max(lens), D ** -0.5, causal)
        torch.cuda.synchronize()
        want = torch.zeros(T, H, D, dtype=torch.float64)
        for s, n in enumerate(lens):
            a, b = int(cu[s]), int(cu[s + 1])
            for h in range(H):
                logits = (q[a:b, h].double() @ k[a:b, h // 2].double().T) * D ** -0.5
                if causal:
                    logits = logits.masked_fill(torch.ones(n, n).triu(1).bool(), float("-inf"))
                want[a:b, h] = torch.softmax(logits, 1) @ v[a:b, h // 2].double()
        check_attention(out.cpu(), want)


def test_mfma32_block_numbers_beyond_the_stated_extent_are_clamped_not_followed(ops):
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


@pytest.mark.parametrize("block_size", [16, 32])
def test_mfma32_contexts_longer_than_one_block_number_chunk(ops, block_size):
    """The body keeps the block numbers of 64 tiles (4 096 keys) in a register and reloads it every 64 tiles: a chunk
    over a context of more than 64 tiles has to pick up the second chunk's numbers."""
    inp = make_prefill_inputs(8, 2, 128, block_size, [4500, 8300], [130, 70], dtype=torch.bfloat16, seed=51)
    out = run_hip(ops, inp)
    assert torch.isfinite(out).all()
    check_attention(out, run_oracle(inp), dense_prefill_fp64(inp))


@pytest.mark.parametrize("shape", [([2080] * 2, [32] * 2), ([1040] * 4, [16] * 4), ([4128], [32]),
                                   ([700, 2080, 130], [40, 64, 9]), ([1500], [100])])
@pytest.mark.parametrize("block_size", [16, 32])
def test_partitioned_key_walk_matches_single_pass_and_the_oracle(ops, shape, block_size):
    """Few workgroups over long contexts: with max_seq_len stated (and the scratch the op allocates) the key walk is cut
    across workgroups (csrc/prefill_partitions.h) and merged per row over the partitions the row's own horizon
    reaches.  Same bar as the single pass; against it the partial results are rounded once more to the model dtype."""
    seq, ql = shape
    inp = make_prefill_inputs(8, 2, 128, block_size, seq, ql, dtype=torch.bfloat16, seed=7)
    need = torch.ops._C_amd.paged_prefill_workspace_bytes(len(seq), sum(ql), max(ql), 8, 2, 128, max(seq))
    assert need > 0  # these launches are cut
    d = to_dev(inp)

    def run(max_seq_len):
        out = torch.full_like(d["query"], float("nan"))
        ops.paged_prefill_attention(out, d["query"], d["key_cache"], d["value_cache"], 2, inp["scale"],
                                    d["block_tables"], d["seq_lens"], d["query_start_loc"], inp["max_query_len"],
                                    block_size, None, 0, 0.0, "auto", True, 1.0, 1.0, max_seq_len)
        torch.cuda.synchronize()
        return out

    cut, whole = run(max(seq)), run(0)
    want, ref64 = run_oracle(inp), dense_prefill_fp64(inp)
    check_attention(cut, want, ref64)
    check_attention(whole, want, ref64)
    assert not torch.equal(cut.view(torch.int16), whole.view(torch.int16))  # (the cut is observable)
    assert float((cut.float() - whole.float()).abs().max()) <= 2e-2 * float(whole.float().abs().max())


def test_partitioned_key_walk_ignores_garbage_past_the_sequences(ops):
    inp = make_prefill_inputs(8, 2, 128, 16, [1030, 517], [20, 33], dtype=torch.bfloat16, seed=9, garbage=float("nan"))
    d = to_dev(inp)
    out = torch.full_like(d["query"], float("nan"))
    ops.paged_prefill_attention(out, d["query"], d["key_cache"], d["value_cache"], 2, inp["scale"], d["block_tables"],
                                d["seq_lens"], d["query_start_loc"], inp["max_query_len"], 16, None, 0, 0.0, "auto",
                                True, 1.0, 1.0, 1030)
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    check_attention(out, run_oracle(inp), dense_prefill_fp64(inp))


def test_partitioned_key_walk_with_an_understated_bound_still_walks_every_key(ops):
    """max_seq_len is the caller's promise; a context longer than it (a graph captured for shorter ones) is still walked
    to its end -- the last partition is open-ended."""
    inp = make_prefill_inputs(8, 2, 128, 16, [2080, 1500], [32, 20], dtype=torch.bfloat16, seed=13)
    d = to_dev(inp)
    out = torch.full_like(d["query"], float("nan"))
    ops.paged_prefill_attention(out, d["query"], d["key_cache"], d["value_cache"], 2, inp["scale"], d["block_tables"],
                                d["seq_lens"], d["query_start_loc"], inp["max_query_len"], 16, None, 0, 0.0, "auto",
                                True, 1.0, 1.0, 1024)  # half the real length
    torch.cuda.synchronize()
    check_attention(out, run_oracle(inp), dense_prefill_fp64(inp))
