"""Parity of the mixed-step walk (csrc/prefill_chunk.h: launches that are mostly one-token sequences beside short prompt
chunks -- the steps of chunked prefill, flash_attn.py:538-555) against the CPU oracle and the fp64 dense statement,
through the same op and C-ABI entry as every other prompt launch (torch.ops._C_amd.paged_prefill_attention ->
lvllm_paged_prefill_attention_ws).  Bar: that of tests/test_prefill_gpu.py.  Most cases FORCE the walk
(prefill_chunk_max_avg_x8 raised) on launches the shipped rule would leave to the prefill bodies, to cover ragged
chunks; test_dispatch_rule pins the rule itself."""
import pytest
import torch

from helpers import dense_prefill_fp64, make_prefill_inputs
from oracle import oracle
from test_ops_gpu import check_attention, to_dev

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def run_hip(ops, inp, max_seq_len=0, chunk=64, force=True):
    d = to_dev(inp)
    out = torch.full_like(d["query"], float("nan"))
    torch.ops._C_amd.set_tuning("prefill_chunk_max_query", chunk)
    if force:
        torch.ops._C_amd.set_tuning("prefill_chunk_max_avg_x8", 1 << 20)
    try:
        ops.paged_prefill_attention(out, d["query"], d["key_cache"], d["value_cache"], inp["num_kv_heads"], inp["scale"],
                                    d["block_tables"], d["seq_lens"], d["query_start_loc"], inp["max_query_len"],
                                    inp["block_size"], None, 0, 0.0, "auto", True, 1.0, 1.0, max_seq_len)
        torch.cuda.synchronize()
    finally:
        torch.ops._C_amd.set_tuning("prefill_chunk_max_query", 64)
        torch.ops._C_amd.set_tuning("prefill_chunk_max_avg_x8", 16)
    return out


def run_oracle(inp):
    out = torch.zeros_like(inp["query"])
    oracle.paged_prefill_attention(out, inp["query"], inp["key_cache"], inp["value_cache"], inp["num_kv_heads"],
                                   inp["scale"], inp["block_tables"], inp["seq_lens"], inp["query_start_loc"],
                                   inp["block_size"])
    return out


MIXED = dict(seq=[37, 200, 5, 1, 129, 64, 48, 333], ql=[1, 32, 5, 1, 17, 0, 16, 31])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("block_size", [16, 32])
@pytest.mark.parametrize("heads", [(8, 2), (4, 4), (16, 1), (6, 2), (8, 8)])
@pytest.mark.parametrize("head_size", [64, 128])
def test_short_chunks_match_the_oracle(ops, dtype, block_size, heads, head_size):
    """Ragged mixed launch: one-token sequences, chunks of 5..32 tokens, an empty chunk, contexts from 1 to 333 tokens;
    GQA groups of 1, 2, 3 (padded to 4), 4 and 16 heads."""
    H, KVH = heads
    inp = make_prefill_inputs(H, KVH, head_size, block_size, MIXED["seq"], MIXED["ql"], dtype=dtype, seed=5)
    out = run_hip(ops, inp)
    assert torch.isfinite(out.float()).all()
    check_attention(out, run_oracle(inp), dense_prefill_fp64(inp))


def test_short_chunks_take_the_new_walk_and_the_old_bodies_agree(ops):
    """The same launch through prefill_mfma.h (prefill_chunk_max_query = 0) meets the same bar, and the two differ (a
    different order of additions): the dispatch is observable."""
    inp = make_prefill_inputs(32, 8, 128, 16, [1040] * 4 + [500, 77], [16] * 4 + [1, 32], dtype=torch.bfloat16, seed=6)
    new = run_hip(ops, inp)
    default_32 = int(torch.ops._C_amd.get_tuning("prefill_mfma32_min_query"))
    torch.ops._C_amd.set_tuning("prefill_mfma32_min_query", 0)
    try:
        old = run_hip(ops, inp, chunk=0)
    finally:
        torch.ops._C_amd.set_tuning("prefill_mfma32_min_query", default_32)
    want, ref64 = run_oracle(inp), dense_prefill_fp64(inp)
    check_attention(new, want, ref64)
    check_attention(old, want, ref64)
    assert not torch.equal(new.view(torch.int16), old.view(torch.int16))


@pytest.mark.parametrize("shape", [([1040] * 2, [16] * 2), ([2080], [32]), ([700, 2080, 130], [1, 32, 9]),
                                   ([4100], [20])])
def test_partitioned_walk_matches_single_pass_and_the_oracle(ops, shape):
    """Few workgroups and long contexts: with max_seq_len stated the key walk is cut across workgroups (blockIdx.z) and
    merged by prefill_chunk_reduce_kernel; rows whose own horizon ends before a partition never read it.  Same bar;
    against the single pass the partial results are rounded once more to the model dtype."""
    seq, ql = shape
    inp = make_prefill_inputs(8, 2, 128, 16, seq, ql, dtype=torch.bfloat16, seed=7)
    torch.ops._C_amd.set_tuning("prefill_chunk_max_avg_x8", 1 << 20)
    need = torch.ops._C_amd.paged_prefill_workspace_bytes(len(seq), sum(ql), max(ql), 8, 2, 128, max(seq))
    torch.ops._C_amd.set_tuning("prefill_chunk_max_avg_x8", 16)
    assert need > 0  # these shapes are cut
    cut = run_hip(ops, inp, max_seq_len=max(seq))
    whole = run_hip(ops, inp)
    want, ref64 = run_oracle(inp), dense_prefill_fp64(inp)
    check_attention(cut, want, ref64)
    check_attention(whole, want, ref64)
    assert float((cut.float() - whole.float()).abs().max()) <= 2e-2 * float(whole.float().abs().max())


def test_short_chunk_garbage_beyond_the_sequences_is_ignored(ops):
    """Cache slots past seq_len (rest of the last block, padding blocks of the table) may hold NaN."""
    inp = make_prefill_inputs(8, 2, 128, 16, [35, 70, 17, 300], [3, 21, 1, 32], dtype=torch.bfloat16, seed=9,
                              garbage=float("nan"))
    for msl in (0, 300):
        out = run_hip(ops, inp, max_seq_len=msl)
        assert torch.isfinite(out.float()).all()
        check_attention(out, run_oracle(inp), dense_prefill_fp64(inp))


def test_short_chunk_late_spike_rescales_only_its_columns(ops):
    """A key near the end of the context with a much larger logit: the accumulators of the columns that see it are
    rescaled, the others (earlier tokens of the chunk whose horizon ends before it) are not touched."""
    inp = make_prefill_inputs(4, 1, 128, 16, [400], [32], dtype=torch.bfloat16, seed=21)
    b = int(inp["block_tables"][0, 390 // 16])
    k = torch.ones(128, dtype=torch.bfloat16) * 1.5
    inp["key_cache"][b, 0, :, 390 % 16, :] = k.view(16, 8)
    inp["k_dense"][0][390, 0] = k
    inp["query"][:, 0] = 1.0
    out = run_hip(ops, inp)
    check_attention(out, run_oracle(inp), dense_prefill_fp64(inp))


def test_one_token_chunks_match_decode(ops):
    """query_len 1 everywhere: what paged_attention_v1 computes (flash_attn.py:554 runs decode tokens this way)."""
    inp = make_prefill_inputs(32, 8, 128, 16, [100, 257, 16, 1], [1, 1, 1, 1], dtype=torch.bfloat16, seed=8)
    out = run_hip(ops, inp)
    d = to_dev(inp)
    dec = torch.zeros_like(d["query"])
    ops.paged_attention_v1(dec, d["query"], d["key_cache"], d["value_cache"], 8, inp["scale"], d["block_tables"],
                           d["seq_lens"], 16, 257, None, "auto", 1.0, 1.0)
    check_attention(out, dec.cpu(), None, tol=5e-3)


def test_short_chunks_compose_to_the_whole_prompt_within_the_bar(ops):
    """A 300-token prompt in chunks of 32: row for row within the bar of the single-chunk run (a different body walks
    the long chunk: equality of bits is not promised across bodies, tests/test_prefill_gpu.py has it within one)."""
    inp = make_prefill_inputs(8, 2, 128, 16, [300], [300], dtype=torch.bfloat16, seed=4)
    whole = run_hip(ops, inp)
    pieces, done = [], 0
    while done < 300:
        n = min(32, 300 - done)
        part = dict(inp)
        part["query"] = inp["query"][done:done + n].contiguous()
        part["seq_lens"] = torch.tensor([done + n], dtype=torch.int32)
        part["query_start_loc"] = torch.tensor([0, n], dtype=torch.int32)
        part["max_query_len"] = n
        pieces.append(run_hip(ops, part))
        done += n
    got = torch.cat(pieces)
    check_attention(got, whole.cpu(), dense_prefill_fp64(inp))


def test_dispatch_rule(ops):
    """As shipped the walk takes launches with at most 2 query tokens per sequence on average (the caller states the
    token count) and no chunk over 64 tokens: a mixed step does, a launch of chunks only does not, and either way the
    result is bit-identical to the body the rule names."""
    assert int(torch.ops._C_amd.get_tuning("prefill_chunk_max_query")) == 64
    assert int(torch.ops._C_amd.get_tuning("prefill_chunk_max_avg_x8")) == 16
    mixed = make_prefill_inputs(8, 2, 128, 16, [200] * 30 + [90, 300], [1] * 30 + [2, 30], dtype=torch.bfloat16, seed=11)
    chunks = make_prefill_inputs(8, 2, 128, 16, [200] * 8, [16] * 8, dtype=torch.bfloat16, seed=12)
    for inp, takes in ((mixed, True), (chunks, False)):
        shipped = run_hip(ops, inp, force=False)
        check_attention(shipped, run_oracle(inp), dense_prefill_fp64(inp))
        pinned = run_hip(ops, inp) if takes else run_hip(ops, inp, chunk=0)
        assert torch.equal(shipped.view(torch.int16), pinned.view(torch.int16)), takes


def test_workspace_rule(ops):
    """lvllm_paged_prefill_workspace_bytes: 0 for launches that are not cut (many workgroups, short contexts, unknown
    bound, launches the walk does not take), > 0 otherwise; a workspace too small means a single pass, not an error."""
    f = torch.ops._C_amd.paged_prefill_workspace_bytes
    assert f(64, 96, 32, 32, 8, 128, 2048) == 0     # a full mixed step: hundreds of workgroups
    assert f(2, 3, 2, 32, 8, 128, 200) == 0         # nothing to cut
    assert f(2, 3, 2, 32, 8, 128, 0) == 0           # no bound
    assert f(2, 64, 32, 32, 8, 128, 4096) > 0       # chunks only: the 32x32 body, which cuts its walk too
    assert f(64, 2048, 32, 32, 8, 128, 4096) == 0   # ... unless the launch fills the CUs by itself
    assert f(2, 3, 2, 32, 8, 128, 4096) > 0


def test_partitioned_walk_with_an_understated_bound_still_walks_every_key(ops):
    """max_seq_len is the caller's promise; a context longer than it is still walked to its end (the last partition is
    open-ended)."""
    inp = make_prefill_inputs(8, 2, 128, 16, [2080, 1500, 900], [2, 1, 1], dtype=torch.bfloat16, seed=13)
    out = run_hip(ops, inp, max_seq_len=1024)  # half the real length
    check_attention(out, run_oracle(inp), dense_prefill_fp64(inp))
