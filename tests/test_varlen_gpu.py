"""The prefill-only (no KV cache) attention path on the GPU: torch.ops._C_amd.varlen_attention ->
lvllm_varlen_attention (placement + reshape_and_cache into scratch tiles + the paged MFMA kernel)
against the golden outputs of the reference's torch-naive backend, the CPU oracle and fp64.

The reference's own bar for this path is cosine similarity within 1e-2 between backends
(tests/prefill_only/attention/test_basic_correctness.py:81-89); here additionally
max-abs <= 2e-2 * max|out| as for every attention kernel of this package."""
import numpy as np
import pytest
import torch

from oracle import oracle
from test_oracle import _load_prefill_only_cases
from test_ops_gpu import check_attention

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def run_backend(c, dtype, attn_type_name):
    from light_vllm_amd.attention.prefill_only import AttentionType, PrefillOnlyHIPVarlenBackend
    H, KVH, D = c["H"], c["KVH"], 64
    backend = PrefillOnlyHIPVarlenBackend(AttentionType.attn_type_name_to_enum(attn_type_name))
    impl = backend.get_impl_cls()(H, D, D ** -0.5, KVH, None, None, "auto")
    md = backend.make_metadata_builder()(seq_lens=c["seq_lens"]).to(DEV)
    q, k, v = (c[n].to(dtype).to(DEV) for n in ("q", "k", "v"))
    out = impl.forward(q, k, v, None, md, attn_type=backend.attn_type)
    torch.cuda.synchronize()
    return out.cpu()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("attn_type", ["DECODER", "ENCODER"])
def test_backend_vs_reference_torch_naive_golden(dtype, attn_type):
    for c in _load_prefill_only_cases():
        out = run_backend(c, dtype, attn_type)
        want = c[attn_type.lower()]
        assert torch.isfinite(out).all()
        cos = torch.nn.functional.cosine_similarity(out.float(), want, dim=1)  # per token, as the reference test
        assert bool(((cos >= 1 - 1e-2) & (cos <= 1 + 1e-2)).all()), (c["tag"], float(cos.min()))
        H = c["H"]
        check_attention(out.view(-1, H, 64), want.view(-1, H, 64))


@pytest.mark.parametrize("causal", [True, False])
@pytest.mark.parametrize("H,KVH,D", [(8, 2, 128), (12, 12, 64), (14, 2, 64), (4, 1, 256), (5, 1, 80)])
def test_varlen_op_vs_oracle_ragged(ops, causal, H, KVH, D):
    """Ragged batch incl. lengths around the 16/32-token tile edges; strided q/k/v views of one
    fused qkv projection; stale NaNs in the scratch workspace must not leak."""
    seq_lens = [1, 15, 16, 17, 31, 32, 33, 64, 100, 257, 5]
    T = sum(seq_lens)
    g = torch.Generator().manual_seed(H * 100 + D)
    qkv = (torch.randn(T, (H + 2 * KVH) * D, generator=g) * 0.5).to(torch.bfloat16)
    q, k, v = qkv.split([H * D, KVH * D, KVH * D], dim=1)
    q, k, v = q.view(T, H, D), k.view(T, KVH, D), v.view(T, KVH, D)
    cu = torch.tensor([0] + list(np.cumsum(seq_lens)), dtype=torch.int32)
    want = torch.zeros(T, H, D, dtype=torch.bfloat16)
    oracle.varlen_attention(want, q, k, v, cu, D ** -0.5, causal)
    dq = qkv.to(DEV)
    q_d, k_d, v_d = dq.split([H * D, KVH * D, KVH * D], dim=1)
    need = torch.ops._C_amd.varlen_attention_workspace_bytes(T, len(seq_lens), max(seq_lens), KVH, D)
    ws = torch.full((need // 2,), float("nan"), dtype=torch.bfloat16, device=DEV).view(torch.uint8)
    out = torch.full((T, H, D), float("nan"), dtype=torch.bfloat16, device=DEV)
    ops.varlen_attention(out, q_d.view(T, H, D), k_d.view(T, KVH, D), v_d.view(T, KVH, D), cu.to(DEV),
                         max(seq_lens), D ** -0.5, causal, workspace=ws)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    check_attention(out, want)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("causal", [True, False])
@pytest.mark.parametrize("H,KVH,D", [(16, 16, 64), (8, 2, 128), (12, 4, 64), (4, 4, 128)])
def test_rows_read_in_place_equal_the_pack_pass_bit_for_bit(ops, dtype, causal, H, KVH, D):
    """Launches the 32x32 body takes read K / V where the caller left them (row-major LDS images, transposed reads of
    V; `varlen_dense`, on by default) -- the same MFMAs on the same operands as behind the pack pass, so the same bits.
    Strided views of one fused qkv buffer; lengths around the 64-key tile, the 8-key group and the 16-row copy share
    of a wave; Inf / NaN in the rows of the NEXT sequence and behind the last one must not leak into a sequence.
    (Head size 128 keeps the pack pass whatever the knob says: the case checks that it is left alone.)"""
    seq_lens = [64, 65, 127, 128, 129, 200, 1, 7, 63, 333, 512, 72]
    T = sum(seq_lens)
    g = torch.Generator().manual_seed(H * 1000 + D + causal)
    tail = 40  # rows behind the last sequence: another caller's data
    qkv = (torch.randn(T + tail, (H + 2 * KVH) * D, generator=g) * 0.5).to(dtype)
    qkv[T:] = float("nan")
    dq = qkv.to(DEV)
    q_d, k_d, v_d = (x[:T] for x in dq.split([H * D, KVH * D, KVH * D], dim=1))
    cu = torch.tensor([0] + list(np.cumsum(seq_lens)), dtype=torch.int32)
    outs = []
    for dense, waves in ((1, 8), (0, 8), (1, 4)):  # (4 waves per workgroup: the other shape of the in-place launch)
        torch.ops._C_amd.set_tuning("varlen_dense", dense)
        torch.ops._C_amd.set_tuning("varlen_dense_waves", waves)
        try:
            out = torch.full((T, H, D), float("nan"), dtype=dtype, device=DEV)
            ops.varlen_attention(out, q_d.view(T, H, D), k_d.view(T, KVH, D), v_d.view(T, KVH, D), cu.to(DEV),
                                 max(seq_lens), D ** -0.5, causal)
            torch.cuda.synchronize()
        finally:
            torch.ops._C_amd.set_tuning("varlen_dense", 1)
            torch.ops._C_amd.set_tuning("varlen_dense_waves", 0)
        outs.append(out.cpu())
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    assert torch.equal(outs[0].view(torch.int16), outs[2].view(torch.int16))
    q, k, v = (x[:T] for x in qkv.split([H * D, KVH * D, KVH * D], dim=1))
    want = torch.zeros(T, H, D, dtype=dtype)
    oracle.varlen_attention(want, q.reshape(T, H, D), k.reshape(T, KVH, D), v.reshape(T, KVH, D), cu, D ** -0.5,
                            causal)
    check_attention(outs[0], want)
    # a sequence whose neighbour holds Inf: the neighbour's rows are in its last tile's image, never in its result
    seq = 4  # 129 tokens: its third tile holds 63 rows of sequence 5
    poisoned = dq.clone()
    poisoned[int(cu[seq + 1]):int(cu[seq + 2]), H * D:] = float("inf")
    pq, pk, pv = (x[:T] for x in poisoned.split([H * D, KVH * D, KVH * D], dim=1))
    out2 = torch.empty((T, H, D), dtype=dtype, device=DEV)
    ops.varlen_attention(out2, pq.view(T, H, D), pk.view(T, KVH, D), pv.view(T, KVH, D), cu.to(DEV), max(seq_lens),
                         D ** -0.5, causal)
    sl = slice(int(cu[seq]), int(cu[seq + 1]))
    assert torch.equal(out2[sl].cpu().view(torch.int16), outs[0][sl].view(torch.int16))


def test_rows_read_in_place_with_different_key_and_value_strides(ops):
    """K and V of different buffers, different token strides, V not at the start of its rows: the in-place launch
    takes each operand's own stride and base (bit-identical to the pack pass, as above)."""
    H, KVH, D = 8, 8, 64
    seq_lens = [300, 64, 97]
    T = sum(seq_lens)
    g = torch.Generator().manual_seed(21)
    q = (torch.randn(T, H, D, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    kbuf = (torch.randn(T, 2 * KVH * D + 64, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    vbuf = (torch.randn(T, KVH * D + 8, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    k = kbuf[:, KVH * D:2 * KVH * D].view(T, KVH, D)   # stride 2 KVH D + 64, offset KVH D
    v = vbuf[:, 8:].view(T, KVH, D)                     # stride KVH D + 8, offset 8 elements (16 bytes)
    cu = torch.tensor([0] + list(np.cumsum(seq_lens)), dtype=torch.int32, device=DEV)
    outs = []
    for dense in (1, 0):
        torch.ops._C_amd.set_tuning("varlen_dense", dense)
        try:
            out = torch.empty(T, H, D, dtype=torch.bfloat16, device=DEV)
            ops.varlen_attention(out, q, k, v, cu, max(seq_lens), D ** -0.5, False)
            torch.cuda.synchronize()
        finally:
            torch.ops._C_amd.set_tuning("varlen_dense", 1)
        outs.append(out.cpu())
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    want = torch.zeros(T, H, D, dtype=torch.bfloat16)
    oracle.varlen_attention(want, q.cpu(), k.cpu().contiguous(), v.cpu().contiguous(), cu.cpu(), D ** -0.5, False)
    check_attention(outs[0], want)


def test_varlen_long_sequences_properties(ops):
    """bge-m3-like encoder batch (8 x 2048 tokens, 16 heads of 64): sampled rows vs fp64, and the
    encoder output of a sequence does not depend on its neighbours in the batch (bit-exact)."""
    H, D, L, B = 16, 64, 2048, 8
    g = torch.Generator().manual_seed(3)
    q, k, v = ((torch.randn(B * L, H, D, generator=g) * 0.5).to(torch.bfloat16) for _ in range(3))
    cu = (torch.arange(B + 1) * L).to(torch.int32)
    out = torch.empty(B * L, H, D, dtype=torch.bfloat16, device=DEV)
    ops.varlen_attention(out, q.to(DEV), k.to(DEV), v.to(DEV), cu.to(DEV), L, D ** -0.5, False)
    o = out.cpu()
    for s, t, h in ((0, 0, 0), (3, 1000, 7), (7, 2047, 15)):
        kk, vv = k[s * L:(s + 1) * L, h].double(), v[s * L:(s + 1) * L, h].double()
        p = torch.softmax((kk @ q[s * L + t, h].double()) * D ** -0.5, 0)
        want = p @ vv
        assert float((o[s * L + t, h].double() - want).abs().max()) <= 2e-2 * max(float(want.abs().max()), 1e-3)
    one = torch.empty(L, H, D, dtype=torch.bfloat16, device=DEV)
    sl = slice(3 * L, 4 * L)
    ops.varlen_attention(one, q[sl].to(DEV), k[sl].to(DEV), v[sl].to(DEV), cu[:2].to(DEV), L, D ** -0.5, False)
    assert torch.equal(one.cpu().view(torch.int16), o[sl].view(torch.int16))


def test_varlen_argument_errors(ops):
    q = torch.zeros(4, 2, 64, dtype=torch.bfloat16, device=DEV)
    cu = torch.tensor([0, 4], dtype=torch.int32, device=DEV)
    ws = torch.empty(16, dtype=torch.uint8, device=DEV)
    with pytest.raises(RuntimeError, match="workspace"):
        ops.varlen_attention(q.clone(), q, q, q, cu, 4, 1.0, True, workspace=ws)
    with pytest.raises(RuntimeError, match="causal"):
        ops.varlen_attention(q.clone(), q, q, q, cu, 4, 1.0, False, sliding_window=2)
    with pytest.raises(RuntimeError, match="int32"):
        ops.varlen_attention(q.clone(), q, q, q, cu.long(), 4, 1.0, True)
