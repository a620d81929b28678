"""fp8 (OCP e4m3fn) KV cache: oracle conversions against torch.float8_e4m3fn on the CPU; on the GPU
reshape_and_cache bit-exact against the oracle and paged_attention_v1/v2 over an fp8 cache against
the oracle and fp64 (csrc/cache_kernels.cu:194-202, csrc/attention/attention_kernels.cu with
KV_DTYPE = kFp8E4M3, csrc/quantization/fp8/nvidia/quant_utils.cuh:295-300,458-489)."""
import math

import pytest
import torch

from helpers import dense_attention_fp64, make_paged_inputs, quantize_paged_inputs_fp8, v2_scratch
from oracle import oracle

DEV = "cuda:0"


# ------------------------------------------------------------------ CPU: the oracle itself
def test_e4m3_decode_all_codes_and_encode_vs_torch():
    codes = torch.arange(256, dtype=torch.uint8)
    ref = codes.view(torch.float8_e4m3fn).float()
    for v in range(256):
        a, b = oracle.e4m3_to_f32(v), float(ref[v])
        assert (math.isnan(a) and math.isnan(b)) or a == b, v
    g = torch.Generator().manual_seed(0)
    xs = torch.cat([torch.randn(6000, generator=g) * 3, torch.randn(2000, generator=g) * 300,
                    torch.randn(2000, generator=g) * 0.01,
                    torch.tensor([0.0, -0.0, 448.0, 449.0, 1e9, -1e9, 2.0 ** -9, 2.0 ** -10, 2.0 ** -10 * 1.0001,
                                  2.0 ** -10 * 3, 464.0, 447.9, 15.5, 17.0, 2.0 ** -11])])
    want = xs.clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)  # SATFINITE + round-to-nearest-even
    got = [oracle.f32_to_e4m3(x) for x in xs.tolist()]
    assert got == want.tolist()
    assert oracle.f32_to_e4m3(float("nan")) & 0x7f == 0x7f


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("k_scale,v_scale", [(1.0, 1.0), (0.5, 0.25), (0.37, 1.9)])
def test_oracle_fp8_cache_write_and_attention(dtype, k_scale, v_scale):
    """reshape_and_cache(fp8) stores e4m3(x / scale); attention over it equals the fp64 attention
    over the dequantised values T(fp8 * scale)."""
    S, H, KVH, D, BS = 3, 4, 2, 64, 16
    lens = [40, 17, 1]
    inp = make_paged_inputs(S, H, KVH, D, BS, lens, dtype=dtype, seed=2)
    # write the dense K/V through the oracle's fp8 reshape_and_cache and compare with the torch-built twin
    twin = quantize_paged_inputs_fp8(inp, k_scale, v_scale)
    kc = torch.zeros_like(twin["key_cache"])
    vc = torch.zeros_like(twin["value_cache"])
    for s, n in enumerate(lens):
        tok = torch.arange(n)
        slots = (inp["block_tables"][s, tok // BS].long() * BS + tok % BS)
        oracle.reshape_and_cache_fp8(inp["k_dense"][s], inp["v_dense"][s], kc, vc, slots, k_scale, v_scale)
    used = torch.zeros(kc.shape[0], dtype=torch.bool)
    for s, n in enumerate(lens):
        used[inp["block_tables"][s, : (n + BS - 1) // BS].long()] = True
    # slots past a sequence's end are zero in both; compare whole used blocks
    assert torch.equal(kc[used], twin["key_cache"][used])
    assert torch.equal(vc[used], twin["value_cache"][used])
    out = torch.zeros_like(inp["query"])
    oracle.set_kv_cache_fp8(True, k_scale, v_scale)
    try:
        oracle.paged_attention_v1(out, inp["query"], kc, vc, KVH, inp["scale"], inp["block_tables"],
                                  inp["seq_lens"], BS, max(lens))
    finally:
        oracle.set_kv_cache_fp8(False)
    want = dense_attention_fp64(twin)
    tol = (2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -9) * max(1.0, float(want.abs().max()))
    assert float((out.double() - want).abs().max()) <= tol


# ------------------------------------------------------------------ GPU
def _run_attn(ops, twin, version, dev=DEV):
    q = twin["query"].to(dev)
    out = torch.full_like(q, float("nan"))
    args = (q, twin["key_cache"].to(dev), twin["value_cache"].to(dev), twin["num_kv_heads"], twin["scale"],
            twin["block_tables"].to(dev), twin["seq_lens"].to(dev), twin["block_size"], twin["max_seq_len"], None,
            "fp8", twin["k_scale"], twin["v_scale"])
    if version == "v1":
        ops.paged_attention_v1(out, *args)
    else:
        es, ml, tmp = v2_scratch(q.shape[0], q.shape[1], q.shape[2], twin["max_seq_len"], q.dtype, dev)
        ops.paged_attention_v2(out, es, ml, tmp, *args)
    torch.cuda.synchronize()
    return out.cpu()


def _oracle_attn(twin):
    out = torch.zeros_like(twin["query"])
    oracle.set_kv_cache_fp8(True, twin["k_scale"], twin["v_scale"])
    try:
        oracle.paged_attention_v1(out, twin["query"], twin["key_cache"], twin["value_cache"], twin["num_kv_heads"],
                                  twin["scale"], twin["block_tables"], twin["seq_lens"], twin["block_size"],
                                  twin["max_seq_len"])
    finally:
        oracle.set_kv_cache_fp8(False)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("block_size", [16, 32])
@pytest.mark.parametrize("k_scale,v_scale", [(1.0, 1.0), (0.37, 1.9)])
def test_reshape_and_cache_fp8_bit_exact(ops, dtype, block_size, k_scale, v_scale):
    g = torch.Generator().manual_seed(5)
    T, KVH, D, NB = 70, 3, 128, 9
    key = (torch.randn(T, KVH, D, generator=g) * 3).to(dtype)
    value = (torch.randn(T, KVH, D, generator=g) * 3).to(dtype)
    key[0, 0, :6] = torch.tensor([1e4, -1e4, 448.0, 460.0, 0.0, -0.0]).to(dtype)  # saturation, signed zero
    value[1, 1, :4] = torch.tensor([float("nan"), 2.0 ** -9, 2.0 ** -10, 3 * 2.0 ** -10]).to(dtype)
    slots = torch.randperm(NB * block_size, generator=g)[:T].to(torch.int64)
    slots[7] = -1
    kc = torch.randint(0, 255, (NB, KVH, D // 16, block_size, 16), generator=g, dtype=torch.uint8)
    vc = torch.randint(0, 255, (NB, KVH, D, block_size), generator=g, dtype=torch.uint8)
    kc_o, vc_o = kc.clone(), vc.clone()
    oracle.reshape_and_cache_fp8(key, value, kc_o, vc_o, slots, k_scale, v_scale)
    kc_d, vc_d = kc.to(DEV), vc.to(DEV)
    ops.reshape_and_cache(key.to(DEV), value.to(DEV), kc_d, vc_d, slots.to(DEV), "fp8", k_scale, v_scale)
    torch.cuda.synchronize()
    assert torch.equal(kc_d.cpu(), kc_o)
    assert torch.equal(vc_d.cpu(), vc_o)


@pytest.mark.gpu
@pytest.mark.parametrize("version", ["v1", "v2"])
@pytest.mark.parametrize("head_size", [64, 80, 96, 112, 128, 192, 256])
def test_fp8_attention_head_sizes(ops, version, head_size):
    from test_ops_gpu import check_attention
    lens = [530, 100, 17, 1, 0, 64]
    inp = make_paged_inputs(len(lens), 8, 2, head_size, 16, lens, dtype=torch.bfloat16, seed=head_size)
    twin = quantize_paged_inputs_fp8(inp, 0.5, 2.0)
    out = _run_attn(ops, twin, version)
    assert torch.isfinite(out).all()
    check_attention(out, _oracle_attn(twin), dense_attention_fp64(twin))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("block_size", [16, 32])
@pytest.mark.parametrize("H,KVH", [(8, 8), (8, 2), (14, 2), (16, 1), (20, 1)])
def test_fp8_attention_gqa_blocks_dtypes(ops, dtype, block_size, H, KVH):
    from test_ops_gpu import check_attention
    lens = [700, 33, 16, 15, 129]
    inp = make_paged_inputs(len(lens), H, KVH, 128, block_size, lens, dtype=dtype, seed=H + KVH + block_size)
    twin = quantize_paged_inputs_fp8(inp, 1.0, 1.0)
    o1 = _run_attn(ops, twin, "v1")
    o2 = _run_attn(ops, twin, "v2")
    want = _oracle_attn(twin)
    check_attention(o1, want, dense_attention_fp64(twin))
    check_attention(o2, want)


@pytest.mark.gpu
def test_fp8_attention_full_size_vs_bf16_cache(ops):
    """BASELINE shape (bs 32, 1024 tokens, H32/KVH8/D128): the fp8-cache result equals the bf16-cache
    kernel run on the dequantised values (same kernel arithmetic after conversion) to 2 ulp at row scale."""
    from test_ops_gpu import rows_close_in_ulp
    lens = [1024] * 32
    inp = make_paged_inputs(32, 32, 8, 128, 16, lens, dtype=torch.bfloat16, seed=1)
    twin = quantize_paged_inputs_fp8(inp, 1.0, 1.0)
    o8 = _run_attn(ops, twin, "v1")
    # bf16 cache holding exactly the dequantised values
    deq = dict(inp)
    NB, KVH, _, BS, _ = inp["key_cache"].shape
    kd = twin["key_cache"].permute(0, 1, 3, 2, 4).reshape(NB, KVH, BS, 128).view(torch.float8_e4m3fn).to(torch.bfloat16)
    deq["key_cache"] = kd.view(NB, KVH, BS, 16, 8).permute(0, 1, 3, 2, 4).contiguous()
    deq["value_cache"] = twin["value_cache"].view(torch.float8_e4m3fn).to(torch.bfloat16)
    q = deq["query"].to(DEV)
    o16 = torch.zeros_like(q)
    ops.paged_attention_v1(o16, q, deq["key_cache"].to(DEV), deq["value_cache"].to(DEV), 8, inp["scale"],
                           inp["block_tables"].to(DEV), inp["seq_lens"].to(DEV), 16, 1024, None, "auto", 1.0, 1.0)
    torch.cuda.synchronize()
    assert rows_close_in_ulp(o8, o16.cpu(), 2)


@pytest.mark.gpu
def test_fp8_argument_errors(ops):
    inp = make_paged_inputs(1, 4, 2, 128, 16, [20], dtype=torch.bfloat16)
    q = inp["query"].to(DEV)
    out = torch.zeros_like(q)
    kc, vc = inp["key_cache"].to(DEV), inp["value_cache"].to(DEV)
    with pytest.raises(RuntimeError, match="one-byte"):
        ops.paged_attention_v1(out, q, kc, vc, 2, 1.0, inp["block_tables"].to(DEV), inp["seq_lens"].to(DEV), 16, 20,
                               None, "fp8", 1.0, 1.0)
    with pytest.raises(RuntimeError, match="Unsupported data type of kv cache"):
        ops.paged_attention_v1(out, q, kc, vc, 2, 1.0, inp["block_tables"].to(DEV), inp["seq_lens"].to(DEV), 16, 20,
                               None, "fp8_e5m2", 1.0, 1.0)
    with pytest.raises(RuntimeError, match="'auto' needs a cache of dtype"):
        ops.paged_attention_v1(out, q, kc.view(torch.uint8), vc.view(torch.uint8), 2, 1.0,
                               inp["block_tables"].to(DEV), inp["seq_lens"].to(DEV), 16, 20, None, "auto", 1.0, 1.0)


# ------------------------------------------------------------------ fp8 activation quantisation
def _fp8_inputs(dtype, tokens, hidden, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(tokens, hidden, generator=g) * 4).to(dtype)
    x[0, :5] = torch.tensor([1e4, -1e4, 0.0, -0.0, float("nan")]).to(dtype)  # clamp, zeros, NaN -> +448
    return x


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_fp8_quant_oracle_vs_torch(dtype):
    """static: e4m3(clamp(x * (1/scale))); per token: scale = max(absmax/448, 1/(448*512))."""
    x = _fp8_inputs(dtype, 5, 96)
    x[0, 4] = 1.0  # (torch's clamp keeps NaN; checked separately below)
    scale = torch.tensor([0.037], dtype=torch.float32)
    out = torch.zeros(x.shape, dtype=torch.uint8)
    oracle.static_scaled_fp8_quant(out, x, scale)
    want = (x.float() * (1.0 / scale)).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    assert torch.equal(out, want)
    scales = torch.zeros(5, 1, dtype=torch.float32)
    oracle.dynamic_per_token_scaled_fp8_quant(out, scales, x)
    want_s = (x.float().abs().amax(dim=1, keepdim=True) / 448.0).clamp_min(1.0 / (448.0 * 512.0))
    assert torch.equal(scales, want_s)
    assert torch.equal(out, (x.float() / want_s).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8))
    nan_in = torch.tensor([[float("nan"), 1.0]], dtype=dtype)
    o = torch.zeros(1, 2, dtype=torch.uint8)
    oracle.static_scaled_fp8_quant(o, nan_in, torch.tensor([1.0]))
    assert o[0, 0] == 0x7e  # fmax(-448, fmin(NaN, 448)) = 448


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("tokens,hidden", [(1, 16), (7, 4096), (33, 14336), (5, 100), (3, 7)])
def test_fp8_quant_ops_bit_exact(ops, dtype, tokens, hidden):
    x = _fp8_inputs(dtype, tokens, max(hidden, 5))[:, :hidden].contiguous()
    xd = x.to(DEV)
    # static
    scale = torch.tensor([0.05], dtype=torch.float32)
    want = torch.zeros(x.shape, dtype=torch.uint8)
    oracle.static_scaled_fp8_quant(want, x, scale)
    got, _ = ops.scaled_fp8_quant(xd, scale.to(DEV))
    assert got.dtype == torch.float8_e4m3fn and torch.equal(got.view(torch.uint8).cpu(), want)
    # dynamic per tensor
    s_o = torch.zeros(1, dtype=torch.float32)
    oracle.dynamic_scaled_fp8_quant(want, x, s_o)
    got, s = ops.scaled_fp8_quant(xd)
    if torch.isnan(x.float()).any():  # max(|x|) ignores NaN on both sides (fmaxf)
        pass
    assert torch.equal(s.cpu(), s_o) and torch.equal(got.view(torch.uint8).cpu(), want)
    # dynamic per token, with and without an upper bound
    for ub in (None, torch.tensor([3.0], dtype=torch.float32)):
        s_o = torch.zeros(tokens, 1, dtype=torch.float32)
        oracle.dynamic_per_token_scaled_fp8_quant(want, s_o, x, ub)
        got, s = ops.scaled_fp8_quant(xd, scale_ub=ub.to(DEV) if ub is not None else None,
                                      use_per_token_if_dynamic=True)
        assert torch.equal(s.cpu(), s_o)
        assert torch.equal(got.view(torch.uint8).cpu(), want)
    # padding of the token dimension (static path keeps rows beyond the input untouched)
    got, s = ops.scaled_fp8_quant(xd, num_token_padding=tokens + 3, use_per_token_if_dynamic=True)
    assert got.shape == (tokens + 3, hidden) and s.shape == (tokens + 3, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("per_token", [False, True])
def test_apply_fp8_linear_vs_dequantised_matmul(ops, per_token):
    """W8A8 linear (w8a8_utils.py:103-189): quantise with the HIP kernels, multiply with
    torch._scaled_mm, compare with the fp32 product of the dequantised operands."""
    from light_vllm_amd.quantization import apply_fp8_linear, per_tensor_quantize_weight
    g = torch.Generator().manual_seed(0)
    M, K, N = 37, 512, 256
    x = (torch.randn(M, K, generator=g)).to(torch.bfloat16).to(DEV)
    w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).to(DEV)
    bias = torch.randn(N, generator=g).to(torch.bfloat16).to(DEV)
    wq, ws = per_tensor_quantize_weight(w)
    try:
        y = apply_fp8_linear(x, wq, ws, bias=bias, use_per_token_if_dynamic=per_token)
    except RuntimeError as e:  # a torch build without fp8 _scaled_mm for this GPU
        pytest.skip(f"torch._scaled_mm unavailable: {e}")
    xq, xs = ops.scaled_fp8_quant(x, use_per_token_if_dynamic=per_token)
    want = (xq.float() * xs) @ (wq.float() * ws) + bias.float()
    assert y.shape == (M, N) and y.dtype == torch.bfloat16
    assert float((y.float() - want).abs().max()) <= 2e-2 * float(want.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_convert_fp8_both_directions(ops, dtype):
    """T -> fp8(x / scale) bit-exact with the oracle's conversion (= torch.float8_e4m3fn after the
    saturating clamp); fp8 -> T(float(fp8) * scale) for all 256 codes."""
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(9, 2, 64, generator=g) * 50).to(dtype)
    scale = 0.75
    want = (x.float() / scale).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    got = torch.zeros(x.shape, dtype=torch.uint8, device=DEV)
    ops.convert_fp8(got, x.to(DEV), scale, "fp8")
    assert torch.equal(got.cpu(), want)
    codes = torch.arange(256, dtype=torch.uint8).view(4, 64)
    back = torch.zeros(4, 64, dtype=dtype, device=DEV)
    ops.convert_fp8(back, codes.to(DEV), scale, "fp8_e4m3")
    ref = (codes.view(torch.float8_e4m3fn).float() * scale).to(dtype)
    b = back.cpu()
    ok = (b == ref) | (torch.isnan(b.float()) & torch.isnan(ref.float()))
    assert bool(ok.all())
    with pytest.raises(RuntimeError, match="Unsupported data type"):
        ops.convert_fp8(back, codes.to(DEV), scale, "auto")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(32, 128256, 4096), (1, 256, 128), (17, 32000, 1024), (64, 4096, 2048), (33, 512, 8192)])
def test_w8a8_argmax_epilogue_gives_torch_argmax_of_the_projection(ops, dtype, M, N, K):
    from light_vllm_amd.quantization import pack_fp8_weight, skinny_fp8_linear
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g, device=DEV).to(dtype)
    w = torch.randn(N, K, generator=g, device=DEV) * 0.05
    w[N // 2] = w[3]  # an exact tie between two tiles
    w[N - 1] = w[3]
    w_scale = (w.abs().max() / 448.0).reshape(1).float()
    wp = pack_fp8_weight((w / w_scale).clamp(-448, 448).to(torch.float8_e4m3fn))
    x_scale = (x.float().abs().max() / 448.0).reshape(1)
    logits = skinny_fp8_linear(x, wp, w_scale, x_scale, N, K, None)
    tokens = torch.ops._C_amd.skinny_linear_w8a8_argmax(x, wp, w_scale, x_scale, N, K)
    first = (logits == logits.max(dim=-1, keepdim=True).values).float().argmax(-1)
    assert tokens.dtype == torch.int64 and torch.equal(tokens, first)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,inter,K", [(32, 14336, 4096), (1, 256, 128), (17, 1040, 1024), (64, 2048, 2048),
                                       (33, 512, 8192)])
def test_w8a8_swiglu_epilogue_is_bit_identical_to_the_two_launches(ops, dtype, M, inter, K):
    from light_vllm_amd.quantization import pack_fp8_weight, skinny_fp8_linear
    g = torch.Generator(device=DEV).manual_seed(M + inter + K)
    x = torch.randn(M, K, generator=g, device=DEV).to(dtype)
    w = torch.randn(2 * inter, K, generator=g, device=DEV) * 0.05
    bias = torch.randn(2 * inter, generator=g, device=DEV).to(dtype)
    w_scale = (w.abs().max() / 448.0).reshape(1).float()
    wp = pack_fp8_weight((w / w_scale).clamp(-448, 448).to(torch.float8_e4m3fn))
    x_scale = (x.float().abs().max() / 448.0).reshape(1)
    for b in (None, bias):
        gate_up = skinny_fp8_linear(x, wp, w_scale, x_scale, 2 * inter, K, b)
        ref = torch.empty(M, inter, dtype=dtype, device=DEV)
        ops.silu_and_mul(ref, gate_up)
        out = torch.ops._C_amd.skinny_linear_w8a8_swiglu(x, wp, w_scale, x_scale, 2 * inter, K, b)
        assert out.shape == (M, inter)
        assert torch.equal(out.view(torch.int16), ref.view(torch.int16))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M", [1, 7, 16, 32, 64])
@pytest.mark.parametrize("N,K", [(6144, 4096), (4096, 4096), (256, 14336), (512, 64), (48, 8192)])
def test_skinny_w8a8_vs_dequantised_matmul(ops, dtype, M, N, K):
    """The W8A8 weight-streaming kernel against the fp64 product of the very fp8 values it multiplies
    (activation quantised by the oracle's static_scaled_fp8_quant arithmetic), scales and bias applied
    as torch._scaled_mm does.  fp8 x fp8 products are exact in fp32; only the summation order and the
    final rounding to T differ."""
    from light_vllm_amd.quantization import pack_fp8_weight, skinny_fp8_linear
    g = torch.Generator().manual_seed(N + K + M)
    x = torch.randn(M, K, generator=g).to(dtype)
    w = (torch.randn(N, K, generator=g) * 0.05)
    bias = torch.randn(N, generator=g).to(dtype)
    w_scale = (w.abs().max() / 448.0).reshape(1).float()
    wq = (w / w_scale).clamp(-448, 448).to(torch.float8_e4m3fn)
    x_scale = torch.tensor([float(x.float().abs().max()) / 448.0 * 0.8])  # some values saturate on purpose
    xq = torch.zeros(M, K, dtype=torch.uint8)
    oracle.static_scaled_fp8_quant(xq, x, x_scale)
    want = (xq.view(torch.float8_e4m3fn).double() @ wq.double().T) * float(x_scale) * float(w_scale) + bias.double()
    wp = pack_fp8_weight(wq.to(DEV))
    y = skinny_fp8_linear(x.to(DEV), wp, w_scale.to(DEV), x_scale.to(DEV), N, K, bias.to(DEV))
    torch.cuda.synchronize()
    assert y.shape == (M, N) and y.dtype == dtype
    err = (y.cpu().double() - want).abs().max().item()
    tol = (2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10) * max(1.0, want.abs().max().item())
    assert err <= tol, (err, tol)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("block_size", [16, 32])
@pytest.mark.parametrize("head_size,H,KVH", [(128, 8, 2), (64, 4, 4), (256, 4, 1), (192, 6, 2)])
def test_fp8_prefill_over_the_paged_cache(ops, dtype, block_size, head_size, H, KVH):
    """Chunked prefill / prefix hits over an fp8 cache: the LDS prefill kernel's KV8 instantiation
    against the oracle (dequantised element = T(fp8 * scale)) and fp64 over the dequantised values."""
    from helpers import dense_prefill_fp64, make_prefill_inputs
    from test_ops_gpu import check_attention
    seq, ql = [37, 200, 5, 129, 64], [37, 40, 5, 129, 17]
    inp = make_prefill_inputs(H, KVH, head_size, block_size, seq, ql, dtype=dtype, seed=head_size + block_size)
    twin = quantize_paged_inputs_fp8(inp, 0.5, 2.0)
    want = torch.zeros_like(twin["query"])
    oracle.set_kv_cache_fp8(True, 0.5, 2.0)
    try:
        oracle.paged_prefill_attention(want, twin["query"], twin["key_cache"], twin["value_cache"], KVH, twin["scale"],
                                       twin["block_tables"], twin["seq_lens"], twin["query_start_loc"], block_size)
    finally:
        oracle.set_kv_cache_fp8(False)
    out = torch.full_like(twin["query"], float("nan")).to(DEV)
    ops.paged_prefill_attention(out, twin["query"].to(DEV), twin["key_cache"].to(DEV), twin["value_cache"].to(DEV),
                                KVH, twin["scale"], twin["block_tables"].to(DEV), twin["seq_lens"].to(DEV),
                                twin["query_start_loc"].to(DEV), max(ql), block_size, None, 0, 0.0, "fp8", True,
                                0.5, 2.0)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    check_attention(out, want, dense_prefill_fp64(twin))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(32, 4096, 14336), (64, 4096, 14336), (7, 512, 8192), (33, 1024, 6144)])
def test_w8a8_partials_into_add_norm_are_bit_identical_to_the_separate_launches(ops, dtype, M, N, K):
    """W8A8 projection whose K is split over workgroups: leaving the raw fp32 partials to
    fused_add_rms_norm_splitk_scaled == the projection's own reduce pass (scale, round) followed by
    fused_add_rms_norm -- normed output and updated residual, bit for bit.  A shape whose K is not split returns
    an empty tensor."""
    from light_vllm_amd.quantization import pack_fp8_weight, skinny_fp8_linear
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g, device=DEV).to(dtype)
    w = torch.randn(N, K, generator=g, device=DEV) * 0.05
    w_scale = (w.abs().max() / 448.0).reshape(1).float()
    wp = pack_fp8_weight((w / w_scale).clamp(-448, 448).to(torch.float8_e4m3fn))
    x_scale = (x.float().abs().max() / 448.0).reshape(1)
    res = torch.randn(M, N, generator=g, device=DEV).to(dtype)
    weight = (1 + 0.1 * torch.randn(N, generator=g, device=DEV)).to(dtype)
    part = torch.ops._C_amd.skinny_linear_w8a8_partials(x, wp, w_scale, x_scale, N, K)
    assert part.dim() == 3 and part.shape[1:] == (M, N) and part.shape[0] > 1
    out, res1 = torch.empty_like(res), res.clone()
    torch.ops._C_amd.fused_add_rms_norm_splitk_scaled(out, res1, part, weight, 1e-5, x_scale, w_scale)
    y = skinny_fp8_linear(x, wp, w_scale, x_scale, N, K, None)
    res2 = res.clone()
    ops.fused_add_rms_norm(y, res2, weight, 1e-5)
    assert torch.equal(res1.view(torch.int16), res2.view(torch.int16))
    assert torch.equal(out.view(torch.int16), y.view(torch.int16))
    assert torch.ops._C_amd.skinny_linear_w8a8_partials(x[:, :4096].contiguous(), wp.view(-1)[: N * 4096], w_scale, x_scale,
                                                        N, 4096).numel() == 0


# ------------------------------------------------------------------ W8A8 decode step: activations quantised ONCE
def _w8(N, K, g):
    from light_vllm_amd.quantization import pack_fp8_weight
    w = torch.randn(N, K, generator=g, device=DEV) * 0.05
    w_scale = (w.abs().max() / 448.0).reshape(1).float()
    return pack_fp8_weight((w / w_scale).clamp(-448, 448).to(torch.float8_e4m3fn)), w_scale


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,hidden", [(32, 4096), (1, 256), (7, 1024), (19, 2048)])
def test_norm_launches_with_an_fp8_twin_equal_norm_then_quant(ops, dtype, M, hidden):
    """rms_norm_fp8 / fused_add_rms_norm_fp8 / fused_add_rms_norm_splitk_fp8: the bytes static_scaled_fp8_quant writes
    for the normalised rows of rms_norm / fused_add_rms_norm / fused_add_rms_norm_splitk[_scaled], and the same
    residual, bit for bit (some values saturate on purpose)."""
    g = torch.Generator(device=DEV).manual_seed(M + hidden)
    x = torch.randn(M, hidden, generator=g, device=DEV).to(dtype)
    res = torch.randn(M, hidden, generator=g, device=DEV).to(dtype)
    weight = (1 + 0.1 * torch.randn(hidden, generator=g, device=DEV)).to(dtype)
    qs = torch.tensor([0.004], device=DEV)

    def quant(t):
        out = torch.empty(t.shape, dtype=torch.float8_e4m3fn, device=DEV)
        torch.ops._C.static_scaled_fp8_quant(out, t.contiguous(), qs)
        return out.view(torch.uint8)
    normed = torch.empty_like(x)
    ops.rms_norm(normed, x, weight, 1e-5)
    assert torch.equal(torch.ops._C_amd.rms_norm_fp8(x, weight, 1e-5, qs), quant(normed))
    y, r1 = x.clone(), res.clone()
    ops.fused_add_rms_norm(y, r1, weight, 1e-5)
    r2 = res.clone()
    got = torch.ops._C_amd.fused_add_rms_norm_fp8(x, r2, weight, 1e-5, qs)
    assert torch.equal(got, quant(y)) and torch.equal(r1.view(torch.int16), r2.view(torch.int16))
    assert int((got.view(torch.int8).abs() == 0x7e).sum()) > 0 or hidden < 1024  # 448 = 0x7e: saturation happened
    part = torch.randn(3, M, hidden, generator=g, device=DEV) * 0.7
    xs, ws = torch.tensor([0.03], device=DEV), torch.tensor([0.5], device=DEV)
    for scales in ((None, None), (xs, ws)):
        o1, r1 = torch.empty_like(res), res.clone()
        torch.ops._C_amd.fused_add_rms_norm_splitk_scaled(o1, r1, part, weight, 1e-5, *scales)
        r2 = res.clone()
        got = torch.ops._C_amd.fused_add_rms_norm_splitk_fp8(r2, part, weight, 1e-5, scales[0], scales[1], qs)
        assert torch.equal(got, quant(o1)) and torch.equal(r1.view(torch.int16), r2.view(torch.int16))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M", [1, 7, 16, 17, 32])
@pytest.mark.parametrize("N,K", [(6144, 4096), (4096, 4096), (512, 64), (256, 1024), (4096, 14336), (48, 8192)])
def test_w8a8_projection_of_prequantised_activations_is_bit_identical(ops, dtype, M, N, K):
    """skinny_linear_w8a8_q(static_scaled_fp8_quant(x)) == skinny_linear_w8a8(x): same bytes into the same MFMAs; the
    raw split-K partials too (K beyond one workgroup), with and without a bias."""
    from light_vllm_amd.quantization import skinny_fp8_linear
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g, device=DEV).to(dtype)
    wp, w_scale = _w8(N, K, g)
    bias = torch.randn(N, generator=g, device=DEV).to(dtype)
    x_scale = (x.float().abs().max() / 448.0 * 0.8).reshape(1)
    x8 = torch.empty(M, K, dtype=torch.float8_e4m3fn, device=DEV)
    torch.ops._C.static_scaled_fp8_quant(x8, x, x_scale)
    x8 = x8.view(torch.uint8)
    for b in (None, bias):
        want = skinny_fp8_linear(x, wp, w_scale, x_scale, N, K, b)
        got = torch.ops._C_amd.skinny_linear_w8a8_q(x8, wp, w_scale, x_scale, N, K, b, dtype)
        assert got.dtype == dtype and torch.equal(got.view(torch.int16), want.view(torch.int16))
    p1 = torch.ops._C_amd.skinny_linear_w8a8_partials(x, wp, w_scale, x_scale, N, K)
    p2 = torch.ops._C_amd.skinny_linear_w8a8_q_partials(x8, wp, w_scale, x_scale, N, K)
    assert p1.shape == p2.shape and torch.equal(p1, p2) and (p1.numel() > 0) == (K > 4096)
    # rows of a wider buffer (row stride > K bytes)
    wide = torch.zeros(M, K + 64, dtype=torch.uint8, device=DEV)
    wide[:, :K] = x8
    got = torch.ops._C_amd.skinny_linear_w8a8_q(wide[:, :K], wp, w_scale, x_scale, N, K, None, dtype)
    assert torch.equal(got.view(torch.int16), skinny_fp8_linear(x, wp, w_scale, x_scale, N, K, None).view(torch.int16))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,inter,K", [(32, 14336, 4096), (1, 256, 128), (17, 1040, 1024), (9, 2048, 2048)])
def test_w8a8_swiglu_epilogue_with_an_fp8_result_is_bit_identical_to_the_launches_it_replaces(ops, dtype, M, inter, K):
    """gate_up projection of pre-quantised activations + silu_and_mul + static_scaled_fp8_quant in one launch."""
    g = torch.Generator(device=DEV).manual_seed(M + inter + K)
    x = torch.randn(M, K, generator=g, device=DEV).to(dtype)
    wp, w_scale = _w8(2 * inter, K, g)
    x_scale = (x.float().abs().max() / 448.0).reshape(1)
    x8 = torch.empty(M, K, dtype=torch.float8_e4m3fn, device=DEV)
    torch.ops._C.static_scaled_fp8_quant(x8, x, x_scale)
    act = torch.ops._C_amd.skinny_linear_w8a8_swiglu(x, wp, w_scale, x_scale, 2 * inter, K, None)
    qs = (act.float().abs().max() / 448.0 * 0.7).reshape(1)
    want = torch.empty(M, inter, dtype=torch.float8_e4m3fn, device=DEV)
    torch.ops._C.static_scaled_fp8_quant(want, act, qs)
    got = torch.ops._C_amd.skinny_linear_w8a8_q_swiglu_fp8(x8.view(torch.uint8), wp, w_scale, x_scale, 2 * inter, K, qs, dtype)
    assert got.shape == (M, inter) and torch.equal(got, want.view(torch.uint8))


@pytest.mark.gpu
@pytest.mark.parametrize("kv", ["auto", "fp8"])
def test_attention_launch_with_an_fp8_twin_of_its_result(ops, kv):
    """paged_attention_v2_q / rope_cache_paged_attention(out_fp8=...): `out` as the plain launch writes it and
    out_fp8 = static_scaled_fp8_quant(out, scale), for a W8A8 output projection; a launch that would be cut into
    shares is refused (False, nothing written) instead of silently skipping the twin."""
    S, H, KVH, D, BS = 32, 32, 8, 128, 16
    lens = [1024 - 7 * i for i in range(S)]
    inp = make_paged_inputs(S, H, KVH, D, BS, lens, dtype=torch.bfloat16, seed=5)
    if kv == "fp8":
        inp = quantize_paged_inputs_fp8(inp, 1.0, 1.0)
    d = {k: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for k, v in inp.items()}
    es, ml, tmp = v2_scratch(S, H, D, max(lens), torch.bfloat16, DEV)
    want = torch.zeros_like(d["query"])
    ops.paged_attention_v2(want, es, ml, tmp, d["query"], d["key_cache"], d["value_cache"], KVH, inp["scale"],
                           d["block_tables"], d["seq_lens"], BS, max(lens), None, kv, 1.0, 1.0)
    qs = (want.float().abs().max() / 448.0 * 0.6).reshape(1)
    want8 = torch.empty(want.shape, dtype=torch.float8_e4m3fn, device=DEV)
    torch.ops._C.static_scaled_fp8_quant(want8, want, qs)
    out, out8 = torch.zeros_like(want), torch.zeros(want.shape, dtype=torch.uint8, device=DEV)
    assert torch.ops._C_amd.paged_attention_v2_q(out, out8, qs, es, ml, tmp, d["query"], d["key_cache"], d["value_cache"], KVH,
                                                 inp["scale"], d["block_tables"], d["seq_lens"], BS, max(lens), kv, 1.0, 1.0)
    assert torch.equal(out.view(torch.int16), want.view(torch.int16)) and torch.equal(out8, want8.view(torch.uint8))
    # two sequences of 1024 tokens: the library cuts them into shares -> no twin, nothing launched
    out8.zero_()
    assert not torch.ops._C_amd.paged_attention_v2_q(out[:2], out8[:2], qs, es[:2], ml[:2], tmp[:2], d["query"][:2],
                                                     d["key_cache"], d["value_cache"], KVH, inp["scale"],
                                                     d["block_tables"][:2], d["seq_lens"][:2], BS, max(lens), kv, 1.0, 1.0)
    assert int(out8.sum()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("kv", ["auto", "fp8"])
def test_w8a8_model_forward_with_activations_quantised_once_is_bit_identical(kv):
    """Two layers at Llama-3-8B widths, W8A8: the decode forward with ModelConfig.fp8_activations_once (norm launches,
    the attention launch and the SwiGLU epilogue hand fp8 to the projections) against the forward in which every
    projection quantises its own input -- hidden states equal bit for bit, 1, 8 and 32 rows, 16-bit and fp8 KV cache."""
    import light_vllm_amd  # noqa: F401
    from light_vllm_amd.attention.backend import PagedAttnImpl, PagedAttnMetadata
    from light_vllm_amd.engine.config import ModelConfig
    from light_vllm_amd.engine.model import DecoderModel
    cfg = ModelConfig.llama3_8b()
    cfg.num_hidden_layers, cfg.vocab_size, cfg.quantization = 2, 1024, "fp8"
    attn = PagedAttnImpl(cfg.num_attention_heads, cfg.head_dim, cfg.head_dim ** -0.5, cfg.num_key_value_heads, None, None, kv)
    model = DecoderModel(cfg, attn, DEV, seed=0)
    NB, BS = 64, 16
    for n in (1, 8, 32):
        g = torch.Generator(device=DEV).manual_seed(n)
        caches = [(torch.randn(2, NB, BS * cfg.num_key_value_heads * cfg.head_dim, generator=g, device=DEV) * 0.3).to(cfg.dtype)
                  for _ in range(2)]
        if kv == "fp8":
            caches = [c.to(torch.float8_e4m3fn).view(torch.uint8) for c in caches]
        ids = torch.randint(0, cfg.vocab_size, (n,), generator=g, device=DEV)
        lens = torch.randint(1, 2 * BS, (n,), generator=g, device=DEV).to(torch.int32)
        bt = torch.randperm(NB, generator=g, device=DEV)[: 2 * n].view(n, 2).to(torch.int32)
        pos = (lens - 1).long()
        slots = bt[torch.arange(n, device=DEV), (pos // BS)].long() * BS + pos % BS
        md = PagedAttnMetadata(num_prefills=0, num_prefill_tokens=0, num_decode_tokens=n, slot_mapping=slots, seq_lens=None,
                               seq_lens_tensor=lens, max_query_len=1, max_prefill_seq_len=0, max_decode_seq_len=2 * BS,
                               query_start_loc=None, seq_start_loc=None, context_lens_tensor=None, block_tables=bt)
        outs = []
        for once in (True, False):
            cfg.fp8_activations_once = once
            outs.append(model.forward(ids, pos, [t.clone() for t in caches], md))
        torch.cuda.synchronize()
        assert torch.isfinite(outs[0].float()).all()
        assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16)), n
    cfg.fp8_activations_once = True
