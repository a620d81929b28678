"""Synthetic inputs of the paged-attention decode path shared by the tests,
the golden-vector generator and bench.py.  Pure torch-CPU construction; callers move the
tensors where they need them."""
import math

import torch


def make_paged_inputs(num_seqs, num_heads, num_kv_heads, head_size, block_size, seq_lens,
                      dtype=torch.bfloat16, seed=0, extra_blocks=7, q_in_qkv=False):
    """Returns a dict with query, key_cache, value_cache (paged layouts of the reference,
    csrc/cache_kernels.cu:152-204), block_tables, seq_lens, scale and the dense k/v they hold.
    Block tables are a random permutation (defeats locality); K/V ~ N(0,1)*0.5."""
    g = torch.Generator().manual_seed(seed)
    seq_lens = list(seq_lens)
    assert len(seq_lens) == num_seqs
    x = 16 // torch.tensor([], dtype=dtype).element_size()
    max_len = max(max(seq_lens), 1)
    blocks_per_seq = (max_len + block_size - 1) // block_size
    num_blocks = num_seqs * blocks_per_seq + extra_blocks
    perm = torch.randperm(num_blocks, generator=g)[: num_seqs * blocks_per_seq]
    block_tables = perm.view(num_seqs, blocks_per_seq).to(torch.int32).contiguous()
    key_cache = torch.zeros(num_blocks, num_kv_heads, head_size // x, block_size, x, dtype=dtype)
    value_cache = torch.zeros(num_blocks, num_kv_heads, head_size, block_size, dtype=dtype)
    k_dense, v_dense = [], []
    for s, n in enumerate(seq_lens):
        k = (torch.randn(n, num_kv_heads, head_size, generator=g) * 0.5).to(dtype)
        v = (torch.randn(n, num_kv_heads, head_size, generator=g) * 0.5).to(dtype)
        k_dense.append(k)
        v_dense.append(v)
        if n == 0:
            continue
        tok = torch.arange(n)
        blk = block_tables[s, tok // block_size].long()
        off = tok % block_size
        # key_cache[blk, h, d // x, off, d % x] = k[tok, h, d]
        key_cache[blk, :, :, off, :] = k.view(n, num_kv_heads, head_size // x, x)
        value_cache[blk, :, :, off] = v
    if q_in_qkv:  # q as a strided view of a fused qkv projection (qwen2.py:151-152)
        qkv = (torch.randn(num_seqs, (num_heads + 2 * num_kv_heads) * head_size, generator=g) * 0.5).to(dtype)
        query = qkv[:, : num_heads * head_size].view(num_seqs, num_heads, head_size)
    else:
        query = (torch.randn(num_seqs, num_heads, head_size, generator=g) * 0.5).to(dtype)
    return dict(query=query, key_cache=key_cache, value_cache=value_cache,
                block_tables=block_tables, seq_lens=torch.tensor(seq_lens, dtype=torch.int32),
                scale=1.0 / math.sqrt(head_size), k_dense=k_dense, v_dense=v_dense,
                block_size=block_size, num_kv_heads=num_kv_heads, max_seq_len=max(seq_lens))


def dense_attention_fp64(inp, alibi_slopes=None):
    """Independent fp64 gather-softmax of the same problem (no paging arithmetic shared
    with the oracle): out[s, h] = softmax(scale * q.K^T + alibi) V."""
    q = inp["query"]
    S, H, D = q.shape
    KVH = inp["num_kv_heads"]
    out = torch.zeros(S, H, D, dtype=torch.float64)
    for s in range(S):
        k, v = inp["k_dense"][s].double(), inp["v_dense"][s].double()
        n = k.shape[0]
        if n == 0:
            continue
        for h in range(H):
            kv = h // (H // KVH)
            logits = (k[:, kv] @ q[s, h].double()) * inp["scale"]
            if alibi_slopes is not None:
                logits = logits + alibi_slopes[h].double() * (torch.arange(n).double() - n + 1)
            p = torch.softmax(logits, dim=0)
            out[s, h] = p @ v[:, kv]
    return out


def max_partitions(max_seq_len, partition=512):
    return max(1, (max_seq_len + partition - 1) // partition)


def v2_scratch(num_seqs, num_heads, head_size, max_seq_len, dtype, device="cpu"):
    P = max_partitions(max_seq_len)
    tmp_out = torch.zeros(num_seqs, num_heads, P, head_size, dtype=dtype, device=device)
    exp_sums = torch.zeros(num_seqs, num_heads, P, dtype=torch.float32, device=device)
    max_logits = torch.zeros_like(exp_sums)
    return exp_sums, max_logits, tmp_out


def make_prefill_inputs(num_heads, num_kv_heads, head_size, block_size, seq_lens, query_lens,
                        dtype=torch.bfloat16, seed=0, garbage=None):
    """Prompt chunks over a paged cache: sequence i has seq_lens[i] tokens in the cache, the last
    query_lens[i] of which are the chunk being computed.  Adds query [T, H, D], query_start_loc,
    max_query_len to make_paged_inputs' dict.  `garbage`: value written into the cache slots past
    each sequence's end (NaN tests)."""
    inp = make_paged_inputs(len(seq_lens), num_heads, num_kv_heads, head_size, block_size, seq_lens,
                            dtype=dtype, seed=seed)
    assert all(0 <= q <= s for q, s in zip(query_lens, seq_lens))
    g = torch.Generator().manual_seed(seed + 1000)
    T = sum(query_lens)
    inp["query"] = (torch.randn(T, num_heads, head_size, generator=g) * 0.5).to(dtype)
    qsl = [0]
    for q in query_lens:
        qsl.append(qsl[-1] + q)
    inp["query_start_loc"] = torch.tensor(qsl, dtype=torch.int32)
    inp["query_lens"] = list(query_lens)
    inp["max_query_len"] = max(query_lens) if query_lens else 0
    if garbage is not None:
        for s, n in enumerate(seq_lens):
            nb = inp["block_tables"].shape[1]
            for t in range(n, nb * block_size):
                b = int(inp["block_tables"][s, t // block_size])
                inp["key_cache"][b, :, :, t % block_size, :] = garbage
                inp["value_cache"][b, :, :, t % block_size] = garbage
    return inp


def dense_prefill_fp64(inp, alibi_slopes=None, sliding_window=0, softcap=0.0):
    """Independent fp64 statement of causal varlen attention, bottom-right aligned:
    query t of a chunk of L tokens in a context of S tokens sits at position S - L + t."""
    q = inp["query"]
    T, H, D = q.shape
    KVH = inp["num_kv_heads"]
    out = torch.zeros(T, H, D, dtype=torch.float64)
    qsl = inp["query_start_loc"].tolist()
    for s, S in enumerate(inp["seq_lens"].tolist()):
        L = qsl[s + 1] - qsl[s]
        if L == 0:
            continue
        k, v = inp["k_dense"][s].double(), inp["v_dense"][s].double()
        pos = torch.arange(S - L, S)
        keys = torch.arange(S)
        mask = keys[None, :] <= pos[:, None]
        if sliding_window > 0:
            mask &= keys[None, :] > pos[:, None] - sliding_window
        for h in range(H):
            kv = h // (H // KVH)
            logits = (q[qsl[s]:qsl[s + 1], h].double() @ k[:, kv].T) * inp["scale"]
            if softcap > 0:
                logits = softcap * torch.tanh(logits / softcap)
            if alibi_slopes is not None:
                logits = logits + alibi_slopes[h].double() * (keys[None, :] - pos[:, None]).double()
            logits = logits.masked_fill(~mask, float("-inf"))
            out[qsl[s]:qsl[s + 1], h] = torch.softmax(logits, dim=1) @ v[:, kv]
    return out


def quantize_paged_inputs_fp8(inp, k_scale=1.0, v_scale=1.0):
    """fp8 (OCP e4m3fn) twin of a make_paged_inputs / make_prefill_inputs dict: caches become uint8
    with x = 16 ([NB, KVH, D/16, BS, 16] / [NB, KVH, D, BS]), elements e4m3(float(x) / scale) with
    saturation at +-448; k_dense / v_dense become the values the cache now represents (fp64)."""
    out = dict(inp)
    kc, vc = inp["key_cache"], inp["value_cache"]
    NB, KVH, _, BS, x = kc.shape
    D = vc.shape[2]
    assert D % 16 == 0
    kd = kc.permute(0, 1, 3, 2, 4).reshape(NB, KVH, BS, D).float()  # [NB, KVH, BS, D]
    q = (kd / k_scale).clamp(-448, 448).to(torch.float8_e4m3fn)
    out["key_cache"] = q.view(torch.uint8).view(NB, KVH, BS, D // 16, 16).permute(0, 1, 3, 2, 4).contiguous()
    out["value_cache"] = (vc.float() / v_scale).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8).contiguous()
    out["k_scale"], out["v_scale"] = k_scale, v_scale
    dt = inp["query"].dtype
    out["k_dense"] = [((k.float() / k_scale).clamp(-448, 448).to(torch.float8_e4m3fn).float() * k_scale).to(dt)
                      for k in inp["k_dense"]]
    out["v_dense"] = [((v.float() / v_scale).clamp(-448, 448).to(torch.float8_e4m3fn).float() * v_scale).to(dt)
                      for v in inp["v_dense"]]
    return out
