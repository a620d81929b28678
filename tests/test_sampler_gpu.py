"""The device-side sampler (csrc/sampler.hip, lvllm_sample_rows through torch.ops._C_amd.sample_rows) on the GPU:
 * against golden outputs of the reference's own functions (tests/golden/sampler_front_half.npz: _apply_penalties,
   _apply_top_k_top_p, _apply_min_p of light_vllm/decoding/backends/sampler.py:281-347) -- penalised logits, kept-token
   sets, min-p output;
 * against torch statements of the same stages at the model's vocabulary size (light_vllm_amd.sampling, itself pinned
   to the golden file on the CPU);
 * the draw: an exponential-race multinomial (frequencies follow the probabilities), repeatable per (seed, step),
   confined to the kept set; the state the kernel keeps on the device (output counts, step counter)."""
import os
import struct

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _params(**kw):
    from light_vllm_amd.engine.sampling_params import SamplingParams
    return SamplingParams(**kw)


def make_sampler(vocab, rows_params, prompts=None, outputs=None, eos=None, seeds=None):
    """A DeviceSampler with one slot per row, built from SamplingParams and (padded) token histories."""
    import light_vllm_amd  # noqa: F401
    from light_vllm_amd.device_sampler import DeviceSampler
    ds = DeviceSampler(vocab, DEV, num_slots=len(rows_params), seed=0)
    slots = []
    for i, sp in enumerate(rows_params):
        if seeds is not None:
            sp.seed = seeds[i]
        p = [t for t in (prompts[i] if prompts is not None else []) if t < vocab]
        o = [t for t in (outputs[i] if outputs is not None else []) if t < vocab]
        slots.append(ds.ensure(i, sp, p, o, eos))
    return ds, torch.tensor(slots, dtype=torch.int32, device=DEV)


def load_golden():
    z = np.load(os.path.join(GOLDEN, "sampler_front_half.npz"))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def run(ds, slots, logits, update_state=False):
    processed = torch.full(logits.shape, float("nan"), dtype=torch.float32, device=DEV)
    tok = ds.sample(logits.to(DEV), slots, processed_out=processed, update_state=update_state)
    torch.cuda.synchronize()
    return tok.cpu(), processed.cpu()


def test_penalised_logits_match_the_reference_golden():
    z = load_golden()
    N, V = z["logits"].shape
    ps = [_params(temperature=1.0, presence_penalty=float(z["pres"][i]), frequency_penalty=float(z["freq"][i]),
                  repetition_penalty=float(z["rep"][i])) for i in range(N)]
    ds, slots = make_sampler(V, ps, z["prompt"].tolist(), z["output"].tolist())
    _, processed = run(ds, slots, z["logits"])
    assert torch.allclose(processed, z["penalised"], atol=1e-6, rtol=1e-6)
    # greedy rows with penalties: the arg-max of the reference's penalised logits
    gs = [_params(temperature=0.0, presence_penalty=float(z["pres"][i]), frequency_penalty=float(z["freq"][i]),
                  repetition_penalty=float(z["rep"][i])) for i in range(N)]
    ds, slots = make_sampler(V, gs, z["prompt"].tolist(), z["output"].tolist())
    tok, _ = run(ds, slots, z["logits"])
    assert tok.tolist() == z["penalised"].argmax(-1).tolist()


def test_top_k_top_p_keep_the_reference_golden_sets():
    z = load_golden()
    N, V = z["logits"].shape
    ps = [_params(temperature=1.0, top_p=float(z["top_p"][i]), top_k=int(z["top_k"][i]) if z["top_k"][i] < V else -1)
          for i in range(N)]
    ds, slots = make_sampler(V, ps)
    tok, processed = run(ds, slots, z["logits"])
    assert torch.equal(torch.isinf(processed), torch.isinf(z["filtered"]))
    keep = ~torch.isinf(processed)
    assert torch.equal(processed[keep], z["filtered"][keep])
    assert keep[torch.arange(N), tok].all()  # the draw lands inside the kept set
    assert int(keep[4].sum()) == 1 and tok[4] == z["logits"][4].argmax()  # top_k = 1


def test_min_p_matches_the_reference_golden():
    z = load_golden()
    N, V = z["logits"].shape
    ds, slots = make_sampler(V, [_params(temperature=1.0, min_p=float(z["min_p"][i])) for i in range(N)])
    _, processed = run(ds, slots, z["logits"])
    assert torch.equal(processed, z["min_p_out"])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_plain_greedy_rows_are_torch_argmax_with_ties(dtype):
    """Rows without a state slot: arg-max of the logits, the smaller index on exact ties (torch.argmax), at the model's
    vocabulary size; with and without a state_slot tensor."""
    V = 128256
    g = torch.Generator().manual_seed(0)
    logits = (torch.randn(9, V, generator=g) * 2).to(dtype)
    logits[1, 77] = logits[1, 90000] = logits[1].float().max() + 1   # a tie for the maximum
    logits[2, V - 1] = 1e4 if dtype != torch.float16 else 6e4        # the very last element
    logits[3, :] = 0.5                                               # everything equal: index 0
    want = logits.float().argmax(-1)
    assert want[1] == 77 and want[2] == V - 1 and want[3] == 0
    tok = torch.empty(9, dtype=torch.long, device=DEV)
    torch.ops._C_amd.sample_rows(tok, logits.to(DEV), None, None, None, None, None, False)
    assert tok.cpu().tolist() == want.tolist()
    ds, _ = make_sampler(V, [_params(temperature=0.7)])
    slots = torch.full((9,), -1, dtype=torch.int32, device=DEV)
    assert ds.sample(logits.to(DEV), slots).cpu().tolist() == want.tolist()


def test_filters_at_the_model_vocabulary_match_the_torch_statement():
    """bf16 logits of a 128 256-token vocabulary (many exact ties), top-k and top-p and min-p together: kept sets
    against light_vllm_amd.sampling's torch statement (pinned to the reference's golden outputs on the CPU).  Exact
    ties at the top-p cut are kept as a group here and one by one there: sets may differ only by elements tied with
    the cut value."""
    import light_vllm_amd  # noqa: F401
    from light_vllm_amd import sampling
    V = 128256
    g = torch.Generator().manual_seed(1)
    logits = (torch.randn(8, V, generator=g) * 3).to(torch.bfloat16)
    cfgs = [(1.0, 50, 0.9, 0.0), (0.7, -1, 0.95, 0.0), (1.3, 1000, 1.0, 0.0), (1.0, -1, 1.0, 0.05),
            (0.5, 20, 0.5, 0.1), (2.0, -1, 0.8, 0.0), (1.0, 1, 1.0, 0.0), (1.0, 3, 0.1, 0.0)]
    ps = [_params(temperature=t, top_k=k, top_p=p, min_p=mp) for t, k, p, mp in cfgs]
    ds, slots = make_sampler(V, ps)
    tok, processed = run(ds, slots, logits)
    x = logits.float().to(DEV) / torch.tensor([c[0] for c in cfgs], device=DEV)[:, None]
    want = sampling.apply_top_k_top_p(x, torch.tensor([c[2] for c in cfgs], device=DEV),
                                      torch.tensor([V if c[1] == -1 else c[1] for c in cfgs], device=DEV))
    want = sampling.apply_min_p(want, torch.tensor([c[3] for c in cfgs], device=DEV)).cpu()
    for r in range(8):
        a, b = ~torch.isinf(processed[r]), ~torch.isinf(want[r])
        diff = a ^ b
        if diff.any():  # only elements exactly tied with the smallest kept value may differ
            cut = min(processed[r][a].min().item(), want[r][b].min().item())
            assert (x[r].cpu()[diff] == cut).all(), (r, int(diff.sum()))
        assert a[tok[r]], r
        assert torch.equal(processed[r][a & b], want[r][a & b])
    assert int((~torch.isinf(processed[6])).sum()) == 1 and int((~torch.isinf(processed[7])).sum()) <= 3


def test_the_draw_is_multinomial_and_repeatable():
    """4096 rows with the same logits and different seeds: token frequencies follow softmax(logits / T) over the kept
    set; the same (seed, step) draws the same token, the next step another stream; update_state advances the step
    counter and the output counts on the device."""
    V, R = 50, 4096
    g = torch.Generator().manual_seed(2)
    row = torch.randn(V, generator=g) * 1.5
    logits = row[None].repeat(R, 1).contiguous()
    ps = [_params(temperature=0.8, top_k=10, seed=1000 + i) for i in range(R)]
    ds, slots = make_sampler(V, ps)
    t1, processed = run(ds, slots, logits)
    t2, _ = run(ds, slots, logits)
    assert torch.equal(t1, t2)  # nothing advanced: same seed, same step
    kept = ~torch.isinf(processed[0])
    assert int(kept.sum()) == 10 and kept[t1].all()
    probs = torch.softmax((row / 0.8).masked_fill(~kept, float("-inf")), -1)
    freq = torch.bincount(t1, minlength=V).float() / R
    assert float((freq - probs).abs().max()) < 0.03, (freq - probs).abs().max()
    # the state advances on the device
    t3, _ = run(ds, slots, logits, update_state=True)
    assert torch.equal(t3, t1)
    rec = ds.params.cpu().numpy()
    out_len = np.array([struct.unpack_from("<i", rec[i].tobytes(), 40)[0] for i in range(R)])
    assert (out_len == 1).all()
    counts = ds.counts.cpu()
    assert torch.equal(counts[torch.arange(R), t1], torch.ones(R, dtype=torch.int32)) and int(counts.sum()) == R
    t4, _ = run(ds, slots, logits, update_state=True)
    assert (t4 != t3).float().mean() > 0.5  # step 1 draws from another stream
    freq4 = torch.bincount(t4, minlength=V).float() / R
    assert float((freq4 - probs).abs().max()) < 0.03
    # adjacent seeds at adjacent steps are unrelated streams (a seed + step sum would make them equal)
    assert (t4[:-1] != t3[1:]).float().mean() > 0.5


def test_min_tokens_bans_stop_tokens_until_enough_outputs_exist():
    V = 64
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(3, V, generator=g)
    eos = int(logits[0].argmax())
    stop = int(logits[1].argmax())
    ps = [_params(temperature=0.0, min_tokens=3), _params(temperature=0.0, min_tokens=2, stop_token_ids=[stop]),
          _params(temperature=0.0, min_tokens=1)]
    logits[2, eos] = 50.0
    ds, slots = make_sampler(V, ps, outputs=[[1, 2], [5], [9]], eos=eos)
    tok, processed = run(ds, slots, logits)
    assert tok[0] != eos and tok[0] == logits[0].topk(2).indices[1] and processed[0, eos] == float("-inf")
    assert tok[1] != stop and processed[1, stop] == float("-inf")
    assert tok[2] == eos  # min_tokens reached: nothing banned


def test_penalty_state_follows_the_tokens_the_kernel_draws():
    """Greedy with a frequency penalty, state kept on the device over 12 steps == the torch statement fed the growing
    output history from the host."""
    import light_vllm_amd  # noqa: F401
    from light_vllm_amd import sampling
    V = 40
    g = torch.Generator().manual_seed(4)
    logits = torch.randn(2, V, generator=g)
    prompt = [[1, 2, 3], [4]]
    ps = [_params(temperature=0.0, frequency_penalty=0.9, repetition_penalty=1.2), _params(temperature=0.0, presence_penalty=1.5)]
    ds, slots = make_sampler(V, ps, prompts=prompt, outputs=[[], []])
    outs = [[], []]
    for _ in range(12):
        tok, _ = run(ds, slots, logits, update_state=True)
        pt = torch.tensor([p + [V] * (3 - len(p)) for p in prompt])
        ot = torch.tensor([o + [V] * (12 - len(o)) for o in outs])
        want = sampling.apply_penalties(logits.clone(), pt, ot, torch.tensor([0.0, 1.5]), torch.tensor([0.9, 0.0]),
                                        torch.tensor([1.2, 1.0])).argmax(-1)
        assert tok.tolist() == want.tolist()
        for i in range(2):
            outs[i].append(int(tok[i]))
    assert len(set(outs[0])) > 1


def test_an_evicted_slot_is_rebuilt_to_the_same_state():
    """Three requests, two slots: the least recently used slot that is not pinned is evicted; ensure() rebuilds the
    evicted request's state from its histories -- the same bytes -- and keeps its seed."""
    import light_vllm_amd  # noqa: F401
    from light_vllm_amd.device_sampler import DeviceSampler
    V = 97
    ds = DeviceSampler(V, DEV, num_slots=2, seed=3)
    sp = [_params(temperature=0.9, frequency_penalty=0.5), _params(temperature=0.7, top_k=5), _params(temperature=1.1, seed=42)]
    hist = [([1, 2, 3, 3], [7, 7, 8]), ([4], []), ([5, 6], [9])]
    s0 = ds.ensure(10, sp[0], *hist[0], None)
    s1 = ds.ensure(11, sp[1], *hist[1], None)
    torch.cuda.synchronize()
    rec0, cnt0 = ds.params[s0].cpu().clone(), ds.counts[s0].cpu().clone()
    assert cnt0[3] == -2147483648 and cnt0[7] == 2 and cnt0[8] == 1 and cnt0[1] == -2147483648 and int((cnt0 != 0).sum()) == 5
    s2 = ds.ensure(12, sp[2], *hist[2], None, pinned=[11])  # evicts request 10 (LRU, not pinned)
    assert s2 == s0 and ds.slot_of(10) is None and ds.evictions == 1
    with pytest.raises(RuntimeError):
        ds.ensure(13, sp[0], [1], [], None, pinned=lambda: {11, 12})
    ds.release(12)
    s0b = ds.ensure(10, sp[0], *hist[0], None)
    torch.cuda.synchronize()
    assert torch.equal(ds.params[s0b].cpu(), rec0) and torch.equal(ds.counts[s0b].cpu(), cnt0)  # same seed, same counts


def test_first_pass_shared_by_workgroups_gives_the_same_tokens_and_state():
    """With the tail behind the vocabulary in its scratch rows (DeviceSampler.new_scratch) and a multiple of 8 rows, up to
    8 workgroups share a row's first pass and meet through one L2; without the tail one workgroup does it all.  Same
    tokens, same processed logits, same device state, launch after launch (the meeting counter is left at zero)."""
    V, R = 128256, 16
    g = torch.Generator().manual_seed(7)
    logits = [(torch.randn(R, V, generator=g) * 3).to(torch.bfloat16).to(DEV) for _ in range(3)]
    kinds = [dict(temperature=0.8, top_k=40, top_p=0.9), dict(temperature=0.0, repetition_penalty=1.3),
             dict(temperature=1.1, min_p=0.02, frequency_penalty=0.4, presence_penalty=0.2), dict(temperature=0.6)]
    prompts = [torch.randint(0, V, (60,), generator=g).tolist() for _ in range(R)]
    results = []
    for shared in (True, False):
        ds, slots = make_sampler(V, [_params(**kinds[i % 4]) for i in range(R)], prompts=prompts,
                                 seeds=[100 + i for i in range(R)])
        slots[3] = -1  # plain greedy rows among them
        slots[12] = -1
        scratch = ds.new_scratch(R) if shared else torch.empty(R, V, dtype=torch.float32, device=DEV)
        toks, procs = [], []
        for step in range(3):
            processed = torch.full((R, V), float("nan"), dtype=torch.float32, device=DEV)
            toks.append(ds.sample(logits[step], slots, scratch=scratch, processed_out=processed, update_state=True).cpu())
            procs.append(processed.cpu())
        torch.cuda.synchronize()
        if shared:
            Vp = (V + 3) & ~3
            assert not scratch[:, [Vp + 16, Vp + 48]].view(torch.int32).any()  # the arrival counters are back at zero
        results.append((toks, procs, ds.counts.cpu(), ds.params.cpu()))
    (ta, pa, ca, qa), (tb, pb, cb, qb) = results
    for a, b in zip(ta, tb):
        assert torch.equal(a, b)
    for a, b in zip(pa, pb):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    assert torch.equal(ca, cb) and torch.equal(qa, qb)


def test_the_uniform_of_the_draw_never_reaches_its_ends():
    """with 24 random bits + 0.5 the sum 16 777 215.5 rounded to 2^24, u was exactly 1.0 once in
    2^24 draws, q = -ln(1) = 0 and the race score +inf: that token won whatever its probability (~0.8 % of the draws
    over a 128 k vocabulary).  The device arithmetic itself (lvllm_sampler_draw_probe = the draw's own helpers) at the
    ends of the range and on random words: 0 < u < 1, q > 0, the score term finite."""
    import ctypes
    from light_vllm_amd import _native
    lib = _native.load_hip_library()
    lib.lvllm_sampler_draw_probe.restype = ctypes.c_int
    g = torch.Generator().manual_seed(5)
    words = torch.cat([torch.tensor([0, 1, 0x1FF, 0x200, 0x7FFFFFFF, 0x80000000, 0xFFFFFDFF, 0xFFFFFE00, 0xFFFFFEFF,
                                     0xFFFFFF00, 0xFFFFFFFE, 0xFFFFFFFF], dtype=torch.int64),
                       torch.randint(0, 2**32, (4084,), generator=g, dtype=torch.int64)])
    r = words & 0xFFFFFFFF
    r32 = torch.from_numpy(r.numpy().astype(np.uint32).view(np.int32)).to(DEV)
    out = torch.full((r32.numel(), 3), float("nan"), dtype=torch.float32, device=DEV)
    rc = lib.lvllm_sampler_draw_probe(ctypes.c_void_p(r32.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                      ctypes.c_int(r32.numel()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, lib.lvllm_last_error()
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    u, q, lq = o[:, 0], o[:, 1], o[:, 2]
    want_u = ((r.numpy().astype(np.uint32) >> 9).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -23)
    assert np.array_equal(u, want_u)                       # exact: 23 bits + 0.5 fits fp32
    assert u.min() > 0.0 and u.max() < 1.0, (u.min(), u.max())
    assert u[11] == np.float32(1.0) - np.float32(2.0 ** -24) and u[0] == np.float32(2.0 ** -24)
    assert np.isfinite(q).all() and (q > 0).all()
    assert np.isfinite(lq).all(), lq[~np.isfinite(lq)]
    # the round-3 form for comparison: it does reach 1.0 in fp32
    old = (np.float32(0xFFFFFFFF >> 8) + np.float32(0.5)) * np.float32(2.0 ** -24)
    assert old == np.float32(1.0)
