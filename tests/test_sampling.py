"""Sampler front half against golden outputs of the reference's own functions
(tests/golden/sampler_front_half.npz, generator oracle/make_golden.py sampler)."""
import os

import numpy as np
import pytest
import torch

import light_vllm_amd  # noqa: F401
from light_vllm_amd import sampling

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load():
    z = np.load(os.path.join(GOLDEN, "sampler_front_half.npz"))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def test_penalties_match_the_reference():
    z = _load()
    got = sampling.apply_penalties(z["logits"].clone(), z["prompt"], z["output"], z["pres"], z["freq"], z["rep"])
    assert torch.allclose(got, z["penalised"], atol=1e-6, rtol=1e-6)


def test_top_k_top_p_keep_the_same_tokens():
    z = _load()
    got = sampling.apply_top_k_top_p(z["logits"].clone(), z["top_p"], z["top_k"])
    assert torch.equal(torch.isinf(got), torch.isinf(z["filtered"]))
    keep = ~torch.isinf(got)
    assert torch.equal(got[keep], z["filtered"][keep])
    assert (keep.sum(1) >= 1).all() and int(keep[4].sum()) == 1 and int(keep[3].sum()) <= 3


def test_min_p_matches_the_reference():
    z = _load()
    got = sampling.apply_min_p(z["logits"].clone(), z["min_p"])
    assert torch.equal(got, z["min_p_out"])


def test_sample_modes():
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(4, 50, generator=g)
    assert torch.equal(sampling.sample(logits), logits.argmax(-1))
    t = torch.tensor([0.0, 1.0, 1.0, 0.7])
    tok = sampling.sample(logits, t, top_k=torch.tensor([50, 1, 50, 5]), generator=torch.Generator().manual_seed(1))
    assert tok[0] == logits[0].argmax() and tok[1] == logits[1].argmax()  # temperature 0; top-1
    assert tok[3] in logits[3].topk(5).indices
    # the exponential race is a multinomial draw: empirical frequencies follow the probabilities
    probs = torch.tensor([[0.6, 0.3, 0.1]]).repeat(20000, 1)
    draws = sampling.random_sample(probs, torch.Generator().manual_seed(2))
    freq = torch.bincount(draws, minlength=3).float() / 20000
    assert torch.allclose(freq, probs[0], atol=0.02)


# ---- SamplingBatch: the sampler's order of operations on one step's rows (CPU) ----
def _params(**kw):
    from light_vllm_amd.engine.sampling_params import SamplingParams
    return SamplingParams(**kw)


def test_sampling_batch_follows_the_reference_order_of_operations():
    """penalties -> temperature -> top-k / top-p -> min-p -> softmax -> draw (sampler.py:90-200), each stage
    against the golden outputs of the reference's own functions."""
    import numpy as np
    import os
    from light_vllm_amd.sampling import SamplingBatch, apply_penalties, apply_top_k_top_p
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "sampler_front_half.npz"))
    logits = torch.from_numpy(z["logits"])
    N, V = logits.shape
    prompt, output = z["prompt"], z["output"]
    rows = []
    for i in range(N):
        rows.append(dict(params=_params(temperature=0.0, presence_penalty=float(z["pres"][i]),
                                        frequency_penalty=float(z["freq"][i]), repetition_penalty=float(z["rep"][i])),
                         prompt=[t for t in prompt[i].tolist() if t < V], output=[t for t in output[i].tolist() if t < V],
                         eos=None))
    b = SamplingBatch(rows, V, "cpu")
    # greedy with penalties = arg-max of the reference's penalised logits
    assert b.sample(logits.clone()).tolist() == torch.from_numpy(z["penalised"]).argmax(-1).tolist()
    # top-k = 1 at any temperature is the arg-max of the (unpenalised) logits
    rows1 = [dict(params=_params(temperature=0.7, top_k=1), prompt=[], output=[], eos=None) for _ in range(N)]
    assert SamplingBatch(rows1, V, "cpu").sample(logits.clone()).tolist() == logits.argmax(-1).tolist()
    # draws only land on tokens the reference's top-k / top-p filter keeps
    rows2 = [dict(params=_params(temperature=1.0, top_p=float(z["top_p"][i]), top_k=int(z["top_k"][i]) if z["top_k"][i] < V else -1),
                  prompt=[], output=[], eos=None) for i in range(N)]
    keep = torch.from_numpy(z["filtered"]) > float("-inf")
    g = torch.Generator().manual_seed(0)
    for _ in range(50):
        drawn = SamplingBatch(rows2, V, "cpu", g).sample(logits.clone())
        assert keep[torch.arange(N), drawn].all()


def test_sampling_batch_min_tokens_seed_and_greedy_rows():
    from light_vllm_amd.sampling import SamplingBatch
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(4, 50, generator=g)
    eos = int(logits[0].argmax())
    rows = [dict(params=_params(temperature=0.0, min_tokens=3), prompt=[1], output=[2], eos=eos),   # eos banned
            dict(params=_params(temperature=0.0, min_tokens=1), prompt=[1], output=[2], eos=int(logits[1].argmax())),
            dict(params=_params(temperature=1.0, seed=11), prompt=[1], output=[2, 3], eos=None),
            dict(params=None, prompt=[1], output=[], eos=None)]
    b = SamplingBatch(rows, 50, "cpu", torch.Generator().manual_seed(1))
    b._steps = [1, 1, 2, 0]
    out = b.sample(logits.clone())
    assert out[0] != eos and out[0] == logits[0].topk(2).indices[1]   # the runner-up
    assert out[1] == logits[1].argmax() and out[3] == logits[3].argmax()
    # a seeded row repeats whatever the shared generator did before
    b2 = SamplingBatch(rows, 50, "cpu", torch.Generator().manual_seed(999))
    b2._steps = [1, 1, 2, 0]
    assert b2.sample(logits.clone())[2] == out[2]


def test_sampling_params_validation_and_greedy_normalisation():
    import pytest
    p = _params(temperature=0.0, top_p=0.5, top_k=7, min_p=0.3)
    assert p.greedy and p.plain_greedy and (p.top_p, p.top_k, p.min_p) == (1.0, -1, 0.0)
    assert not _params(temperature=0.0, repetition_penalty=1.2).plain_greedy
    for bad in (dict(temperature=-1), dict(top_p=0.0), dict(top_k=0), dict(min_p=1.5), dict(repetition_penalty=0.0),
                dict(presence_penalty=3.0), dict(max_tokens=0)):
        with pytest.raises(ValueError):
            _params(**bad)


def test_the_draw_uniform_is_strictly_inside_the_unit_interval():
    """The fp32 arithmetic of the device draw (csrc/sampler.hip draw_uniform), restated in numpy: 23 random bits + 0.5
    scaled by 2^-23 is exact and stays in [2^-24, 1 - 2^-24]; the round-3 form (24 bits + 0.5, scaled by 2^-24)
    rounds to 1.0 at the top word, which made -ln(u) = 0 and the race score infinite."""
    import numpy as np
    r = np.array([0, 1, 0x1FF, 0x200, 0xFFFFFE00, 0xFFFFFFFF], dtype=np.uint32)
    u = ((r >> 9).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -23)
    assert u.dtype == np.float32 and u.min() == np.float32(2.0 ** -24) and u.max() == np.float32(1.0) - np.float32(2.0 ** -24)
    q = -np.log(u.astype(np.float64))
    assert (q > 0).all() and np.isfinite(np.log(q)).all()
    old = ((r >> 8).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -24)
    assert old.max() == np.float32(1.0)


def test_sample_logprobs_follow_the_reference_rank_and_top_n():
    """The sample half of get_logprobs (sampler.py:726-990): {sampled token: (logprob, rank)} updated with the n most
    likely tokens at ranks 1 .. n; rank = 1 + the number of strictly larger log-probabilities (the reference's own
    _get_ranks, imported where the reference is present); rows that did not ask get None; one request with n = 0 gets
    the sampled token alone."""
    g = torch.Generator().manual_seed(5)
    logits = torch.randn(6, 97, generator=g)
    logits[2, 10] = logits[2, 11] = logits[2].max() + 1.0  # a tie at the top: both "rank 1"
    lp = torch.log_softmax(logits, dim=-1)
    toks = torch.tensor([3, 96, 11, 0, 50, 7])
    nums = [5, None, 2, 0, 1, None]
    got = sampling.sample_logprobs(lp, toks, nums)
    assert got[1] is None and got[5] is None
    for i, n in enumerate(nums):
        if n is None:
            continue
        d, t = got[i], int(toks[i])
        want_rank = int((lp[i] > lp[i, t]).sum()) + 1
        top = torch.topk(lp[i], n) if n > 0 else None
        if n > 0 and t in top.indices.tolist():  # the top-n entry replaces the sampled token's (dict.update)
            want_rank = top.indices.tolist().index(t) + 1
        assert d[t][0] == pytest.approx(float(lp[i, t])) and d[t][1] == want_rank, (i, d[t], want_rank)
        assert len(d) == n + (0 if n > 0 and t in top.indices.tolist() else 1)
        if n > 0:
            for r, (tid, val) in enumerate(zip(top.indices.tolist(), top.values.tolist()), start=1):
                assert d[tid] == (pytest.approx(val), r)
    assert got[3] == {0: (pytest.approx(float(lp[3, 0])), int((lp[3] > lp[3, 0]).sum()) + 1)}
    if os.path.isdir("/root/reference/light_vllm"):  # dev container: the reference's own rank function
        from oracle import ref_block_manager
        ref_block_manager.load()  # registers the light_vllm namespace + stubs
        from light_vllm.decoding.backends import sampler as S
        rows = [i for i, n in enumerate(nums) if n == 0]
        assert [got[i][int(toks[i])][1] for i in rows] == S._get_ranks(lp[rows], toks[rows]).tolist()


def test_sampling_batch_logprobs_are_the_log_softmax_the_draw_uses():
    """SamplingBatch.logprobs = log_softmax of the logits after the min-tokens mask, the penalties, the temperature
    and the filters (sampler.py:158-180): greedy rows see temperature 1, filtered tokens are -inf, each row sums to 1."""
    from light_vllm_amd.engine.sampling_params import SamplingParams
    V = 64
    g = torch.Generator().manual_seed(9)
    logits = torch.randn(3, V, generator=g) * 3
    rows = [dict(params=SamplingParams(temperature=0.0, logprobs=2), prompt=[1, 2], output=[3], eos=None),
            dict(params=SamplingParams(temperature=0.5, top_k=4, logprobs=1), prompt=[5], output=[], eos=None),
            dict(params=SamplingParams(temperature=1.0, repetition_penalty=1.5, presence_penalty=0.5, logprobs=0),
                 prompt=[7, 8], output=[9, 9], eos=None)]
    b = sampling.SamplingBatch(rows, V, "cpu")
    lp = b.logprobs(logits.clone())
    assert torch.allclose(lp.exp().sum(-1), torch.ones(3), atol=1e-5)
    assert torch.allclose(lp[0], torch.log_softmax(logits[0], -1), atol=1e-6)       # greedy, no penalties: raw
    assert int(torch.isfinite(lp[1]).sum()) == 4                                    # top-k 4
    kept = torch.topk(logits[1], 4).indices
    assert torch.allclose(lp[1, kept], torch.log_softmax(logits[1, kept] / 0.5, -1), atol=1e-5)
    x = logits[2].clone()
    seen = torch.tensor([7, 8, 9])
    x[seen] = torch.where(x[seen] > 0, x[seen] / 1.5, x[seen] * 1.5)
    x[9] -= 0.5
    assert torch.allclose(lp[2], torch.log_softmax(x, -1), atol=1e-5)
    with pytest.raises(ValueError, match="logprobs must be non-negative"):
        SamplingParams(logprobs=-1)
    with pytest.raises(ValueError, match="prompt_logprobs must be non-negative"):
        SamplingParams(prompt_logprobs=-2)


def test_sampling_params_n_and_best_of():
    """sampling_params.py:285-297, _verify_greedy_sampling: best_of >= n, several sequences need a temperature."""
    from light_vllm_amd.engine.sampling_params import SamplingParams
    assert SamplingParams().num_samples == 1
    assert SamplingParams(n=3, temperature=0.7).num_samples == 3
    assert SamplingParams(n=2, best_of=5, temperature=0.7).num_samples == 5
    with pytest.raises(ValueError, match="best_of must be greater than or equal to n"):
        SamplingParams(n=3, best_of=2, temperature=0.7)
    with pytest.raises(ValueError, match="best_of must be 1 when using greedy sampling"):
        SamplingParams(n=2, temperature=0.0)
    with pytest.raises(ValueError, match="n must be at least 1"):
        SamplingParams(n=0)
