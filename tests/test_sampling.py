"""Sampler front half against golden outputs of the reference's own functions
(tests/golden/sampler_front_half.npz, generator oracle/make_golden.py sampler)."""
import os

import numpy as np
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
