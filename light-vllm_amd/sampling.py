"""Sampler front half: what happens to the logits of the sampled rows before a token is drawn
(light_vllm/decoding/backends/sampler.py:217-347, 434-454): repetition / frequency / presence
penalties from the token histories, temperature, top-k, top-p, min-p, then greedy argmax or an
exponential-race multinomial.  Torch ops on the device, as in the reference (no custom kernel there
either); formulated independently (descending order, cumulative mass from the top) and checked
against golden outputs of the reference functions (tests/golden/sampler_front_half.npz).
The engine of this package samples greedily inside the captured step; this module serves hosts that
want the other modes."""
from typing import Optional

import torch


def token_counts(tokens: torch.Tensor, vocab_size: int) -> torch.Tensor:
    """[num_seqs, vocab] occurrence counts of padded token histories (pad id = vocab_size)."""
    num_seqs = tokens.shape[0]
    counts = torch.zeros(num_seqs, vocab_size + 1, dtype=torch.long, device=tokens.device)
    counts.scatter_add_(1, tokens, torch.ones_like(tokens))
    return counts[:, :vocab_size]


def apply_penalties(logits: torch.Tensor, prompt_tokens: torch.Tensor, output_tokens: torch.Tensor,
                    presence_penalties: torch.Tensor, frequency_penalties: torch.Tensor,
                    repetition_penalties: torch.Tensor) -> torch.Tensor:
    """sampler.py:281-301: tokens seen in the prompt or the output have positive logits divided and
    negative ones multiplied by the repetition penalty; every output occurrence costs the frequency
    penalty, any occurrence the presence penalty (OpenAI definition)."""
    vocab = logits.shape[1]
    out_counts = token_counts(output_tokens, vocab)
    seen = (token_counts(prompt_tokens, vocab) > 0) | (out_counts > 0)
    rep = torch.where(seen, repetition_penalties[:, None].to(logits.dtype), torch.ones((), dtype=logits.dtype,
                                                                                        device=logits.device))
    logits = torch.where(logits > 0, logits / rep, logits * rep)
    logits = logits - frequency_penalties[:, None] * out_counts
    logits = logits - presence_penalties[:, None] * (out_counts > 0)
    return logits


def apply_top_k_top_p(logits: torch.Tensor, p: torch.Tensor, k: torch.Tensor) -> torch.Tensor:
    """sampler.py:304-330.  Keep the k largest logits of a row, then the smallest set of them whose
    probability mass exceeds 1 - (1 - p) from the top; the rest become -inf.  (The reference sorts
    ascending and drops the prefix whose cumulative mass is <= 1 - p; same set.)"""
    srt, idx = logits.sort(dim=-1, descending=True)
    # top-k: everything strictly below the k-th largest value goes (ties with it stay, as in the reference)
    kth = srt.gather(1, (k.to(torch.long) - 1).clamp_(0, logits.shape[1] - 1).unsqueeze(1))
    srt = srt.masked_fill(srt < kth, float("-inf"))
    # top-p on what is left: a token goes when the mass of it and everything below it is <= 1 - p
    probs = srt.softmax(dim=-1)
    mass_from_bottom = probs.flip(-1).cumsum(-1).flip(-1)
    drop = mass_from_bottom <= (1 - p).unsqueeze(1)
    drop[:, 0] = False  # at least one token survives
    srt = srt.masked_fill(drop, float("-inf"))
    return torch.empty_like(srt).scatter_(-1, idx, srt)


def apply_min_p(logits: torch.Tensor, min_p: torch.Tensor) -> torch.Tensor:
    """sampler.py:333-347: tokens whose probability is below min_p times the top probability go."""
    probs = torch.softmax(logits, dim=-1)
    top = probs.amax(dim=-1, keepdim=True)
    return logits.masked_fill(probs < min_p[:, None] * top, float("-inf"))


def greedy_sample(logits: torch.Tensor) -> torch.Tensor:
    return torch.argmax(logits, dim=-1)


def random_sample(probs: torch.Tensor, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """sampler.py:434-454 (_multinomial): argmax of probs / Exp(1) noise -- a multinomial draw with no
    host synchronisation."""
    q = torch.empty_like(probs).exponential_(generator=generator)
    return (probs / q).argmax(dim=-1)


def sample(logits: torch.Tensor, temperature: Optional[torch.Tensor] = None, top_p: Optional[torch.Tensor] = None,
           top_k: Optional[torch.Tensor] = None, min_p: Optional[torch.Tensor] = None,
           generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """Sampler.forward's order (sampler.py:90-200) for one step: temperature -> top-k/top-p -> min-p ->
    softmax -> draw; rows with temperature 0 are greedy."""
    logits = logits.float()
    greedy = greedy_sample(logits)
    if temperature is None:
        return greedy
    t = torch.where(temperature > 0, temperature, torch.ones_like(temperature))
    x = logits / t[:, None]
    if top_p is not None or top_k is not None:
        V = logits.shape[1]
        x = apply_top_k_top_p(x, top_p if top_p is not None else torch.ones_like(t),
                              top_k if top_k is not None else torch.full_like(t, V, dtype=torch.long))
    if min_p is not None:
        x = apply_min_p(x, min_p)
    drawn = random_sample(torch.softmax(x, dim=-1), generator)
    return torch.where(temperature > 0, drawn, greedy)
