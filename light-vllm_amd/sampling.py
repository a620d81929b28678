"""Sampler front half: what happens to the logits of the sampled rows before a token is drawn
(light_vllm/decoding/backends/sampler.py:217-347, 434-454): repetition / frequency / presence
penalties from the token histories, temperature, top-k, top-p, min-p, then greedy argmax or an
exponential-race multinomial.  Torch ops on the device, as in the reference (no custom kernel there
either); formulated independently (descending order, cumulative mass from the top) and checked
against golden outputs of the reference functions (tests/golden/sampler_front_half.npz).
The engine samples plain-greedy requests inside the captured step (arg-max epilogue of the lm_head
projection) and every other request through the device-side sampler (csrc/sampler.hip, device_sampler.py: one launch,
inside the captured step and the multi-step burst); this module is the torch statement of the same stages that the
kernel is compared with (tests/test_sampler_gpu.py) and that the memory-profiling run uses."""
from typing import List, Optional

import torch

_SAMPLING_EPS = 1e-5  # sampling_params.py:14: below it a request is greedy


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
    t = torch.where(temperature >= _SAMPLING_EPS, temperature, torch.ones_like(temperature))
    x = logits / t[:, None]
    if top_p is not None or top_k is not None:
        V = logits.shape[1]
        x = apply_top_k_top_p(x, top_p if top_p is not None else torch.ones_like(t),
                              top_k if top_k is not None else torch.full_like(t, V, dtype=torch.long))
    if min_p is not None:
        x = apply_min_p(x, min_p)
    drawn = random_sample(torch.softmax(x, dim=-1), generator)
    return torch.where(temperature >= _SAMPLING_EPS, drawn, greedy)


class SamplingBatch:
    """Per-row sampling state of one step (the job of SamplingMetadata / SamplingTensors in the reference,
    decoding/backends/sampling_metadata.py): one row per sequence that samples, in the order of the step's
    sampled rows.  `sample(logits)` is Sampler.forward's sequence (sampler.py:90-200): min-tokens mask ->
    penalties -> temperature -> top-k / top-p -> min-p -> softmax -> draw, greedy rows by arg-max."""

    def __init__(self, rows: List[dict], vocab_size: int, device, generator: Optional[torch.Generator] = None):
        """rows[i]: dict(params=SamplingParams | None, prompt=[token ids], output=[token ids], eos=int | None)."""
        from .engine.sampling_params import SamplingParams
        self.n, self.vocab_size, self.device, self.generator = len(rows), vocab_size, device, generator
        ps = [r["params"] or SamplingParams(temperature=0.0) for r in rows]
        f = lambda vals, dt=torch.float32: torch.tensor(vals, dtype=dt, device=device)
        self.temperature = f([p.temperature for p in ps])
        self.top_p = f([p.top_p for p in ps])
        self.top_k = f([vocab_size if p.top_k == -1 else min(p.top_k, vocab_size) for p in ps], torch.long)
        self.min_p = f([p.min_p for p in ps])
        self.presence, self.frequency = f([p.presence_penalty for p in ps]), f([p.frequency_penalty for p in ps])
        self.repetition = f([p.repetition_penalty for p in ps])
        self.do_penalties = any(p.presence_penalty != 0.0 or p.frequency_penalty != 0.0 or p.repetition_penalty != 1.0
                                for p in ps)
        self.do_top = any(p.top_p < 1.0 or p.top_k != -1 for p in ps)
        self.do_min_p = any(p.min_p > 0.0 for p in ps)
        self.all_greedy = all(p.greedy for p in ps)
        self.seeds = [p.seed for p in ps]
        if self.do_penalties:  # padded histories, pad id = vocab_size (sampler.py:_get_bin_counts_and_mask)
            lp = max(1, max(len(r["prompt"]) for r in rows))
            lo = max(1, max(len(r["output"]) for r in rows))
            self.prompt = f([list(r["prompt"]) + [vocab_size] * (lp - len(r["prompt"])) for r in rows], torch.long)
            self.output = f([list(r["output"]) + [vocab_size] * (lo - len(r["output"])) for r in rows], torch.long)
        # min_tokens: stop tokens cannot be sampled before min_tokens outputs exist (sampler.py:238-277)
        self.banned = [(i, t) for i, (p, r) in enumerate(zip(ps, rows)) if p.min_tokens > len(r["output"])
                       for t in set(p.stop_token_ids) | ({r["eos"]} if r.get("eos") is not None and not p.ignore_eos else set())]

    def penalised(self, logits: torch.Tensor) -> torch.Tensor:
        """The logits after the min-tokens mask and the penalties (what a greedy row takes its arg-max of)."""
        assert logits.shape[0] == self.n
        x = logits.float()
        if self.banned:
            rows, toks = zip(*self.banned)
            x[list(rows), list(toks)] = float("-inf")
        if self.do_penalties:
            x = apply_penalties(x, self.prompt, self.output, self.presence, self.frequency, self.repetition)
        return x

    def shaped(self, x: torch.Tensor) -> torch.Tensor:
        """Temperature, top-k / top-p and min-p on penalised logits: what the softmax of Sampler.forward sees
        (sampler.py:158-176) -- the probabilities the draw uses and the log-probabilities `get_logprobs` reports."""
        if self.all_greedy:
            return x
        t = torch.where(self.temperature >= _SAMPLING_EPS, self.temperature, torch.ones_like(self.temperature))
        x = x / t[:, None]
        if self.do_top:
            x = apply_top_k_top_p(x, self.top_p, self.top_k)
        if self.do_min_p:
            x = apply_min_p(x, self.min_p)
        return x

    def logprobs(self, logits: torch.Tensor) -> torch.Tensor:
        """log_softmax of the adjusted logits, fp32 (sampler.py:178-180)."""
        return torch.log_softmax(self.shaped(self.penalised(logits)), dim=-1, dtype=torch.float)

    def sample(self, logits: torch.Tensor) -> torch.Tensor:
        x = self.penalised(logits)
        greedy = greedy_sample(x)
        if self.all_greedy:
            return greedy
        x = self.shaped(x)
        probs = torch.softmax(x, dim=-1)
        drawn = random_sample(probs, self.generator)
        for i, seed in enumerate(self.seeds):  # a seeded request draws from its own stream (sampler.py:479-493)
            if seed is not None:
                # (seed, step) mixed, not added: seed s at step t + 1 must not repeat seed s + 1 at step t
                mixed = (seed * 0x9E3779B97F4A7C15 + int(self._step_of(i)) * 0xD1B54A32D192ED03 + 0x2545F4914F6CDD1D) % (1 << 63)
                g = torch.Generator(device=probs.device).manual_seed(mixed)
                drawn[i] = random_sample(probs[i:i + 1], g)[0]
        return torch.where(self.temperature >= _SAMPLING_EPS, drawn, greedy)

    def _step_of(self, i: int) -> int:
        return self._steps[i] if hasattr(self, "_steps") else 0



def sample_logprobs(logprobs: torch.Tensor, tokens: torch.Tensor, nums: List[Optional[int]]) -> List[Optional[dict]]:
    """The sample half of get_logprobs (sampler.py:726-990) for one sampled token per row: row i with nums[i] = n
    (SamplingParams.logprobs) gets {sampled token: (logprob, rank)} updated with its n most likely tokens
    {token: (logprob, 1 .. n)}; rank = 1 + the number of strictly larger log-probabilities of the row (_get_ranks,
    sampler.py:705-723).  Rows with nums[i] None get None.  One top-k of the largest n for the whole batch, as there."""
    rows = [i for i, n in enumerate(nums) if n is not None]
    out: List[Optional[dict]] = [None] * len(nums)
    if not rows:
        return out
    idx = torch.tensor(rows, dtype=torch.long, device=logprobs.device)
    lp = logprobs.index_select(0, idx)
    tok = tokens.to(lp.device).long().index_select(0, idx)
    sel = lp.gather(1, tok[:, None])[:, 0]
    ranks = (lp > sel[:, None]).sum(1).add_(1)
    largest = max(nums[i] for i in rows)
    sel_l, rank_l, tok_l = sel.tolist(), ranks.tolist(), tok.tolist()
    if largest > 0:
        top_lp, top_id = torch.topk(lp, min(largest, lp.shape[1]), dim=-1)
        top_lp, top_id = top_lp.tolist(), top_id.tolist()
    for j, i in enumerate(rows):
        d = {tok_l[j]: (sel_l[j], rank_l[j])}
        n = min(nums[i], lp.shape[1])
        if n > 0:
            d.update({t: (v, r) for t, v, r in zip(top_id[j][:n], top_lp[j][:n], range(1, n + 1))})
        out[i] = d
    return out
