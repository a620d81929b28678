"""Sampling parameters of a request: the numeric subset of light_vllm/decoding/backends/sampling_params.py:94-216
that the sampler's front half reads (penalties, temperature, top-k / top-p / min-p, seed) plus the stop criteria
the output processor checks (max_tokens, stop_token_ids, ignore_eos) and `logprobs` / `prompt_logprobs` (get_logprobs, sampler.py:726-990).  Same names, defaults and validation
messages; text-side fields (stop strings, detokenisation, guided decoding) belong to the tokenizer front end,
which is out of this path's scope."""
from dataclasses import dataclass, field
from typing import List, Optional

_SAMPLING_EPS = 1e-5  # sampling_params.py:17


@dataclass
class SamplingParams:
    n: int = 1                      # sequences returned per prompt
    best_of: Optional[int] = None   # sequences generated per prompt (>= n; the n most likely are returned); None = n
    presence_penalty: float = 0.0
    frequency_penalty: float = 0.0
    repetition_penalty: float = 1.0
    temperature: float = 1.0
    top_p: float = 1.0
    top_k: int = -1
    min_p: float = 0.0
    seed: Optional[int] = None
    stop_token_ids: List[int] = field(default_factory=list)
    ignore_eos: bool = False
    max_tokens: Optional[int] = 16
    min_tokens: int = 0
    logprobs: Optional[int] = None         # log-probabilities of the sampled token and of the n most likely ones
    prompt_logprobs: Optional[int] = None  # the same for every prompt token but the first

    def __post_init__(self) -> None:  # sampling_params.py:_verify_args
        if self.n < 1:
            raise ValueError(f"n must be at least 1, got {self.n}.")
        if self.best_of is not None and self.best_of < self.n:
            raise ValueError(f"best_of must be greater than or equal to n, got n={self.n} and best_of={self.best_of}.")
        if not -2.0 <= self.presence_penalty <= 2.0:
            raise ValueError(f"presence_penalty must be in [-2, 2], got {self.presence_penalty}.")
        if not -2.0 <= self.frequency_penalty <= 2.0:
            raise ValueError(f"frequency_penalty must be in [-2, 2], got {self.frequency_penalty}.")
        if not 0.0 < self.repetition_penalty <= 2.0:
            raise ValueError(f"repetition_penalty must be in (0, 2], got {self.repetition_penalty}.")
        if self.temperature < 0.0:
            raise ValueError(f"temperature must be non-negative, got {self.temperature}.")
        if not 0.0 < self.top_p <= 1.0:
            raise ValueError(f"top_p must be in (0, 1], got {self.top_p}.")
        if self.top_k < -1 or self.top_k == 0:
            raise ValueError(f"top_k must be -1 (disable), or at least 1, got {self.top_k}.")
        if not 0.0 <= self.min_p <= 1.0:
            raise ValueError(f"min_p must be in [0, 1], got {self.min_p}.")
        if self.max_tokens is not None and self.max_tokens < 1:
            raise ValueError(f"max_tokens must be at least 1, got {self.max_tokens}.")
        if self.min_tokens < 0:
            raise ValueError(f"min_tokens must be greater than or equal to 0, got {self.min_tokens}.")
        if self.logprobs is not None and self.logprobs < 0:
            raise ValueError(f"logprobs must be non-negative, got {self.logprobs}.")
        if self.prompt_logprobs is not None and self.prompt_logprobs < 0:
            raise ValueError(f"prompt_logprobs must be non-negative, got {self.prompt_logprobs}.")
        if self.temperature < _SAMPLING_EPS:  # zero temperature means greedy sampling (sampling_params.py:320-326)
            self.top_p, self.top_k, self.min_p = 1.0, -1, 0.0
            if self.num_samples > 1:  # sampling_params.py:_verify_greedy_sampling
                raise ValueError(f"best_of must be 1 when using greedy sampling, got {self.num_samples}.")

    @property
    def num_samples(self) -> int:
        """Sequences forked from the prompt (the reference folds best_of into n, sampling_params.py:285-297)."""
        return self.best_of if self.best_of is not None else self.n

    @property
    def greedy(self) -> bool:
        return self.temperature < _SAMPLING_EPS

    @property
    def plain_greedy(self) -> bool:
        """Greedy with no logit adjustment at all: what the captured step's arg-max epilogue computes."""
        return (self.greedy and self.presence_penalty == 0.0 and self.frequency_penalty == 0.0
                and self.repetition_penalty == 1.0 and self.min_tokens == 0)
