"""Scheduler configuration of the prefill-only workflow.

Same constructor arguments, defaults, derived values and failure messages as the reference's
PrefillOnlySchedulerConfig (light_vllm/prefill_only/config.py:11-73); written here as a dataclass
whose derived fields are filled in once after construction."""
from dataclasses import dataclass, field
from typing import ClassVar, Optional, Tuple

_DEFAULT_IN_FLIGHT = {"double_buffer": 3}  # every other scheduling mode keeps two steps in flight


@dataclass
class PrefillOnlySchedulerConfig:
    max_model_len: int
    max_num_batched_tokens: Optional[int] = None
    max_num_requests: Optional[int] = None
    max_num_seqs: Optional[int] = None  # alias of max_num_requests; wins when both are given
    max_num_on_the_fly: Optional[int] = None
    scheduling: str = "async"

    supported_scheduling: ClassVar[Tuple[str, ...]] = ("sync", "simple_async", "async", "double_buffer")
    _requests: int = field(init=False, repr=False, default=0)

    def __post_init__(self) -> None:
        if self.max_num_on_the_fly is None:
            self.max_num_on_the_fly = _DEFAULT_IN_FLIGHT.get(self.scheduling, 2)
        self.set_args(self.max_num_batched_tokens, self.max_num_requests, self.max_num_seqs)

    def set_args(self, max_num_batched_tokens: Optional[int] = None, max_num_requests: Optional[int] = None,
                 max_num_seqs: Optional[int] = None) -> None:
        requests = max_num_requests if max_num_seqs is None else max_num_seqs
        tokens = max_num_batched_tokens
        if tokens is None:  # room for a full-length prompt per request
            tokens = self.max_model_len * requests
        self.max_num_requests = self.max_num_seqs = requests
        self.max_num_batched_tokens = tokens
        problems = []
        if tokens < self.max_model_len:
            problems.append(f"max_num_batched_tokens ({tokens}) must be greater than or equal to "
                            f"max_model_len ({self.max_model_len}).")
        if self.max_num_on_the_fly < 2:
            problems.append(f"max_num_on_the_fly {self.max_num_on_the_fly} must be greater than 1")
        if self.scheduling not in self.supported_scheduling:
            problems.append(f"scheduling {self.scheduling} must in {list(self.supported_scheduling)}")
        if problems:
            raise ValueError(problems[0])
