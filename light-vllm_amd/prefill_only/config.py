"""PrefillOnlySchedulerConfig: light_vllm/prefill_only/config.py:11-73 (same arguments, defaults and
checks)."""
from typing import Optional


class PrefillOnlySchedulerConfig:
    supported_scheduling = ["sync", "simple_async", "async", "double_buffer"]

    def __init__(self, max_model_len: int, max_num_batched_tokens: Optional[int] = None,
                 max_num_requests: Optional[int] = None, max_num_seqs: Optional[int] = None,
                 max_num_on_the_fly: Optional[int] = None, scheduling: str = "async") -> None:
        self.max_model_len = max_model_len
        self.max_num_requests: int = 0
        self.max_num_batched_tokens: int = 0
        self.scheduling = scheduling
        if max_num_on_the_fly is None:
            self.max_num_on_the_fly = 3 if scheduling == "double_buffer" else 2
        else:
            self.max_num_on_the_fly = max_num_on_the_fly
        self.set_args(max_num_batched_tokens, max_num_requests, max_num_seqs)

    def set_args(self, max_num_batched_tokens: Optional[int] = None, max_num_requests: Optional[int] = None,
                 max_num_seqs: Optional[int] = None) -> None:
        self.max_num_requests = max_num_seqs if max_num_seqs is not None else max_num_requests
        if max_num_batched_tokens is not None:
            self.max_num_batched_tokens = max_num_batched_tokens
        else:
            self.max_num_batched_tokens = self.max_model_len * self.max_num_requests
        self._verify_args()

    def _verify_args(self) -> None:
        if self.max_num_batched_tokens < self.max_model_len:
            raise ValueError(f"max_num_batched_tokens ({self.max_num_batched_tokens}) must be greater than or "
                             f"equal to max_model_len ({self.max_model_len}).")
        if self.max_num_on_the_fly < 2:
            raise ValueError(f"max_num_on_the_fly {self.max_num_on_the_fly} must be greater than 1")
        if self.scheduling not in self.supported_scheduling:
            raise ValueError(f"scheduling {self.scheduling} must in {self.supported_scheduling}")

    @property
    def max_num_seqs(self) -> int:
        return self.max_num_requests
