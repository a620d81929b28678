"""Request schemas and the scheduler of the prefill-only workflow.

Scheduler: light_vllm/core/scheduler.py:14-89 (request set, aborts, free_finished_request) +
light_vllm/prefill_only/scheduler.py:14-100 (one pass over the waiting queue under a token budget
and a request budget; prompts longer than max_model_len are ignored).  Schemas:
light_vllm/core/schema/engine_io.py and prefill_only/schema/engine_io.py:13-53."""
from collections import deque
from dataclasses import dataclass, field
from typing import Callable, Deque, Iterable, List, Optional, Set, Union

import torch


@dataclass
class Request:
    request_id: str
    arrival_time: float = 0.0


@dataclass
class SchedulableRequest(Request):
    prompt_token_ids: List[int] = field(default_factory=list)

    @property
    def num_new_tokens(self) -> int:
        return len(self.prompt_token_ids)


@dataclass
class PrefillOnlySchedulerOutput:
    scheduled_requests: List[SchedulableRequest]
    ignored_requests: List[SchedulableRequest]

    def is_empty(self) -> bool:
        return not self.scheduled_requests


class PrefillOnlyRequestOutput:

    def __init__(self, request_id: str, outputs: Optional[torch.Tensor], prompt_token_ids: List[int],
                 finished: bool, arrival_time: float = 0.0):
        self.request_id = request_id
        self.prompt_token_ids = prompt_token_ids
        self.finished = finished
        self.outputs = outputs
        self.arrival_time = arrival_time

    def __repr__(self):
        return (f"PrefillOnlyRequestOutput(request_id='{self.request_id}', outputs={self.outputs!r}, "
                f"prompt_token_ids={self.prompt_token_ids}, finished={self.finished})")


@dataclass
class PrefillOnlySchedulingBudget:
    token_budget: int
    max_num_requests: int
    _curr_requests: Set[str] = field(default_factory=set)
    _num_batched_tokens: int = 0

    def can_schedule(self, *, num_new_tokens: int, num_new_request: int = 1) -> bool:
        assert num_new_tokens != 0
        assert num_new_request != 0
        return (self.num_batched_tokens + num_new_tokens <= self.token_budget
                and self.num_curr_request + num_new_request <= self.max_num_requests)

    def add_num_batched_tokens(self, req_id: str, num_batched_tokens: int) -> None:
        if req_id in self._curr_requests:
            return
        self._curr_requests.add(req_id)
        self._num_batched_tokens += num_batched_tokens

    @property
    def num_batched_tokens(self) -> int:
        return self._num_batched_tokens

    @property
    def num_curr_request(self) -> int:
        return len(self._curr_requests)


class PrefillOnlyScheduler:
    support_scheduling = ["sync_scheduling", "async_scheduling"]

    def __init__(self, scheduler_config, request_processor: Optional[Callable[[Request], SchedulableRequest]] = None):
        self.scheduler_config = scheduler_config
        self.request_processor = request_processor
        self.waiting: Deque[Request] = deque()
        self.requests: Set[str] = set()
        self.aborted_requests: Set[str] = set()

    # ---- light_vllm/core/scheduler.py ----
    def add_request(self, request: Request) -> None:
        if request.request_id in self.requests or request.request_id in self.aborted_requests:
            return  # request_id conflict (the reference logs a warning)
        self.waiting.append(request)
        self.requests.add(request.request_id)

    def abort_request(self, request_id: Union[str, Iterable[str]]) -> None:
        if isinstance(request_id, str):
            request_id = (request_id,)
        request_ids = set(request_id)
        self.requests -= request_ids
        self.aborted_requests |= request_ids

    def remove_abort_request(self, request_outputs: List[PrefillOnlyRequestOutput]) -> List[PrefillOnlyRequestOutput]:
        if not self.aborted_requests:
            return request_outputs
        need_abort = self.aborted_requests & {r.request_id for r in request_outputs}
        if not need_abort:
            return request_outputs
        self.aborted_requests -= need_abort
        return [r for r in request_outputs if r.request_id not in need_abort]

    def has_unfinished_requests(self) -> bool:
        return len(self.requests) != 0

    def get_num_unfinished_requests(self) -> int:
        return len(self.requests)

    def free_finished_request(self, request_outputs) -> None:
        self.requests -= {r.request_id for r in request_outputs if r.finished}

    # ---- light_vllm/prefill_only/scheduler.py:57-100 ----
    def schedule(self) -> PrefillOnlySchedulerOutput:
        budget = PrefillOnlySchedulingBudget(token_budget=self.scheduler_config.max_num_batched_tokens,
                                             max_num_requests=self.scheduler_config.max_num_seqs)
        waiting_queue = self.waiting
        scheduled_requests: List[SchedulableRequest] = []
        ignored_requests: List[SchedulableRequest] = []
        while waiting_queue:
            request = waiting_queue[0]
            if request.request_id in self.aborted_requests:
                self.aborted_requests.remove(request.request_id)
                waiting_queue.popleft()
                continue
            if not isinstance(request, SchedulableRequest):
                request = self.request_processor(request)
                waiting_queue[0] = request
            num_new_tokens = request.num_new_tokens
            if num_new_tokens > self.scheduler_config.max_model_len:
                self.requests.remove(request.request_id)
                waiting_queue.popleft()
                ignored_requests.append(request)
                continue
            if not budget.can_schedule(num_new_tokens=num_new_tokens):
                break
            budget.add_num_batched_tokens(request.request_id, num_new_tokens)
            waiting_queue.popleft()
            scheduled_requests.append(request)
        return PrefillOnlySchedulerOutput(scheduled_requests=scheduled_requests, ignored_requests=ignored_requests)
