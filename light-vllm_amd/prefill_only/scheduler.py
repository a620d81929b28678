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


class PrefillOnlySchedulingBudget:
    """What one scheduling pass may still admit: prompt tokens and requests.  A request is
    charged once however often it is offered (prefill_only/scheduler.py:14-43 of the reference)."""

    def __init__(self, token_budget: int, max_num_requests: int):
        self.token_budget, self.max_num_requests = token_budget, max_num_requests
        self._charged = {}  # request id -> tokens

    @property
    def num_batched_tokens(self) -> int:
        return sum(self._charged.values())

    @property
    def num_curr_request(self) -> int:
        return len(self._charged)

    def can_schedule(self, *, num_new_tokens: int, num_new_request: int = 1) -> bool:
        assert num_new_tokens != 0 and num_new_request != 0
        tokens_ok = self.num_batched_tokens + num_new_tokens <= self.token_budget
        return tokens_ok and self.num_curr_request + num_new_request <= self.max_num_requests

    def add_num_batched_tokens(self, req_id: str, num_batched_tokens: int) -> None:
        self._charged.setdefault(req_id, num_batched_tokens)


class PrefillOnlyScheduler:
    """Waiting queue + the set of live request ids.  Aborts are lazy: an aborted id is dropped
    when the queue reaches it, or filtered out of the outputs of a step already in flight
    (light_vllm/core/scheduler.py:14-89)."""
    support_scheduling = ["sync_scheduling", "async_scheduling"]

    def __init__(self, scheduler_config, request_processor: Optional[Callable[[Request], SchedulableRequest]] = None):
        self.scheduler_config = scheduler_config
        self.request_processor = request_processor
        self.waiting: Deque[Request] = deque()
        self.requests: Set[str] = set()
        self.aborted_requests: Set[str] = set()

    def add_request(self, request: Request) -> None:
        rid = request.request_id
        if rid in self.requests or rid in self.aborted_requests:
            return  # duplicate id: the reference warns and drops it
        self.requests.add(rid)
        self.waiting.append(request)

    def abort_request(self, request_id: Union[str, Iterable[str]]) -> None:
        ids = {request_id} if isinstance(request_id, str) else set(request_id)
        self.aborted_requests |= ids
        self.requests -= ids

    def remove_abort_request(self, request_outputs: List[PrefillOnlyRequestOutput]) -> List[PrefillOnlyRequestOutput]:
        hit = self.aborted_requests.intersection(r.request_id for r in request_outputs)
        if hit:
            self.aborted_requests -= hit
            request_outputs = [r for r in request_outputs if r.request_id not in hit]
        return request_outputs

    def has_unfinished_requests(self) -> bool:
        return bool(self.requests)

    def get_num_unfinished_requests(self) -> int:
        return len(self.requests)

    def free_finished_request(self, request_outputs) -> None:
        self.requests.difference_update(r.request_id for r in request_outputs if r.finished)

    def schedule(self) -> PrefillOnlySchedulerOutput:
        """One pass in arrival order (prefill_only/scheduler.py:57-100 of the reference): stop at the
        first prompt that does not fit the token / request budget; prompts over max_model_len
        are returned as ignored and forgotten."""
        cfg = self.scheduler_config
        budget = PrefillOnlySchedulingBudget(cfg.max_num_batched_tokens, cfg.max_num_seqs)
        admitted: List[SchedulableRequest] = []
        too_long: List[SchedulableRequest] = []
        queue = self.waiting
        while queue:
            head = queue[0]
            rid = head.request_id
            if rid in self.aborted_requests:
                self.aborted_requests.discard(rid)
                queue.popleft()
                continue
            if not isinstance(head, SchedulableRequest):  # tokenise once, keep the result queued
                head = queue[0] = self.request_processor(head)
            n = head.num_new_tokens
            if n > cfg.max_model_len:
                queue.popleft()
                self.requests.remove(rid)
                too_long.append(head)
            elif budget.can_schedule(num_new_tokens=n):
                queue.popleft()
                budget.add_num_batched_tokens(rid, n)
                admitted.append(head)
            else:
                break
        return PrefillOnlySchedulerOutput(scheduled_requests=admitted, ignored_requests=too_long)
