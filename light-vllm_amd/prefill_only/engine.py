"""Encode-only engine: scheduler -> model input (tokens of the scheduled prompts back to back +
seq_lens metadata) -> one forward pass -> per-request embedding.

Step structure of light_vllm/core/llm_engine.py (sync step :119-130, async step :132-176) with the
prefill-only executor's overlap of host->device copies, compute and device->host copies
(prefill_only/executor/gpu_executor.py:109-262: three streams) obtained the same way as in the
decoding engine of this package: every step in flight owns a stream, so one step's copies run beside
another step's kernels.  One engine = one GPU; `engine.replicas` shards requests over GPUs."""
import queue
import threading
import time
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch

from .config import PrefillOnlySchedulerConfig
from .model import EncoderConfig, EncoderModel
from .scheduler import (PrefillOnlyRequestOutput, PrefillOnlyScheduler, PrefillOnlySchedulerOutput,
                        SchedulableRequest)


@dataclass
class ModelInput:
    """prefill_only/processor/model_input_builder.py:17-52: flattened tokens + positions + metadata."""
    input_ids: torch.Tensor
    positions: torch.Tensor
    attn_metadata: object
    seq_lens: List[int]

    def to(self, device, non_blocking=True):
        self.input_ids = self.input_ids.to(device, non_blocking=non_blocking)
        self.positions = self.positions.to(device, non_blocking=non_blocking)
        self.attn_metadata.to(device, non_blocking=non_blocking)
        return self


class PrefillOnlyWorker:
    """The model on its GPU, the per-step input building and the step itself
    (prefill_only/worker/gpu_worker.py + runner/model_runner.py of the reference).  `pooling`: "cls" (bge-m3's dense
    embedding: first token, L2-normalised, fp32), "mean", or "last_hidden_states" (the whole [len, hidden] block per
    request)."""

    def __init__(self, model_config: EncoderConfig, device: str = "cuda:0", pooling: str = "cls", seed: int = 0):
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.model_config = model_config
        self.model = EncoderModel(model_config, device, seed)
        self.builder = self.model.backend.make_metadata_builder()
        assert pooling in ("cls", "mean", "last_hidden_states")
        self.pooling = pooling

    def build(self, token_ids: torch.Tensor, seq_lens: List[int]) -> ModelInput:
        """prefill_only/processor/model_input_builder.py:17-52: the scheduled prompts back to back, positions
        restarting at every prompt, the attention metadata from the lengths."""
        pos = torch.cat([torch.arange(n, dtype=torch.long) for n in seq_lens]) if seq_lens else token_ids.new_empty(0)
        if torch.cuda.is_available():
            token_ids, pos = token_ids.pin_memory(), pos.pin_memory()
        return ModelInput(token_ids, pos, self.builder(seq_lens=list(seq_lens)), list(seq_lens))

    @torch.inference_mode()
    def execute(self, mi: ModelInput) -> torch.Tensor:
        """On the current stream: H2D, forward, pooling, D2H into pinned memory (not yet complete)."""
        mi.to(self.device)
        hidden = self.model.forward(mi.input_ids, mi.positions, mi.attn_metadata)
        if self.pooling == "last_hidden_states":
            out = hidden
        else:
            start = mi.attn_metadata.seq_start_loc[:-1].long()
            if self.pooling == "cls":
                pooled = hidden[start].float()
            else:
                csum = torch.cat([hidden.new_zeros(1, hidden.shape[1], dtype=torch.float32),
                                  hidden.float().cumsum(0)])
                end = mi.attn_metadata.seq_start_loc[1:].long()
                pooled = (csum[end] - csum[start]) / (end - start).unsqueeze(1)
            out = torch.nn.functional.normalize(pooled, dim=-1)
        host = torch.empty(out.shape, dtype=out.dtype, pin_memory=True)
        host.copy_(out, non_blocking=True)
        return host


class PrefillOnlyStepExecutor:
    """A worker's execute loop over queues it shares with other workers (the data-parallel front end,
    dp_executor.py; reference: Executor.async_execute_loop, prefill_only/executor/gpu_executor.py:109-262).  `slots`
    steps are in flight on their own streams: one step's copies run beside another's kernels.  A slot is taken
    BEFORE an item is taken off the shared queue, so a worker never holds a step it cannot start."""

    def __init__(self, worker: PrefillOnlyWorker, slots: int = 2):
        self.worker = worker
        self.slots = max(1, slots)
        self.streams = [torch.cuda.Stream(worker.device) for _ in range(self.slots)]

    def execute_loop(self, executor_in, executor_out, rank: int = 0) -> None:
        from .dp_executor import ExecuteOutput
        torch.cuda.set_device(self.worker.device)
        free: "queue.Queue" = queue.Queue()
        for i in range(self.slots):
            free.put(i)
        done_q: "queue.Queue" = queue.Queue()

        def done_loop():
            torch.cuda.set_device(self.worker.device)
            while True:
                item = done_q.get()
                if item is None:
                    return
                slot, ev, step_id, host, t0 = item
                try:
                    ev.synchronize()
                    out = ExecuteOutput(step_id, rank, host, execute_begin_ts=t0, execute_end_ts=time.time())
                except Exception as e:
                    out = ExecuteOutput(step_id, rank, None, error=repr(e))
                free.put(slot)
                executor_out.put(out)

        waiter = threading.Thread(target=done_loop, daemon=True)
        waiter.start()
        while True:
            slot = free.get()
            item = executor_in.get()
            if item is None:
                free.put(slot)
                break
            t0 = time.time()
            try:
                with torch.cuda.stream(self.streams[slot]):
                    host = self.worker.execute(self.worker.build(item.token_ids, item.seq_lens))
                    ev = torch.cuda.Event()
                    ev.record(self.streams[slot])
                done_q.put((slot, ev, item.step_id, host, t0))
            except Exception as e:
                free.put(slot)
                executor_out.put(ExecuteOutput(item.step_id, rank, None, error=repr(e)))
        done_q.put(None)
        waiter.join(timeout=30)


class PrefillOnlyEngine:
    """One engine = one GPU: scheduler + worker in one process (`dp_executor.DataParallelEncodeEngine` puts N worker
    processes behind one scheduler).  `pooling`: see PrefillOnlyWorker."""

    def __init__(self, model_config: EncoderConfig, scheduler_config: PrefillOnlySchedulerConfig,
                 device: str = "cuda:0", pooling: str = "cls", seed: int = 0):
        self.worker = PrefillOnlyWorker(model_config, device, pooling, seed)
        self.device = self.worker.device
        self.model_config, self.scheduler_config = model_config, scheduler_config
        self.model = self.worker.model
        self.scheduler = PrefillOnlyScheduler(scheduler_config)
        self.pooling = pooling
        self.num_slots = (max(1, scheduler_config.max_num_on_the_fly)
                          if scheduler_config.scheduling in ("async", "double_buffer") else 1)
        self.streams = [torch.cuda.Stream(self.device) for _ in range(self.num_slots)]
        self.free_slots: "queue.Queue" = queue.Queue()
        for i in range(self.num_slots):
            self.free_slots.put(i)
        self.executor_in: "queue.Queue" = queue.Queue()
        self.executor_out: "queue.Queue" = queue.Queue()
        self._done_q: "queue.Queue" = queue.Queue()
        self._threads: List[threading.Thread] = []
        self.num_on_the_fly = 0

    # ---- requests ----
    def add_request(self, request_id: str, prompt_token_ids: List[int]) -> None:
        self.scheduler.add_request(SchedulableRequest(request_id, time.time(), list(prompt_token_ids)))

    def abort_request(self, request_id) -> None:
        self.scheduler.abort_request(request_id)

    def has_unfinished_requests(self) -> bool:
        return self.scheduler.has_unfinished_requests()

    # ---- one step ----
    def _build(self, sched: PrefillOnlySchedulerOutput) -> ModelInput:
        toks = torch.tensor([t for r in sched.scheduled_requests for t in r.prompt_token_ids], dtype=torch.long)
        return self.worker.build(toks, [r.num_new_tokens for r in sched.scheduled_requests])

    def _execute(self, mi: ModelInput) -> torch.Tensor:
        return self.worker.execute(mi)

    def _process(self, sched: PrefillOnlySchedulerOutput, host: Optional[torch.Tensor], lens: List[int]):
        outs: List[PrefillOnlyRequestOutput] = []
        off = 0
        for i, r in enumerate(sched.scheduled_requests):
            if self.pooling == "last_hidden_states":
                o = host[off:off + lens[i]]
                off += lens[i]
            else:
                o = host[i]
            outs.append(PrefillOnlyRequestOutput(r.request_id, o, r.prompt_token_ids, True, r.arrival_time))
        for r in sched.ignored_requests:
            outs.append(PrefillOnlyRequestOutput(r.request_id, None, r.prompt_token_ids, True, r.arrival_time))
        outs = self.scheduler.remove_abort_request(outs)
        self.scheduler.free_finished_request(outs)
        return outs

    def step(self) -> List[PrefillOnlyRequestOutput]:
        sched = self.scheduler.schedule()
        if sched.is_empty():
            return self._process(sched, None, []) if sched.ignored_requests else []
        mi = self._build(sched)
        with torch.cuda.stream(self.streams[0]):
            host = self._execute(mi)
        self.streams[0].synchronize()
        return self._process(sched, host, mi.seq_lens)

    # ---- steps in flight ----
    def _execute_loop(self) -> None:
        torch.cuda.set_device(self.device)
        while True:
            item = self.executor_in.get()
            if item is None:
                self._done_q.put(None)
                return
            sched, mi = item
            slot = self.free_slots.get()
            try:
                with torch.cuda.stream(self.streams[slot]):
                    host = self._execute(mi)
                    ev = torch.cuda.Event()
                    ev.record(self.streams[slot])
                self._done_q.put((slot, ev, sched, host, mi.seq_lens))
            except Exception as e:
                self.free_slots.put(slot)
                e.failed_request_ids = [r.request_id for r in sched.scheduled_requests]
                self.executor_out.put(e)

    def _done_loop(self) -> None:
        torch.cuda.set_device(self.device)
        while True:
            item = self._done_q.get()
            if item is None:
                return
            slot, ev, sched, host, lens = item
            ev.synchronize()
            self.free_slots.put(slot)
            self.executor_out.put((sched, host, lens))

    def async_step(self) -> List[PrefillOnlyRequestOutput]:
        if not self._threads:
            self._threads = [threading.Thread(target=self._execute_loop, daemon=True),
                             threading.Thread(target=self._done_loop, daemon=True)]
            for t in self._threads:
                t.start()
        outs: List[PrefillOnlyRequestOutput] = []
        while self.num_on_the_fly < self.scheduler_config.max_num_on_the_fly:
            sched = self.scheduler.schedule()
            if sched.ignored_requests:
                outs.extend(self._process(PrefillOnlySchedulerOutput([], sched.ignored_requests), None, []))
            if sched.is_empty():
                break
            self.executor_in.put((sched, self._build(sched)))
            self.num_on_the_fly += 1
        if self.num_on_the_fly == 0:
            return outs
        item = self.executor_out.get()
        self.num_on_the_fly -= 1  # the failed step is no longer in flight either: encode()'s loop must end
        if isinstance(item, Exception):  # its requests will never produce an output: they leave the books
            self.scheduler.requests.difference_update(getattr(item, "failed_request_ids", ()))
            raise item
        sched, host, lens = item
        return outs + self._process(PrefillOnlySchedulerOutput(sched.scheduled_requests, []), host, lens)

    def shutdown(self) -> None:
        if self._threads:
            self.executor_in.put(None)
            for t in self._threads:
                t.join(timeout=5)
            self._threads = []

    def encode(self, prompts: List[List[int]], use_async: Optional[bool] = None) -> Dict[str, torch.Tensor]:
        """Convenience: run `prompts` to completion, return {index: embedding}."""
        for i, p in enumerate(prompts):
            self.add_request(str(i), p)
        use_async = self.num_slots > 1 if use_async is None else use_async
        res: Dict[str, torch.Tensor] = {}
        while self.has_unfinished_requests() or self.num_on_the_fly > 0:
            for o in (self.async_step() if use_async else self.step()):
                res[o.request_id] = o.outputs
        return res
