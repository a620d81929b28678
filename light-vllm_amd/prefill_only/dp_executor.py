"""Queue-sharing data-parallel front end of the prefill-only workflow.

What the reference does (light_vllm/prefill_only/executor/gpu_data_parallelism_executor.py:17-81,
prefill_only/workflow.py:31-41): ONE engine -- one scheduler, one `executor_in`, one `executor_out` -- and
`data_parallel_size` workers, each owning a GPU and a full copy of the model, all pulling `ExecuteInput`s from the
same queue; `max_num_on_the_fly` is multiplied by the number of workers so that the scheduler keeps every worker's
pipeline fed.  A worker that finishes early simply takes the next step: a slow step (long prompts) never idles the
other GPUs the way a static shard of the request list does.

What is different here: the workers are PROCESSES, not threads.  Every worker runs its own Python (input
building, launches, completion polling) off the front end's GIL, which is what one process per GPU means on an
8-GPU MI355X node; the queues are `multiprocessing` queues; what travels is small (token ids in, pooled embeddings
out).  Processes are started with the "spawn" method BEFORE the front-end process has touched a GPU -- the front end
never does: it owns the scheduler only -- and never by exec.  No collective, no RCCL: replicas share nothing but the
two queues.

    engine = DataParallelEncodeEngine(EncoderConfig.bge_m3(), PrefillOnlySchedulerConfig(...), data_parallel_size=8)
    embeddings = engine.encode(prompts)          # {request id: tensor}
    engine.shutdown()

`worker_factory`: a picklable callable `(rank) -> executor` whose `execute_loop(executor_in, executor_out, rank)`
serves the queue; the default builds the gfx950 worker (model on `cuda:<rank>`, this package's HIP kernels) and fails
loudly without a GPU.  Tests of the queue discipline inject a host-only stand-in (tests/test_dp_executor.py).
"""
import atexit
import queue
import time
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence

import torch

from .config import PrefillOnlySchedulerConfig
from .scheduler import (PrefillOnlyRequestOutput, PrefillOnlyScheduler, PrefillOnlySchedulerOutput,
                        SchedulableRequest)


def _pack(t: Optional[torch.Tensor]):
    """A CPU tensor as (dtype name, shape, bytes): pickled by value.  (torch registers shared-memory reductions for
    tensors on every multiprocessing queue -- one file descriptor per tensor the receiver keeps alive, which a
    front end that hands embeddings to its caller would run out of.)"""
    if t is None:
        return None
    t = t.detach().contiguous().cpu()
    return (str(t.dtype).split(".")[1], tuple(t.shape), bytes(t.view(torch.uint8).reshape(-1).numpy()))


def _unpack(p) -> Optional[torch.Tensor]:
    if p is None:
        return None
    name, shape, raw = p
    dt = getattr(torch, name)
    if len(raw) == 0:
        return torch.empty(shape, dtype=dt)
    return torch.frombuffer(bytearray(raw), dtype=torch.uint8).view(dt).reshape(shape)


@dataclass
class ExecuteInput:
    """One scheduled step on its way to whichever worker takes it (core/schema/execute_io.py of the reference:
    the model input travels, the scheduler's bookkeeping stays with the front end under `step_id`)."""
    step_id: int
    token_ids: torch.Tensor          # int64 [sum(seq_lens)], the scheduled prompts back to back
    seq_lens: List[int]

    def __getstate__(self):
        return (self.step_id, _pack(self.token_ids), self.seq_lens)

    def __setstate__(self, st):
        self.step_id, self.token_ids, self.seq_lens = st[0], _unpack(st[1]), st[2]


@dataclass
class ExecuteOutput:
    step_id: int
    rank: int
    outputs: Optional[torch.Tensor]  # pooled: [num requests, hidden] fp32; "last_hidden_states": [tokens, hidden]
    error: Optional[str] = None
    execute_begin_ts: float = 0.0
    execute_end_ts: float = 0.0

    def __getstate__(self):
        return (self.step_id, self.rank, _pack(self.outputs), self.error, self.execute_begin_ts, self.execute_end_ts)

    def __setstate__(self, st):
        self.step_id, self.rank, self.error, self.execute_begin_ts, self.execute_end_ts = st[0], st[1], st[3], st[4], st[5]
        self.outputs = _unpack(st[2])


class GPUWorkerFactory:
    """Builds a worker's executor inside the worker process: the encoder on its GPU plus the step pipeline of
    prefill_only/executor/gpu_executor.py:109-262 (`slots` steps in flight on their own streams)."""

    def __init__(self, model_config, pooling: str = "cls", seed: int = 0, slots: int = 2,
                 devices: Optional[Sequence[int]] = None):
        self.model_config, self.pooling, self.seed, self.slots = model_config, pooling, seed, slots
        self.devices = list(devices) if devices is not None else None

    def __call__(self, rank: int):
        from .engine import PrefillOnlyStepExecutor, PrefillOnlyWorker
        index = self.devices[rank] if self.devices is not None else rank
        if not torch.cuda.is_available():
            raise RuntimeError("the prefill-only GPU worker needs an MI355X (no HIP device visible)")
        worker = PrefillOnlyWorker(self.model_config, f"cuda:{index}", self.pooling, self.seed)
        return PrefillOnlyStepExecutor(worker, self.slots)


def _worker_main(rank: int, factory: Callable, executor_in, executor_out) -> None:
    """Target of a worker process (gpu_data_parallelism_executor.py:41-62: create the worker, load the model, run the
    execute loop until the `None` that shuts it down)."""
    try:
        executor = factory(rank)
    except Exception as e:  # the front end must not wait for ever on a worker that never came up
        executor_out.put(ExecuteOutput(-1, rank, None, error=f"worker {rank} failed to start: {e!r}"))
        return
    executor_out.put(ExecuteOutput(-1, rank, None))  # ready
    executor.execute_loop(executor_in, executor_out, rank)


class DataParallelEncodeEngine:
    """The prefill-only engine with `data_parallel_size` worker processes behind one scheduler."""

    def __init__(self, model_config, scheduler_config: PrefillOnlySchedulerConfig, data_parallel_size: int,
                 pooling: str = "cls", seed: int = 0, devices: Optional[Sequence[int]] = None,
                 worker_factory: Optional[Callable] = None, start_timeout_s: float = 600.0):
        assert data_parallel_size > 0
        # workflow.py:33-36: the data-parallel executor serves the modes that keep steps in flight
        assert scheduler_config.scheduling in ("async", "double_buffer"), scheduler_config.scheduling
        self.data_parallel_size = data_parallel_size
        self.scheduler_config = scheduler_config
        self.per_worker_on_the_fly = scheduler_config.max_num_on_the_fly
        # workflow.py:37-38: max_num_on_the_fly *= data_parallel_size
        self.max_num_on_the_fly = scheduler_config.max_num_on_the_fly * data_parallel_size
        self.scheduler = PrefillOnlyScheduler(scheduler_config)
        self.pooling = pooling
        self.factory = worker_factory or GPUWorkerFactory(model_config, pooling, seed,
                                                          slots=self.per_worker_on_the_fly, devices=devices)
        self.start_timeout_s = start_timeout_s
        self._ctx = torch.multiprocessing.get_context("spawn")
        self.executor_in = self._ctx.Queue()
        self.executor_out = self._ctx.Queue()
        self.procs: Optional[list] = None
        self.num_on_the_fly = 0
        self._next_step = 0
        self._steps: Dict[int, PrefillOnlySchedulerOutput] = {}
        self.steps_by_rank: Dict[int, int] = {}      # how many steps each worker took (work sharing, observable)
        self.step_timeout_s = 600.0

    # ---- workers ----
    def ensure_start_execute_loop(self) -> None:
        if self.procs is not None:
            return
        self.procs = []
        for rank in range(self.data_parallel_size):
            p = self._ctx.Process(target=_worker_main, args=(rank, self.factory, self.executor_in, self.executor_out),
                                  daemon=True)
            p.start()
            self.procs.append(p)
        atexit.register(self.shutdown)
        ready = 0
        deadline = time.time() + self.start_timeout_s
        while ready < self.data_parallel_size:
            try:
                msg = self.executor_out.get(timeout=max(0.1, deadline - time.time()))
            except queue.Empty:
                self.shutdown()
                raise RuntimeError(f"only {ready} of {self.data_parallel_size} workers came up "
                                   f"within {self.start_timeout_s:.0f} s")
            if msg.error is not None:
                self.shutdown()
                raise RuntimeError(msg.error)
            ready += 1

    def shutdown(self) -> None:
        if self.procs is None:
            return
        for _ in self.procs:
            self.executor_in.put(None)
        for p in self.procs:
            p.join(timeout=10)
            if p.is_alive():
                p.terminate()  # the exact process this engine started
        self.procs = None
        try:
            atexit.unregister(self.shutdown)
        except Exception:
            pass

    # ---- requests ----
    def add_request(self, request_id: str, prompt_token_ids: List[int]) -> None:
        self.scheduler.add_request(SchedulableRequest(request_id, time.time(), list(prompt_token_ids)))

    def abort_request(self, request_id) -> None:
        self.scheduler.abort_request(request_id)

    def has_unfinished_requests(self) -> bool:
        return self.scheduler.has_unfinished_requests()

    # ---- the step (core/llm_engine.py:132-176 over the shared queues) ----
    def _finish(self, sched: PrefillOnlySchedulerOutput, out: Optional[torch.Tensor],
                lens: List[int]) -> List[PrefillOnlyRequestOutput]:
        outs: List[PrefillOnlyRequestOutput] = []
        off = 0
        for i, r in enumerate(sched.scheduled_requests):
            if self.pooling == "last_hidden_states":
                o = out[off:off + lens[i]]
                off += lens[i]
            else:
                o = out[i]
            outs.append(PrefillOnlyRequestOutput(r.request_id, o, r.prompt_token_ids, True, r.arrival_time))
        for r in sched.ignored_requests:
            outs.append(PrefillOnlyRequestOutput(r.request_id, None, r.prompt_token_ids, True, r.arrival_time))
        outs = self.scheduler.remove_abort_request(outs)
        self.scheduler.free_finished_request(outs)
        return outs

    def step(self) -> List[PrefillOnlyRequestOutput]:
        self.ensure_start_execute_loop()
        outs: List[PrefillOnlyRequestOutput] = []
        while self.num_on_the_fly < self.max_num_on_the_fly:
            sched = self.scheduler.schedule()
            if sched.ignored_requests:
                outs.extend(self._finish(PrefillOnlySchedulerOutput([], sched.ignored_requests), None, []))
            if sched.is_empty():
                break
            lens = [r.num_new_tokens for r in sched.scheduled_requests]
            toks = torch.tensor([t for r in sched.scheduled_requests for t in r.prompt_token_ids], dtype=torch.long)
            sid = self._next_step
            self._next_step += 1
            self._steps[sid] = PrefillOnlySchedulerOutput(sched.scheduled_requests, [])
            self.executor_in.put(ExecuteInput(sid, toks, lens))
            self.num_on_the_fly += 1
        if self.num_on_the_fly == 0:
            return outs
        # Wake up every second and look at the workers: one that died on a GPU fault or an out-of-memory kill after
        # it took a step off the shared queue will never answer, and the queue does not say which step it held --
        # so every outstanding step is failed at once, its requests leave the scheduler's books (as on the error
        # path below) and the caller hears about it within a second, not after `step_timeout_s`.
        res: Optional[ExecuteOutput] = None
        waited = 0.0
        while res is None:
            try:
                res = self.executor_out.get(timeout=min(1.0, self.step_timeout_s))
            except queue.Empty:
                waited += min(1.0, self.step_timeout_s)
                dead = [(i, p.exitcode) for i, p in enumerate(self.procs or []) if not p.is_alive()]
                if dead or waited >= self.step_timeout_s:
                    lost = sorted(self._steps)
                    for sid in lost:
                        sched = self._steps.pop(sid)
                        self.scheduler.requests.difference_update(r.request_id for r in sched.scheduled_requests)
                    self.num_on_the_fly = 0
                    if dead:
                        raise RuntimeError(f"worker(s) died (rank, exit code): {dead}; steps {lost} failed and their "
                                           f"requests were dropped")
                    raise RuntimeError(f"no step completed within {self.step_timeout_s:.0f} s; steps {lost} failed and "
                                       f"their requests were dropped")
        self.num_on_the_fly -= 1
        sched = self._steps.pop(res.step_id)
        if res.error is not None:  # the step's requests will never produce an output: they leave the books
            self.scheduler.requests.difference_update(r.request_id for r in sched.scheduled_requests)
            raise RuntimeError(f"worker {res.rank}: {res.error}")
        self.steps_by_rank[res.rank] = self.steps_by_rank.get(res.rank, 0) + 1
        return outs + self._finish(sched, res.outputs, [r.num_new_tokens for r in sched.scheduled_requests])

    def encode(self, prompts: List[List[int]]) -> Dict[str, torch.Tensor]:
        for i, p in enumerate(prompts):
            self.add_request(str(i), p)
        res: Dict[str, torch.Tensor] = {}
        while self.has_unfinished_requests() or self.num_on_the_fly > 0:
            for o in self.step():
                res[o.request_id] = o.outputs
        return res
