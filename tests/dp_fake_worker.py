"""Host-only stand-in for a prefill-only GPU worker, for the queue-discipline tests of the data-parallel front end
(tests/test_dp_executor.py).  NOT a CPU path of the product: it computes a checksum, not an embedding."""
import time

import torch


class FakeExecutor:
    def __init__(self, rank, us_per_token, fail_token, die_token=None):
        self.rank, self.us_per_token, self.fail_token, self.die_token = rank, us_per_token, fail_token, die_token

    def execute_loop(self, executor_in, executor_out, rank=0):
        from light_vllm_amd.prefill_only.dp_executor import ExecuteOutput
        while True:
            item = executor_in.get()
            if item is None:
                return
            t0 = time.time()
            if self.die_token is not None and bool((item.token_ids == self.die_token).any()):
                import os
                os._exit(17)  # the process is gone with the step it took: no answer will ever come (a GPU fault, an OOM kill)
            try:
                if self.fail_token is not None and bool((item.token_ids == self.fail_token).any()):
                    raise ValueError(f"poisoned step {item.step_id}")
                time.sleep(self.us_per_token * 1e-6 * int(item.token_ids.numel()))
                rows, off = [], 0
                for n in item.seq_lens:
                    t = item.token_ids[off:off + n]
                    rows.append([float(t.sum()), float(n), float(t[0]), float(rank)])
                    off += n
                out = torch.tensor(rows, dtype=torch.float32)
                executor_out.put(ExecuteOutput(item.step_id, rank, out, execute_begin_ts=t0, execute_end_ts=time.time()))
            except Exception as e:
                executor_out.put(ExecuteOutput(item.step_id, rank, None, error=repr(e)))


class FakeWorkerFactory:
    """rank r sleeps us_per_token[r] microseconds per token of a step; `fail_token` in a step raises in the worker;
    `die_token` in a step kills the worker process on the spot; `fail_start` = rank whose construction raises."""

    def __init__(self, us_per_token, fail_token=None, fail_start=None, die_token=None):
        self.us_per_token, self.fail_token, self.fail_start = list(us_per_token), fail_token, fail_start
        self.die_token = die_token

    def __call__(self, rank):
        if self.fail_start == rank:
            raise RuntimeError("no such device")
        return FakeExecutor(rank, self.us_per_token[rank], self.fail_token, self.die_token)
