"""Data-parallel replicas: the only multi-GPU strategy of the path.

Every GPU runs an independent engine (own weights, KV cache, block manager, scheduler) in its own
process; requests are sharded over replicas by the front end and **no collective touches the data
path**.  The process group exists for run control only: a barrier around timed regions, the maximum
of a per-rank clock, and the sum of per-rank counters.  The reference's counterpart is its
thread-per-GPU executor (light_vllm/prefill_only/executor/gpu_data_parallelism_executor.py:17-81);
processes instead of threads keep each replica's Python scheduler off a shared GIL.
"""
import os
from typing import List, Optional, Sequence

import torch


class ReplicaGroup:

    def __init__(self, backend: Optional[str] = None, device: Optional[torch.device] = None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world_size = int(os.environ.get("WORLD_SIZE", "1"))
        self.device = device if device is not None else torch.device("cpu")
        self._dist = None
        if self.world_size > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            if not dist.is_initialized():
                backend = backend or ("nccl" if self.device.type == "cuda" else "gloo")
                kw = {"device_id": self.device} if backend == "nccl" else {}
                dist.init_process_group(backend, rank=self.rank, world_size=self.world_size, **kw)
            self._dist = dist
            if dist.get_backend() == "gloo":  # gloo reduces host tensors
                self.device = torch.device("cpu")

    def barrier(self) -> None:
        if self._dist is not None:
            self._dist.barrier()

    def _reduce(self, value: float, op) -> float:
        if self._dist is None:
            return value
        t = torch.tensor([value], dtype=torch.float64, device=self.device)
        self._dist.all_reduce(t, op=op)
        return float(t.item())

    def max(self, value: float) -> float:
        return self._reduce(value, self._dist.ReduceOp.MAX) if self._dist else value

    def sum(self, value: float) -> float:
        return self._reduce(value, self._dist.ReduceOp.SUM) if self._dist else value

    def shutdown(self) -> None:
        if self._dist is not None and self._dist.is_initialized():
            self._dist.barrier()
            self._dist.destroy_process_group()
            self._dist = None


def shard_requests(request_ids: Sequence, world_size: int, rank: int) -> List:
    """Round-robin sharding of a request stream over replicas (what a front end does)."""
    return [r for i, r in enumerate(request_ids) if i % world_size == rank]
