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


# ---- CPU placement of a replica (N Python schedulers on one host are the scaling risk) ----
def parse_cpulist(text: str) -> List[int]:
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the format of /sys/devices/system/node/node*/cpulist)."""
    out: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def gpu_numa_nodes(sysfs: str = "/sys") -> List[int]:
    """NUMA node of every AMD GPU of the host in PCI address order (the order HIP enumerates them in), read from
    sysfs -- no GPU call is made; -1 where the platform does not say.  HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES
    given as a list of indices select and order the entries as the runtime will."""
    base = os.path.join(sysfs, "bus", "pci", "devices")
    found = []
    try:
        names = sorted(os.listdir(base))
    except OSError:
        return []
    for name in names:
        d = os.path.join(base, name)
        try:
            with open(os.path.join(d, "vendor")) as f:
                if f.read().strip().lower() != "0x1002":
                    continue
            with open(os.path.join(d, "class")) as f:
                cls = int(f.read().strip(), 16) >> 8
            if cls not in (0x0300, 0x0302, 0x0380, 0x1200):  # display controllers / processing accelerators
                continue
            with open(os.path.join(d, "numa_node")) as f:
                found.append(int(f.read().strip()))
        except (OSError, ValueError):
            continue
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        sel = os.environ.get(var)
        if sel:
            try:
                idx = [int(t) for t in sel.split(",") if t.strip() != ""]
                found = [found[i] for i in idx if 0 <= i < len(found)]
            except ValueError:
                pass  # UUIDs: keep the PCI order
            break
    return found


def plan_affinity(local_rank: int, local_world: int, allowed: Sequence[int], gpu_nodes: Sequence[int],
                  node_cpus: dict) -> List[int]:
    """The CPUs rank `local_rank` of `local_world` ranks on this host should run on: the cores of its GPU's NUMA node
    that this process may use, cut evenly among the ranks whose GPUs share that node (each replica's engine thread,
    launcher and completion pollers then stay off the other replicas' cores); where the platform does not say which
    node a GPU hangs off, an even cut of the allowed set.  Never empty: with fewer cores than ranks, ranks share."""
    allowed = sorted(set(allowed))
    if local_world <= 1 or not allowed:
        return allowed
    node = gpu_nodes[local_rank] if local_rank < len(gpu_nodes) else -1
    pool, peers = allowed, list(range(local_world))
    if node >= 0 and node in node_cpus:
        on_node = [c for c in allowed if c in set(node_cpus[node])]
        if on_node:
            pool = on_node
            peers = [r for r in range(local_world) if (gpu_nodes[r] if r < len(gpu_nodes) else -1) == node]
    pos, n = peers.index(local_rank), len(peers)
    lo, hi = pos * len(pool) // n, (pos + 1) * len(pool) // n
    return pool[lo:hi] if hi > lo else [pool[lo]]  # fewer cores than ranks: neighbours share one


def pin_to_gpu_numa(local_rank: Optional[int] = None, local_world: Optional[int] = None, sysfs: str = "/sys") -> dict:
    """Pin THIS process (a call, not a re-exec) to the cores `plan_affinity` gives it; to be called before the process
    touches its GPU, so that the runtime's helper threads inherit the mask.  Returns what was done, for the report."""
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if local_rank is None else local_rank
    if local_world is None:
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    if not hasattr(os, "sched_setaffinity"):
        return {"pinned": False, "reason": "no sched_setaffinity on this platform"}
    allowed = sorted(os.sched_getaffinity(0))
    if local_world <= 1:
        return {"pinned": False, "cpus": len(allowed), "reason": "one replica on this host"}
    nodes = gpu_numa_nodes(sysfs)
    node_cpus = {}
    for n in set(nodes):
        if n >= 0:
            try:
                with open(os.path.join(sysfs, "devices", "system", "node", f"node{n}", "cpulist")) as f:
                    node_cpus[n] = parse_cpulist(f.read())
            except OSError:
                pass
    cpus = plan_affinity(local_rank, local_world, allowed, nodes, node_cpus)
    os.sched_setaffinity(0, cpus)
    return {"pinned": True, "cpus": len(cpus), "first_cpu": cpus[0], "last_cpu": cpus[-1],
            "numa_node": nodes[local_rank] if local_rank < len(nodes) else -1}
