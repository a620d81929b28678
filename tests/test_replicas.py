"""N > 1 protocol of bench.py / a replicated deployment, rehearsed on the CPU with gloo
(world_size 2): request sharding is disjoint and complete, the barrier / max-clock / sum-counter
used around timed regions work, and no collective is needed by the per-replica host logic."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import light_vllm_amd  # noqa: F401
    from light_vllm_amd.engine.config import CacheConfig, SchedulerConfig
    from light_vllm_amd.engine.replicas import ReplicaGroup, shard_requests
    from light_vllm_amd.engine.scheduler import DecodingScheduler
    from light_vllm_amd.engine.sequence import Sequence, SequenceGroup

    group = ReplicaGroup(backend="gloo")
    mine = shard_requests(list(range(10)), group.world_size, group.rank)
    # every replica runs its own scheduler + block manager on its shard, no communication
    s = DecodingScheduler(SchedulerConfig(max_num_batched_tokens=64, max_num_seqs=8, max_model_len=64),
                          CacheConfig(block_size=4, num_gpu_blocks=32, num_cpu_blocks=0))
    for r in mine:
        s.add_request(SequenceGroup(str(r), [Sequence(r, list(range(r + 1)), 4)]))
    out = s.schedule()
    scheduled = sorted(int(sg.seq_group.request_id) for sg in out.scheduled_seq_groups)
    group.barrier()
    elapsed = group.max(1.0 + rank)           # slowest replica's clock
    total = group.sum(float(len(scheduled)))  # whole-job count
    q.put((rank, mine, scheduled, elapsed, total))
    group.shutdown()


def test_two_replicas_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, mine0, sch0, el0, tot0), (r1, mine1, sch1, el1, tot1) = results
    assert mine0 == [0, 2, 4, 6, 8] and mine1 == [1, 3, 5, 7, 9]
    assert sch0 == mine0 and sch1 == mine1
    assert el0 == el1 == 2.0 and tot0 == tot1 == 10.0
