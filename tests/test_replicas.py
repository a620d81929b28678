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


def _protocol_worker(rank, world, port, q):
    """bench.py's timed-region protocol and nothing else: pin, barrier, a per-rank clock, barrier, MAX of the clocks,
    SUM of the per-rank token counts -- what `value` is made of at N ranks."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import light_vllm_amd  # noqa: F401
    from light_vllm_amd.engine.replicas import ReplicaGroup, pin_to_gpu_numa
    before = sorted(os.sched_getaffinity(0))
    pin = pin_to_gpu_numa()  # a call in this process, before anything else
    after = sorted(os.sched_getaffinity(0))
    group = ReplicaGroup(backend="gloo")
    group.barrier()
    time.sleep(0.02 * (1 + rank % 3))  # ranks finish at different times
    arrived = time.time()  # one host: every rank reads the same clock
    group.barrier()
    left = time.time()
    mine = (arrived, left)
    elapsed = group.max(0.5 + 0.25 * rank)       # a known per-rank clock: the slowest wins
    tokens = group.sum(float(20 * 32))            # --steps 20 x bs 32 per replica
    q.put((rank, elapsed, tokens, mine, pin, before, after))
    group.shutdown()


def test_eight_replicas_gloo_rehearse_the_bench_protocol():
    """World 8 on the CPU (the N = 8 case itself is the driver's to launch on a GPU node): every rank sees the same
    MAX clock and the same SUM, the barrier holds the fast ranks for the slowest, and each rank pinned itself to a
    disjoint, non-empty share of the cores this container allows."""
    world = 8
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_protocol_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[0] for r in results] == list(range(world))
    assert all(r[1] == 0.5 + 0.25 * (world - 1) for r in results)
    assert all(r[2] == world * 640.0 for r in results)
    assert min(r[3][1] for r in results) >= max(r[3][0] for r in results)  # nobody left before the slowest arrived
    allowed = results[0][5]
    if len(allowed) >= world:
        shares = [tuple(r[6]) for r in results]
        assert all(len(s) >= 1 and set(s) <= set(allowed) for s in shares)
        assert len(set(c for s in shares for c in s)) == sum(len(s) for s in shares), shares  # disjoint
        assert all(r[4]["pinned"] for r in results)


def test_affinity_plan_follows_the_gpus_numa_nodes():
    from light_vllm_amd.engine.replicas import parse_cpulist, plan_affinity
    assert parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    nodes = [0, 0, 0, 0, 1, 1, 1, 1]
    node_cpus = {0: parse_cpulist("0-63,128-191"), 1: parse_cpulist("64-127,192-255")}
    allowed = list(range(256))
    shares = [plan_affinity(r, 8, allowed, nodes, node_cpus) for r in range(8)]
    assert all(len(s) == 32 for s in shares)
    assert all(set(shares[r]) <= set(node_cpus[nodes[r]]) for r in range(8))
    assert len(set(c for s in shares for c in s)) == 256
    # a cgroup that allows only part of a node: the cut is made of what is allowed
    few = list(range(0, 16)) + list(range(64, 72))
    s0, s5 = plan_affinity(0, 8, few, nodes, node_cpus), plan_affinity(5, 8, few, nodes, node_cpus)
    assert s0 == [0, 1, 2, 3] and s5 == [66, 67]
    # unknown topology: an even cut of the allowed set; fewer cores than ranks: never an empty mask
    assert plan_affinity(3, 4, list(range(8)), [], {}) == [6, 7]
    assert plan_affinity(5, 8, [0, 1], [-1] * 8, {}) == [1]
    assert plan_affinity(0, 1, [4, 5], [0], {0: [4, 5]}) == [4, 5]
