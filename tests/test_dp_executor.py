"""The queue-sharing data-parallel front end of the prefill-only workflow (dp_executor.py; reference:
prefill_only/executor/gpu_data_parallelism_executor.py:41-72, prefill_only/workflow.py:31-41) on the CPU: two worker
PROCESSES drain one executor_in queue behind one scheduler.  The workers here are host-only stand-ins
(tests/dp_fake_worker.py: a checksum per request, a sleep per token) -- the subject is the queue discipline, not the
model; the GPU twin runs the real worker (test_dp_encode_two_workers_on_the_gpu)."""
import pytest
import torch

from dp_fake_worker import FakeWorkerFactory


def make(factory, dp=2, max_seqs=4, on_the_fly=2, budget=None):
    import light_vllm_amd  # noqa: F401
    from light_vllm_amd.prefill_only import PrefillOnlySchedulerConfig
    from light_vllm_amd.prefill_only.dp_executor import DataParallelEncodeEngine
    cfg = PrefillOnlySchedulerConfig(max_model_len=512, max_num_seqs=max_seqs, max_num_batched_tokens=budget,
                                     max_num_on_the_fly=on_the_fly, scheduling="async")
    return DataParallelEncodeEngine(None, cfg, data_parallel_size=dp, worker_factory=factory, start_timeout_s=120)


def uneven_prompts(n=60, seed=3):
    g = torch.Generator().manual_seed(seed)
    lens = [int(torch.randint(1, 400, (1,), generator=g)) if i % 5 else 500 for i in range(n)]
    return [torch.randint(2, 1000, (m,), generator=g).tolist() for m in lens]


def test_two_workers_drain_one_queue_and_the_faster_one_takes_more():
    """Uneven request lengths, worker 0 four times slower per token than worker 1: every request comes back with its
    own result, both workers served steps of the ONE queue, the faster worker served more of them (a static shard
    would have given each half), and no more than max_num_on_the_fly x workers steps were ever outstanding."""
    eng = make(FakeWorkerFactory([400, 100]))
    assert eng.max_num_on_the_fly == 4  # workflow.py:37-38
    ps = uneven_prompts() + [[7] * 600]  # the last one exceeds max_model_len: ignored, outputs None
    peak = 0
    for i, p in enumerate(ps):
        eng.add_request(str(i), p)
    res = {}
    while eng.has_unfinished_requests() or eng.num_on_the_fly > 0:
        for o in eng.step():
            res[o.request_id] = o.outputs
        peak = max(peak, eng.num_on_the_fly)
    eng.shutdown()
    assert len(res) == len(ps) and res[str(len(ps) - 1)] is None
    served = set()
    for i, p in enumerate(ps[:-1]):
        r = res[str(i)]
        assert r[0] == float(sum(p)) and r[1] == float(len(p)) and r[2] == float(p[0]), i
        served.add(int(r[3]))
    assert served == {0, 1}
    assert peak <= 4
    assert eng.steps_by_rank[1] > eng.steps_by_rank[0] >= 1, eng.steps_by_rank
    assert sum(eng.steps_by_rank.values()) == eng._next_step


def test_one_worker_is_the_plain_async_engine():
    eng = make(FakeWorkerFactory([50]), dp=1)
    ps = uneven_prompts(20)
    res = eng.encode(ps)
    eng.shutdown()
    assert all(res[str(i)][0] == float(sum(p)) for i, p in enumerate(ps))
    assert eng.steps_by_rank == {0: eng._next_step}


def test_a_failing_step_surfaces_and_its_requests_leave_the_books():
    eng = make(FakeWorkerFactory([50, 50], fail_token=999999), max_seqs=2)
    good = [[5, 6, 7], [8, 9], [10] * 40, [11] * 3]
    for i, p in enumerate(good[:2]):
        eng.add_request(str(i), p)
    eng.add_request("bad", [3, 999999, 4])
    eng.add_request("bad2", [12, 13])
    for i, p in enumerate(good[2:], start=2):
        eng.add_request(str(i), p)
    res, errors = {}, 0
    while eng.has_unfinished_requests() or eng.num_on_the_fly > 0:
        try:
            for o in eng.step():
                res[o.request_id] = o.outputs
        except RuntimeError as e:
            assert "poisoned" in str(e)
            errors += 1
    eng.shutdown()
    assert errors == 1
    assert set(res) == {"0", "1", "2", "3"}  # the poisoned step's two requests produced nothing, the rest all did
    assert not eng.has_unfinished_requests()


def test_a_worker_that_dies_with_a_step_is_noticed_within_seconds():
    """a worker process that dies after taking a step off the shared queue never answers.  The front end
    looks at its workers every second, fails every outstanding step, drops their requests from the books and raises
    -- long before the 600 s step timeout."""
    import time
    eng = make(FakeWorkerFactory([50], die_token=424242), dp=1, max_seqs=2, on_the_fly=2)
    eng.step_timeout_s = 600.0
    eng.add_request("a", [1, 2, 3])
    eng.add_request("b", [4, 424242, 5])
    t0 = time.time()
    with pytest.raises(RuntimeError, match="died"):
        while eng.has_unfinished_requests() or eng.num_on_the_fly > 0:
            eng.step()
    assert time.time() - t0 < 30
    assert eng.num_on_the_fly == 0 and not eng.has_unfinished_requests()
    eng.shutdown()


def test_a_worker_that_fails_to_start_is_reported():
    eng = make(FakeWorkerFactory([50, 50], fail_start=1))
    eng.add_request("0", [1, 2, 3])
    with pytest.raises(RuntimeError, match="worker 1 failed to start"):
        eng.step()
    assert eng.procs is None  # the workers that did start were shut down


def test_the_gpu_worker_fails_loudly_without_a_gpu():
    """The default factory builds the gfx950 worker; in a container without a GPU the front end reports that,
    it does not fall back to anything."""
    if torch.cuda.is_available():
        pytest.skip("needs a container without a GPU")
    import light_vllm_amd  # noqa: F401
    from light_vllm_amd.prefill_only import PrefillOnlySchedulerConfig
    from light_vllm_amd.prefill_only.dp_executor import DataParallelEncodeEngine
    from light_vllm_amd.prefill_only.model import EncoderConfig
    eng = DataParallelEncodeEngine(EncoderConfig.tiny(), PrefillOnlySchedulerConfig(max_model_len=64, max_num_seqs=2),
                                   data_parallel_size=1, start_timeout_s=120)
    eng.add_request("0", [1, 2, 3])
    with pytest.raises(RuntimeError, match="needs an MI355X"):
        eng.step()


@pytest.mark.gpu
def test_dp_encode_two_workers_on_the_gpu():
    """Two real worker processes on the one GPU of the box (devices [0, 0]) behind one scheduler: embeddings equal
    the single-process engine's, both workers served steps."""
    import light_vllm_amd  # noqa: F401
    from light_vllm_amd.prefill_only import PrefillOnlySchedulerConfig
    from light_vllm_amd.prefill_only.dp_executor import DataParallelEncodeEngine
    from light_vllm_amd.prefill_only.engine import PrefillOnlyEngine
    from light_vllm_amd.prefill_only.model import EncoderConfig
    ps = uneven_prompts(40)
    cfg = lambda: PrefillOnlySchedulerConfig(max_model_len=512, max_num_seqs=4, scheduling="async")  # noqa: E731
    dp = DataParallelEncodeEngine(EncoderConfig.tiny(), cfg(), data_parallel_size=2, devices=[0, 0], seed=0)
    got = dp.encode([[min(t, 511) for t in p] for p in ps])
    dp.shutdown()
    assert set(dp.steps_by_rank) == {0, 1}
    one = PrefillOnlyEngine(EncoderConfig.tiny(), cfg(), device="cuda:0", seed=0)
    want = one.encode([[min(t, 511) for t in p] for p in ps], use_async=True)
    one.shutdown()
    for k in want:
        assert float((got[k] - want[k]).abs().max()) <= 2e-2, k
