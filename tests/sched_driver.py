"""Seeded programs for the continuous-batching scheduler (light_vllm/decoding/scheduler.py:235-1132), the caller of
the block manager and the producer of the per-step block tables / swap / copy lists the kernels consume.

`run_program(make_scheduler, adapter, config, seed)` drives ANY scheduler with the reference's interface
(add_request / schedule / free_seq / free_finished_request) through the engine's synchronous step loop
(core/llm_engine.py:119-130): requests arrive over time, every scheduled step is "executed" (computed tokens
advance, every decoding sequence gets a seeded token), sequences finish at their token budget, the pool is small
enough to preempt.  It records what each schedule() returned: the groups in order, prompt / decode, chunk size,
do_sample, every block table, the computed-prefix blocks, the swap-in / swap-out / copy lists, the ignored
requests, the token count and the preemption counter.

`oracle/make_golden.py scheduler` records the traces of the REFERENCE's DecodingScheduler (imported through the stub
loader) into tests/golden/scheduler_*.json; tests/test_scheduler.py replays the same programs on this package's
scheduler and compares step by step.  Synchronous stepping only: with steps in flight this package's scheduler
deliberately differs from the reference (it never evicts a group whose step is still executing).

TEST INFRASTRUCTURE ONLY.
"""
import random
from typing import Any, Dict, List

CONFIGS = [
    # name, config
    ("default_v1", dict(version="v1", block_size=16, num_gpu_blocks=64, num_cpu_blocks=16, max_num_seqs=8,
                        max_num_batched_tokens=256, max_model_len=256, chunked=False, preemption_mode=None,
                        enable_caching=False, n_requests=40, seed=1)),
    ("default_v2", dict(version="v2", block_size=16, num_gpu_blocks=64, num_cpu_blocks=0, max_num_seqs=8,
                        max_num_batched_tokens=256, max_model_len=256, chunked=False, preemption_mode=None,
                        enable_caching=False, n_requests=40, seed=2)),
    ("tight_recompute_v1", dict(version="v1", block_size=8, num_gpu_blocks=22, num_cpu_blocks=0, max_num_seqs=6,
                                max_num_batched_tokens=128, max_model_len=128, chunked=False,
                                preemption_mode="recompute", enable_caching=False, n_requests=50, seed=3, max_out=60)),
    ("tight_swap_v1", dict(version="v1", block_size=8, num_gpu_blocks=22, num_cpu_blocks=64, max_num_seqs=6,
                           max_num_batched_tokens=128, max_model_len=128, chunked=False, preemption_mode="swap",
                           enable_caching=False, n_requests=50, seed=4, max_out=60)),
    ("tight_default_mode_v2", dict(version="v2", block_size=8, num_gpu_blocks=22, num_cpu_blocks=0, max_num_seqs=6,
                                   max_num_batched_tokens=128, max_model_len=128, chunked=False, preemption_mode=None,
                                   enable_caching=False, n_requests=50, seed=9, max_out=60)),
    ("chunked_v1", dict(version="v1", block_size=16, num_gpu_blocks=96, num_cpu_blocks=0, max_num_seqs=8,
                        max_num_batched_tokens=48, max_model_len=256, chunked=True, preemption_mode=None,
                        enable_caching=False, n_requests=40, seed=5)),
    ("chunked_tight_v2", dict(version="v2", block_size=8, num_gpu_blocks=24, num_cpu_blocks=0, max_num_seqs=6,
                              max_num_batched_tokens=32, max_model_len=128, chunked=True, preemption_mode=None,
                              enable_caching=False, n_requests=40, seed=6, max_out=60)),
    ("prefix_cache_v1", dict(version="v1", block_size=8, num_gpu_blocks=64, num_cpu_blocks=0, max_num_seqs=8,
                             max_num_batched_tokens=256, max_model_len=128, chunked=False, preemption_mode=None,
                             enable_caching=True, n_requests=40, seed=7)),
    ("lookahead_v2", dict(version="v2", block_size=8, num_gpu_blocks=64, num_cpu_blocks=0, max_num_seqs=6,
                          max_num_batched_tokens=128, max_model_len=128, chunked=False, preemption_mode=None,
                          enable_caching=False, n_requests=40, seed=8, lookahead=3)),
    # multi-step decode over the prefix-caching allocator of the v2 manager (block_manager_v2.py:199-237)
    ("lookahead_prefix_cache_v2", dict(version="v2", block_size=8, num_gpu_blocks=40, num_cpu_blocks=0, max_num_seqs=6,
                                       max_num_batched_tokens=128, max_model_len=128, chunked=False, preemption_mode=None,
                                       enable_caching=True, n_requests=50, seed=10, lookahead=3, max_out=40)),
]


class _Finished:
    """What free_finished_request receives: objects with a request_id (the reference passes RequestOutputs)."""

    def __init__(self, request_id):
        self.request_id = request_id


def run_program(make_scheduler, adapter, config: Dict[str, Any], max_steps: int = 600, free_hook=None,
                recorded=None) -> List[dict]:
    """free_hook(block_manager, recorded_order | None) -> finish(): wraps one step (schedule + output processing);
    finish() returns the order in which blocks were released during it.  The reference's v1 manager releases the
    blocks of a freed table in `set()` order (object addresses, block_manager_v1.py:553-557): the recorder logs the
    order it happened to use (`free_order` of a step), the replay imposes it."""
    rng = random.Random(config["seed"])
    bs = config["block_size"]
    sched = make_scheduler(config)
    S = adapter.status
    vocab = 1000
    shared = [rng.randrange(vocab) for _ in range(3 * bs)]
    # arrivals: (step, prompt, max_tokens)
    arrivals = []
    step_at = 0
    for i in range(config["n_requests"]):
        step_at += rng.choice([0, 0, 0, 1, 2, 5])
        n = rng.choice([1, 3, bs - 1, bs, bs + 1, 2 * bs + 3, rng.randint(1, min(6 * bs, config["max_model_len"] - 20))])
        if rng.random() < 0.05:
            n = config["max_model_len"] + 5  # too long: ignored
        prompt = [rng.randrange(vocab) for _ in range(n)]
        if config["enable_caching"] and rng.random() < 0.5 and n > len(shared):
            prompt[:len(shared)] = shared
        arrivals.append((step_at, prompt, rng.randint(1, config.get("max_out", 3 * bs))))
    groups: Dict[str, Any] = {}
    budgets: Dict[str, int] = {}
    next_arrival = 0
    trace: List[dict] = []
    la = config.get("lookahead", 0)
    for step in range(max_steps):
        while next_arrival < len(arrivals) and arrivals[next_arrival][0] <= step:
            _, prompt, max_tokens = arrivals[next_arrival]
            rid = str(next_arrival)
            seq = adapter.seq(next_arrival, prompt, bs)
            g = adapter.group(rid, [seq])
            groups[rid], budgets[rid] = g, max_tokens
            sched.add_request(g)
            next_arrival += 1
        rec_prev = recorded[len(trace)] if recorded is not None and len(trace) < len(recorded) else None
        finish = free_hook(sched.block_manager, rec_prev.get("free_order") if rec_prev else None) if free_hook else None
        out = sched.schedule()
        if out is None:
            if finish is not None:
                finish()
            if next_arrival >= len(arrivals):
                break
            trace.append({"step": step, "none": True})
            continue
        rec: Dict[str, Any] = {"step": step, "groups": [],
                               "swap_in": [list(p) for p in out.blocks_to_swap_in],
                               "swap_out": [list(p) for p in out.blocks_to_swap_out],
                               "copy": [list(p) for p in out.blocks_to_copy],
                               "ignored": [g.request_id for g in out.ignored_seq_groups],
                               "num_batched_tokens": out.num_batched_tokens,
                               "num_prefill_groups": out.num_prefill_groups,
                               "num_lookahead_slots": out.num_lookahead_slots,
                               "preempted": out.preempted,
                               "cumulative_preemption": sched.num_cumulative_preemption}
        for m in out.seq_group_metadata_list:
            rec["groups"].append({"id": m.request_id, "is_prompt": bool(m.is_prompt), "chunk": m.token_chunk_size,
                                  "do_sample": bool(m.do_sample),
                                  "tables": {str(k): list(v) for k, v in m.block_tables.items()},
                                  "computed": list(m.computed_block_nums or [])})
        rec["free_gpu"] = sched.block_manager.get_num_free_gpu_blocks()
        trace.append(rec)
        # "execute" the step and process its outputs (output_processor.py: computed tokens advance, a sampled token
        # per running sequence of a group that sampled; with lookahead slots a burst of 1 + lookahead tokens)
        finished: List[_Finished] = []
        for s, m in zip(out.scheduled_seq_groups, out.seq_group_metadata_list):
            g = s.seq_group
            g.update_num_computed_tokens(s.token_chunk_size)
            if m.do_sample:
                for seq in g.get_seqs(status=S("RUNNING")):
                    burst = 1 + (la if not m.is_prompt else 0)
                    for b in range(burst):
                        if b:
                            seq.data.update_num_computed_tokens(1)
                        adapter.append(seq, rng.randrange(vocab))
                        if seq.get_output_len() >= budgets[g.request_id] or seq.get_len() >= config["max_model_len"]:
                            seq.status = S("FINISHED_LENGTH_CAPPED")
                            sched.free_seq(seq)
                            break
            finished.append(_Finished(g.request_id))
        for g in out.ignored_seq_groups:
            groups.pop(g.request_id, None)
        adapter.free_finished(sched, finished)
        if finish is not None:
            order = finish()
            if order:
                rec["free_order"] = order
        if next_arrival >= len(arrivals) and not sched.has_unfinished_requests():
            break
    return trace
