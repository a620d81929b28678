"""Host logic of the decode loop on the CPU: scheduler policy, block tables handed to the kernels,
slot mapping, metadata layout.  No GPU, no kernels."""
import torch

from light_vllm_amd.attention.backend import (PagedAttnBackend, compute_slot_mapping,
                                              compute_slot_mapping_start_idx)
from light_vllm_amd.engine.config import CacheConfig, SchedulerConfig
from light_vllm_amd.engine.input_builder import ModelInputBuilder
from light_vllm_amd.engine.scheduler import DecodingScheduler
from light_vllm_amd.engine.sequence import Sequence, SequenceGroup, SequenceStatus


def make(num_gpu=64, num_cpu=16, max_seqs=4, max_tokens=64, chunked=False, **kw):
    cc = CacheConfig(block_size=4, num_gpu_blocks=num_gpu, num_cpu_blocks=num_cpu, **kw)
    sc = SchedulerConfig(max_num_batched_tokens=max_tokens, max_num_seqs=max_seqs, max_model_len=256)
    return DecodingScheduler(sc, cc, chunked_prefill_enabled=chunked), sc, cc


def add(s, i, n, **kw):
    g = SequenceGroup(str(i), [Sequence(i, list(range(100 * i, 100 * i + n)), 4)], **kw)
    s.add_request(g)
    return g


def finish_step(s, out, token=7):
    """What the output processor does after a step."""
    for sg in out.scheduled_seq_groups:
        g = sg.seq_group
        g.update_num_computed_tokens(sg.token_chunk_size)
        if not g.is_prefill():
            for seq in g.get_seqs(status=SequenceStatus.RUNNING):
                seq.append_token_id(token)
    s.free_finished_request([sg.seq_group.request_id for sg in out.scheduled_seq_groups])


def test_prefill_then_decode_and_block_tables():
    s, sc, cc = make()
    gs = [add(s, i, n) for i, n in enumerate((5, 9, 3))]
    out = s.schedule()
    assert out.num_prefill_groups == 3 and out.num_batched_tokens == 17
    assert [m.is_prompt for m in out.seq_group_metadata_list] == [True] * 3
    # uncached v1 allocator hands out ids from the top (block_manager_v1.py:186-193)
    assert [m.block_tables[i] for i, m in enumerate(out.seq_group_metadata_list)] == [[63, 62], [61, 60, 59], [58]]
    assert all(g.busy for g in gs)
    assert s.schedule() is None  # everything in flight: nothing to schedule (async overlap rule)
    finish_step(s, out)
    out2 = s.schedule()
    assert out2.num_prefill_groups == 0 and len(out2.scheduled_seq_groups) == 3
    assert all(sg.token_chunk_size == 1 for sg in out2.scheduled_seq_groups)
    # seq 1 had 9 tokens + 1 sampled = 10 -> still 3 blocks; seq 0: 6 tokens -> 2 blocks; seq 2: 4 -> 1 block
    finish_step(s, out2)
    out3 = s.schedule()
    tables = {m.request_id: list(m.block_tables.values())[0] for m in out3.seq_group_metadata_list}
    assert tables["2"] == [58, 57]  # 5 tokens now: a second block was appended


def test_token_budget_and_max_seqs():
    s, sc, cc = make(max_seqs=2, max_tokens=10)
    for i, n in enumerate((6, 6, 3)):
        add(s, i, n)
    out = s.schedule()
    assert [sg.seq_group.request_id for sg in out.scheduled_seq_groups] == ["0"]  # 6 + 6 > 10 tokens
    finish_step(s, out)
    out = s.schedule()
    assert [sg.seq_group.request_id for sg in out.scheduled_seq_groups] == ["1"]  # prefills go first
    finish_step(s, out)
    out = s.schedule()
    # two running sequences fill max_num_seqs: the third prompt waits, decodes run
    assert out.num_prefill_groups == 0 and len(out.scheduled_seq_groups) == 2
    # a prompt longer than the step budget can never run: ignored (scheduler.py:611-622)
    s2, _, _ = make(max_seqs=2, max_tokens=10)
    g = add(s2, 9, 200)
    ok = add(s2, 10, 4)
    out = s2.schedule()
    assert g in out.ignored_seq_groups and g.seqs[0].status == SequenceStatus.FINISHED_IGNORED
    assert [sg.seq_group.request_id for sg in out.scheduled_seq_groups] == ["10"]


def test_preemption_by_recompute_and_swap():
    s, sc, cc = make(num_gpu=6, num_cpu=8)
    s.block_manager.watermark_blocks = 0
    a, b = add(s, 0, 8), add(s, 1, 8)   # 2 + 2 blocks, both exactly full
    out = s.schedule()
    assert len(out.scheduled_seq_groups) == 2
    finish_step(s, out)                   # 9 tokens each -> each needs a 3rd block: 6 blocks, fits
    out = s.schedule()
    assert out.preempted == 0
    for _ in range(3):
        finish_step(s, out)
        out = s.schedule()
    # all 6 blocks are in use: the conservative rule "one free block per running sequence"
    # (block_manager_v1.py:355-359) fails, the younger group is preempted and will be recomputed
    assert out.preempted == 1 and b.seqs[0].status == SequenceStatus.WAITING
    assert b.seqs[0].data.get_num_computed_tokens() == 0 and s.waiting[0] is b
    assert s.block_manager.get_num_free_gpu_blocks() == 3
    # swap mode
    s2, _, _ = make(num_gpu=6, num_cpu=8)
    s2.block_manager.watermark_blocks = 0
    s2.user_specified_preemption_mode = "swap"
    a2, b2 = add(s2, 0, 8), add(s2, 1, 8)
    out = s2.schedule()
    for _ in range(5):
        finish_step(s2, out)
        out = s2.schedule()
        if out.preempted:
            break
    assert out.preempted == 1 and b2.seqs[0].status == SequenceStatus.SWAPPED
    assert len(out.blocks_to_swap_out) == 3 and s2.swapped[0] is b2
    # finish a2 -> its blocks free up -> b2 swaps back in
    finish_step(s2, out)
    a2.seqs[0].status = SequenceStatus.FINISHED_STOPPED
    s2.free_seq(a2.seqs[0])
    s2.free_finished_request(["0"])
    out = s2.schedule()
    assert len(out.blocks_to_swap_in) == 3 and b2.seqs[0].status == SequenceStatus.RUNNING


def test_async_two_batches_in_flight():
    """max_num_seqs bounds a STEP; while batch A is busy a second batch can be admitted and run
    (the reference's double buffering, scheduler.py:388-391,680-684)."""
    s, sc, cc = make(max_seqs=2, max_tokens=64)
    for i in range(4):
        add(s, i, 4)
    a = s.schedule()
    assert [sg.seq_group.request_id for sg in a.scheduled_seq_groups] == ["0", "1"]
    b = s.schedule()  # A still busy
    assert [sg.seq_group.request_id for sg in b.scheduled_seq_groups] == ["2", "3"]
    assert s.schedule() is None
    finish_step(s, a)
    a2 = s.schedule()
    assert sorted(sg.seq_group.request_id for sg in a2.scheduled_seq_groups) == ["0", "1"]
    assert all(sg.token_chunk_size == 1 for sg in a2.scheduled_seq_groups)


def test_chunked_prefill_mixes_decodes_and_prompt_chunks():
    s, sc, cc = make(max_seqs=4, max_tokens=8, chunked=True)
    g0 = add(s, 0, 5)
    out = s.schedule()
    finish_step(s, out)
    g1 = add(s, 1, 20)
    out = s.schedule()
    kinds = [(sg.seq_group.request_id, sg.token_chunk_size) for sg in out.scheduled_seq_groups]
    assert kinds == [("1", 7), ("0", 1)]  # prompt chunk cut to the 8-token budget left by the decode
    metas = {m.request_id: m for m in out.seq_group_metadata_list}
    assert metas["1"].do_sample is False and metas["0"].do_sample is True


def test_slot_mapping_and_metadata():
    assert compute_slot_mapping_start_idx(True, 10, 0, 8, False) == 2
    assert compute_slot_mapping_start_idx(False, 1, 9, 8, False) == 0
    sm = []
    compute_slot_mapping(False, sm, 0, 10, 0, 2, 4, {0: [7, 3, 5]})
    assert sm == [-1, -1, 30, 31, 12, 13, 14, 15, 20, 21]  # docstring example of backends/utils.py:60-65 shape
    sm = []
    compute_slot_mapping(True, sm, 0, 3, 0, 0, 4, None)
    assert sm == [-1, -1, -1]

    s, sc, cc = make()
    add(s, 0, 6)
    add(s, 1, 3)
    builder = ModelInputBuilder(sc, cc, PagedAttnBackend())
    out = s.schedule()
    ei = builder(out)
    mi = ei.model_input
    md = mi.attn_metadata
    assert mi.input_tokens.tolist() == list(range(0, 6)) + list(range(100, 103))
    assert mi.input_positions.tolist() == [0, 1, 2, 3, 4, 5, 0, 1, 2]
    assert md.num_prefills == 2 and md.num_prefill_tokens == 9 and md.num_decode_tokens == 0
    assert md.slot_mapping.tolist() == [63 * 4 + i for i in range(4)] + [62 * 4, 62 * 4 + 1] + [61 * 4 + i for i in range(3)]
    assert md.slot_mapping.dtype == torch.int64 and md.block_tables.dtype == torch.int32
    assert md.query_start_loc.tolist() == [0, 6, 9] and md.seq_start_loc.tolist() == [0, 6, 9]
    assert mi.sample_indices == [5, 8] and ei.worker_input.blocks_to_copy.shape == (0, 2)
    finish_step(s, out, token=42)
    out = s.schedule()
    mi = builder(out).model_input
    md = mi.attn_metadata
    assert mi.decode_only and mi.input_tokens.tolist() == [42, 42] and mi.input_positions.tolist() == [6, 3]
    assert md.seq_lens_tensor.tolist() == [7, 4] and md.max_decode_seq_len == 7
    assert md.block_tables.tolist() == [[63, 62], [61, 0]]  # padded with 0 (flash_attn.py:317-321)
    assert md.slot_mapping.tolist() == [62 * 4 + 2, 61 * 4 + 3]
    dm = md.decode_metadata
    assert dm.num_decode_tokens == 2 and dm.block_tables.shape == (2, 2) and md.prefill_metadata is None


def test_prefix_cache_hit_skips_computed_blocks():
    s, sc, cc = make(enable_prefix_caching=True)
    builder = ModelInputBuilder(sc, cc, PagedAttnBackend())
    g0 = SequenceGroup("a", [Sequence(0, list(range(10)), 4)])
    s.add_request(g0)
    out = s.schedule()
    finish_step(s, out)
    for _ in range(2):  # decode steps mark the full blocks of "a" computed (all but its last full one)
        out = s.schedule()
        finish_step(s, out)
    g1 = SequenceGroup("b", [Sequence(1, list(range(8)) + [50, 51, 52], 4)])  # shares two full blocks
    s.add_request(g1)
    out = s.schedule()
    m = out.seq_group_metadata_list[0]
    assert m.request_id == "b" and len(m.computed_block_nums) == 2
    assert m.block_tables[1][:2] == out.seq_group_metadata_list[0].block_tables[1][:2]
    mi = builder(out).model_input
    assert mi.input_tokens.tolist() == [50, 51, 52] and mi.input_positions.tolist() == [8, 9, 10]
    assert mi.attn_metadata.context_lens == [8] and mi.attn_metadata.query_lens == [3]


def test_decode_step_arrays_equal_the_general_input_builder():
    """The staging arrays of the decode fast path (DecodeStepArrays) against ModelInputBuilder +
    the attention metadata builder, step by step over a run with admissions, preemptions under
    memory pressure, finishing sequences and changing batch composition: bit-equal every step."""
    import random

    import numpy as np

    from light_vllm_amd.engine.input_builder import DecodeStepArrays
    for version, mode in ((False, "swap"), (True, "swap"), (False, None), (True, "recompute")):
        rng = random.Random(5)
        cc = CacheConfig(block_size=4, num_gpu_blocks=40, num_cpu_blocks=16)
        sc = SchedulerConfig(max_num_batched_tokens=64, max_num_seqs=6, max_model_len=256,
                             use_v2_block_manager=version, preemption_mode=mode)
        s = DecodingScheduler(sc, cc)
        builder = ModelInputBuilder(sc, cc, PagedAttnBackend())
        arrays = DecodeStepArrays(8, 64, 4)
        next_id, checked = 0, 0
        for step in range(300):
            if rng.random() < 0.15 and next_id < 40:
                add(s, next_id, rng.randint(1, 14), max_tokens=rng.randint(2, 30))
                next_id += 1
            out = s.schedule()
            if out is None or out.is_empty():
                continue
            metas = out.seq_group_metadata_list
            plain = not (out.blocks_to_swap_in or out.blocks_to_swap_out or out.blocks_to_copy)
            if DecodeStepArrays.eligible(metas, plain, None):
                ids = arrays.fill(metas)
                mi = builder(out).model_input
                n = len(metas)
                md = mi.attn_metadata
                assert ids == mi.sample_seq_ids
                assert np.array_equal(arrays.input_ids[:n], mi.input_tokens.numpy())
                assert np.array_equal(arrays.positions[:n], mi.input_positions.numpy())
                assert np.array_equal(arrays.slot_mapping[:n], md.slot_mapping.numpy())
                assert np.array_equal(arrays.seq_lens[:n], md.seq_lens_tensor.numpy())
                w = md.block_tables.shape[1]
                lens = md.seq_lens_tensor.numpy()
                for i in range(n):  # entries a kernel may read: the blocks the sequence owns
                    k = (int(lens[i]) + 3) // 4
                    assert np.array_equal(arrays.block_tables[i, :k], md.block_tables[i, :k].numpy()), (step, i)
                assert (arrays.slot_mapping[n:] == -1).all() and (arrays.seq_lens[n:] == 0).all()
                assert w <= 64
                checked += 1
            # finish the step: random tokens, sequences end at max_tokens
            for sg in out.scheduled_seq_groups:
                g = sg.seq_group
                g.update_num_computed_tokens(sg.token_chunk_size)
                if not g.is_prefill():
                    for seq in g.get_seqs(status=SequenceStatus.RUNNING):
                        seq.append_token_id(rng.randint(0, 999))
                        if seq.get_output_len() >= g.max_tokens:
                            seq.status = SequenceStatus.FINISHED_LENGTH_CAPPED
                            s.free_seq(seq)
            s.free_finished_request([sg.seq_group.request_id for sg in out.scheduled_seq_groups])
        assert checked > 100, checked


def test_mixed_step_arrays_equal_the_general_input_builder():
    """MixedStepArrays (chunked-prefill steps staged for the captured mixed graph) against
    ModelInputBuilder + the attention metadata builder over a run with prompt chunks of every size,
    decode tokens, finishing sequences and preemption: bit-equal every step."""
    import random

    import numpy as np

    from light_vllm_amd.engine.input_builder import MixedStepArrays
    for version, mode in ((False, None), (True, None), (False, "swap")):
        rng = random.Random(11)
        cc = CacheConfig(block_size=4, num_gpu_blocks=48, num_cpu_blocks=16)
        sc = SchedulerConfig(max_num_batched_tokens=16, max_num_seqs=8, max_model_len=256,
                             use_v2_block_manager=version, preemption_mode=mode, chunked_prefill_enabled=True)
        s = DecodingScheduler(sc, cc, chunked_prefill_enabled=True)
        builder = ModelInputBuilder(sc, cc, PagedAttnBackend(), chunked_prefill_enabled=True)
        arrays = MixedStepArrays(16, 8, 64, 4)
        next_id, checked, with_prompts = 0, 0, 0
        for step in range(400):
            if rng.random() < 0.2 and next_id < 40:
                add(s, next_id, rng.randint(1, 40), max_tokens=rng.randint(2, 20))
                next_id += 1
            out = s.schedule()
            if out is None or out.is_empty():
                continue
            metas = out.seq_group_metadata_list
            plain = not (out.blocks_to_swap_in or out.blocks_to_swap_out or out.blocks_to_copy)
            if MixedStepArrays.eligible(metas, plain, None):
                filled = arrays.fill(metas)
                assert filled is not None
                ids, n = filled
                mi = builder(out).model_input
                md = mi.attn_metadata
                ns = len(metas)
                assert n == mi.input_tokens.shape[0] and ids == mi.sample_seq_ids
                assert np.array_equal(arrays.input_ids[:n], mi.input_tokens.numpy())
                assert np.array_equal(arrays.positions[:n], mi.input_positions.numpy())
                assert np.array_equal(arrays.slot_mapping[:n], md.slot_mapping.numpy())
                assert np.array_equal(arrays.seq_lens[:ns], md.seq_lens_tensor.numpy())
                assert np.array_equal(arrays.query_start_loc[:ns + 1], md.query_start_loc.numpy())
                assert list(arrays.sample_rows[:len(ids)]) == mi.sample_indices
                lens = md.seq_lens_tensor.numpy()
                for i in range(ns):
                    k = (int(lens[i]) + 3) // 4
                    assert np.array_equal(arrays.block_tables[i, :k], md.block_tables[i, :k].numpy()), (step, i)
                assert (arrays.slot_mapping[n:] == -1).all() and (arrays.seq_lens[ns:] == 0).all()
                assert (arrays.query_start_loc[ns + 1:] == n).all()
                checked += 1
                with_prompts += any(m.is_prompt for m in metas)
            for sg in out.scheduled_seq_groups:
                g = sg.seq_group
                was_prefill = g.is_prefill()
                g.update_num_computed_tokens(sg.token_chunk_size)
                if not g.is_prefill():  # the chunk reached the end of the prompt, or a decode step
                    for seq in g.get_seqs(status=SequenceStatus.RUNNING):
                        seq.append_token_id(rng.randint(0, 999))
                        if seq.get_output_len() >= g.max_tokens:
                            seq.status = SequenceStatus.FINISHED_LENGTH_CAPPED
                            s.free_seq(seq)
            s.free_finished_request([sg.seq_group.request_id for sg in out.scheduled_seq_groups])
        assert checked > 150 and with_prompts > 30, (checked, with_prompts)


def test_lookahead_slots_reach_the_block_manager_and_the_output():
    """num_scheduler_steps = k reserves k - 1 slots beyond the known tokens of every decoding sequence at each
    decode schedule (scheduler.py:978-982,1095-1106 with num_lookahead_slots), none for prompts."""
    cc = CacheConfig(block_size=4, num_gpu_blocks=64, num_cpu_blocks=0)
    sc = SchedulerConfig(max_num_batched_tokens=64, max_num_seqs=4, max_model_len=256, use_v2_block_manager=True,
                         num_scheduler_steps=4)
    assert sc.num_lookahead_slots == 3
    s = DecodingScheduler(sc, cc)
    g = add(s, 0, 6)          # 6 prompt tokens: 2 blocks
    out = s.schedule()
    assert out.num_prefill_groups == 1 and out.num_lookahead_slots == 0
    assert len(out.seq_group_metadata_list[0].block_tables[0]) == 2
    finish_step(s, out)       # +1 sampled token: 7 known tokens
    out = s.schedule()
    assert out.num_prefill_groups == 0 and out.num_lookahead_slots == 3
    # 7 tokens + 3 lookahead = 10 slots -> 3 blocks
    assert len(out.seq_group_metadata_list[0].block_tables[0]) == 3
    # the burst appends 4 tokens: 11 known; the next schedule reserves up to 14 slots -> 4 blocks
    for sg in out.scheduled_seq_groups:
        for _ in range(4):
            sg.seq_group.seqs[0].data.update_num_computed_tokens(1)
            sg.seq_group.seqs[0].append_token_id(9)
    s.free_finished_request([g.request_id])
    out = s.schedule()
    assert len(out.seq_group_metadata_list[0].block_tables[0]) == 4
    free_before = s.block_manager.get_num_free_gpu_blocks()
    g.seqs[0].status = SequenceStatus.FINISHED_STOPPED
    s.free_seq(g.seqs[0])
    assert s.block_manager.get_num_free_gpu_blocks() == free_before + 4 == 64


def test_lookahead_needs_v2():
    import pytest
    with pytest.raises(ValueError):
        SchedulerConfig(num_lookahead_slots=2)


# ---- the reference's DecodingScheduler, step by step (tests/golden/scheduler_*.json) ----
import glob as _glob
import json as _json
import os as _os

import pytest as _pytest

_SCHED_TRACES = sorted(_glob.glob(_os.path.join(_os.path.dirname(__file__), "golden", "scheduler_*.json")))


def _make_product_scheduler(cfg):
    cc = CacheConfig(block_size=cfg["block_size"], num_gpu_blocks=cfg["num_gpu_blocks"],
                     num_cpu_blocks=cfg["num_cpu_blocks"], enable_prefix_caching=cfg["enable_caching"])
    sc = SchedulerConfig(max_num_batched_tokens=cfg["max_num_batched_tokens"], max_num_seqs=cfg["max_num_seqs"],
                         max_model_len=cfg["max_model_len"], use_v2_block_manager=cfg["version"] == "v2",
                         preemption_mode=cfg["preemption_mode"], chunked_prefill_enabled=cfg["chunked"],
                         num_lookahead_slots=cfg.get("lookahead", 0))
    return DecodingScheduler(sc, cc, chunked_prefill_enabled=cfg["chunked"])


@_pytest.mark.parametrize("path", _SCHED_TRACES, ids=[_os.path.basename(p)[10:-5] for p in _SCHED_TRACES])
def test_scheduler_replays_the_reference_scheduler_step_by_step(path):
    """Every schedule() of the seeded programs returns what the reference's DecodingScheduler returned: the same
    groups in the same order, prompt / decode, chunk sizes, do_sample, block tables, computed-prefix blocks, swap-in /
    swap-out / copy lists, ignored requests, token counts, preemption counters and free-block counts (integer work:
    exact).  Default and chunked-prefill policies, v1 / v2 managers, recompute and swap preemption on pools too small
    for their load, prefix caching, lookahead slots; synchronous stepping (see tests/sched_driver.py)."""
    import bm_driver
    import sched_driver
    with open(path) as f:
        gold = _json.load(f)

    def replay_free_hook(bm, recorded_order):
        """The reference's v1 manager releases a freed table's blocks in `set()` order (object addresses): the
        trace carries the order it used in each step and the replay imposes it, table by table, after checking it
        names the same blocks."""
        if not hasattr(bm, "_free_order"):
            return None
        default = bm._free_order
        log = [b for _dev, b in (recorded_order or [])]
        pos = [0]

        def order(blocks):
            n = len(set(blocks))
            want = log[pos[0]:pos[0] + n]
            pos[0] += n
            assert sorted(want) == sorted(set(blocks)), (want, blocks)
            return want
        if recorded_order:
            bm._free_order = order

        def finish():
            bm._free_order = default
            assert not recorded_order or pos[0] == len(log), "the reference released blocks this replay did not"
            return recorded_order
        return finish

    trace = sched_driver.run_program(_make_product_scheduler, bm_driver.ProductAdapter(), gold["config"],
                                     free_hook=replay_free_hook, recorded=gold["trace"])
    assert len(trace) == len(gold["trace"]), (len(trace), len(gold["trace"]))
    for i, (got, want) in enumerate(zip(trace, gold["trace"])):
        assert got == want, f"step {i}: first difference {[k for k in want if got.get(k) != want[k]]}\n got {got}\n want {want}"


def test_scheduler_traces_exist_and_preempt():
    assert len(_SCHED_TRACES) >= 8
    preempt = swapped = 0
    for p in _SCHED_TRACES:
        with open(p) as f:
            steps = [t for t in _json.load(f)["trace"] if "groups" in t]
        preempt += steps[-1]["cumulative_preemption"]
        swapped += sum(len(t["swap_out"]) for t in steps)
    assert preempt > 100 and swapped > 100
