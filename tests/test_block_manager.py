"""Block-manager parity: this package's BlockSpaceManager replays the seeded programs of
bm_driver.py and must reproduce the traces recorded from the REFERENCE's block manager
(tests/golden/block_manager_*.json, written by oracle/make_golden.py) bit for bit: every
block table, verdict, CoW / swap pair, free-block count and computed-block list."""
import glob
import json
import os

import pytest

import bm_driver

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TRACES = sorted(glob.glob(os.path.join(GOLDEN, "block_manager_*.json")))


def make_product_manager(cfg):
    from light_vllm_amd.block_manager import BlockSpaceManager
    cls = BlockSpaceManager.get_block_space_manager_class(cfg["version"])
    return cls(block_size=cfg["block_size"], num_gpu_blocks=cfg["num_gpu_blocks"],
               num_cpu_blocks=cfg["num_cpu_blocks"], watermark=cfg["watermark"],
               sliding_window=cfg["sliding_window"], enable_caching=cfg["enable_caching"])


def replay_free_hook(bm, op_idx, recorded_order):
    """The reference releases the blocks of a freed table in `set()` order (an artefact of
    object addresses, block_manager_v1.py:553-557); the trace carries the order it happened
    to use and the replay imposes it, after checking it is a permutation of the same blocks."""
    if not hasattr(bm, "_free_order"):
        return None
    default = bm._free_order
    if recorded_order:
        def order(blocks, _rec=recorded_order):
            want = [b for _dev, b in _rec]
            assert sorted(want) == sorted(set(blocks)), (want, blocks)
            return want
        bm._free_order = order

    def finish():
        bm._free_order = default
        return recorded_order
    return finish


@pytest.mark.parametrize("path", TRACES, ids=[os.path.basename(p)[14:-5] for p in TRACES])
def test_replay_matches_reference_trace(path):
    with open(path) as f:
        gold = json.load(f)
    trace = bm_driver.run_program(make_product_manager, bm_driver.ProductAdapter(), gold["config"],
                                  gold["seed"], gold["num_ops"], free_hook=replay_free_hook,
                                  recorded=gold["trace"])
    assert len(trace) == len(gold["trace"])
    for i, (got, want) in enumerate(zip(trace, gold["trace"])):
        assert got == want, f"op {i}: {got} != {want}"


def test_traces_exist():
    assert len(TRACES) >= 5


def test_survey_known_answers():
    """SURVEY.md §8c sample: block 16, 64 GPU + 8 CPU blocks, prompts of 5/16/33 tokens
    range(n), 20 decode appends each, then free sequence 1."""
    from light_vllm_amd.block_manager.v1 import BlockSpaceManagerV1
    from light_vllm_amd.engine.sequence import Sequence, SequenceGroup, SequenceStatus
    expected = {False: ([[63], [62], [61, 60, 59]], [[63, 57], [62, 58, 55], [61, 60, 59, 56]]),
                True: ([[0], [1], [1, 2, 3]], [[0, 5], [1, 4, 7], [1, 2, 3, 6]])}
    for caching, (first, after) in expected.items():
        bm = BlockSpaceManagerV1(16, 64, 8, enable_caching=caching)
        seqs = []
        for i, n in enumerate((5, 16, 33)):
            s = Sequence(i, list(range(n)), 16)
            g = SequenceGroup(str(i), [s])
            assert bm.can_allocate(g).name == "OK"
            bm.allocate(g)
            s.status = SequenceStatus.RUNNING
            seqs.append(s)
        assert [bm.get_block_table(s) for s in seqs] == first
        for step in range(20):
            for s in seqs:
                s.append_token_id(1000 + step)
                assert bm.append_slots(s) == []
        assert [bm.get_block_table(s) for s in seqs] == after
        bm.free(seqs[1])
        assert bm.get_num_free_gpu_blocks() == 58


def test_alloc_status_and_errors():
    from light_vllm_amd.block_manager.interfaces import AllocStatus, BlockSpaceManager
    from light_vllm_amd.block_manager.v1 import BlockSpaceManagerV1
    from light_vllm_amd.engine.sequence import Sequence, SequenceGroup, SequenceStatus
    bm = BlockSpaceManagerV1(4, 8, 2, watermark=0.25)  # 2 watermark blocks
    big = SequenceGroup("a", [Sequence(0, list(range(4 * 7)), 4)])
    assert bm.can_allocate(big) == AllocStatus.NEVER        # 8 - 7 < 2
    ok = SequenceGroup("b", [Sequence(1, list(range(4 * 5)), 4)])
    assert bm.can_allocate(ok) == AllocStatus.OK
    bm.allocate(ok)
    ok.seqs[0].status = SequenceStatus.RUNNING
    later = SequenceGroup("c", [Sequence(2, list(range(4 * 2)), 4)])
    assert bm.can_allocate(later) == AllocStatus.LATER      # 3 free - 2 < 2
    with pytest.raises(ValueError):
        BlockSpaceManager.get_block_space_manager_class("v3")
    with pytest.raises(NotImplementedError):
        BlockSpaceManagerV1(4, 8, 2, sliding_window=8, enable_caching=True)
    bm.free(ok.seqs[0])
    bm.free(ok.seqs[0])  # second free of an unknown table is a no-op (block_manager_v1.py:560-563)
    assert bm.get_num_free_gpu_blocks() == 8
