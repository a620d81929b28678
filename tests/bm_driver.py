"""Seeded random programs over a BlockSpaceManager, recorded as JSON traces.

The same program runs against the reference's block manager (oracle/make_golden.py, in the dev
container, to write tests/golden/block_manager_*.json) and against this package's block manager
(tests/test_block_manager.py, to compare every observation bit for bit): every block table
after every operation, every AllocStatus / bool verdict, every CoW and swap pair, the free
block counts, the common computed block ids.

The program follows the call protocol of the decoding scheduler
(light_vllm/decoding/scheduler.py): can_allocate -> allocate -> [append token; can_append_slots
-> append_slots]* with fork / swap_out / swap_in / free interleaved, and after each "schedule"
access_all_blocks_in_seq + get_common_computed_block_ids + mark_blocks_as_computed
(scheduler.py:881-926).

`adapter` hides which Sequence implementation is in use.
"""
import json
import random
import zlib
from typing import Any, Dict, List, Optional


class ProductAdapter:
    """This package's Sequence / SequenceGroup."""

    def __init__(self):
        from light_vllm_amd.engine.sequence import Sequence, SequenceGroup, SequenceStatus
        self.Sequence, self.SequenceGroup, self.S = Sequence, SequenceGroup, SequenceStatus

    def seq(self, seq_id, tokens, block_size):
        return self.Sequence(seq_id, list(tokens), block_size)

    def group(self, request_id, seqs):
        return self.SequenceGroup(request_id, seqs, 0.0)

    def append(self, seq, token):
        seq.append_token_id(token, 0.0)

    def fork(self, seq, new_id):
        return seq.fork(new_id)

    def status(self, name):
        return getattr(self.S, name)

    def free_finished(self, scheduler, finished):
        scheduler.free_finished_request([f.request_id for f in finished])


class ReferenceAdapter:
    """The reference's Sequence / SequenceGroup (via oracle/ref_block_manager.py)."""

    def __init__(self, ns):
        self.ns = ns

    def seq(self, seq_id, tokens, block_size):
        return self.ns.Sequence(seq_id, self.ns.TextOnlyInputs(prompt_token_ids=list(tokens), prompt=None),
                                block_size)

    def group(self, request_id, seqs):
        return self.ns.SequenceGroup(request_id, seqs, 0.0)

    def append(self, seq, token):
        seq.append_token_id(token, {token: self.ns.Logprob(0.0)})

    def fork(self, seq, new_id):
        return seq.fork(new_id)

    def status(self, name):
        return getattr(self.ns.SequenceStatus, name)

    def free_finished(self, scheduler, finished):
        scheduler.free_finished_request(finished)  # the reference takes RequestOutputs (anything with .request_id)


def run_program(make_manager, adapter, config: Dict[str, Any], seed: int, num_ops: int,
                free_hook=None, recorded: Optional[List[dict]] = None) -> List[dict]:
    """Runs the seeded program; returns the list of observations (one dict per operation).

    free_hook(manager, op_index, recorded_order|None) -> callable() returning the order in
    which the manager released blocks during the op just executed, or None.  It lets the
    recorder log the reference's `set()`-ordered frees and the replayer impose that order.
    """
    rng = random.Random(seed)
    bs = config["block_size"]
    bm = make_manager(config)
    trace: List[dict] = []
    groups: Dict[str, Any] = {}
    next_seq_id = [0]
    next_group = [0]
    clock = [0.0]
    vocab = 50
    shared_prefixes = [[rng.randrange(vocab) for _ in range(rng.choice([bs, 2 * bs, 3 * bs + 3]))]
                       for _ in range(3)]
    S = adapter.status

    def live_seqs():
        out = []
        for g in groups.values():
            for s in g.get_seqs():
                if not s.is_finished():
                    out.append(s)
        return out

    def observe(op: dict, hint_idx: int):
        tables = {}
        for s in live_seqs():
            try:
                tables[str(s.seq_id)] = list(bm.get_block_table(s))
            except KeyError:
                pass
        # tables of the sequences this operation touched, in full; all others by checksum
        touched = set()
        if "group" in op and op["group"] in groups:
            touched = {str(s.seq_id) for s in groups[op["group"]].get_seqs()}
        op["tables"] = {k: v for k, v in tables.items() if k in touched}
        op["all_tables_crc"] = zlib.crc32(json.dumps(tables, sort_keys=True).encode())
        op["free_gpu"] = bm.get_num_free_gpu_blocks()
        op["free_cpu"] = bm.get_num_free_cpu_blocks()
        trace.append(op)

    def schedule_side_effects(g, op):
        clock[0] += 1.0
        running = g.get_seqs(status=S("RUNNING"))
        for s in running:
            bm.access_all_blocks_in_seq(s, clock[0])
        op["common_computed"] = list(bm.get_common_computed_block_ids(running))
        bm.mark_blocks_as_computed(g)

    for op_idx in range(num_ops):
        rec = recorded[op_idx] if recorded is not None and op_idx < len(recorded) else None
        hook = free_hook(bm, op_idx, rec.get("free_order") if rec else None) if free_hook else None
        running_groups = [g for g in groups.values() if g.get_seqs(status=S("RUNNING"))]
        swapped_groups = [g for g in groups.values() if g.get_seqs(status=S("SWAPPED"))]
        r = rng.random()
        op: Dict[str, Any] = {}
        if r < 0.22 or not groups:
            # ---- new prompt ----
            if rng.random() < 0.5:
                toks = list(rng.choice(shared_prefixes)) + [rng.randrange(vocab) for _ in range(rng.randrange(0, 2 * bs))]
            else:
                toks = [rng.randrange(vocab) for _ in range(rng.randrange(1, 5 * bs))]
            sid = next_seq_id[0]
            next_seq_id[0] += 1
            gid = str(next_group[0])
            next_group[0] += 1
            seq = adapter.seq(sid, toks, bs)
            g = adapter.group(gid, [seq])
            verdict = bm.can_allocate(g).name
            op.update(op="allocate", group=gid, seq=sid, tokens=toks, verdict=verdict)
            if verdict == "OK":
                bm.allocate(g)
                seq.status = S("RUNNING")
                groups[gid] = g
                # the prompt step: computed tokens advance, first output token is sampled
                schedule_side_effects(g, op)
                seq.data.update_num_computed_tokens(len(toks))
                adapter.append(seq, rng.randrange(vocab))
        elif r < 0.62 and running_groups:
            # ---- decode step of one running group ----
            g = rng.choice(running_groups)
            # "lookahead" programs (multi-step decode): the manager reserves `la` slots beyond the known
            # tokens (block_manager_v2.py:183-239) and the step then appends 1 .. la + 1 tokens at once
            la = config.get("lookahead", 0)
            ok = bm.can_append_slots(g, la)
            op.update(op="decode", group=g.request_id, can_append=bool(ok))
            if la:
                op["lookahead"] = la
            if ok:
                cows = []
                for s in g.get_seqs(status=S("RUNNING")):
                    cows.extend([list(p) for p in bm.append_slots(s, la)])
                op["cows"] = cows
                schedule_side_effects(g, op)
                n_new = rng.randint(1, la + 1) if la else 1
                for s in g.get_seqs(status=S("RUNNING")):
                    for _ in range(n_new):
                        s.data.update_num_computed_tokens(1)
                        adapter.append(s, rng.randrange(vocab))
        elif r < 0.70 and running_groups and not config.get("no_fork"):
            # ---- fork (parallel sampling / beam): child shares the parent's blocks ----
            g = rng.choice(running_groups)
            parent = rng.choice(g.get_seqs(status=S("RUNNING")))
            child = adapter.fork(parent, next_seq_id[0])
            next_seq_id[0] += 1
            g.add(child)
            bm.fork(parent, child)
            op.update(op="fork", group=g.request_id, parent=parent.seq_id, child=child.seq_id)
        elif r < 0.78 and running_groups and config["num_cpu_blocks"] > 0:
            # ---- preemption by swap ----
            g = rng.choice(running_groups)
            ok = bm.can_swap_out(g)
            op.update(op="swap_out", group=g.request_id, can=bool(ok))
            if ok:
                op["mapping"] = [list(p) for p in bm.swap_out(g)]
                for s in g.get_seqs(status=S("RUNNING")):
                    s.status = S("SWAPPED")
        elif r < 0.88 and swapped_groups:
            g = rng.choice(swapped_groups)
            verdict = bm.can_swap_in(g, config.get("lookahead", 0)).name
            op.update(op="swap_in", group=g.request_id, verdict=verdict)
            if verdict == "OK":
                op["mapping"] = [list(p) for p in bm.swap_in(g)]
                for s in g.get_seqs(status=S("SWAPPED")):
                    s.status = S("RUNNING")
        else:
            # ---- a sequence finishes (or a whole group is preempted by recompute) ----
            cands = live_seqs()
            if config.get("free_running_only"):
                cands = [s for s in cands if s.status == S("RUNNING")]
            if not cands:
                op.update(op="noop")
            else:
                s = rng.choice(cands)
                s.status = S("FINISHED_STOPPED")
                bm.free(s)
                op.update(op="free", seq=s.seq_id)
                for gid in [k for k, g in groups.items() if g.is_finished()]:
                    del groups[gid]
        if hook is not None:
            order = hook()
            if order:
                op["free_order"] = order
        observe(op, op_idx)
    return trace


DEFAULT_CONFIGS = [
    # name, config, seed, num_ops
    ("v1_uncached", dict(version="v1", block_size=16, num_gpu_blocks=96, num_cpu_blocks=32, watermark=0.01,
                         sliding_window=None, enable_caching=False), 1, 400),
    ("v1_uncached_tight", dict(version="v1", block_size=8, num_gpu_blocks=40, num_cpu_blocks=16, watermark=0.05,
                               sliding_window=None, enable_caching=False), 2, 400),
    ("v1_cached", dict(version="v1", block_size=16, num_gpu_blocks=64, num_cpu_blocks=24, watermark=0.01,
                       sliding_window=None, enable_caching=True), 3, 500),
    ("v1_cached_tight", dict(version="v1", block_size=4, num_gpu_blocks=48, num_cpu_blocks=48, watermark=0.0,
                             sliding_window=None, enable_caching=True), 4, 600),
    # no swap space: the reference double-frees when it swaps a sliding-window table
    ("v1_sliding_window", dict(version="v1", block_size=8, num_gpu_blocks=64, num_cpu_blocks=0, watermark=0.01,
                               sliding_window=20, enable_caching=False), 5, 400),
]


# version 2 (block_manager_v2.py).  Options avoid states in which the REFERENCE itself raises:
#   fork and swap never meet in one program: v2 swaps a group sequence by sequence and allocates a
#     destination block per sequence even for blocks the group shares, while can_swap_out counts each
#     shared block once -> NoFreeBlocksError inside swap_out (cpu_gpu_block_allocator.py:236-262)
#   free_running_only: freeing a swapped-out sequence with prefix caching stamps CPU block ids on the
#     GPU allocator's tracker (KeyError, block_manager_v2.py:241-247)
#   sliding window: a table holds null blocks that fork and swap cannot handle
V2_CONFIGS = [
    ("v2_naive_fork", dict(version="v2", block_size=16, num_gpu_blocks=96, num_cpu_blocks=0, watermark=0.01,
                           sliding_window=None, enable_caching=False), 11, 400),
    ("v2_naive_swap", dict(version="v2", block_size=8, num_gpu_blocks=40, num_cpu_blocks=24, watermark=0.05,
                           sliding_window=None, enable_caching=False, no_fork=True), 12, 500),
    ("v2_cached_fork", dict(version="v2", block_size=16, num_gpu_blocks=64, num_cpu_blocks=0, watermark=0.01,
                            sliding_window=None, enable_caching=True), 13, 500),
    ("v2_cached_swap", dict(version="v2", block_size=4, num_gpu_blocks=48, num_cpu_blocks=48, watermark=0.0,
                            sliding_window=None, enable_caching=True, free_running_only=True, no_fork=True), 14, 700),
    ("v2_sliding_window", dict(version="v2", block_size=8, num_gpu_blocks=64, num_cpu_blocks=0, watermark=0.01,
                               sliding_window=20, enable_caching=False, no_fork=True), 15, 400),
    # multi-step decode: lookahead slots reserved at every decode, several tokens appended per step
    # (no swap space: the reference's own swap_out raises IndexError on a table that ends in reserved,
    # still empty lookahead blocks -- block/common.py:207 via naive_block.py:335)
    ("v2_naive_lookahead3", dict(version="v2", block_size=8, num_gpu_blocks=56, num_cpu_blocks=0, watermark=0.05,
                                 sliding_window=None, enable_caching=False, no_fork=True, lookahead=3), 16, 500),
    ("v2_naive_lookahead7_fork", dict(version="v2", block_size=16, num_gpu_blocks=96, num_cpu_blocks=0, watermark=0.01,
                                      sliding_window=None, enable_caching=False, lookahead=7), 17, 400),
    # multi-step decode over the prefix-caching allocator (block_manager_v2.py:199-237 with
    # PrefixCachingBlockAllocator): lookahead blocks are mutable, content hashes appear as blocks fill up
    ("v2_cached_lookahead3", dict(version="v2", block_size=8, num_gpu_blocks=64, num_cpu_blocks=0, watermark=0.02,
                                  sliding_window=None, enable_caching=True, no_fork=True, lookahead=3), 18, 600),
    ("v2_cached_lookahead7_tight", dict(version="v2", block_size=4, num_gpu_blocks=48, num_cpu_blocks=0, watermark=0.0,
                                        sliding_window=None, enable_caching=True, no_fork=True, lookahead=7), 19, 600),
]
