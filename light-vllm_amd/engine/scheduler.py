"""Continuous-batching scheduler of the decode engine: the caller of the block manager.

Policy and call protocol of light_vllm/decoding/scheduler.py:235-1132, restated:
  default policy (:663-740)   prefills first (a step is either all-prompt or all-decode), then
                              running decodes, then swapped-in groups; under memory pressure
                              the youngest running group is preempted (recompute for single-
                              sequence groups, swap otherwise, :1003-1028)
  chunked prefill (:742-815)  decodes first, then unfinished prefills, swapped, new prefills,
                              all inside one token budget; prompts are cut to the budget
  async overlap               `schedule()` marks every scheduled group busy (:874) and
                              `_schedule_running` steps over busy groups (:388-391);
                              `free_finished_request` clears the flag (:939-951)
The block lists it emits (blocks_to_swap_in / swap_out / copy) are what CacheEngine executes
before the forward pass; the block tables it attaches are what the attention kernels walk.
"""
import enum
import time
from collections import deque
from dataclasses import dataclass, field
from typing import Deque, Dict, Iterable, List, Optional, Set, Tuple

from ..block_manager.interfaces import AllocStatus, BlockSpaceManager
from ..block_manager.v1 import BlockSpaceManagerV1
from .config import CacheConfig, SchedulerConfig
from .sequence import Sequence, SequenceData, SequenceGroup, SequenceStatus


class PreemptionMode(enum.Enum):
    SWAP = enum.auto()       # blocks go to CPU memory and come back later
    RECOMPUTE = enum.auto()  # blocks are dropped, the sequence is prefilled again


class SchedulingBudget:
    """Token and sequence slots of one step; updates are idempotent per request id
    (scheduler.py:42-98)."""

    def __init__(self, token_budget: int, max_num_seqs: int) -> None:
        self.token_budget = token_budget
        self.max_num_seqs = max_num_seqs
        self._tok_ids: Set[str] = set()
        self._seq_ids: Set[str] = set()
        self.num_batched_tokens = 0
        self.num_curr_seqs = 0

    def can_schedule(self, *, num_new_tokens: int, num_new_seqs: int) -> bool:
        assert num_new_tokens != 0 and num_new_seqs != 0
        return (self.num_batched_tokens + num_new_tokens <= self.token_budget
                and self.num_curr_seqs + num_new_seqs <= self.max_num_seqs)

    def remaining_token_budget(self) -> int:
        return self.token_budget - self.num_batched_tokens

    def add_num_batched_tokens(self, req_id: str, n: int) -> None:
        if req_id not in self._tok_ids:
            self._tok_ids.add(req_id)
            self.num_batched_tokens += n

    def subtract_num_batched_tokens(self, req_id: str, n: int) -> None:
        if req_id in self._tok_ids:
            self._tok_ids.remove(req_id)
            self.num_batched_tokens -= n

    def add_num_seqs(self, req_id: str, n: int) -> None:
        if req_id not in self._seq_ids:
            self._seq_ids.add(req_id)
            self.num_curr_seqs += n

    def subtract_num_seqs(self, req_id: str, n: int) -> None:
        if req_id in self._seq_ids:
            self._seq_ids.remove(req_id)
            self.num_curr_seqs -= n


@dataclass
class ScheduledSequenceGroup:
    seq_group: SequenceGroup
    token_chunk_size: int  # tokens of this group processed in the step (1 for decode)


@dataclass
class SequenceGroupMetadata:
    """What the input builder needs for one scheduled group (sequence.py:577-643)."""
    request_id: str
    is_prompt: bool
    seq_data: Dict[int, SequenceData]
    block_tables: Dict[int, List[int]]
    do_sample: bool
    token_chunk_size: int
    computed_block_nums: List[int]


@dataclass
class SchedulerOutput:
    scheduled_seq_groups: List[ScheduledSequenceGroup]
    num_prefill_groups: int
    num_batched_tokens: int
    blocks_to_swap_in: List[Tuple[int, int]]
    blocks_to_swap_out: List[Tuple[int, int]]
    blocks_to_copy: List[Tuple[int, int]]
    ignored_seq_groups: List[SequenceGroup]
    num_lookahead_slots: int
    running_queue_size: int
    preempted: int
    seq_group_metadata_list: List[SequenceGroupMetadata] = field(default_factory=list)

    def is_empty(self) -> bool:
        return (not self.scheduled_seq_groups and not self.blocks_to_swap_in
                and not self.blocks_to_swap_out and not self.blocks_to_copy)


class DecodingScheduler:

    def __init__(self, scheduler_config: SchedulerConfig, cache_config: CacheConfig,
                 chunked_prefill_enabled: bool = False) -> None:
        self.scheduler_config = scheduler_config
        self.cache_config = cache_config
        self.chunked_prefill_enabled = chunked_prefill_enabled
        cls = BlockSpaceManager.get_block_space_manager_class(
            "v2" if scheduler_config.use_v2_block_manager else "v1")
        self.block_manager = cls(block_size=cache_config.block_size,
                                 num_gpu_blocks=cache_config.num_gpu_blocks,
                                 num_cpu_blocks=cache_config.num_cpu_blocks or 0,
                                 sliding_window=cache_config.sliding_window,
                                 enable_caching=cache_config.enable_prefix_caching)
        self.waiting: Deque[SequenceGroup] = deque()
        self.running: Deque[SequenceGroup] = deque()
        self.swapped: Deque[SequenceGroup] = deque()
        self.user_specified_preemption_mode = scheduler_config.preemption_mode
        self.num_cumulative_preemption = 0

    # ---- queue management ----
    def add_request(self, seq_group: SequenceGroup) -> None:
        self.waiting.append(seq_group)

    def abort_request(self, request_id) -> None:
        ids = {request_id} if isinstance(request_id, str) else set(request_id)
        for queue in (self.waiting, self.running, self.swapped):
            hit = [g for g in queue if g.request_id in ids]
            for g in hit:
                queue.remove(g)
                ids.discard(g.request_id)
                for seq in g.get_seqs():
                    if not seq.is_finished():
                        seq.status = SequenceStatus.FINISHED_ABORTED
                        self.free_seq(seq)

    def has_unfinished_requests(self) -> bool:
        return bool(self.waiting or self.running or self.swapped)

    def get_num_unfinished_requests(self) -> int:
        return len(self.waiting) + len(self.running) + len(self.swapped)

    # ---- the three sources of work ----
    def _get_num_new_tokens(self, seq_group: SequenceGroup, status: SequenceStatus,
                            enable_chunking: bool, budget: SchedulingBudget) -> int:
        seqs = seq_group.get_seqs(status=status)
        num_new_tokens = seqs[0].get_num_new_tokens() if len(seqs) == 1 else sum(s.get_num_new_tokens() for s in seqs)
        assert num_new_tokens > 0
        # only single-sequence groups (prompts) are ever chunked (scheduler.py:1102-1132)
        if enable_chunking and len(seqs) == 1:
            num_new_tokens = min(num_new_tokens, budget.remaining_token_budget())
        return num_new_tokens

    def _schedule_running(self, budget: SchedulingBudget, enable_chunking: bool = False):
        blocks_to_swap_out: List[Tuple[int, int]] = []
        blocks_to_copy: List[Tuple[int, int]] = []
        busy: List[SequenceGroup] = []
        decodes: List[ScheduledSequenceGroup] = []
        prefills: List[ScheduledSequenceGroup] = []
        preempted: List[SequenceGroup] = []
        swapped_out: List[SequenceGroup] = []
        q = self.running
        num_scheduled_seqs = 0
        deferred: List[SequenceGroup] = []
        lookahead = self._get_num_lookahead_slots(is_prefill=False)
        while q:
            seq_group = q[0]
            if seq_group.busy:  # a step holding this group is still in flight
                q.popleft()
                busy.append(seq_group)
                continue
            # The common case -- one running sequence in its decode stage, no chunking -- is the general
            # body below with n_seqs = 1 and num_running_tokens = 1, without its list building.
            seqs = seq_group.seqs
            if (len(seqs) == 1 and seq_group.n <= 1
                    and seqs[0].status == SequenceStatus.RUNNING and not seqs[0].is_prefill()
                    and num_scheduled_seqs + 1 <= budget.max_num_seqs
                    and (not enable_chunking or budget.remaining_token_budget() >= 1)
                    and self.block_manager.can_append_slots(seq_group, lookahead)):
                q.popleft()
                blocks_to_copy.extend(self.block_manager.append_slots(seqs[0], lookahead))
                num_scheduled_seqs += 1
                decodes.append(ScheduledSequenceGroup(seq_group, 1))
                budget.add_num_batched_tokens(seq_group.request_id, 1)
                if enable_chunking:
                    budget.add_num_seqs(seq_group.request_id, 1)
                continue
            # One step never carries more than max_num_seqs sequences.  The reference relies on
            # admission control for this (prompts are admitted under the same budget); here it is
            # enforced, so that more idle running groups than one step can hold simply wait.
            n_seqs = seq_group.get_max_num_running_seqs()
            if num_scheduled_seqs + n_seqs > budget.max_num_seqs:
                q.popleft()
                deferred.append(seq_group)
                continue
            num_running_tokens = self._get_num_new_tokens(seq_group, SequenceStatus.RUNNING,
                                                          enable_chunking, budget)
            if num_running_tokens == 0:
                break
            q.popleft()
            while not self.block_manager.can_append_slots(seq_group, lookahead):
                budget.subtract_num_batched_tokens(seq_group.request_id, num_running_tokens)
                budget.subtract_num_seqs(seq_group.request_id, seq_group.get_max_num_running_seqs())
                # evict the most recently arrived group first -- but never one that a step still in
                # flight is computing on (the reference pops the tail whatever its `busy` flag,
                # scheduler.py:408-416: under memory pressure with two steps in flight it then
                # frees the blocks of a sequence the GPU is writing to)
                victim = None
                for idx in range(len(q) - 1, -1, -1):
                    if not q[idx].busy:
                        victim = q[idx]
                        del q[idx]
                        break
                if victim is not None:
                    mode = self._preempt(victim, blocks_to_swap_out)
                    (preempted if mode == PreemptionMode.RECOMPUTE else swapped_out).append(victim)
                else:  # nothing else to evict: this group itself goes
                    mode = self._preempt(seq_group, blocks_to_swap_out)
                    (preempted if mode == PreemptionMode.RECOMPUTE else swapped_out).append(seq_group)
                    break
            else:
                self._append_slots(seq_group, blocks_to_copy)
                num_scheduled_seqs += n_seqs
                if seq_group.is_prefill():
                    prefills.append(ScheduledSequenceGroup(seq_group, num_running_tokens))
                else:
                    decodes.append(ScheduledSequenceGroup(seq_group, 1))
                budget.add_num_batched_tokens(seq_group.request_id, num_running_tokens)
                if enable_chunking:
                    budget.add_num_seqs(seq_group.request_id, seq_group.get_max_num_running_seqs())
        self.running.extend(deferred)  # ahead of the busy groups: they run in the next step
        self.running.extend(busy)
        return decodes, prefills, preempted, swapped_out, blocks_to_swap_out, blocks_to_copy

    def _schedule_swapped(self, budget: SchedulingBudget, enable_chunking: bool = False):
        blocks_to_swap_in: List[Tuple[int, int]] = []
        blocks_to_copy: List[Tuple[int, int]] = []
        decodes: List[ScheduledSequenceGroup] = []
        prefills: List[ScheduledSequenceGroup] = []
        infeasible: List[SequenceGroup] = []
        q = self.swapped
        while q:
            seq_group = q[0]
            status = self.block_manager.can_swap_in(seq_group, self._get_num_lookahead_slots(seq_group.is_prefill()))
            if status == AllocStatus.LATER:
                break
            if status == AllocStatus.NEVER:
                for seq in seq_group.get_seqs():
                    seq.status = SequenceStatus.FINISHED_IGNORED
                infeasible.append(seq_group)
                q.popleft()
                continue
            num_new_seqs = seq_group.get_max_num_running_seqs()
            num_new_tokens = self._get_num_new_tokens(seq_group, SequenceStatus.SWAPPED,
                                                      enable_chunking, budget)
            if num_new_tokens == 0 or not budget.can_schedule(num_new_tokens=num_new_tokens,
                                                              num_new_seqs=num_new_seqs):
                break
            q.popleft()
            blocks_to_swap_in.extend(self.block_manager.swap_in(seq_group))
            for seq in seq_group.get_seqs(status=SequenceStatus.SWAPPED):
                seq.status = SequenceStatus.RUNNING
            self._append_slots(seq_group, blocks_to_copy)
            if seq_group.is_prefill():
                prefills.append(ScheduledSequenceGroup(seq_group, num_new_tokens))
            else:
                decodes.append(ScheduledSequenceGroup(seq_group, 1))
            budget.add_num_batched_tokens(seq_group.request_id, num_new_tokens)
            budget.add_num_seqs(seq_group.request_id, num_new_seqs)
        return decodes, prefills, blocks_to_swap_in, blocks_to_copy, infeasible

    def _get_prompt_limit(self) -> int:
        if self.chunked_prefill_enabled:
            return self.scheduler_config.max_model_len
        return min(self.scheduler_config.max_model_len, self.scheduler_config.max_num_batched_tokens)

    def _schedule_prefills(self, budget: SchedulingBudget, enable_chunking: bool = False):
        ignored: List[SequenceGroup] = []
        scheduled: List[ScheduledSequenceGroup] = []
        q = self.waiting
        while q:
            seq_group = q[0]
            waiting_seqs = seq_group.get_seqs(status=SequenceStatus.WAITING)
            assert len(waiting_seqs) == 1, "Waiting sequence group should have only one prompt sequence."
            num_new_tokens = self._get_num_new_tokens(seq_group, SequenceStatus.WAITING,
                                                      enable_chunking, budget)
            if not enable_chunking:
                assert num_new_tokens == waiting_seqs[0].get_len()
            if num_new_tokens > self._get_prompt_limit():
                for seq in waiting_seqs:
                    seq.status = SequenceStatus.FINISHED_IGNORED
                ignored.append(seq_group)
                q.popleft()
                continue
            can_allocate = self.block_manager.can_allocate(seq_group)
            if can_allocate == AllocStatus.LATER:
                break
            if can_allocate == AllocStatus.NEVER:
                for seq in waiting_seqs:
                    seq.status = SequenceStatus.FINISHED_IGNORED
                ignored.append(seq_group)
                q.popleft()
                continue
            num_new_seqs = seq_group.get_max_num_running_seqs()
            if num_new_tokens == 0 or not budget.can_schedule(num_new_tokens=num_new_tokens,
                                                              num_new_seqs=num_new_seqs):
                break
            q.popleft()
            self.block_manager.allocate(seq_group)
            for seq in waiting_seqs:
                seq.status = SequenceStatus.RUNNING
            scheduled.append(ScheduledSequenceGroup(seq_group, num_new_tokens))
            budget.add_num_batched_tokens(seq_group.request_id, num_new_tokens)
            budget.add_num_seqs(seq_group.request_id, num_new_seqs)
        return scheduled, ignored

    # ---- the two policies ----
    def _schedule_default(self) -> SchedulerOutput:
        cfg = self.scheduler_config
        budget = SchedulingBudget(cfg.max_num_batched_tokens, cfg.max_num_seqs)
        for g in self.running:
            if not g.busy:
                budget.add_num_seqs(g.request_id, g.get_max_num_running_seqs())
        prefills: List[ScheduledSequenceGroup] = []
        ignored: List[SequenceGroup] = []
        run = ([], [], [], [], [], [])
        swp = ([], [], [], [], [])
        if not self.swapped:
            prefills, ignored = self._schedule_prefills(budget, enable_chunking=False)
        if not prefills:
            run = self._schedule_running(budget, enable_chunking=False)
            if len(run[2]) + len(run[3]) == 0:  # nothing was preempted: try to bring groups back
                swp = self._schedule_swapped(budget)
        decodes, run_prefills, preempted, swapped_out, swap_out_blocks, copy_blocks = run
        sw_decodes, sw_prefills, swap_in_blocks, sw_copy, infeasible = swp
        assert budget.num_batched_tokens <= cfg.max_num_batched_tokens
        assert sum(s.seq_group.get_max_num_running_seqs()
                   for s in prefills + decodes + sw_decodes) <= max(cfg.max_num_seqs, 1) or not decodes
        assert not run_prefills and not sw_prefills
        self.waiting.extendleft(preempted)
        self.running.extend(s.seq_group for s in prefills)
        self.running.extend(s.seq_group for s in decodes)
        self.running.extend(s.seq_group for s in sw_decodes)
        self.swapped.extend(swapped_out)
        return SchedulerOutput(
            scheduled_seq_groups=prefills + decodes + sw_decodes, num_prefill_groups=len(prefills),
            num_batched_tokens=budget.num_batched_tokens, blocks_to_swap_in=swap_in_blocks,
            blocks_to_swap_out=swap_out_blocks, blocks_to_copy=copy_blocks + sw_copy,
            ignored_seq_groups=ignored + infeasible,
            num_lookahead_slots=self._get_num_lookahead_slots(is_prefill=bool(prefills)),
            running_queue_size=len(self.running), preempted=len(preempted) + len(swapped_out))

    def _schedule_chunked_prefill(self) -> SchedulerOutput:
        cfg = self.scheduler_config
        budget = SchedulingBudget(cfg.max_num_batched_tokens, cfg.max_num_seqs)
        decodes, run_prefills, preempted, swapped_out, swap_out_blocks, copy_blocks = \
            self._schedule_running(budget, enable_chunking=True)
        swp = ([], [], [], [], [])
        if len(preempted) + len(swapped_out) == 0:
            swp = self._schedule_swapped(budget)
        sw_decodes, sw_prefills, swap_in_blocks, sw_copy, infeasible = swp
        prefills, ignored = self._schedule_prefills(budget, enable_chunking=True)
        assert budget.num_batched_tokens <= cfg.max_num_batched_tokens
        assert budget.num_curr_seqs <= cfg.max_num_seqs
        self.waiting.extendleft(preempted)
        for lst in (sw_decodes, sw_prefills, decodes, run_prefills, prefills):
            self.running.extend(s.seq_group for s in lst)
        self.swapped.extend(swapped_out)
        return SchedulerOutput(
            scheduled_seq_groups=prefills + run_prefills + sw_prefills + decodes + sw_decodes,
            num_prefill_groups=len(prefills) + len(sw_prefills) + len(run_prefills),
            num_batched_tokens=budget.num_batched_tokens, blocks_to_swap_in=swap_in_blocks,
            blocks_to_swap_out=swap_out_blocks, blocks_to_copy=copy_blocks + sw_copy,
            ignored_seq_groups=ignored + infeasible,
            num_lookahead_slots=self._get_num_lookahead_slots(is_prefill=False),
            running_queue_size=len(self.running), preempted=len(preempted) + len(swapped_out))

    def need_scheduling(self) -> bool:
        if self.waiting or self.swapped:
            return True
        return any(not g.busy for g in self.running)

    def schedule(self) -> Optional[SchedulerOutput]:
        if not self.need_scheduling():
            return None
        out = self._schedule_chunked_prefill() if self.chunked_prefill_enabled else self._schedule_default()
        now = time.time()
        metas: List[SequenceGroupMetadata] = []
        # without prefix caching the manager's three cache hooks below are no-ops (v1: block_manager_v1.py
        # :640-707 all start with `if self.enable_caching`): skip the calls, not the semantics
        bm = self.block_manager
        hooks = not (isinstance(bm, BlockSpaceManagerV1) and not bm.enable_caching)
        for sched in out.scheduled_seq_groups:
            g = sched.seq_group
            g.maybe_set_first_scheduled_time(now)
            g.busy = True
            seqs = g.seqs
            if not hooks and len(seqs) == 1 and seqs[0].status == SequenceStatus.RUNNING and not seqs[0].is_prefill():
                seq = seqs[0]  # one decoding sequence, no prefix-cache hooks: the record below, directly
                metas.append(SequenceGroupMetadata(
                    request_id=g.request_id, is_prompt=False, seq_data={seq.seq_id: seq.data},
                    block_tables={seq.seq_id: bm.get_block_table(seq)}, do_sample=True,
                    token_chunk_size=sched.token_chunk_size, computed_block_nums=[]))
                continue
            seq_data: Dict[int, SequenceData] = {}
            block_tables: Dict[int, List[int]] = {}
            running = g.get_seqs(status=SequenceStatus.RUNNING)
            for seq in running:
                seq_data[seq.seq_id] = seq.data
                block_tables[seq.seq_id] = bm.get_block_table(seq)
                if hooks:
                    bm.access_all_blocks_in_seq(seq, now)
            common = list(bm.get_common_computed_block_ids(running)) if hooks else []
            do_sample = True
            if g.is_prefill():
                seqs = g.get_seqs()
                assert len(seqs) == 1
                # a prompt chunk that does not reach the end of the prompt samples nothing
                if sched.token_chunk_size + seqs[0].data.get_num_computed_tokens() < seqs[0].data.get_len():
                    do_sample = False
            metas.append(SequenceGroupMetadata(
                request_id=g.request_id, is_prompt=g.is_prefill(), seq_data=seq_data,
                block_tables=block_tables, do_sample=do_sample,
                token_chunk_size=sched.token_chunk_size, computed_block_nums=common))
        if hooks:
            for sched in out.scheduled_seq_groups:
                bm.mark_blocks_as_computed(sched.seq_group)
        out.seq_group_metadata_list = metas
        return out

    # ---- helpers shared with the output processor ----
    def fork_seq(self, parent_seq: Sequence, child_seq: Sequence) -> None:
        self.block_manager.fork(parent_seq, child_seq)

    def free_seq(self, seq: Sequence) -> None:
        self.block_manager.free(seq)

    def free_finished_request(self, request_ids: Iterable[str]) -> None:
        """Called once the outputs of a step were processed: finished groups leave the
        running queue, the others become schedulable again."""
        ids = set(request_ids)
        remaining: Deque[SequenceGroup] = deque()
        for g in self.running:
            if not g.is_finished():
                remaining.append(g)
            if g.request_id in ids:
                g.busy = False
        self.running = remaining

    def _get_num_lookahead_slots(self, is_prefill: bool) -> int:
        """Slots to allocate per sequence per step beyond the known token ids: none for prompts, the
        configured number for decoding sequences (scheduler.py:1095-1106)."""
        if is_prefill:
            return 0
        return self.scheduler_config.num_lookahead_slots

    def _append_slots(self, seq_group: SequenceGroup, blocks_to_copy: List[Tuple[int, int]]) -> None:
        num_lookahead_slots = self._get_num_lookahead_slots(is_prefill=False)  # scheduler.py:978-982
        for seq in seq_group.get_seqs(status=SequenceStatus.RUNNING):
            blocks_to_copy.extend(self.block_manager.append_slots(seq, num_lookahead_slots))

    def _preempt(self, seq_group: SequenceGroup, blocks_to_swap_out: List[Tuple[int, int]]) -> PreemptionMode:
        if self.user_specified_preemption_mode is None:
            # recomputation is cheaper for one sequence; groups of several cannot be recomputed
            mode = (PreemptionMode.RECOMPUTE if seq_group.get_max_num_running_seqs() == 1
                    else PreemptionMode.SWAP)
        elif self.user_specified_preemption_mode == "swap":
            mode = PreemptionMode.SWAP
        else:
            mode = PreemptionMode.RECOMPUTE
        self.num_cumulative_preemption += 1
        if mode == PreemptionMode.RECOMPUTE:
            seqs = seq_group.get_seqs(status=SequenceStatus.RUNNING)
            assert len(seqs) == 1
            for seq in seqs:
                seq.status = SequenceStatus.WAITING
                self.free_seq(seq)
                seq.reset_state_for_recompute()
        else:
            if not self.block_manager.can_swap_out(seq_group):
                raise RuntimeError("Aborted due to the lack of CPU swap space. Please increase "
                                   "the swap space to avoid this error.")
            blocks_to_swap_out.extend(self.block_manager.swap_out(seq_group))
            for seq in seq_group.get_seqs(status=SequenceStatus.RUNNING):
                seq.status = SequenceStatus.SWAPPED
        return mode
