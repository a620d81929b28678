"""KV-cache block manager, version 1 (the reference's default: `use_v2_block_manager=False`).

Observable behaviour -- every block number it hands out, every CoW / swap pair, every
AllocStatus -- restates light_vllm/decoding/core/block_manager_v1.py:216-707 with its two
allocators (:67-154 cached, :157-213 uncached) and the LRU evictor (evictor_v1.py:53-99).
The data structure is this build's own: a pool is a struct of arrays indexed by block
number (ref count, content hash, hashed-token count, last access, computed flag) and a
block table is a list of ints, instead of a graph of PhysicalTokenBlock objects.

  uncached pool   free list used as a stack: blocks are handed out from the tail, so ids
                  descend from N-1; a freed block goes back on the tail (:186-201).
  cached pool     ids ascend from 0 until the pool is full, then the evictor chooses
                  (:93-108); content hash -> block dict; a freed block parks in the
                  evictor and can be resurrected by hash (:115-122,132-142).
  evictor         insertion-ordered; evicts the least recently accessed block, among equals
                  the one with most hashed tokens, among those the oldest entry (:67-84).

One point of the reference is not a function of its inputs: freeing a block table walks
`set(blocks)` (:553-557), whose order depends on object addresses.  Here the order is the
table order with duplicates dropped (`_free_order`), which tests can override to replay a
trace recorded from the reference.
"""
import math
from collections import OrderedDict
from itertools import count
from os.path import commonprefix
from typing import Callable, Dict, List, Optional, Sequence as GenericSequence, Tuple

from .interfaces import AllocStatus, BlockSpaceManager

try:  # status constants of whichever Sequence implementation is in use (same integers)
    from ..engine.sequence import SequenceStatus
except Exception:  # pragma: no cover
    SequenceStatus = None

GPU, CPU = 0, 1
DEFAULT_LAST_ACCESSED_TIME = -1  # block/block.py:6


class _Pool:
    """State shared by both allocators: per-block arrays."""

    def __init__(self, num_blocks: int) -> None:
        self.num_blocks = num_blocks
        self.ref_count = [0] * num_blocks
        self.block_hash = [-1] * num_blocks
        self.num_hashed_tokens = [0] * num_blocks
        self.last_accessed = [DEFAULT_LAST_ACCESSED_TIME] * num_blocks
        self.computed = [False] * num_blocks

    def get_num_total_blocks(self) -> int:
        return self.num_blocks


class UncachedPool(_Pool):

    def __init__(self, num_blocks: int) -> None:
        super().__init__(num_blocks)
        self.free_blocks: List[int] = list(range(num_blocks))

    def allocate(self, block_hash: Optional[int] = None, num_hashed_tokens: int = 0) -> int:
        if not self.free_blocks:
            raise ValueError("Out of memory! No free blocks are available.")
        b = self.free_blocks.pop()
        self.ref_count[b] = 1
        return b

    def free(self, b: int) -> None:
        if self.ref_count[b] == 0:
            raise ValueError(f"Double free! block {b} is already freed.")
        self.ref_count[b] -= 1
        if self.ref_count[b] == 0:
            self.free_blocks.append(b)

    def get_num_free_blocks(self) -> int:
        return len(self.free_blocks)

    def contains_block(self, block_hash: int) -> bool:
        raise NotImplementedError("Invalid codepath for uncached block allocator.")

    def update_hash(self, block_hash: int, b: int) -> None:
        raise NotImplementedError("Invalid codepath for uncached block allocator.")


class CachedPool(_Pool):

    def __init__(self, num_blocks: int) -> None:
        super().__init__(num_blocks)
        self.current_num_blocks = 0
        self.cached_blocks: Dict[int, int] = {}           # content hash -> block (ref > 0)
        self.evictor: "OrderedDict[int, int]" = OrderedDict()  # content hash -> block (ref == 0)
        self.default_hash_ctr = count()

    def _evict(self) -> int:
        if not self.evictor:
            raise ValueError("No usable cache memory left")
        it = iter(self.evictor.values())
        victim = next(it)
        for b in self.evictor.values():
            if self.last_accessed[victim] < self.last_accessed[b]:
                break
            if self.num_hashed_tokens[victim] < self.num_hashed_tokens[b]:
                victim = b
        del self.evictor[self.block_hash[victim]]
        self.computed[victim] = False
        return victim

    def _new_block(self, block_hash: int, num_hashed_tokens: int) -> int:
        if self.current_num_blocks == self.num_blocks:
            b = self._evict()
        else:
            b = self.current_num_blocks
            self.current_num_blocks += 1
        self.block_hash[b] = block_hash
        self.num_hashed_tokens[b] = num_hashed_tokens
        return b

    def allocate(self, block_hash: Optional[int] = None, num_hashed_tokens: int = 0) -> int:
        if block_hash is None:
            block_hash = next(self.default_hash_ctr)
        if block_hash in self.evictor:
            assert block_hash not in self.cached_blocks
            b = self.evictor.pop(block_hash)
            assert self.ref_count[b] == 0
            self.cached_blocks[block_hash] = b
            self.ref_count[b] += 1
            return b
        if block_hash not in self.cached_blocks:
            self.cached_blocks[block_hash] = self._new_block(block_hash, num_hashed_tokens)
        b = self.cached_blocks[block_hash]
        self.ref_count[b] += 1
        return b

    def free(self, b: int) -> None:
        if self.ref_count[b] == 0:
            raise ValueError(f"Double free! block {b} is already freed.")
        self.ref_count[b] -= 1
        if self.ref_count[b] == 0:
            h = self.block_hash[b]
            assert h not in self.evictor
            self.evictor[h] = b
            del self.cached_blocks[h]

    def get_num_free_blocks(self) -> int:
        return self.num_blocks - self.current_num_blocks + len(self.evictor)

    def contains_block(self, block_hash: int) -> bool:
        return block_hash in self.cached_blocks or block_hash in self.evictor

    def update_hash(self, block_hash: int, b: int) -> None:
        assert not self.contains_block(block_hash)
        old = self.block_hash[b]
        self.block_hash[b] = block_hash
        del self.cached_blocks[old]
        self.cached_blocks[block_hash] = b


def _dedup_in_order(blocks: List[int]) -> List[int]:
    return list(dict.fromkeys(blocks))


class _Table:
    """Block table of one sequence: physical block numbers + the device they live on."""
    __slots__ = ("blocks", "device")

    def __init__(self, blocks: List[int], device: int) -> None:
        self.blocks = blocks
        self.device = device


class BlockSpaceManagerV1(BlockSpaceManager):
    """Manages the mapping between logical and physical token blocks."""

    def __init__(self, block_size: int, num_gpu_blocks: int, num_cpu_blocks: int,
                 watermark: float = 0.01, sliding_window: Optional[int] = None,
                 enable_caching: bool = False) -> None:
        self.block_size = block_size
        self.num_total_gpu_blocks = num_gpu_blocks
        self.num_total_cpu_blocks = num_cpu_blocks
        if enable_caching and sliding_window is not None:
            raise NotImplementedError("Sliding window is not allowed with prefix caching enabled!")
        self.block_sliding_window = None
        if sliding_window is not None:
            # rounded up to whole blocks (block_manager_v1.py:236-240)
            self.block_sliding_window = math.ceil(sliding_window / block_size)
        self.watermark = watermark
        assert watermark >= 0.0
        self.enable_caching = enable_caching
        self.watermark_blocks = int(watermark * num_gpu_blocks)
        pool_cls = CachedPool if enable_caching else UncachedPool
        self.gpu_allocator = pool_cls(num_gpu_blocks)
        self.cpu_allocator = pool_cls(num_cpu_blocks)
        self._pools = (self.gpu_allocator, self.cpu_allocator)
        self.block_tables: Dict[int, _Table] = {}
        # order in which the distinct blocks of a table are released (see module docstring)
        self._free_order: Callable[[List[int]], List[int]] = _dedup_in_order

    # ---- helpers ----
    @staticmethod
    def _status(name: str):
        return getattr(SequenceStatus, name)

    def _waiting(self, seq_group):
        return seq_group.get_seqs(status=self._status("WAITING"))

    # ---- allocation of a prompt ----
    def can_allocate(self, seq_group) -> AllocStatus:
        num_required_blocks = self._waiting(seq_group)[0].n_blocks
        if self.block_sliding_window is not None:
            num_required_blocks = min(num_required_blocks, self.block_sliding_window)
        num_free_gpu_blocks = self.gpu_allocator.get_num_free_blocks()
        # the watermark keeps a few blocks back to avoid constant eviction (:285-292)
        if self.num_total_gpu_blocks - num_required_blocks < self.watermark_blocks:
            return AllocStatus.NEVER
        if num_free_gpu_blocks - num_required_blocks >= self.watermark_blocks:
            return AllocStatus.OK
        return AllocStatus.LATER

    def _allocate_sequence(self, seq, ref_count: int) -> List[int]:
        table: List[int] = []
        pool = self.gpu_allocator
        for logical_idx in range(seq.n_blocks):
            if self.block_sliding_window is not None and logical_idx >= self.block_sliding_window:
                b = table[logical_idx % self.block_sliding_window]  # reuse inside the window
                pool.ref_count[b] = ref_count
            elif self.enable_caching:
                b = pool.allocate(seq.hash_of_block(logical_idx),
                                  seq.num_hashed_tokens_of_block(logical_idx))
            else:
                b = pool.allocate()
                pool.ref_count[b] = ref_count
            table.append(b)
        return table

    def allocate(self, seq_group) -> None:
        waiting = self._waiting(seq_group)
        table = self._allocate_sequence(waiting[0], seq_group.num_seqs())
        for seq in waiting:
            self.block_tables[seq.seq_id] = _Table(list(table), GPU)

    # ---- decode: one more slot ----
    def can_append_slots(self, seq_group, num_lookahead_slots: int = 0) -> bool:
        assert num_lookahead_slots == 0, "lookahead allocation not supported in BlockSpaceManagerV1"
        # one free block per running sequence is enough (:355-359)
        num_seqs = seq_group.num_seqs(status=self._status("RUNNING"))
        return num_seqs <= self.gpu_allocator.get_num_free_blocks()

    def _is_last_block_full(self, seq) -> bool:
        n = seq.data.get_len()
        return n > 0 and n % seq.block_size == 0

    def _promote_last_block(self, seq, last_block: int) -> int:
        pool = self.gpu_allocator
        new_hash = seq.hash_of_block(seq.n_blocks - 1)
        if pool.contains_block(new_hash):  # an identical block exists: share it (:376-381)
            pool.free(last_block)
            return pool.allocate(new_hash)
        pool.update_hash(new_hash, last_block)
        return last_block

    def _allocate_last_physical_block(self, seq) -> int:
        if not self.enable_caching:
            return self.gpu_allocator.allocate()
        n_blocks = seq.n_blocks
        block_hash = seq.hash_of_block(n_blocks - 1) if self._is_last_block_full(seq) else None
        b = self.gpu_allocator.allocate(block_hash, seq.num_hashed_tokens_of_block(n_blocks - 1))
        if block_hash is None:
            assert self.gpu_allocator.ref_count[b] == 1
        return b

    def append_slots(self, seq, num_lookahead_slots: int = 0) -> List[Tuple[int, int]]:
        """Make room for the sequence's newest token; returns CoW (src, dst) pairs."""
        n_blocks = seq.n_blocks
        table = self.block_tables[seq.seq_id].blocks
        pool = self.gpu_allocator
        if len(table) < n_blocks:
            assert len(table) == n_blocks - 1
            if self.block_sliding_window and len(table) >= self.block_sliding_window:
                table.append(table[len(table) % self.block_sliding_window])
            else:
                table.append(self._allocate_last_physical_block(seq))
                return []
        last_block = table[-1]
        assert self.block_tables[seq.seq_id].device == GPU
        if pool.ref_count[last_block] == 1:
            if self.enable_caching and self._is_last_block_full(seq):
                table[-1] = self._promote_last_block(seq, last_block)
            return []
        # shared with another sequence: copy on write (:459-467)
        new_block = self._allocate_last_physical_block(seq)
        table[-1] = new_block
        pool.free(last_block)
        return [(last_block, new_block)]

    def fork(self, parent_seq, child_seq) -> None:
        if parent_seq.seq_id not in self.block_tables:
            return
        src = self.block_tables[parent_seq.seq_id]
        self.block_tables[child_seq.seq_id] = _Table(list(src.blocks), src.device)
        pool = self._pools[src.device]
        for b in set(src.blocks):  # a sliding-window table repeats blocks: count each once
            pool.ref_count[b] += 1

    # ---- swapping ----
    def _get_physical_blocks(self, seq_group) -> set:
        blocks = set()
        for seq in seq_group.get_seqs():
            if seq.is_finished():
                continue
            blocks.update(self.block_tables[seq.seq_id].blocks)
        return blocks

    def can_swap_in(self, seq_group, num_lookahead_slots: int = 0) -> AllocStatus:
        assert num_lookahead_slots == 0, "BlockSpaceManagerV1 does not support lookahead allocation"
        blocks = self._get_physical_blocks(seq_group)
        num_swapped_seqs = seq_group.num_seqs(status=self._status("SWAPPED"))
        num_free_blocks = self.gpu_allocator.get_num_free_blocks()
        # every swapped-in sequence is assumed to need one more block right away (:505-508)
        num_required_blocks = len(blocks) + num_swapped_seqs
        if self.gpu_allocator.get_num_total_blocks() < num_required_blocks:
            return AllocStatus.NEVER
        if num_free_blocks - num_required_blocks >= self.watermark_blocks:
            return AllocStatus.OK
        return AllocStatus.LATER

    def _swap_block_table(self, table: _Table, src_pool, dst_pool, dst_device: int,
                          mapping: Dict[int, int]) -> _Table:
        new_blocks: List[int] = []
        for from_block in table.blocks:
            if from_block in mapping:
                to_block = mapping[from_block]
                dst_pool.ref_count[to_block] += 1
            else:
                to_block = dst_pool.allocate(src_pool.block_hash[from_block],
                                             src_pool.num_hashed_tokens[from_block])
                mapping[from_block] = to_block
            new_blocks.append(to_block)
            src_pool.free(from_block)
        return _Table(new_blocks, dst_device)

    def swap_in(self, seq_group) -> List[Tuple[int, int]]:
        mapping: Dict[int, int] = {}  # cpu block -> gpu block, in order of first use
        for seq in seq_group.get_seqs(status=self._status("SWAPPED")):
            self.block_tables[seq.seq_id] = self._swap_block_table(
                self.block_tables[seq.seq_id], self.cpu_allocator, self.gpu_allocator, GPU, mapping)
        return list(mapping.items())

    def can_swap_out(self, seq_group) -> bool:
        return len(self._get_physical_blocks(seq_group)) <= self.cpu_allocator.get_num_free_blocks()

    def swap_out(self, seq_group) -> List[Tuple[int, int]]:
        mapping: Dict[int, int] = {}  # gpu block -> cpu block
        for seq in seq_group.get_seqs(status=self._status("RUNNING")):
            self.block_tables[seq.seq_id] = self._swap_block_table(
                self.block_tables[seq.seq_id], self.gpu_allocator, self.cpu_allocator, CPU, mapping)
        return list(mapping.items())

    # ---- release ----
    def _free_block_table(self, table: _Table) -> None:
        # inside a sliding window only the last `window` entries are distinct allocations (:545-552)
        blocks = (table.blocks[-self.block_sliding_window:]
                  if self.block_sliding_window is not None else table.blocks)
        pool = self._pools[table.device]
        for b in self._free_order(blocks):
            pool.free(b)

    def free(self, seq) -> None:
        table = self.block_tables.pop(seq.seq_id, None)
        if table is not None:  # else: already freed or never scheduled
            self._free_block_table(table)

    def reset(self) -> None:
        for table in self.block_tables.values():
            self._free_block_table(table)
        self.block_tables.clear()

    # ---- queries ----
    def get_block_table(self, seq) -> List[int]:
        return list(self.block_tables[seq.seq_id].blocks)

    def get_num_free_gpu_blocks(self) -> int:
        return self.gpu_allocator.get_num_free_blocks()

    def get_num_free_cpu_blocks(self) -> int:
        return self.cpu_allocator.get_num_free_blocks()

    # ---- prefix-cache bookkeeping ----
    def access_all_blocks_in_seq(self, seq, access_time: float) -> None:
        if self.enable_caching:
            table = self.block_tables[seq.seq_id]
            la = self._pools[table.device].last_accessed
            for b in table.blocks:
                la[b] = access_time

    def compute_full_blocks_in_seq(self, seq) -> None:
        table = self.block_tables.get(seq.seq_id)
        if table is None:
            return
        max_full_block = seq.get_len() // self.block_size - 1
        if max_full_block == -1:
            return
        computed = self._pools[table.device].computed
        for i in reversed(range(max_full_block)):
            if computed[table.blocks[i]]:
                break
            computed[table.blocks[i]] = True

    def get_all_computed_blocks(self, seq) -> List[int]:
        table = self.block_tables.get(seq.seq_id)
        if table is None:
            return []
        computed = self._pools[table.device].computed
        out: List[int] = []
        # the last block is left out so that a fully cached prompt still runs one block (:670-676)
        for b in table.blocks[:-1]:
            if not computed[b]:
                break
            out.append(b)
        return out

    def get_common_computed_block_ids(self, seqs) -> GenericSequence[int]:
        if not self.enable_caching:
            return []
        ids_list = [self.get_all_computed_blocks(seq) for seq in seqs]
        return commonprefix([ids for ids in ids_list if ids != []])

    def mark_blocks_as_computed(self, seq_group) -> None:
        if self.enable_caching:
            for seq in seq_group.seqs_dict.values():
                self.compute_full_blocks_in_seq(seq)
