"""KV-cache block manager, version 2 (`use_v2_block_manager=True`).

Observable behaviour restates light_vllm/decoding/core/block_manager_v2.py:24-500 and the
allocator stack under it (block/block_table.py, block/naive_block.py,
block/prefix_caching_block.py, block/cpu_gpu_block_allocator.py, block/common.py,
evictor_v2.py): block tables, CoW pairs, swap maps, AllocStatus verdicts, computed-block lists.

Block ids are ABSOLUTE: GPU blocks are [0, NG), CPU blocks [NG, NG+NC)
(cpu_gpu_block_allocator.py:57-93); swap maps are reported device-relative
(`get_physical_block_id`, :236-262).

This build's structure (not the reference's class graph): one `_Pool` per device holding the
free deque, the reference counts and -- with prefix caching -- the content-hash table, the
LRU evictor and the per-block access/computed trackers; a block table is a list of light
`_Blk` records (id, tokens, chained content hash, back pointer).

  naive pool          free ids in a deque: handed out from the LEFT, a freed id returns to the
                      LEFT (naive_block.py:132-148) -> the most recently freed id is reused first
  prefix-caching pool full blocks are content addressed: hash((is_first, prev_hash, *tokens))
                      (prefix_caching_block.py:815-836); a full block whose hash is known shares
                      the cached id (:145-175, :413-437); a freed cached block parks in the
                      evictor keyed by block id and is evicted least-recently-used, ties ->
                      more hashed tokens, then oldest entry (evictor_v2.py:76-96)
  sliding window      blocks that fell out of the window are replaced by one shared null
                      block (block_table.py:133-143)
"""
import math
from collections import OrderedDict, deque
from os.path import commonprefix
from typing import Deque, Dict, List, Optional, Sequence as GenericSequence, Tuple

from .interfaces import AllocStatus, BlockSpaceManager

try:
    from ..engine.sequence import SequenceStatus
except Exception:  # pragma: no cover
    SequenceStatus = None

GPU, CPU = 0, 1
_DEFAULT_LAST_ACCESSED_TIME = -1


class NoFreeBlocksError(ValueError):
    pass


def _cdiv(a: int, b: int) -> int:
    return -(-a // b)


class _Blk:
    """One logical block of one sequence."""
    __slots__ = ("block_id", "tokens", "prev", "block_size", "_hash", "num_tokens_total", "computed",
                 "is_null")

    def __init__(self, prev: Optional["_Blk"], tokens: List[int], block_size: int,
                 block_id: Optional[int]) -> None:
        self.prev = prev
        self.tokens = list(tokens)
        self.block_size = block_size
        self.block_id = block_id
        self._hash: Optional[int] = None
        self.computed = False
        self.is_null = False
        self.num_tokens_total = (prev.num_tokens_total if prev is not None else 0) + len(self.tokens)

    @property
    def is_full(self) -> bool:
        return len(self.tokens) == self.block_size

    @property
    def num_empty_slots(self) -> int:
        return self.block_size - len(self.tokens)

    def content_hash(self) -> Optional[int]:
        """Chained hash of a full block whose predecessors are all hashed
        (prefix_caching_block.py:790-836); cached once known."""
        if self._hash is not None:
            return self._hash
        if not self.is_full:
            return None
        is_first = self.prev is None
        prev_hash = None if is_first else self.prev.content_hash()
        if prev_hash is None and not is_first:
            return None
        self._hash = hash((is_first, prev_hash, *self.tokens))
        return self._hash


class _Pool:
    """Allocator of one device."""

    def __init__(self, block_ids: List[int], block_size: int, caching: bool) -> None:
        self.block_size = block_size
        self.caching = caching
        self.free: Deque[int] = deque(block_ids)
        self.all_ids = frozenset(block_ids)
        self.base = min(block_ids) if block_ids else 0
        self.ref: Dict[int, int] = {i: 0 for i in block_ids}
        self.cows: List[Tuple[int, int]] = []
        # prefix caching state
        self.cached: Dict[int, int] = {}                       # content hash -> block id
        self.evictor: "OrderedDict[int, List]" = OrderedDict()  # block id -> [hash, n_tokens, last_accessed]
        self.trk_active = {i: False for i in block_ids}
        self.trk_last = {i: _DEFAULT_LAST_ACCESSED_TIME for i in block_ids}
        self.trk_computed = {i: False for i in block_ids}

    # ---- ids ----
    def get_num_free_blocks(self) -> int:
        return len(self.free) + (len(self.evictor) if self.caching else 0)

    def get_num_total_blocks(self) -> int:
        return len(self.all_ids)

    def physical(self, absolute_id: int) -> int:
        return absolute_id - self.base

    def _track(self, block_id: int, computed: bool) -> None:
        assert not self.trk_active[block_id]
        self.trk_active[block_id] = True
        self.trk_last[block_id] = _DEFAULT_LAST_ACCESSED_TIME
        self.trk_computed[block_id] = computed

    def _untrack(self, block_id: int) -> None:
        assert self.trk_active[block_id]
        self.trk_active[block_id] = False
        self.trk_last[block_id] = _DEFAULT_LAST_ACCESSED_TIME
        self.trk_computed[block_id] = False

    def _evict(self) -> Tuple[int, int]:
        it = iter(self.evictor.items())
        victim_id, victim = next(it)
        for bid, meta in self.evictor.items():
            if victim[2] > meta[2] or (victim[2] == meta[2] and victim[1] < meta[1]):
                victim_id, victim = bid, meta
        del self.evictor[victim_id]
        return victim_id, victim[0]

    def alloc_id(self) -> int:
        if self.free:
            b = self.free.popleft()
            self.ref[b] += 1
            if self.caching:
                self._track(b, computed=False)
            return b
        if self.caching and self.evictor:
            b, h = self._evict()
            assert self.cached[h] == b and self.ref[b] == 0
            del self.cached[h]
            self.ref[b] += 1
            self._track(b, computed=False)
            return b
        raise NoFreeBlocksError()

    def _free_hashless(self, blk: _Blk) -> None:
        b = blk.block_id
        if self.caching and self.ref[b] == 1:
            self._untrack(b)
        self.ref[b] -= 1
        if self.ref[b] == 0:
            self.free.appendleft(b)
        blk.block_id = None

    def _incr_cached(self, blk: _Blk) -> None:
        blk.computed = True
        b = blk.block_id
        self.ref[b] += 1
        if self.ref[b] == 1:  # resurrected from the evictor
            if b in self.evictor:
                del self.evictor[b]
            self._track(b, computed=True)

    def _decr_cached(self, blk: _Blk) -> None:
        b = blk.block_id
        self.ref[b] -= 1
        if self.ref[b] == 0:
            h = blk.content_hash()
            assert h in self.cached
            self.evictor[b] = [h, blk.num_tokens_total, self.trk_last[b]]
            self._untrack(b)
        blk.block_id = None

    def free_blk(self, blk: _Blk) -> None:
        assert blk.block_id is not None, "Freeing unallocated block is undefined"
        if self.caching and blk.content_hash() is not None:
            self._decr_cached(blk)
        else:
            self._free_hashless(blk)

    # ---- block construction ----
    def allocate_mutable(self, prev: Optional[_Blk]) -> _Blk:
        return _Blk(prev, [], self.block_size, self.alloc_id())

    def _promote(self, blk: _Blk) -> None:
        """A block just became full and hashable (prefix_caching_block.py:413-437)."""
        h = blk.content_hash()
        if h not in self.cached:
            self.cached[h] = blk.block_id
            return
        self._free_hashless(blk)       # drop the private copy ...
        blk.block_id = self.cached[h]  # ... and share the cached block
        self._incr_cached(blk)

    def append_tokens(self, blk: _Blk, tokens: List[int]) -> None:
        """Append to a block, copy-on-write if it is shared (naive_block.py:398-409)."""
        if not tokens:
            return
        assert len(tokens) <= blk.num_empty_slots
        blk.tokens.extend(tokens)
        # recomputed from the predecessor's CURRENT total, as the reference does at every append
        # (prefix_caching_block.py _update_num_tokens_total): a lookahead block is created while its predecessor is
        # still filling up, and the evictor breaks last-access ties on this number
        blk.num_tokens_total = (blk.prev.num_tokens_total if blk.prev is not None else 0) + len(blk.tokens)
        if blk.block_id is not None and self.ref[blk.block_id] > 1:
            src = blk.block_id
            # the hash is not known yet for a block that is being written: hashless release
            self._free_hashless(blk)
            blk.block_id = self.alloc_id()
            self.cows.append((src, blk.block_id))
        if self.caching and blk.content_hash() is not None:
            self._promote(blk)

    def allocate_immutable(self, prev: Optional[_Blk], tokens: List[int]) -> _Blk:
        if self.caching:
            blk = _Blk(prev, tokens, self.block_size, None)
            h = blk.content_hash()
            assert h is not None
            cached = self.cached.get(h)
            if cached is not None:
                blk.block_id = cached
                self._incr_cached(blk)
                return blk
        blk = self.allocate_mutable(prev)
        self.append_tokens(blk, tokens)
        return blk

    def allocate_immutable_many(self, prev: Optional[_Blk], chunks: List[List[int]]) -> List[_Blk]:
        out: List[_Blk] = []
        if self.caching:
            for c in chunks:
                prev = self.allocate_immutable(prev, c)
                out.append(prev)
            return out
        ids = [self.alloc_id() for _ in chunks]  # all ids first (naive_block.py:98-115)
        for c, b in zip(chunks, ids):
            prev = _Blk(prev, c, self.block_size, b)
            out.append(prev)
        return out

    def fork_chain(self, last: _Blk) -> List[_Blk]:
        chain: List[_Blk] = []
        b = last
        while b is not None:
            chain.append(b)
            b = b.prev
        chain.reverse()
        out: List[_Blk] = []
        prev = None
        for src in chain:
            assert src.block_id is not None, "can't fork a freed block"
            self.ref[src.block_id] += 1
            assert self.ref[src.block_id] != 1, "can't fork free'd block"
            prev = _Blk(prev, src.tokens, self.block_size, src.block_id)
            out.append(prev)
        return out

    # ---- swapping ----
    def swap_out(self, blocks: List[_Blk]) -> None:
        for blk in blocks:
            self.free_blk(blk)

    def swap_in(self, blocks: List[_Blk]) -> None:
        for blk in blocks:
            if blk.is_full:
                tmp = self.allocate_immutable(blk.prev, blk.tokens)
            else:
                tmp = self.allocate_mutable(blk.prev)
                self.append_tokens(tmp, blk.tokens)
            blk.block_id = tmp.block_id

    def get_num_blocks_touched(self, blocks: List[_Blk], num_lookahead_slots: int = 0) -> int:
        if not self.caching:  # naive_block.py:301-327
            old = set()
            new = 0
            for blk in blocks:
                if not blk.is_full and num_lookahead_slots != 0:
                    if blk.num_empty_slots >= num_lookahead_slots:
                        new += 1
                    else:
                        new += _cdiv(num_lookahead_slots - blk.num_empty_slots, self.block_size)
                else:
                    old.add(blk.block_id)
            return new + len(old)
        n = 0  # prefix_caching_block.py:563-590
        for blk in blocks:
            if not blk.is_full:
                if blk.num_empty_slots >= num_lookahead_slots:
                    n += 1
                else:
                    n += _cdiv(num_lookahead_slots - blk.num_empty_slots, self.block_size)
            elif blk.content_hash() not in self.cached:
                n += 1
        return n

    # ---- prefix-cache bookkeeping ----
    def mark_blocks_as_accessed(self, block_ids: List[int], now: float) -> None:
        if not self.caching:
            return
        for b in block_ids:
            if self.trk_active[b]:
                self.trk_last[b] = now
            elif b in self.evictor:
                self.evictor[b][2] = now
            else:
                raise ValueError("Mark block as accessed which is not belonged to GPU")

    def block_is_computed(self, b: int) -> bool:
        return self.trk_computed[b] if self.trk_active[b] else b in self.evictor

    def get_computed_block_ids(self, prev_computed: List[int], block_ids: List[int],
                               skip_last_block_id: bool = True) -> List[int]:
        if not self.caching:
            return []
        cur = len(block_ids) - (1 if skip_last_block_id else 0)
        assert 0 <= len(prev_computed) <= cur
        ret = prev_computed
        for i in range(len(prev_computed), cur):
            if self.block_is_computed(block_ids[i]):
                ret.append(block_ids[i])
        return ret

    def get_common_computed_block_ids(self, lists: List[List[int]]) -> List[int]:
        if not self.caching:
            return []
        if len(lists) == 1:
            return lists[0]
        return commonprefix([ids for ids in lists if ids])


class _Table:
    """Block table of one sequence (block/block_table.py)."""
    __slots__ = ("blocks", "num_full_slots")

    def __init__(self, blocks: List[_Blk]) -> None:
        self.blocks = blocks
        self.num_full_slots = sum(len(b.tokens) for b in blocks)

    def ids(self) -> List[int]:
        return [b.block_id for b in self.blocks]


class BlockSpaceManagerV2(BlockSpaceManager):

    def __init__(self, block_size: int, num_gpu_blocks: int, num_cpu_blocks: int,
                 watermark: float = 0.01, sliding_window: Optional[int] = None,
                 enable_caching: bool = False) -> None:
        self.block_size = block_size
        self.num_total_gpu_blocks = num_gpu_blocks
        self.num_total_cpu_blocks = num_cpu_blocks
        self.sliding_window = sliding_window
        self.max_block_sliding_window = None
        if sliding_window is not None:
            # +1: the window may straddle a block boundary; +1: the block being generated
            self.max_block_sliding_window = sliding_window // block_size + 1 + 1
        self.watermark = watermark
        assert watermark >= 0.0
        self.enable_caching = enable_caching
        self.watermark_blocks = int(watermark * num_gpu_blocks)
        ids = list(range(num_gpu_blocks + num_cpu_blocks))
        self._pools = {GPU: _Pool(ids[:num_gpu_blocks], block_size, enable_caching),
                       CPU: _Pool(ids[num_gpu_blocks:], block_size, enable_caching)}
        self.block_tables: Dict[int, _Table] = {}
        self._null_block: Optional[_Blk] = None
        # ComputedBlocksTracker / LastAccessBlocksTracker (prefix_caching_block.py:839-964)
        self._computed: Dict[int, Tuple[List[int], bool]] = {}
        self._last_access: Dict[int, Optional[float]] = {}

    # ---- helpers ----
    @staticmethod
    def _status(name: str):
        return getattr(SequenceStatus, name)

    def _pool_of(self, block_id: int) -> _Pool:
        return self._pools[GPU] if block_id < self.num_total_gpu_blocks else self._pools[CPU]

    def _free_block(self, blk: _Blk) -> None:
        if blk.is_null:
            return
        self._pool_of(blk.block_id).free_blk(blk)

    def _add_seq_trackers(self, seq_id: int) -> None:
        assert seq_id not in self._computed and seq_id not in self._last_access
        self._computed[seq_id] = ([], False)
        self._last_access[seq_id] = None

    # ---- prompt allocation ----
    def can_allocate(self, seq_group) -> AllocStatus:
        seq = seq_group.get_seqs(status=self._status("WAITING"))[0]
        num_required_blocks = _cdiv(len(seq.get_token_ids()), self.block_size)
        if self.max_block_sliding_window is not None:
            num_required_blocks = min(num_required_blocks, self.max_block_sliding_window)
        num_free_gpu_blocks = self._pools[GPU].get_num_free_blocks()
        if self.num_total_gpu_blocks - num_required_blocks < self.watermark_blocks:
            return AllocStatus.NEVER
        if num_free_gpu_blocks - num_required_blocks >= self.watermark_blocks:
            return AllocStatus.OK
        return AllocStatus.LATER

    def _allocate_sequence(self, seq) -> _Table:
        token_ids = seq.get_token_ids()
        assert token_ids
        pool = self._pools[GPU]
        bs = self.block_size
        full = [token_ids[i:i + bs] for i in range(0, len(token_ids) - len(token_ids) % bs, bs)]
        tail = token_ids[len(full) * bs:]
        blocks: List[_Blk] = []
        prev = None
        if full:
            blocks.extend(pool.allocate_immutable_many(None, full))
            prev = blocks[-1]
        if tail:
            blk = pool.allocate_mutable(prev)
            pool.append_tokens(blk, tail)
            blocks.append(blk)
        return _Table(blocks)

    def allocate(self, seq_group) -> None:
        waiting = seq_group.get_seqs(status=self._status("WAITING"))
        assert not (set(s.seq_id for s in waiting) & self.block_tables.keys()), "block table already exists"
        seq = waiting[0]
        table = self._allocate_sequence(seq)
        self.block_tables[seq.seq_id] = table
        self._add_seq_trackers(seq.seq_id)
        for other in waiting[1:]:
            self.block_tables[other.seq_id] = _Table(self._pools[GPU].fork_chain(table.blocks[-1]))
            self._add_seq_trackers(other.seq_id)

    # ---- decode ----
    def can_append_slots(self, seq_group, num_lookahead_slots: int) -> bool:
        """Worst case: every touched block needs a new allocation (block_manager_v2.py:183-209)."""
        touched = 0
        for seq in seq_group.get_seqs(status=self._status("RUNNING")):
            table = self.block_tables[seq.seq_id]
            n_tok = len(seq.get_token_ids()) - table.num_full_slots + num_lookahead_slots
            first_chunk = self.block_size - (table.num_full_slots % self.block_size)
            touched += 1 + math.ceil((n_tok - first_chunk) / self.block_size)
        return touched <= self._pools[GPU].get_num_free_blocks()

    def _null(self) -> _Blk:
        if self._null_block is None:
            self._null_block = self._pools[GPU].allocate_mutable(None)
            self._null_block.is_null = True
        return self._null_block

    def append_slots(self, seq, num_lookahead_slots: int) -> List[Tuple[int, int]]:
        table = self.block_tables[seq.seq_id]
        pool = self._pools[GPU]
        bs = self.block_size
        token_ids = seq.get_token_ids()[table.num_full_slots:]
        assert table.blocks, "no blocks have been allocated"
        if self.max_block_sliding_window is not None:
            null_block = self._null()
            end_block_idx = seq.data.get_num_computed_tokens() // bs - self.max_block_sliding_window
            for idx in range(0, end_block_idx):
                b = table.blocks[idx]
                if b is not null_block:
                    self._free_block(b)
                    table.blocks[idx] = null_block
        # make room (block_table.py:158-178)
        need = len(token_ids) + num_lookahead_slots
        empty = len(table.blocks) * bs - table.num_full_slots
        if empty < need:
            for _ in range(_cdiv(need - empty, bs)):
                table.blocks.append(pool.allocate_mutable(table.blocks[-1]))
        # write the tokens block by block
        first_block_idx = table.num_full_slots // bs
        first_chunk = bs - (table.num_full_slots % bs)
        chunks = [token_ids[:first_chunk]] + [token_ids[i:i + bs] for i in range(first_chunk, len(token_ids), bs)]
        if not token_ids:
            # A running prompt still in chunked prefill has no new token, and when its table is
            # exactly full the reference indexes one block past the end here (block_table.py:
            # 150-156 with _chunk_token_blocks_for_append returning [[]]) and raises IndexError;
            # there is nothing to write, so nothing is touched.
            chunks = []
        for i, chunk in enumerate(chunks):
            pool.append_tokens(table.blocks[first_block_idx + i], chunk)
        table.num_full_slots += len(token_ids)
        cows, pool.cows = pool.cows, []
        return cows

    def fork(self, parent_seq, child_seq) -> None:
        if parent_seq.seq_id not in self.block_tables:
            return
        src = self.block_tables[parent_seq.seq_id]
        last = src.blocks[-1]
        assert not last.is_null
        self.block_tables[child_seq.seq_id] = _Table(self._pool_of(last.block_id).fork_chain(last))
        self._add_seq_trackers(child_seq.seq_id)

    # ---- release ----
    def free(self, seq) -> None:
        seq_id = seq.seq_id
        table = self.block_tables.get(seq_id)
        if table is None:
            return
        ts = self._last_access[seq_id]
        if ts is not None:  # stamp the sequence's blocks with its last access time
            self._pools[GPU].mark_blocks_as_accessed(table.ids(), ts)
        del self._last_access[seq_id]
        del self._computed[seq_id]
        for blk in table.blocks:
            self._free_block(blk)
        del self.block_tables[seq_id]

    # ---- queries ----
    def get_block_table(self, seq) -> List[int]:
        return self.block_tables[seq.seq_id].ids()

    def get_num_free_gpu_blocks(self) -> int:
        return self._pools[GPU].get_num_free_blocks()

    def get_num_free_cpu_blocks(self) -> int:
        return self._pools[CPU].get_num_free_blocks()

    def access_all_blocks_in_seq(self, seq, now: float) -> None:
        if self.enable_caching:
            assert seq.seq_id in self._last_access
            self._last_access[seq.seq_id] = now

    def mark_blocks_as_computed(self, seq_group) -> None:
        pass  # computed-ness is derived incrementally (block_manager_v2.py:277-283)

    def get_common_computed_block_ids(self, seqs) -> GenericSequence[int]:
        pool = self._pools[GPU]
        lists: List[List[int]] = []
        for seq in seqs:
            block_ids = self.block_tables[seq.seq_id].ids()
            prev, has_gap = self._computed[seq.seq_id]
            if has_gap:
                lists.append(prev)
                continue
            num_cur = len(block_ids) - 1
            assert num_cur >= 0
            if len(prev) >= num_cur:
                assert len(prev) == num_cur
                lists.append(prev)
                continue
            computed = pool.get_computed_block_ids(prev, block_ids, skip_last_block_id=True)
            self._computed[seq.seq_id] = (computed, len(computed) < num_cur)
            lists.append(computed)
        return pool.get_common_computed_block_ids(lists)

    # ---- swapping ----
    def _can_swap(self, seq_group, device: int, status, num_lookahead_slots: int = 0) -> AllocStatus:
        blocks: List[_Blk] = []
        for seq in seq_group.get_seqs(status=status):
            blocks.extend(self.block_tables[seq.seq_id].blocks)
        pool = self._pools[device]
        touched = pool.get_num_blocks_touched(blocks, num_lookahead_slots)
        watermark = self.watermark_blocks if device == GPU else 0
        if pool.get_num_total_blocks() < touched:
            return AllocStatus.NEVER
        if pool.get_num_free_blocks() - touched >= watermark:
            return AllocStatus.OK
        return AllocStatus.LATER

    def can_swap_in(self, seq_group, num_lookahead_slots: int) -> AllocStatus:
        return self._can_swap(seq_group, GPU, self._status("SWAPPED"), num_lookahead_slots)

    def can_swap_out(self, seq_group) -> bool:
        return self._can_swap(seq_group, CPU, self._status("RUNNING")) == AllocStatus.OK

    def _swap(self, seq_group, status, src: int, dst: int) -> List[Tuple[int, int]]:
        mapping: List[Tuple[int, int]] = []
        for seq in seq_group.get_seqs(status=status):
            blocks = self.block_tables[seq.seq_id].blocks
            if not blocks:
                continue
            src_ids = [b.block_id for b in blocks]
            self._pools[src].swap_out(blocks)
            self._pools[dst].swap_in(blocks)
            per_seq: Dict[int, int] = {}
            for s, b in zip(src_ids, blocks):
                if s is not None and b.block_id is not None:
                    per_seq[self._pools[src].physical(s)] = self._pools[dst].physical(b.block_id)
            mapping.extend(per_seq.items())
        return mapping

    def swap_in(self, seq_group) -> List[Tuple[int, int]]:
        return self._swap(seq_group, self._status("SWAPPED"), CPU, GPU)

    def swap_out(self, seq_group) -> List[Tuple[int, int]]:
        return self._swap(seq_group, self._status("RUNNING"), GPU, CPU)
