"""Per-step model inputs: token ids, positions and the attention metadata (slot_mapping,
block_tables, seq_lens) that the paged-attention kernels consume.

Field-for-field the computation of light_vllm/decoding/processor/model_input_builder.py
(:212-378): per scheduled sequence `context_len` / `seq_len` / `query_len`, the prefix-cache
shortcut (:270-296), the sliding-window clamp of decode sequence lengths (:298-322), then the
attention backend's metadata builder.  Tensors are built on the CPU (pinned when a GPU is
present); the executor moves them with non-blocking copies on its stream.
"""
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch

from .config import CacheConfig, SchedulerConfig
from .scheduler import SchedulerOutput, SequenceGroupMetadata


class InterDataForSeqGroup:
    """Intermediate per-group record read by the attention metadata builder
    (same attribute names as ModelInputForGPUBuilder.InterDataForSeqGroup, :110-157)."""

    def __init__(self, *, request_id: str, seq_ids: List[int], is_prompt: bool,
                 block_tables: Optional[Dict[int, List[int]]], computed_block_nums: List[int]):
        n = len(seq_ids)
        self.request_id = request_id
        self.seq_ids = seq_ids
        self.is_prompt = is_prompt
        self.block_tables = block_tables
        self.computed_block_nums = computed_block_nums
        self.n_seqs = n
        self.input_tokens: List[List[int]] = [[] for _ in range(n)]
        self.input_positions: List[List[int]] = [[] for _ in range(n)]
        self.seq_lens = [0] * n
        self.orig_seq_lens = [0] * n
        self.query_lens = [0] * n
        self.context_lens = [0] * n
        self.curr_sliding_window_blocks = [0] * n
        self.prefix_cache_hit = False


@dataclass
class ModelInput:
    input_tokens: Optional[torch.Tensor] = None      # [T] int64
    input_positions: Optional[torch.Tensor] = None   # [T] int64
    attn_metadata: object = None
    seq_lens: List[int] = field(default_factory=list)
    query_lens: List[int] = field(default_factory=list)
    # rows of the hidden states that feed the sampler, and the sequences they belong to
    sample_indices: List[int] = field(default_factory=list)
    sample_seq_ids: List[int] = field(default_factory=list)
    decode_only: bool = True

    def to(self, device, non_blocking=True) -> "ModelInput":
        if self.input_tokens is not None:
            self.input_tokens = self.input_tokens.to(device, non_blocking=non_blocking)
            self.input_positions = self.input_positions.to(device, non_blocking=non_blocking)
            self.attn_metadata.to(device, non_blocking=non_blocking)
        return self


@dataclass
class WorkerInput:
    """Cache operations to run before the forward pass; [n, 2] int64 CPU tensors
    (model_input_builder.py:84-101)."""
    num_seq_groups: int
    blocks_to_swap_in: torch.Tensor
    blocks_to_swap_out: torch.Tensor
    blocks_to_copy: torch.Tensor


@dataclass
class ExecuteInput:
    worker_input: WorkerInput
    model_input: ModelInput


class ModelInputBuilder:

    def __init__(self, scheduler_config: SchedulerConfig, cache_config: CacheConfig, attn_backend,
                 sliding_window: Optional[int] = None, chunked_prefill_enabled: bool = False):
        self.scheduler_config = scheduler_config
        self.cache_config = cache_config
        self.attn_backend = attn_backend
        self.sliding_window = sliding_window
        self.block_size = cache_config.block_size
        self.chunked_prefill_enabled = chunked_prefill_enabled
        if sliding_window is not None:
            self.sliding_window_blocks = (sliding_window + self.block_size - 1) // self.block_size
            self.block_aligned_sliding_window = self.sliding_window_blocks * self.block_size
        self.inter_data_list: List[InterDataForSeqGroup] = []

    def _add_seq_group(self, meta: SequenceGroupMetadata) -> InterDataForSeqGroup:
        seq_ids = list(meta.seq_data.keys())
        if meta.is_prompt:
            assert len(seq_ids) == 1
        d = InterDataForSeqGroup(request_id=meta.request_id, seq_ids=seq_ids, is_prompt=meta.is_prompt,
                                 block_tables=meta.block_tables,
                                 computed_block_nums=meta.computed_block_nums)
        for i, seq_id in enumerate(seq_ids):
            data = meta.seq_data[seq_id]
            seq_len = data.get_len()
            context_len = data.get_num_computed_tokens() if d.is_prompt else seq_len - 1
            seq_len = min(seq_len, context_len + meta.token_chunk_size)
            if d.is_prompt:
                tokens = data.get_token_ids()[context_len:seq_len]
            else:
                tokens = [data.get_last_token_id()]
            positions = list(range(context_len, seq_len))
            query_len = seq_len - context_len if d.is_prompt else 1
            # prefix-cache hit: the leading computed blocks are skipped (:270-296)
            hit = bool(meta.computed_block_nums) and self.sliding_window is None and d.is_prompt
            d.prefix_cache_hit = hit
            if hit:
                if self.chunked_prefill_enabled:
                    raise RuntimeError("chunked prefill cannot be used with prefix caching now.")
                context_len = len(meta.computed_block_nums) * self.block_size
                tokens = tokens[context_len:]
                positions = positions[context_len:]
                query_len = seq_len - context_len
            d.orig_seq_lens[i] = seq_len
            # sliding window: a decoding sequence only attends to the window (:298-322)
            sw_blocks = 0
            sliding_seq_len = seq_len
            if not d.is_prompt and self.sliding_window is not None:
                sw_blocks = self.sliding_window_blocks
                if self.scheduler_config.use_v2_block_manager:
                    suff = seq_len % self.block_size
                    sliding_seq_len = min(seq_len, self.block_aligned_sliding_window + suff)
                    if suff > 0:
                        sw_blocks += 1
                else:
                    sliding_seq_len = min(seq_len, self.sliding_window)
            d.seq_lens[i] = sliding_seq_len
            d.curr_sliding_window_blocks[i] = sw_blocks
            d.context_lens[i] = context_len
            d.query_lens[i] = query_len
            d.input_tokens[i] = tokens
            d.input_positions[i] = positions
        return d

    def prepare_model_input(self, metas: List[SequenceGroupMetadata]) -> ModelInput:
        self.inter_data_list = [self._add_seq_group(m) for m in metas]
        tokens: List[int] = []
        positions: List[int] = []
        seq_lens: List[int] = []
        query_lens: List[int] = []
        sample_indices: List[int] = []
        sample_seq_ids: List[int] = []
        decode_only = True
        for d, m in zip(self.inter_data_list, metas):
            if d.is_prompt:
                decode_only = False
            for i, seq_id in enumerate(d.seq_ids):
                tokens.extend(d.input_tokens[i])
                positions.extend(d.input_positions[i])
                if m.do_sample:
                    sample_indices.append(len(tokens) - 1)  # last token of the sequence's chunk
                    sample_seq_ids.append(seq_id)
            seq_lens.extend(d.seq_lens)
            query_lens.extend(d.query_lens)
        if not tokens:
            return ModelInput()
        builder = self.attn_backend.make_metadata_builder(self)
        attn_metadata = builder.build(seq_lens, query_lens, -1, len(tokens))
        pin = torch.cuda.is_available()
        t = torch.tensor(tokens, dtype=torch.long)
        p = torch.tensor(positions, dtype=torch.long)
        if pin:
            t, p = t.pin_memory(), p.pin_memory()
        return ModelInput(input_tokens=t, input_positions=p, attn_metadata=attn_metadata,
                          seq_lens=seq_lens, query_lens=query_lens, sample_indices=sample_indices,
                          sample_seq_ids=sample_seq_ids, decode_only=decode_only)

    @staticmethod
    def prepare_worker_input(out: SchedulerOutput) -> WorkerInput:
        def pairs(lst):
            return torch.tensor(lst, dtype=torch.int64).view(-1, 2)
        return WorkerInput(num_seq_groups=len(out.seq_group_metadata_list),
                           blocks_to_swap_in=pairs(out.blocks_to_swap_in),
                           blocks_to_swap_out=pairs(out.blocks_to_swap_out),
                           blocks_to_copy=pairs(out.blocks_to_copy))

    def __call__(self, out: SchedulerOutput) -> ExecuteInput:
        return ExecuteInput(worker_input=self.prepare_worker_input(out),
                            model_input=self.prepare_model_input(out.seq_group_metadata_list))


def token_rows(metas: List[SequenceGroupMetadata], block_size: int, sliding_window: Optional[int]):
    """Where every sequence of a step sits in the token batch `ModelInputBuilder.prepare_model_input` lays out: a list of
    (seq id, meta, first row, rows, context length, end) in batch order -- a prompt chunk contributes its not yet computed
    tokens [context, end) (after a prefix-cache hit: the tokens behind the cached blocks), a decoding sequence one row.
    Used by the log-probability path to find the rows of a prompt's positions (tests/test_input_builder.py checks it
    against the builder on every recorded scenario)."""
    out = []
    row = 0
    for m in metas:
        for sid, data in m.seq_data.items():
            if not m.is_prompt:
                ctx, end, n = data.get_len() - 1, data.get_len(), 1
            else:
                ctx = data.get_num_computed_tokens()
                end = min(data.get_len(), ctx + m.token_chunk_size)
                if m.computed_block_nums and sliding_window is None:
                    ctx = len(m.computed_block_nums) * block_size
                n = end - ctx
            out.append((sid, m, row, n, ctx, end))
            row += n
    return out


class DecodeStepArrays:
    """The five arrays a captured decode step reads -- token ids, positions, slot mapping, sequence
    lengths, block tables -- as numpy views of ONE staging buffer, filled straight from the scheduler's
    metadata.  Same values as ModelInputBuilder + the attention metadata builder produce for a step
    that holds only decode tokens of single-sequence groups without a sliding window
    (model_input_builder.py:212-378 with is_prompt == False: token = last token, position =
    seq_len - 1, slot = table[(seq_len - 1) // block_size] * block_size + (seq_len - 1) % block_size;
    backends/utils.py:31-75), checked against that path by tests/test_input_builder.py; what it
    skips is the Python-list and torch.tensor traffic: the engine thread's turnaround between a
    step's result and the launch of the group's next step is time one stream spends alone on the
    GPU.  Block-table rows are rewritten only when the sequence's table differs from the list this
    row held last time (exact list comparison, so any reallocation -- swap, fork, recompute -- is seen).
    """

    def __init__(self, batch_size: int, max_blocks_per_seq: int, block_size: int, buffer: Optional[np.ndarray] = None):
        B, W = batch_size, max_blocks_per_seq
        self.batch_size, self.width, self.block_size = B, W, block_size
        self.nbytes = self.layout(B, W)[-1]
        buf = np.zeros(self.nbytes, dtype=np.uint8) if buffer is None else buffer
        assert buf.dtype == np.uint8 and buf.size == self.nbytes
        o_ids, o_pos, o_slot, o_len, o_bt, o_state, _ = self.layout(B, W)
        self.buffer = buf
        self.input_ids = buf[o_ids:o_pos].view(np.int64)
        self.positions = buf[o_pos:o_slot].view(np.int64)
        self.slot_mapping = buf[o_slot:o_len].view(np.int64)
        self.seq_lens = buf[o_len:o_bt].view(np.int32)
        self.block_tables = buf[o_bt:o_state].view(np.int32).reshape(B, W)
        # sampler state slot of each row (device_sampler.py): -1 = plain greedy; read by the sampling kernel only
        self.state_slots = buf[o_state:].view(np.int32)
        self.slot_mapping[:] = -1
        self.state_slots[:] = -1
        self._rows: List[Optional[List[int]]] = [None] * B  # the table each row holds
        self._arange = np.arange(B)

    @staticmethod
    def layout(B: int, W: int):
        """Byte offsets of (input_ids i64[B], positions i64[B], slot_mapping i64[B], seq_lens i32[B],
        block_tables i32[B, W], sampler state slots i32[B]) and the total size."""
        return 0, 8 * B, 16 * B, 24 * B, 28 * B, 28 * B + 4 * B * W, 32 * B + 4 * B * W

    @staticmethod
    def eligible(metas, worker_lists_empty: bool, sliding_window) -> bool:
        if not worker_lists_empty or sliding_window is not None or not metas:
            return False
        for m in metas:
            if m.is_prompt or not m.do_sample or len(m.seq_data) != 1 or not m.block_tables:
                return False
        return True

    def fill(self, metas, state_slots: Optional[List[int]] = None) -> Optional[List[int]]:
        """Writes the step into the staging arrays (rows past len(metas) become padding: slot -1,
        length 0) and returns the sequence ids in row order; None when a block table is wider than
        the arrays (the caller then takes the general path).  `state_slots`: the rows' sampler state slots
        (None: every row plain greedy)."""
        n = len(metas)
        assert n <= self.batch_size
        seq_ids: List[int] = []
        lens: List[int] = []
        toks: List[int] = []
        bt, rows, W = self.block_tables, self._rows, self.width
        for i, m in enumerate(metas):
            (seq_id, data), = m.seq_data.items()
            table = m.block_tables[seq_id]
            old = rows[i]
            if old is None or table != old:
                k = len(table)
                if k > W:
                    rows[i] = None  # whatever the row holds now is not a table any more
                    return None
                if old is not None and k == len(old) + 1 and table[:-1] == old:
                    bt[i, k - 1] = table[-1]
                else:
                    bt[i, :k] = table
                rows[i] = list(table)  # a private copy: the comparison must not depend on who owns `table`
            seq_ids.append(seq_id)
            lens.append(data.get_len())
            toks.append(data.get_last_token_id())
        L = np.asarray(lens, dtype=np.int64)
        pos = L - 1
        self.input_ids[:n] = toks
        self.positions[:n] = pos
        self.seq_lens[:n] = L
        blk = bt[self._arange[:n], pos // self.block_size].astype(np.int64)
        self.slot_mapping[:n] = blk * self.block_size + pos % self.block_size
        if n < self.batch_size:
            self.slot_mapping[n:] = -1
            self.seq_lens[n:] = 0
        self.state_slots[:] = -1
        if state_slots is not None:
            self.state_slots[:n] = state_slots
        return seq_ids


class MixedStepArrays:
    """DecodeStepArrays for a step that mixes prompt chunks and decode tokens (chunked prefill under a
    captured graph): token ids, positions, slot mapping [max_tokens]; rows to sample, sequence lengths,
    query_start_loc, block tables [max_seqs] -- numpy views of one staging buffer, filled from the
    scheduler's metadata with the values ModelInputBuilder + the attention metadata builder compute
    (model_input_builder.py:212-378: a prompt chunk covers tokens [computed, min(len, computed + chunk)),
    a decode sequence its last token; slot of position p = table[p // block_size] * block_size +
    p % block_size; a sequence samples from the last row of its chunk).  Single-sequence groups, no
    sliding window, no prefix-cache hit (the general builder refuses that with chunked prefill)."""

    def __init__(self, max_tokens: int, max_seqs: int, max_blocks_per_seq: int, block_size: int,
                 buffer: Optional[np.ndarray] = None):
        T, S, W = max_tokens, max_seqs, max_blocks_per_seq
        self.max_tokens, self.max_seqs, self.width, self.block_size = T, S, W, block_size
        o = self.layout(T, S, W)
        self.nbytes = o[-1]
        buf = np.zeros(self.nbytes, dtype=np.uint8) if buffer is None else buffer
        assert buf.dtype == np.uint8 and buf.size == self.nbytes
        self.buffer = buf
        self.input_ids = buf[o[0]:o[1]].view(np.int64)
        self.positions = buf[o[1]:o[2]].view(np.int64)
        self.slot_mapping = buf[o[2]:o[3]].view(np.int64)
        self.sample_rows = buf[o[3]:o[4]].view(np.int64)
        self.seq_lens = buf[o[4]:o[5]].view(np.int32)
        self.query_start_loc = buf[o[5]:o[6]].view(np.int32)
        self.block_tables = buf[o[6]:o[7]].view(np.int32).reshape(S, W)
        self.slot_mapping[:] = -1
        self._rows: List[Optional[List[int]]] = [None] * S

    @staticmethod
    def layout(T: int, S: int, W: int):
        """Byte offsets of input_ids, positions, slot_mapping (i64[T]), sample_rows (i64[S]), seq_lens
        (i32[S]), query_start_loc (i32[S + 1]), block_tables (i32[S, W]) and the total size."""
        a = [0, 8 * T, 16 * T, 24 * T]
        a.append(a[-1] + 8 * S)
        a.append(a[-1] + 4 * S)
        a.append(a[-1] + 4 * (S + 1))
        a.append(a[-1] + 4 * S * W)
        return a

    @staticmethod
    def eligible(metas, worker_lists_empty: bool, sliding_window) -> bool:
        if not worker_lists_empty or sliding_window is not None or not metas:
            return False
        for m in metas:
            if len(m.seq_data) != 1 or not m.block_tables or (m.is_prompt and m.computed_block_nums):
                return False
        return True

    def fill(self, metas):
        """Returns (sequence ids that sample, number of tokens) or None when the step does not fit."""
        ns = len(metas)
        if ns > self.max_seqs:
            return None
        bs, bt, rows, W = self.block_size, self.block_tables, self._rows, self.width
        cursor = 0
        sample_rows: List[int] = []
        sample_ids: List[int] = []
        qsl = self.query_start_loc
        qsl[0] = 0
        for i, m in enumerate(metas):
            (seq_id, data), = m.seq_data.items()
            table = m.block_tables[seq_id]
            old = rows[i]
            if old is None or table != old:
                k = len(table)
                if k > W:
                    rows[i] = None
                    return None
                bt[i, :k] = table
                rows[i] = list(table)
            if m.is_prompt:
                ctx = data.get_num_computed_tokens()
                end = min(data.get_len(), ctx + m.token_chunk_size)
                q = end - ctx
                if cursor + q > self.max_tokens:
                    return None
                self.input_ids[cursor:cursor + q] = data.get_token_ids()[ctx:end]
                pos = np.arange(ctx, end, dtype=np.int64)
                self.positions[cursor:cursor + q] = pos
                self.slot_mapping[cursor:cursor + q] = bt[i, pos // bs].astype(np.int64) * bs + pos % bs
                self.seq_lens[i] = end
            else:
                L = data.get_len()
                q = 1
                if cursor + 1 > self.max_tokens:
                    return None
                self.input_ids[cursor] = data.get_last_token_id()
                self.positions[cursor] = L - 1
                self.slot_mapping[cursor] = int(bt[i, (L - 1) // bs]) * bs + (L - 1) % bs
                self.seq_lens[i] = L
            cursor += q
            qsl[i + 1] = cursor
            if m.do_sample:
                sample_rows.append(cursor - 1)
                sample_ids.append(seq_id)
        n = cursor
        self.slot_mapping[n:] = -1          # padding tokens write no cache
        self.seq_lens[ns:] = 0
        qsl[ns + 1:] = n                    # padding sequences own no tokens
        k = len(sample_rows)
        self.sample_rows[:k] = sample_rows
        self.sample_rows[k:] = 0
        return sample_ids, n
