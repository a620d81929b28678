"""HIP-graph replay of the decode step.

A decode step is ~290 short kernels (32 layers x 9 launches); launched eagerly it is bound by
the host (3-4 us per launch).  The step is captured once per padded batch size into a HIP
graph over static input buffers (token ids, positions, slot mapping, block tables, sequence
lengths) and replayed: the host then pays one launch per step.  This is the "HIP streams and
graphs instead of a tracing compiler" choice of the design; the reference's decoding workflow
runs eager.

Attention inside a captured step always uses the partitioned kernel (paged_attention_v2) with
the partition count of the longest context the block table can hold: partitions past a
sequence's length exit immediately, so one graph serves every context length.
"""
from typing import Dict, List, Optional

import torch

from ..attention.backend import PagedAttnMetadata
from .input_builder import DecodeStepArrays, MixedStepArrays


class DecodeGraph:
    def __init__(self, model, kv_caches: List[torch.Tensor], batch_size: int,
                 max_blocks_per_seq: int, block_size: int, device, sampler=None):
        """`sampler` (device_sampler.DeviceSampler): the captured step ends with the lm_head's logits and the sampling
        kernel (rows with a state slot are sampled, the others take the arg-max); None: the lm_head's arg-max
        epilogue, no logits tensor (every row plain greedy)."""
        self.model = model
        self.sampler = sampler
        self.kv_caches = kv_caches
        self.batch_size = batch_size
        self.max_blocks_per_seq = max_blocks_per_seq
        self.block_size = block_size
        dev = torch.device(device)
        # the five static inputs are views of one device buffer, mirrored by one pinned staging
        # buffer on the host: a step's inputs arrive with a single copy (DecodeStepArrays)
        B, W = batch_size, max_blocks_per_seq
        o_ids, o_pos, o_slot, o_len, o_bt, o_state, total = DecodeStepArrays.layout(B, W)
        self.packed = torch.zeros(total, dtype=torch.uint8, device=dev)
        self.input_ids = self.packed[o_ids:o_pos].view(torch.int64)
        self.positions = self.packed[o_pos:o_slot].view(torch.int64)
        self.slot_mapping = self.packed[o_slot:o_len].view(torch.int64)
        self.seq_lens = self.packed[o_len:o_bt].view(torch.int32)
        self.block_tables = self.packed[o_bt:o_state].view(torch.int32).view(B, W)
        self.state_slots = self.packed[o_state:].view(torch.int32)
        self.slot_mapping.fill_(-1)
        self.state_slots.fill_(-1)
        self.host = torch.zeros(total, dtype=torch.uint8, pin_memory=dev.type == "cuda")
        self.staging = DecodeStepArrays(B, W, block_size, self.host.numpy())
        # pinned landing buffers of the sampled tokens: a result is read by the host right after its step
        # completes, long before the same graph has run four more steps
        self._host_tokens = [torch.empty(B, dtype=torch.int64, pin_memory=dev.type == "cuda") for _ in range(4)]
        self._host_tokens_next = 0
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.hidden: Optional[torch.Tensor] = None
        self.next_tokens: Optional[torch.Tensor] = None
        # multi-step decode: the tokens of model step j of a burst land in row j, on the device and (one copy
        # per burst) in a pinned twin
        self._scratch = None  # this graph's own paged_attention_v2 scratch (see PagedAttnMetadata.decode_scratch)
        # this graph's own working rows of the sampling launch (allocated and zeroed once, outside the capture)
        self._sampler_scratch = sampler.new_scratch(batch_size, dev) if sampler is not None else None
        self.token_log: Optional[torch.Tensor] = None
        self._host_logs: List[torch.Tensor] = []
        self._host_logs_next = 0

    def replay_steps(self, k: int) -> torch.Tensor:
        """k model steps back to back on the current stream with no host round trip: replay, then
        advance_step on the device (tokens <- sampled ids, positions and lengths + 1, slots through the block
        table: csrc/prepare_inputs/advance_step.cu:14-57), k times; returns the [k, B] tokens.  The block
        tables must already cover the k - 1 positions ahead (lookahead slots of the block manager)."""
        B = self.batch_size
        if self.token_log is None or self.token_log.shape[0] < k:
            self.token_log = torch.zeros(k, B, dtype=torch.int64, device=self.packed.device)
            pin = self.packed.device.type == "cuda"
            self._host_logs = [torch.empty(k, B, dtype=torch.int64, pin_memory=pin) for _ in range(4)]
        for j in range(k):
            self.graph.replay()
            if j + 1 < k:
                torch.ops._C_amd.advance_step_logged(self.block_size, self.input_ids, self.next_tokens, self.positions,
                                                     self.seq_lens, self.slot_mapping, self.block_tables,
                                                     self.token_log[j])
            else:
                self.token_log[j].copy_(self.next_tokens)
        return self.token_log[:k]

    def next_host_log(self, k: int) -> torch.Tensor:
        t = self._host_logs[self._host_logs_next]
        self._host_logs_next = (self._host_logs_next + 1) % len(self._host_logs)
        return t[:k]

    def _metadata(self) -> PagedAttnMetadata:
        if self._scratch is None and hasattr(self.model.attn, "make_v2_scratch"):
            self._scratch = self.model.attn.make_v2_scratch(self.batch_size, self.max_blocks_per_seq * self.block_size,
                                                           self.model.cfg.dtype, self.packed.device)
        return PagedAttnMetadata(
            num_prefills=0, num_prefill_tokens=0, num_decode_tokens=self.batch_size,
            slot_mapping=self.slot_mapping, seq_lens=None, seq_lens_tensor=self.seq_lens,
            max_query_len=1, max_prefill_seq_len=0,
            max_decode_seq_len=self.max_blocks_per_seq * self.block_size,
            query_start_loc=None, seq_start_loc=None, context_lens_tensor=None,
            block_tables=self.block_tables, use_cuda_graph=True, decode_scratch=self._scratch)

    def _step(self):
        hidden = self.model.forward(self.input_ids, self.positions, self.kv_caches, self._metadata())
        if self.sampler is not None:  # logits -> one sampling launch, still inside the captured step
            return hidden, self.sampler.sample(self.model.compute_logits(hidden), self.state_slots,
                                               scratch=self._sampler_scratch)
        return hidden, self.model.greedy_tokens(hidden)  # greedy sampling stays on the device

    def capture(self, stream: Optional[torch.cuda.Stream] = None) -> None:
        s = stream or torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):  # warm up allocators / hipBLASLt heuristics outside the capture
                self._step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: another stream may be replaying (and its waiter thread synchronising an
        # event) while this thread captures
        with torch.cuda.graph(self.graph, stream=s, capture_error_mode="thread_local"):
            self.hidden, self.next_tokens = self._step()
        torch.cuda.synchronize()

    def load(self, input_ids, positions, slot_mapping, block_tables, seq_lens, state_slots=None) -> None:
        """Copy one step's inputs (device or pinned-host tensors) into the static buffers."""
        n = input_ids.shape[0]
        self.state_slots.fill_(-1)
        if state_slots is not None:
            self.state_slots[:n].copy_(state_slots, non_blocking=True)
        self.input_ids[:n].copy_(input_ids, non_blocking=True)
        self.positions[:n].copy_(positions, non_blocking=True)
        self.slot_mapping[:n].copy_(slot_mapping, non_blocking=True)
        self.seq_lens[:n].copy_(seq_lens, non_blocking=True)
        w = block_tables.shape[1]
        self.block_tables[:n, :w].copy_(block_tables, non_blocking=True)
        if n < self.batch_size:  # padding rows: no cache write, empty context
            self.slot_mapping[n:].fill_(-1)
            self.seq_lens[n:].zero_()

    def next_host_tokens(self) -> torch.Tensor:
        t = self._host_tokens[self._host_tokens_next]
        self._host_tokens_next = (self._host_tokens_next + 1) % len(self._host_tokens)
        return t

    def load_staged(self) -> None:
        """The step written into `staging` (DecodeStepArrays.fill) goes to the device in one copy."""
        self.packed.copy_(self.host, non_blocking=True)

    def replay(self) -> torch.Tensor:
        self.graph.replay()
        return self.next_tokens


class MixedGraph:
    """One captured graph for every mixed step (prompt chunks + decode tokens) of at most `max_tokens`
    tokens in at most `max_seqs` sequences.  Everything the step depends on is data in static device
    buffers -- token ids, positions, slot mapping, block tables, sequence lengths, query_start_loc and
    the rows to sample -- because the prefill kernel takes all of it from the device; padding tokens
    write no cache (slot -1) and padding sequences own no tokens."""

    def __init__(self, model, kv_caches: List[torch.Tensor], max_tokens: int, max_seqs: int,
                 max_blocks_per_seq: int, block_size: int, device):
        self.model, self.kv_caches = model, kv_caches
        self.max_tokens, self.max_seqs = max_tokens, max_seqs
        self.max_blocks_per_seq, self.block_size = max_blocks_per_seq, block_size
        dev = torch.device(device)
        # static inputs = views of one device buffer, mirrored by one pinned staging buffer (MixedStepArrays)
        T, S, W = max_tokens, max_seqs, max_blocks_per_seq
        o = MixedStepArrays.layout(T, S, W)
        self.packed = torch.zeros(o[-1], dtype=torch.uint8, device=dev)
        self.input_ids = self.packed[o[0]:o[1]].view(torch.int64)
        self.positions = self.packed[o[1]:o[2]].view(torch.int64)
        self.slot_mapping = self.packed[o[2]:o[3]].view(torch.int64)
        self.sample_rows = self.packed[o[3]:o[4]].view(torch.int64)
        self.seq_lens = self.packed[o[4]:o[5]].view(torch.int32)
        self.query_start_loc = self.packed[o[5]:o[6]].view(torch.int32)
        self.block_tables = self.packed[o[6]:o[7]].view(torch.int32).view(S, W)
        self.slot_mapping.fill_(-1)
        self.host = torch.zeros(o[-1], dtype=torch.uint8, pin_memory=dev.type == "cuda")
        self.staging = MixedStepArrays(T, S, W, block_size, self.host.numpy())
        self._host_tokens = [torch.empty(S, dtype=torch.int64, pin_memory=dev.type == "cuda") for _ in range(4)]
        self._host_tokens_next = 0
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.next_tokens: Optional[torch.Tensor] = None

    def _step(self):
        unified = (self.block_tables, self.seq_lens, self.query_start_loc, self.max_tokens, self.slot_mapping)
        hidden = self.model.forward(self.input_ids, self.positions, self.kv_caches, None, unified=unified)
        return self.model.greedy_tokens(hidden[self.sample_rows])

    def capture(self) -> None:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                self._step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=s, capture_error_mode="thread_local"):
            self.next_tokens = self._step()
        torch.cuda.synchronize()

    def fits(self, num_tokens: int, num_seqs: int, table_width: int) -> bool:
        return (num_tokens <= self.max_tokens and num_seqs <= self.max_seqs
                and table_width <= self.max_blocks_per_seq)

    def load(self, input_ids, positions, slot_mapping, block_tables, seq_lens, query_start_loc,
             sample_indices: List[int]) -> None:
        n, ns = input_ids.shape[0], seq_lens.shape[0]
        self.input_ids[:n].copy_(input_ids, non_blocking=True)
        self.positions[:n].copy_(positions, non_blocking=True)
        self.slot_mapping[:n].copy_(slot_mapping, non_blocking=True)
        self.slot_mapping[n:].fill_(-1)
        self.seq_lens[:ns].copy_(seq_lens, non_blocking=True)
        self.seq_lens[ns:].zero_()
        self.query_start_loc[:ns + 1].copy_(query_start_loc, non_blocking=True)
        self.query_start_loc[ns + 1:].fill_(n)  # padding sequences own no tokens
        self.block_tables[:ns, :block_tables.shape[1]].copy_(block_tables, non_blocking=True)
        rows = torch.tensor(sample_indices + [0] * (self.max_seqs - len(sample_indices)), dtype=torch.int64)
        if torch.cuda.is_available():
            rows = rows.pin_memory()
        self.sample_rows.copy_(rows, non_blocking=True)

    def load_staged(self) -> None:
        """The step written into `staging` (MixedStepArrays.fill) goes to the device in one copy."""
        self.packed.copy_(self.host, non_blocking=True)

    def next_host_tokens(self) -> torch.Tensor:
        t = self._host_tokens[self._host_tokens_next]
        self._host_tokens_next = (self._host_tokens_next + 1) % len(self._host_tokens)
        return t

    def replay(self) -> torch.Tensor:
        self.graph.replay()
        return self.next_tokens


class DecodeGraphPool:
    """One captured graph per padded batch size (powers of two and multiples of 8)."""

    def __init__(self, model, kv_caches, max_blocks_per_seq: int, block_size: int, device):
        self.model, self.kv_caches = model, kv_caches
        self.max_blocks_per_seq, self.block_size, self.device = max_blocks_per_seq, block_size, device
        self.graphs: Dict[int, DecodeGraph] = {}
        self.sampler_graphs: Dict[int, DecodeGraph] = {}  # the flavour that ends with the sampling kernel
        self.mixed: Optional[MixedGraph] = None

    @staticmethod
    def padded(batch_size: int) -> int:
        if batch_size <= 8:
            p = 1
            while p < batch_size:
                p *= 2
            return p
        return (batch_size + 7) // 8 * 8

    def _off_default_stream(self, build):
        """Runs `build()` (construct + capture a graph) on a side stream when the caller sits on the default
        stream: two graphs built there replay one after the other even on different streams (two steps in
        flight measured 27 % slower, as fast as one)."""
        cur = torch.cuda.current_stream(self.device)
        if cur != torch.cuda.default_stream(self.device):
            return build()
        side = torch.cuda.Stream(self.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            g = build()
        cur.wait_stream(side)
        return g

    def get_mixed(self, max_tokens: int, max_seqs: int) -> MixedGraph:
        if self.mixed is None:
            def build():
                g = MixedGraph(self.model, self.kv_caches, max_tokens, max_seqs, self.max_blocks_per_seq,
                               self.block_size, self.device)
                g.capture()
                return g
            self.mixed = self._off_default_stream(build)
        return self.mixed

    def get(self, batch_size: int, sampler=None) -> DecodeGraph:
        """`sampler` (DeviceSampler): the graph flavour whose step ends with logits + the sampling kernel, captured
        the first time a step with a non-greedy request comes by."""
        p = self.padded(batch_size)
        pool = self.graphs if sampler is None else self.sampler_graphs
        g = pool.get(p)
        if g is None:
            def build():
                g = DecodeGraph(self.model, self.kv_caches, p, self.max_blocks_per_seq, self.block_size, self.device,
                                sampler=sampler)
                g.capture()
                return g
            g = self._off_default_stream(build)
            pool[p] = g
        return g
