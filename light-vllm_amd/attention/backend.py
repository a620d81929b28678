"""Paged-attention backend for the decoding workflow (plugin boundary B).

The reference's decoding selector can only return its flash-attn backend
(light_vllm/decoding/backends/attention/selector.py:58-75); its paged path
(ops/paged_attn.py) is never wired.  This module is that wiring, MI355X-native: a
`DecodeOnlyAttentionBackend`-shaped class (abstract.py:15-72) whose Impl writes K/V with
`reshape_and_cache` and decodes with `paged_attention_v1/v2` -- the hand-written gfx950
kernels -- and whose Metadata / Builder carry the same fields as the flash backend's
(flash_attn.py:76-365) so the engine's input builder and executors are unchanged.

Prompt (prefill, chunked prefill, prefix hits) attention runs the HIP varlen kernel over the
paged cache (`PagedAttention.forward_prefix` -> lvllm_paged_prefill_attention); torch SDPA remains
for the memory-profiling run without a cache, sliding windows and fp8 caches.
"""
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Tuple, Type

import numpy as np
import torch

from ..paged_attn import PagedAttention, num_compute_units

PAD_SLOT_ID = -1  # backends/utils.py:13


def is_block_tables_empty(block_tables) -> bool:
    """None, or {seq_id: None, ...}: the memory-profiling run (backends/utils.py:16-26)."""
    if block_tables is None:
        return True
    return isinstance(block_tables, dict) and all(v is None for v in block_tables.values())


def compute_slot_mapping_start_idx(is_prompt: bool, query_len: int, context_len: int,
                                   sliding_window: Optional[int], use_v2_block_manager: bool) -> int:
    """First prompt position whose K/V is written (backends/utils.py:31-45): with a sliding
    window the head of a long prompt is never cached."""
    if is_prompt and sliding_window is not None:
        assert use_v2_block_manager or context_len == 0, (
            "Prefix caching is currently not supported with sliding window attention in V1 block manager")
        return max(0, query_len - sliding_window)
    return 0


def compute_slot_mapping(is_profile_run: bool, slot_mapping: List[int], seq_id: int, seq_len: int,
                         context_len: int, start_idx: int, block_size: int,
                         block_tables: Dict[int, List[int]]) -> None:
    """Append the cache slots of tokens [context_len, seq_len) of one sequence
    (backends/utils.py:48-75): slot = block_table[i // block_size] * block_size + i % block_size,
    PAD_SLOT_ID for tokens before start_idx and for the profile run."""
    if is_profile_run:
        slot_mapping.extend([PAD_SLOT_ID] * seq_len)
        return
    n_pad = max(0, start_idx - context_len)
    if n_pad:
        slot_mapping.extend([PAD_SLOT_ID] * n_pad)
    first = max(start_idx, context_len)
    if first >= seq_len:
        return
    if seq_len - first == 1:  # decode: one token
        i = first
        slot_mapping.append(block_tables[seq_id][i // block_size] * block_size + i % block_size)
        return
    pos = np.arange(first, seq_len, dtype=np.int64)
    table = np.asarray(block_tables[seq_id], dtype=np.int64)
    slot_mapping.extend((table[pos // block_size] * block_size + pos % block_size).tolist())


def make_tensor_with_pad(rows: List[List[int]], pad: int, dtype, max_len: Optional[int] = None,
                         pin_memory: bool = False) -> torch.Tensor:
    """Ragged rows -> [len(rows), max_len] CPU tensor padded with `pad`
    (light_vllm/utils.py make_tensor_with_pad, as used at flash_attn.py:317-321)."""
    width = max((len(r) for r in rows), default=0)
    if max_len is not None:
        width = max(width, max_len)
    arr = np.full((len(rows), width), pad, dtype=np.int64)
    for i, r in enumerate(rows):
        if r:
            arr[i, :len(r)] = r
    t = torch.from_numpy(arr).to(dtype)
    return t.pin_memory() if pin_memory else t


@dataclass
class PagedAttnMetadata:
    """Same fields and meaning as DecodeOnlyFlashAttentionMetadata (flash_attn.py:76-137):
    prefill sequences first, then decode sequences."""
    num_prefills: int
    num_prefill_tokens: int
    num_decode_tokens: int
    slot_mapping: torch.Tensor
    seq_lens: Optional[List[int]]
    seq_lens_tensor: Optional[torch.Tensor]
    max_query_len: Optional[int]
    max_prefill_seq_len: int
    max_decode_seq_len: int
    query_start_loc: Optional[torch.Tensor]
    seq_start_loc: Optional[torch.Tensor]
    context_lens_tensor: Optional[torch.Tensor]
    block_tables: Optional[torch.Tensor]
    use_cuda_graph: bool = False
    _cached_prefill_metadata: Optional["PagedAttnMetadata"] = None
    _cached_decode_metadata: Optional["PagedAttnMetadata"] = None
    # (not in the reference) host copies the prefill path walks without synchronising the device
    query_lens: Optional[List[int]] = None
    context_lens: Optional[List[int]] = None
    # (not in the reference) paged_attention_v2 scratch (exp_sums, max_logits, tmp_out) owned by the caller: a
    # captured step brings its own, so that two graphs replaying on two streams never share one
    decode_scratch: Optional[tuple] = None

    @property
    def prefill_metadata(self) -> Optional["PagedAttnMetadata"]:
        if self.num_prefills == 0:
            return None
        if self._cached_prefill_metadata is None:
            n = self.num_prefills
            self._cached_prefill_metadata = PagedAttnMetadata(
                num_prefills=n, num_prefill_tokens=self.num_prefill_tokens, num_decode_tokens=0,
                slot_mapping=self.slot_mapping[:self.num_prefill_tokens],
                seq_lens=self.seq_lens[:n], seq_lens_tensor=self.seq_lens_tensor[:n],
                max_query_len=self.max_query_len, max_prefill_seq_len=self.max_prefill_seq_len,
                max_decode_seq_len=0, query_start_loc=self.query_start_loc[:n + 1],
                seq_start_loc=self.seq_start_loc[:n + 1],
                context_lens_tensor=self.context_lens_tensor[:n],
                block_tables=self.block_tables[:n], use_cuda_graph=False,
                query_lens=self.query_lens[:n] if self.query_lens is not None else None,
                context_lens=self.context_lens[:n] if self.context_lens is not None else None)
        return self._cached_prefill_metadata

    @property
    def decode_metadata(self) -> Optional["PagedAttnMetadata"]:
        if self.num_decode_tokens == 0:
            return None
        if self._cached_decode_metadata is None:
            n = self.num_prefills
            self._cached_decode_metadata = PagedAttnMetadata(
                num_prefills=0, num_prefill_tokens=0, num_decode_tokens=self.num_decode_tokens,
                slot_mapping=self.slot_mapping[self.num_prefill_tokens:], seq_lens=None,
                seq_lens_tensor=self.seq_lens_tensor[n:], max_query_len=None,
                max_prefill_seq_len=0, max_decode_seq_len=self.max_decode_seq_len,
                query_start_loc=None, seq_start_loc=None, context_lens_tensor=None,
                block_tables=self.block_tables[n:], use_cuda_graph=self.use_cuda_graph,
                decode_scratch=self.decode_scratch)
        return self._cached_decode_metadata

    def asdict_zerocopy(self, skip_fields=None) -> Dict[str, Any]:
        """dataclasses.asdict without the deep copy (abstract.py:105-117)."""
        import dataclasses
        skip = skip_fields or set()
        return {f.name: getattr(self, f.name) for f in dataclasses.fields(self) if f.name not in skip}

    def to(self, device, non_blocking=True):
        for k, v in list(self.__dict__.items()):
            if isinstance(v, torch.Tensor):
                self.__dict__[k] = v.to(device=device, non_blocking=non_blocking)
        self._cached_prefill_metadata = None
        self._cached_decode_metadata = None
        return self


class PagedAttnMetadataBuilder:
    """build(seq_lens, query_lens, cuda_graph_pad_size, batch_size) over the input builder's
    per-sequence-group records, as DecodeOnlyFlashAttentionMetadataBuilder does
    (flash_attn.py:208-365).  Tensors are created on the CPU (pinned when a GPU is
    present) and moved by the executor, exactly as in the reference."""

    def __init__(self, input_builder):
        self.input_builder = input_builder
        self.sliding_window = input_builder.sliding_window
        self.block_size = input_builder.block_size
        self.use_v2_block_manager = input_builder.scheduler_config.use_v2_block_manager
        # Plain prompts get their block table too (the reference leaves it empty and sends them
        # to the dense flash-attn call, flash_attn.py:518-536): the HIP prefill kernel reads the
        # chunk's K/V back from the paged cache, so one code path serves prompts, chunks and
        # prefix hits.  Off with a sliding window (v1 tables are rings there).
        self.prompt_block_tables = getattr(input_builder, "prompt_block_tables", self.sliding_window is None)
        self.slot_mapping: List[int] = []
        self.prefill_seq_lens: List[int] = []
        self.context_lens: List[int] = []
        self.block_tables: List[List[int]] = []
        self.curr_seq_lens: List[int] = []
        self.num_prefills = 0
        self.num_prefill_tokens = 0
        self.num_decode_tokens = 0

    def _add_seq_group(self, inter_data, chunked_prefill_enabled: bool, prefix_cache_hit: bool):
        is_prompt = inter_data.is_prompt
        block_tables = inter_data.block_tables
        is_profile_run = is_block_tables_empty(block_tables)
        for i, seq_id in enumerate(inter_data.seq_ids):
            token_len = len(inter_data.input_tokens[i])
            seq_len = inter_data.orig_seq_lens[i]
            curr_seq_len = inter_data.seq_lens[i]
            query_len = inter_data.query_lens[i]
            context_len = inter_data.context_lens[i]
            curr_sw_blocks = inter_data.curr_sliding_window_blocks[i]
            self.context_lens.append(context_len)
            if is_prompt:
                self.num_prefills += 1
                self.num_prefill_tokens += token_len
                self.prefill_seq_lens.append(seq_len)
            else:
                assert query_len == 1, f"seq_len: {seq_len}, context_len: {context_len}, query_len: {query_len}"
                self.num_decode_tokens += query_len
                self.curr_seq_lens.append(curr_seq_len)
            # block table of the sequence (flash_attn.py:262-273)
            block_table: List[int] = []
            if prefix_cache_hit:
                block_table = block_tables[seq_id]
            elif (chunked_prefill_enabled or not is_prompt or self.prompt_block_tables) and block_tables is not None:
                block_table = block_tables[seq_id][-curr_sw_blocks:] if curr_sw_blocks else block_tables[seq_id]
            self.block_tables.append(block_table)
            start_idx = compute_slot_mapping_start_idx(is_prompt, query_len, context_len,
                                                       self.sliding_window, self.use_v2_block_manager)
            compute_slot_mapping(is_profile_run, self.slot_mapping, seq_id, seq_len, context_len,
                                 start_idx, self.block_size, block_tables)

    def build(self, seq_lens: List[int], query_lens: List[int], cuda_graph_pad_size: int,
              batch_size: int) -> PagedAttnMetadata:
        groups = self.input_builder.inter_data_list
        prefix_cache_hit = any(d.prefix_cache_hit for d in groups)
        for d in groups:
            self._add_seq_group(d, self.input_builder.chunked_prefill_enabled, prefix_cache_hit)
        pin = torch.cuda.is_available()
        max_query_len = max(query_lens)
        assert max_query_len > 0, f"query_lens: {query_lens}"

        def cpu(data, dtype):
            t = torch.tensor(data, dtype=dtype)
            return t.pin_memory() if pin else t

        block_tables = make_tensor_with_pad(self.block_tables, pad=0, dtype=torch.int32, pin_memory=pin)
        seq_lens_tensor = cpu(seq_lens, torch.int32)
        query_start_loc = torch.zeros(len(query_lens) + 1, dtype=torch.int32)
        seq_start_loc = torch.zeros(len(seq_lens) + 1, dtype=torch.int32)
        torch.cumsum(seq_lens_tensor, dim=0, dtype=torch.int32, out=seq_start_loc[1:])
        torch.cumsum(torch.tensor(query_lens, dtype=torch.int64), dim=0, dtype=torch.int32,
                     out=query_start_loc[1:])
        if pin:
            query_start_loc, seq_start_loc = query_start_loc.pin_memory(), seq_start_loc.pin_memory()
        return PagedAttnMetadata(
            num_prefills=self.num_prefills, num_prefill_tokens=self.num_prefill_tokens,
            num_decode_tokens=self.num_decode_tokens,
            slot_mapping=cpu(self.slot_mapping, torch.int64), seq_lens=list(seq_lens),
            seq_lens_tensor=seq_lens_tensor, max_query_len=max_query_len,
            max_prefill_seq_len=max(self.prefill_seq_lens, default=0),
            max_decode_seq_len=max(self.curr_seq_lens, default=0),
            query_start_loc=query_start_loc, seq_start_loc=seq_start_loc,
            context_lens_tensor=cpu(self.context_lens, torch.int32), block_tables=block_tables,
            use_cuda_graph=False, query_lens=list(query_lens), context_lens=list(self.context_lens))

    def __call__(self, *args, **kwargs):
        """AttentionMetadataBuilder's abstract call hook (backends/attention/abstract.py:90-92); the
        decoding workflow never calls it (flash_attn.py:367-368 leaves it empty too)."""
        pass


class PagedAttnImpl:
    """forward(query [T, H*D], key [T, KVH*D], value, kv_cache | None, attn_metadata) -> [T, H*D]
    (abstract.py:137-166).  Token layout: prefill tokens first, then one token per decode
    sequence (flash_attn.py:372-395)."""

    def __init__(self, num_heads: int, head_size: int, scale: float, num_kv_heads: Optional[int] = None,
                 alibi_slopes: Optional[List[float]] = None, sliding_window: Optional[int] = None,
                 kv_cache_dtype: str = "auto", blocksparse_params: Optional[Dict[str, Any]] = None,
                 logits_soft_cap: Optional[float] = None, decode_version: Optional[str] = None) -> None:
        if blocksparse_params is not None:
            raise ValueError("PagedAttn (HIP) does not support block-sparse attention.")
        # Soft cap (cap * tanh(logits / cap); the reference's live decode hands it to flash_attn_with_kvcache,
        # flash_attn.py:554): paged_attention_v1/v2 have no such argument (csrc/ops.h:8-27), so with a cap decode
        # tokens go through the prefill kernel as chunks of one token (`_decode_as_chunks`), which has it.
        self.logits_soft_cap = float(logits_soft_cap) if logits_soft_cap else 0.0
        if head_size not in PagedAttention.get_supported_head_sizes():
            raise ValueError(f"Head size {head_size} is not supported by PagedAttention. "
                             f"Supported head sizes are: {PagedAttention.get_supported_head_sizes()}.")
        self.num_heads = num_heads
        self.head_size = head_size
        self.scale = float(scale)
        self.num_kv_heads = num_heads if num_kv_heads is None else num_kv_heads
        self.alibi_slopes = (torch.tensor(alibi_slopes, dtype=torch.float32)
                             if alibi_slopes is not None else None)
        self.sliding_window = sliding_window
        self.kv_cache_dtype = kv_cache_dtype
        assert self.num_heads % self.num_kv_heads == 0
        self.num_queries_per_kv = self.num_heads // self.num_kv_heads
        self.decode_version = decode_version  # None: heuristic; "v1" | "v2": forced
        self.use_hip_prefill = True  # False: torch SDPA per sequence (kept for A/B tests)
        # rope + cache write + attention in one launch over an fp8 cache too (ModelConfig.rope_in_attention_fp8)
        self.fuse_rope_over_fp8_cache = False
        self._scratch: Dict[Tuple[int, int, int], Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = {}

    def make_v2_scratch(self, num_seqs: int, max_seq_len: int, dtype, device):
        """(exp_sums, max_logits, tmp_out) of paged_attn.py:156-166 for a caller that owns them."""
        parts = (max_seq_len + 511) // 512
        tmp = torch.empty(num_seqs, self.num_heads, parts, self.head_size, dtype=dtype, device=device)
        es = torch.empty(num_seqs, self.num_heads, parts, dtype=torch.float32, device=device)
        return es, torch.empty_like(es), tmp

    def _v2_scratch(self, num_seqs: int, max_seq_len: int, like: torch.Tensor):
        parts = (max_seq_len + 511) // 512
        # per stream: steps in flight on different streams (and the graphs captured on them)
        # must not share scratch
        key = (num_seqs, parts, torch.cuda.current_stream(like.device).cuda_stream)
        s = self._scratch.get(key)
        if s is None or s[2].device != like.device or s[2].dtype != like.dtype:
            tmp = torch.empty(num_seqs, self.num_heads, parts, self.head_size, dtype=like.dtype, device=like.device)
            es = torch.empty(num_seqs, self.num_heads, parts, dtype=torch.float32, device=like.device)
            s = (es, torch.empty_like(es), tmp)
            self._scratch[key] = s
        return s

    def forward(self, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                kv_cache: Optional[torch.Tensor], attn_metadata: PagedAttnMetadata,
                k_scale: float = 1.0, v_scale: float = 1.0, attn_type=None) -> torch.Tensor:
        assert k_scale == 1.0 and v_scale == 1.0, "key/v_scale is not supported in PagedAttn (HIP)."
        num_tokens, hidden_size = query.shape
        query = query.view(-1, self.num_heads, self.head_size)
        key = key.view(-1, self.num_kv_heads, self.head_size)
        value = value.view(-1, self.num_kv_heads, self.head_size)
        key_cache = value_cache = None
        if kv_cache is not None:
            key_cache, value_cache = PagedAttention.split_kv_cache(kv_cache, self.num_kv_heads, self.head_size)
            # new K/V rows go into the paged cache first; decode then reads them back
            PagedAttention.write_to_paged_cache(key, value, key_cache, value_cache,
                                                attn_metadata.slot_mapping, self.kv_cache_dtype,
                                                k_scale, v_scale)
        npt = attn_metadata.num_prefill_tokens
        ndt = attn_metadata.num_decode_tokens
        assert key.shape[0] == npt + ndt and value.shape[0] == npt + ndt
        output = torch.empty_like(query)
        if prefill_meta := attn_metadata.prefill_metadata:
            self._prefill(query[:npt], key[:npt], value[:npt], key_cache, value_cache, prefill_meta,
                          output[:npt])
        if (decode_meta := attn_metadata.decode_metadata) and self.logits_soft_cap > 0.0:
            self._decode_as_chunks(query[npt:], key_cache, value_cache, decode_meta, output[npt:])
        elif decode_meta:
            dq = query[npt:]
            max_len = decode_meta.max_decode_seq_len
            force = self.decode_version
            use_v1 = (force == "v1") if force else PagedAttention.use_v1(
                dq.shape[0], self.num_kv_heads, self.num_heads, max_len, num_compute_units(dq.device))
            scratch = None if use_v1 else (decode_meta.decode_scratch or self._v2_scratch(dq.shape[0], max_len, dq))
            alibi = self.alibi_slopes
            if alibi is not None and alibi.device != dq.device:
                alibi = self.alibi_slopes = alibi.to(dq.device)
            PagedAttention.forward_decode(
                dq, key_cache, value_cache, decode_meta.block_tables, decode_meta.seq_lens_tensor,
                max_len, self.kv_cache_dtype, self.num_kv_heads, self.scale, alibi, k_scale, v_scale,
                force_version="v1" if use_v1 else "v2", scratch=scratch, output=output[npt:])
        return output.view(num_tokens, hidden_size)

    # ---- entry points for a caller that has already written K/V (fused rope + cache write) ----
    def split_kv_cache(self, kv_cache: torch.Tensor):
        return PagedAttention.split_kv_cache(kv_cache, self.num_kv_heads, self.head_size)

    def _decode_as_chunks(self, dq: torch.Tensor, key_cache, value_cache, md: PagedAttnMetadata, out: torch.Tensor) -> None:
        """Decode tokens as chunks of ONE query token over their paged contexts (the prefill kernel: soft cap)."""
        n = dq.shape[0]
        qsl = torch.arange(n + 1, dtype=torch.int32, device=dq.device)
        alibi = self.alibi_slopes
        if alibi is not None and alibi.device != dq.device:
            alibi = self.alibi_slopes = alibi.to(dq.device)
        PagedAttention.forward_prefix(dq, None, None, key_cache, value_cache, md.block_tables, qsl, md.seq_lens_tensor,
                                      None, 1, alibi, None, scale=self.scale, softcap=self.logits_soft_cap,
                                      kv_cache_dtype=self.kv_cache_dtype, output=out)

    def decode_attention(self, query: torch.Tensor, key_cache: torch.Tensor, value_cache: torch.Tensor,
                         attn_metadata: PagedAttnMetadata, fp8_twin_scale: Optional[torch.Tensor] = None):
        """Decode attention only (all tokens are decode tokens, K/V already in the cache).
        `fp8_twin_scale` (one float on the device): the W8A8 output projection's activation scale -- the result is
        then a pair (out, out_fp8 | None): out_fp8 [T, H * D] uint8 = static_scaled_fp8_quant(out, scale), written
        by the attention launch itself when that launch is a single pass (None otherwise: quantise `out`)."""
        num_tokens, hidden_size = query.shape
        dq = query.view(-1, self.num_heads, self.head_size)
        md = attn_metadata.decode_metadata
        if self.logits_soft_cap > 0.0:
            out = torch.empty(dq.shape, dtype=dq.dtype, device=dq.device)
            self._decode_as_chunks(dq, key_cache, value_cache, md, out)
            out = out.view(num_tokens, hidden_size)
            return out if fp8_twin_scale is None else (out, None)
        if fp8_twin_scale is not None:
            pair = None
            if self.alibi_slopes is None and self.decode_version != "v1" and dq.dtype in (torch.bfloat16, torch.float16):
                max_len = md.max_decode_seq_len
                scratch = md.decode_scratch or self._v2_scratch(dq.shape[0], max_len, dq)
                out = torch.empty(dq.shape, dtype=dq.dtype, device=dq.device)
                out8 = torch.empty(dq.shape, dtype=torch.uint8, device=dq.device)
                if torch.ops._C_amd.paged_attention_v2_q(out, out8, fp8_twin_scale, scratch[0], scratch[1], scratch[2], dq,
                                                         key_cache, value_cache, self.num_kv_heads, self.scale,
                                                         md.block_tables, md.seq_lens_tensor, value_cache.shape[3], max_len,
                                                         self.kv_cache_dtype, 1.0, 1.0):
                    pair = (out.view(num_tokens, hidden_size), out8.view(num_tokens, hidden_size))
            return pair if pair is not None else (self.decode_attention(query, key_cache, value_cache, attn_metadata), None)
        max_len = md.max_decode_seq_len
        force = self.decode_version
        use_v1 = (force == "v1") if force else PagedAttention.use_v1(dq.shape[0], self.num_kv_heads, self.num_heads,
                                                                      max_len, num_compute_units(dq.device))
        scratch = None if use_v1 else (md.decode_scratch or self._v2_scratch(dq.shape[0], max_len, dq))
        alibi = self.alibi_slopes
        if alibi is not None and alibi.device != dq.device:
            alibi = self.alibi_slopes = alibi.to(dq.device)
        out = torch.empty(dq.shape, dtype=dq.dtype, device=dq.device)
        PagedAttention.forward_decode(dq, key_cache, value_cache, md.block_tables, md.seq_lens_tensor, max_len,
                                      self.kv_cache_dtype, self.num_kv_heads, self.scale, alibi, 1.0, 1.0,
                                      force_version="v1" if use_v1 else "v2", scratch=scratch, output=out)
        return out.view(num_tokens, hidden_size)

    def rope_cache_decode_attention(self, positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor,
                                    value: torch.Tensor, cos_sin_cache: torch.Tensor, key_cache: torch.Tensor,
                                    value_cache: torch.Tensor, attn_metadata: PagedAttnMetadata,
                                    fp8_twin_scale: Optional[torch.Tensor] = None):
        """rotary_embedding + reshape_and_cache + paged_attention_v2 of a decode-only step in ONE launch
        (lvllm_rope_cache_paged_attention; bit-identical to the three).  None when the shapes are outside the
        fused kernel's envelope -- nothing was done, the caller runs the separate launches."""
        # fp8 caches: the kernel supports them (bit-identical too), but measured SLOWER than the two launches there
        # (13 030 vs 13 300 tokens/s, profiles/r02_tuning.md): the fp8 attention launch is instruction-bound, the
        # prologue and its barrier cost more than the launch they save
        if self.alibi_slopes is not None or self.decode_version == "v1":
            return None
        if self.kv_cache_dtype != "auto" and not self.fuse_rope_over_fp8_cache:
            return None
        # Sliding windows: the fused kernel takes the step's new token for logical position seq_len - 1 of the
        # table it is handed.  The v1 manager's circular table (block_manager/v1.py, block_manager_v1.py:279-295)
        # with seq_len clipped to the window puts it at offset (L - 1) % block_size of some OTHER entry once the
        # sequence has outgrown the window: separate launches there (they read everything from the cache).
        if self.sliding_window is not None or self.logits_soft_cap > 0.0:
            return None
        num_tokens, hidden_size = query.shape
        md = attn_metadata.decode_metadata
        max_len = md.max_decode_seq_len
        scratch = md.decode_scratch or self._v2_scratch(num_tokens, max_len, query)
        out = torch.empty(num_tokens, self.num_heads, self.head_size, dtype=query.dtype, device=query.device)
        args = (out, scratch[0], scratch[1], scratch[2], positions, query, key, value, self.head_size, cos_sin_cache, True,
                key_cache, value_cache, attn_metadata.slot_mapping, self.num_kv_heads, self.scale, md.block_tables,
                md.seq_lens_tensor, value_cache.shape[3], max_len, self.kv_cache_dtype, 1.0, 1.0)
        if fp8_twin_scale is not None:
            # (out, out_fp8 | None): the twin exists when the launch is a single pass; cut into shares the launch is
            # refused WITH a twin (nothing done) and repeated without
            out8 = torch.empty(out.shape, dtype=torch.uint8, device=query.device)
            if torch.ops._C_amd.rope_cache_paged_attention(*args, out8, fp8_twin_scale):
                return out.view(num_tokens, hidden_size), out8.view(num_tokens, hidden_size)
            ok = torch.ops._C_amd.rope_cache_paged_attention(*args)
            return (out.view(num_tokens, hidden_size), None) if ok else None
        ok = torch.ops._C_amd.rope_cache_paged_attention(*args)
        return out.view(num_tokens, hidden_size) if ok else None

    def unified_attention(self, query: torch.Tensor, key_cache: torch.Tensor, value_cache: torch.Tensor,
                          block_tables: torch.Tensor, seq_lens: torch.Tensor, query_start_loc: torch.Tensor,
                          max_query_len: int, output: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Every sequence of a mixed step as a chunk of query_len >= 1 tokens over its paged context
        (a decode token is a chunk of one): one launch of the prefill kernel with device-side
        metadata only, so a mixed step can be captured into a HIP graph.  K/V of all tokens must
        already be in the cache.  `output`: a [num_tokens, hidden] buffer whose padding rows (tokens of no
        sequence, never written here) the caller keeps finite -- the model zeroes ONE buffer per step and hands it to
        every layer instead of paying a fill launch per layer."""
        num_tokens, hidden_size = query.shape
        q = query.view(-1, self.num_heads, self.head_size)
        # rows of padding tokens belong to no sequence: keep them finite
        out = torch.zeros_like(q) if output is None else output.view(-1, self.num_heads, self.head_size)
        alibi = self.alibi_slopes
        if alibi is not None and alibi.device != q.device:
            alibi = self.alibi_slopes = alibi.to(q.device)
        PagedAttention.forward_prefix(q, None, None, key_cache, value_cache, block_tables, query_start_loc,
                                      seq_lens, None, max_query_len, alibi, None, scale=self.scale,
                                      softcap=self.logits_soft_cap, kv_cache_dtype=self.kv_cache_dtype, output=out)
        return out.view(num_tokens, hidden_size)

    # ---- prompt attention ----
    def _prefill(self, q, k, v, key_cache, value_cache, meta: PagedAttnMetadata, out) -> None:
        """With a cache: one launch of the HIP varlen kernel over the paged cache (the chunk's own
        K/V were written just before).  Without one (the memory-profiling run, kv_cache None,
        flash_attn.py:518-536): torch SDPA on the dense prompt."""
        if (self.use_hip_prefill and key_cache is not None
                and (self.kv_cache_dtype == "auto" or self.head_size % 64 == 0)
                and q.dtype in (torch.float16, torch.bfloat16)
                and value_cache.shape[3] in (16, 32) and meta.block_tables.numel() > 0
                and self.sliding_window is None):
            alibi = self.alibi_slopes
            if alibi is not None and alibi.device != q.device:
                alibi = self.alibi_slopes = alibi.to(q.device)
            PagedAttention.forward_prefix(q, k, v, key_cache, value_cache, meta.block_tables,
                                          meta.query_start_loc, meta.seq_lens_tensor,
                                          meta.context_lens_tensor, meta.max_query_len, alibi,
                                          self.sliding_window, scale=self.scale, softcap=self.logits_soft_cap,
                                          kv_cache_dtype=self.kv_cache_dtype, output=out,
                                          max_seq_len=meta.max_prefill_seq_len)
            return
        if self.logits_soft_cap > 0.0 and key_cache is not None:
            raise NotImplementedError("logits_soft_cap needs the HIP prefill kernel (16-bit model dtype, block size 16 / 32, "
                                      "no sliding window)")
        qs = 0
        G = self.num_queries_per_kv
        for i in range(meta.num_prefills):
            qlen = meta.query_lens[i]
            ctx = meta.context_lens[i]
            qi = q[qs:qs + qlen].transpose(0, 1)  # [H, qlen, D]
            ki, vi = k[qs:qs + qlen], v[qs:qs + qlen]
            if ctx > 0:
                # chunked prefill / prefix hit: earlier K/V live in the paged cache
                assert key_cache is not None
                bt = meta.block_tables[i]
                bs = value_cache.shape[3]
                pos = torch.arange(ctx, device=q.device)
                blk = bt[(pos // bs).long()].long()
                off = pos % bs
                kc = key_cache[blk, :, :, off, :].reshape(ctx, self.num_kv_heads, self.head_size)
                vc = value_cache[blk, :, :, off]
                if kc.dtype != q.dtype:  # fp8 cache (k_scale = v_scale = 1): dequantise the gathered context
                    kc = kc.view(torch.float8_e4m3fn).to(q.dtype)
                    vc = vc.view(torch.float8_e4m3fn).to(q.dtype)
                ki = torch.cat([kc, ki], dim=0)
                vi = torch.cat([vc, vi], dim=0)
            ki = ki.transpose(0, 1).repeat_interleave(G, dim=0)  # [H, ctx+qlen, D]
            vi = vi.transpose(0, 1).repeat_interleave(G, dim=0)
            total = ctx + qlen
            mask = torch.ones(qlen, total, dtype=torch.bool, device=q.device).tril(diagonal=ctx)
            if self.sliding_window is not None:
                mask &= ~torch.ones(qlen, total, dtype=torch.bool, device=q.device).tril(
                    diagonal=ctx - self.sliding_window)
            bias = None
            if self.alibi_slopes is not None:
                rel = (torch.arange(total, device=q.device)[None, :] -
                       (torch.arange(qlen, device=q.device)[:, None] + ctx)).float()
                bias = self.alibi_slopes.to(q.device)[:, None, None] * rel[None]
                bias = bias.masked_fill(~mask[None], float("-inf")).to(q.dtype)
            o = torch.nn.functional.scaled_dot_product_attention(
                qi, ki, vi, attn_mask=bias if bias is not None else mask, scale=self.scale)
            out[qs:qs + qlen] = o.transpose(0, 1)
            qs += qlen


class PagedAttnBackend:
    """`DecodeOnlyAttentionBackend` of the paged KV layout (abstract.py:15-72)."""

    def __init__(self, attn_type=None):
        self._attn_type = attn_type

    @property
    def attn_type(self):
        return self._attn_type

    @staticmethod
    def get_name() -> str:
        return "paged-attn-hip"

    @staticmethod
    def get_impl_cls() -> Type[PagedAttnImpl]:
        return PagedAttnImpl

    @staticmethod
    def get_metadata_cls() -> Type[PagedAttnMetadata]:
        return PagedAttnMetadata

    @classmethod
    def make_metadata(cls, *args, **kwargs) -> PagedAttnMetadata:
        return cls.get_metadata_cls()(*args, **kwargs)

    @staticmethod
    def get_builder_cls() -> Type[PagedAttnMetadataBuilder]:
        return PagedAttnMetadataBuilder

    @classmethod
    def make_metadata_builder(cls, *args, **kwargs) -> PagedAttnMetadataBuilder:
        return cls.get_builder_cls()(*args, **kwargs)

    @staticmethod
    def get_kv_cache_shape(num_blocks: int, block_size: int, num_kv_heads: int,
                           head_size: int) -> Tuple[int, ...]:
        return PagedAttention.get_kv_cache_shape(num_blocks, block_size, num_kv_heads, head_size)

    @staticmethod
    def swap_blocks(src_kv_cache: torch.Tensor, dst_kv_cache: torch.Tensor,
                    src_to_dst: torch.Tensor) -> None:
        PagedAttention.swap_blocks(src_kv_cache, dst_kv_cache, src_to_dst)

    @staticmethod
    def copy_blocks(kv_caches: List[torch.Tensor], src_to_dists: torch.Tensor) -> None:
        # the kernel reads the pairs on the device (cache_kernels.cu:79-80): move them explicitly
        if src_to_dists.device != kv_caches[0].device:
            src_to_dists = src_to_dists.to(kv_caches[0].device, non_blocking=True)
        PagedAttention.copy_blocks(kv_caches, src_to_dists)

    @classmethod
    def from_engine(cls, engine):
        """Workflow hook (core/llm_engine.py:30-31): `Workflow.AttnBackend` resolves to this."""
        return cls()
