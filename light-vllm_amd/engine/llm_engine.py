"""Decode engine: scheduler -> input builder -> executor (cache ops + forward + greedy
sampling) -> output processing, in the step structure of light_vllm/core/llm_engine.py:
`sync_step` (:119-130) and `async_step` (:132-176, up to `max_num_on_the_fly` steps in
flight, legal because scheduled groups are marked busy).

One engine = one GPU (replica).  Multi-GPU serving replicates engines, one process per GPU,
with no collective on the kernel path.
"""
import queue
import threading
import time
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

from ..attention.backend import PagedAttnBackend
from .cache_engine import CacheEngine
from .config import CacheConfig, ModelConfig, SchedulerConfig
from .graph_runner import DecodeGraphPool
from .input_builder import ExecuteInput, ModelInputBuilder
from .model import DecoderModel
from .scheduler import DecodingScheduler, SchedulerOutput
from .sequence import Sequence, SequenceGroup, SequenceStatus


@dataclass
class RequestOutput:
    request_id: str
    token_ids: List[int]
    finished: bool
    finish_reason: Optional[str] = None
    # SamplingParams.logprobs: per output token {token id: (logprob, rank)} -- the sampled token and the n most
    # likely ones (the reference's SampleLogprobs, sequence.py Logprob(logprob, rank)); None when not asked for
    logprobs: Optional[List[dict]] = None
    # SamplingParams.prompt_logprobs: [None] + one dictionary per further prompt token computed so far (PromptLogprobs)
    prompt_logprobs: Optional[List[Optional[dict]]] = None
    # SamplingParams.n > 1: the n sequences of the request (of its best_of, by cumulative log-probability), token ids
    # each; `token_ids` is the first of them
    outputs: Optional[List[List[int]]] = None


@dataclass
class ExecuteOutput:
    sampled: Optional[torch.Tensor]  # [num sampled rows] int64 on the CPU (pinned); [num_steps, rows] of a burst
    sample_seq_ids: List[int]
    execute_begin_ts: float = 0.0
    execute_end_ts: float = 0.0
    num_steps: int = 1  # model steps this result covers (multi-step decode)
    logprobs: Optional[List[Optional[dict]]] = None  # per sampled row (sampling.sample_logprobs), rows that asked
    prompt_logprobs: Optional[Dict[int, List[dict]]] = None  # {seq id: dictionaries of this step's prompt positions}
    # {parent seq id: [(token, logprob dictionary or None, logprob), ...]}: the further samples of a prompt whose
    # request asks for several sequences (SamplingParams.n / best_of)
    extra_samples: Optional[Dict[int, list]] = None


class Worker:
    """Owns the model, the KV caches and the stream the step runs on
    (decoding/worker/gpu_worker.py:26-224 + runner/model_runner.py:169-187)."""

    def __init__(self, model_config: ModelConfig, cache_config: CacheConfig, attn_backend,
                 device: str, use_hip_graph: bool = True, decode_version: Optional[str] = None,
                 max_model_len: int = 8192, seed: int = 0, num_slots: int = 1, mixed_graph_tokens: int = 0,
                 mixed_graph_seqs: int = 0):
        self.device = torch.device(device)
        self.model_config = model_config
        self.cache_config = cache_config
        impl_cls = attn_backend.get_impl_cls()
        self.attn_impl = impl_cls(model_config.num_attention_heads, model_config.head_dim,
                                  model_config.head_dim ** -0.5, model_config.num_key_value_heads,
                                  None, cache_config.sliding_window, cache_config.cache_dtype,
                                  decode_version=decode_version)
        if self.device.type == "cuda":  # free memory before the weights (gpu_worker.py:78-80)
            torch.cuda.empty_cache()
            self.init_gpu_memory = torch.cuda.mem_get_info(self.device)[0]
        else:
            self.init_gpu_memory = 0
        self.model = DecoderModel(model_config, self.attn_impl, device, seed)
        self.attn_backend = attn_backend
        self.cache_engine: Optional[CacheEngine] = None
        self.use_hip_graph = use_hip_graph
        self.max_model_len = max_model_len
        # one graph pool (own static input buffers) per execution slot: steps in flight on
        # different streams must not share them
        self.num_slots = max(1, num_slots)
        # > 0: mixed (chunked-prefill) steps of at most this many tokens / sequences replay one graph
        self.mixed_graph_tokens, self.mixed_graph_seqs = mixed_graph_tokens, mixed_graph_seqs
        self.graph_pools: Optional[List[DecodeGraphPool]] = None
        # device-resident sampler state (device_sampler.DeviceSampler), created by the engine with the first request
        # that is not plain greedy; None: every step ends with the arg-max epilogue
        self.sampler = None
        # tests: keep the logits of the sampled rows of the last eager step
        self.capture_logits = False
        self.last_logits: Optional[torch.Tensor] = None

    # ---- KV-cache sizing (decoding/worker/gpu_worker.py:95-144, runner/model_runner.py:111-145) ----
    @staticmethod
    def profile_seq_lens(max_num_batched_tokens: int, max_num_seqs: int) -> List[int]:
        """Prompt lengths of the profile run: the token budget dealt over `max_num_seqs` dummy prompts,
        the remainder one token each to the first ones (model_runner.py:117-123)."""
        return [max_num_batched_tokens // max_num_seqs + (i < max_num_batched_tokens % max_num_seqs)
                for i in range(max_num_seqs)]

    @staticmethod
    def kv_blocks_from_profile(total: int, init_free: int, free_after_load: int, free_after_profile: int,
                               gpu_memory_utilization: float, scheduling: str, block_bytes: int,
                               swap_space_bytes: int, reserve_bytes: int = 0) -> Tuple[int, int]:
        """The arithmetic of gpu_worker.py:100-141 on four memory readings: weights = init_free -
        free_after_load; peak = init_free - free_after_profile; runtime = peak - weights; steps in flight on
        their own streams ("async", "double_buffer") hold two sets of activations, so the runtime part counts
        twice (:116-118).  `reserve_bytes` (not in the reference: HIP-graph pools and GEMM workspaces of the
        captured steps) comes off the top."""
        model_memory_usage = init_free - free_after_load
        peak_memory = init_free - free_after_profile
        runtime_memory = peak_memory - model_memory_usage
        if scheduling in ("async", "double_buffer"):
            peak_memory += runtime_memory
        assert peak_memory > 0, ("Error in memory profiling. "
                                 f"Initial free memory {init_free}, current free memory {free_after_profile}.")
        num_gpu = int((total * gpu_memory_utilization - peak_memory - reserve_bytes) // block_bytes)
        num_cpu = int(swap_space_bytes // block_bytes)
        return max(num_gpu, 0), max(num_cpu, 0)

    @torch.inference_mode()
    def profile_run(self, scheduler_config: SchedulerConfig) -> None:
        """One forward of the largest prompt batch the scheduler can emit with no KV cache
        (`kv_caches = [None] * L`, block tables None -> slot -1 everywhere, model_runner.py:111-145), then the
        logits of one row per sequence through top-k / top-p filtering as the reference's profile sampling
        parameters do (top_p 0.99, top_k vocab - 1: a full sort of [max_num_seqs, vocab])."""
        from ..sampling import apply_top_k_top_p
        from .scheduler import SequenceGroupMetadata
        from .sequence import SequenceData
        lens = self.profile_seq_lens(scheduler_config.max_num_batched_tokens, scheduler_config.max_num_seqs)
        metas = []
        for gid, n in enumerate(lens):
            if n == 0:
                continue
            metas.append(SequenceGroupMetadata(request_id=str(gid), is_prompt=True,
                                               seq_data={gid: SequenceData([0] * n)}, block_tables=None,
                                               do_sample=True, token_chunk_size=n, computed_block_nums=[]))
        builder = ModelInputBuilder(scheduler_config, self.cache_config, self.attn_backend,
                                    self.cache_config.sliding_window)
        mi = builder.prepare_model_input(metas).to(self.device)
        hidden = self.model.forward(mi.input_tokens, mi.input_positions, None, mi.attn_metadata)
        rows = torch.tensor(mi.sample_indices, dtype=torch.long, device=self.device)
        logits = self.model.compute_logits(hidden[rows]).float()
        n = logits.shape[0]
        p = torch.full((n,), 0.99, device=self.device)
        k = torch.full((n,), self.model_config.vocab_size - 1, device=self.device, dtype=torch.long)
        torch.argmax(torch.softmax(apply_top_k_top_p(logits, p, k), dim=-1), dim=-1)
        torch.cuda.synchronize(self.device)

    def graph_reserve_bytes(self, scheduler_config: SchedulerConfig) -> int:
        """Device memory the captured decode steps will take after the cache is sized (not in the
        reference, which runs eager): per slot and captured batch size the activations of one step stay
        allocated in the graph's private pool -- hidden / residual / normed rows, qkv, attention output,
        SwiGLU output, the fp32 split-K partials of the o / down projections (<= 8 slabs) and the lm_head's
        per-workgroup arg-max candidates.  An upper bound, counted for the largest batch (max_num_seqs rows,
        capped at the 64 rows the decode path takes) twice over (a second captured size)."""
        if not self.use_hip_graph:
            return 0
        cfg = self.model_config
        rows = min(scheduler_config.max_num_seqs, 64)
        e = 2
        per_layer = rows * (4 * cfg.hidden_size + (cfg.num_attention_heads + 2 * cfg.num_key_value_heads) * cfg.head_dim
                            + cfg.num_attention_heads * cfg.head_dim + cfg.intermediate_size) * e
        per_layer += 2 * 8 * rows * cfg.hidden_size * 4
        step = per_layer * cfg.num_hidden_layers + rows * 4096 * 16 + (8 << 20)
        return 2 * self.num_slots * step

    def sampler_state_slots(self, scheduler_config: SchedulerConfig) -> int:
        """State slots of the device sampler: every sequence of every step in flight plus one step being built."""
        return max(64, scheduler_config.max_num_seqs * (self.num_slots + 1))

    def sampler_reserve_bytes(self, scheduler_config: SchedulerConfig) -> int:
        """Device memory the device-side sampler takes once a request is not plain greedy (created lazily, after the
        cache is sized: ADVICE r03): its state -- int32 [slots, vocab] counts + 128-byte records -- and, per step in
        flight and captured batch size, the sampled graph flavour's working rows (fp32 [rows, vocab + 64]) and its
        logits (model dtype and the fp32 copy the lm_head path may keep).  An upper bound, like the graph reserve."""
        vocab = self.model_config.vocab_size
        state = self.sampler_state_slots(scheduler_config) * (vocab * 4 + 128)
        rows = min(scheduler_config.max_num_seqs, 64)
        per_flavour = rows * ((((vocab + 3) & ~3) + 64) * 4 + vocab * (2 + 4))
        flavours = 2 * self.num_slots if self.use_hip_graph else 1
        return state + flavours * per_flavour

    def determine_num_available_blocks(self, scheduler_config: Optional[SchedulerConfig] = None) -> Tuple[int, int]:
        """(num_gpu_blocks, num_cpu_blocks) as gpu_worker.py:95-144 computes them: weights = free memory
        before the model minus free memory after; a profile forward with no KV cache; peak = everything the
        caching allocator holds afterwards; the activation part doubled when two steps run on their own
        streams; the rest of `gpu_memory_utilization` x total memory, less the graph reserve, is KV cache."""
        scheduler_config = scheduler_config or SchedulerConfig()
        torch.cuda.empty_cache()
        free_after_load = torch.cuda.mem_get_info(self.device)[0]
        self.profile_run(scheduler_config)
        torch.cuda.synchronize(self.device)
        free_after_profile, total = torch.cuda.mem_get_info(self.device)
        block_bytes = CacheEngine.get_cache_block_footprint(self.cache_config, self.model_config)
        out = self.kv_blocks_from_profile(total, self.init_gpu_memory, free_after_load, free_after_profile,
                                          self.cache_config.gpu_memory_utilization, scheduler_config.scheduling,
                                          block_bytes, self.cache_config.swap_space_bytes,
                                          self.graph_reserve_bytes(scheduler_config)
                                          + self.sampler_reserve_bytes(scheduler_config))
        self.profile = dict(total=total, init_free=self.init_gpu_memory, free_after_load=free_after_load,
                            free_after_profile=free_after_profile, block_bytes=block_bytes)
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        return out

    def initialize_cache(self, num_gpu_blocks: int, num_cpu_blocks: int) -> None:
        self.cache_config.num_gpu_blocks = num_gpu_blocks
        self.cache_config.num_cpu_blocks = num_cpu_blocks
        self.cache_engine = CacheEngine(self.cache_config, self.model_config, self.attn_backend, self.device)
        if self.use_hip_graph:
            max_blocks = (self.max_model_len + self.cache_config.block_size - 1) // self.cache_config.block_size
            self.graph_pools = [DecodeGraphPool(self.model, self.cache_engine.gpu_cache, max_blocks,
                                                self.cache_config.block_size, self.device)
                                for _ in range(self.num_slots)]

    def _slots_tensor(self, slots: List[int]) -> torch.Tensor:
        t = torch.tensor(slots, dtype=torch.int32)
        return (t.pin_memory() if self.device.type == "cuda" else t).to(self.device, non_blocking=True)

    @torch.inference_mode()
    def execute(self, execute_input: ExecuteInput, slot: int = 0, state_slots=None, logprob_rows=None) -> ExecuteOutput:
        """`state_slots`: {seq id: sampler state slot} for a step in which some request is not plain greedy (None:
        all plain greedy): the step's logits then go through the device sampler (one launch: penalties, temperature,
        top-k / top-p / min-p, the draw; rows without a slot take the arg-max).
        `logprob_rows`: {seq id: dict(params, prompt, output, eos)} for a step in which some request asks for
        log-probabilities (SamplingParams.logprobs): the step takes the general path, the adjusted logits of its
        sampled rows are restated in torch (sampling.SamplingBatch: the stages the device sampler applies) and the
        sample half of the reference's get_logprobs is read off them (sampler.py:726-990)."""
        wi, mi = execute_input.worker_input, execute_input.model_input
        ce = self.cache_engine
        if wi.blocks_to_swap_in.numel() > 0:
            ce.swap_in(wi.blocks_to_swap_in)
        if wi.blocks_to_swap_out.numel() > 0:
            ce.swap_out(wi.blocks_to_swap_out)
        if wi.blocks_to_copy.numel() > 0:
            ce.copy(wi.blocks_to_copy.to(self.device, non_blocking=True))
        if mi.input_tokens is None:
            return ExecuteOutput(None, [])
        md = mi.attn_metadata
        graphs = self.graph_pools[slot] if self.graph_pools is not None else None
        row_slots = None if state_slots is None else [state_slots.get(sid, -1) for sid in mi.sample_seq_ids]
        if logprob_rows is not None:
            graphs = None
        if graphs is not None and mi.decode_only and md.block_tables.shape[1] <= graphs.max_blocks_per_seq:
            n = mi.input_tokens.shape[0]
            g = graphs.get(n, sampler=self.sampler if row_slots is not None else None)
            per_row = None
            if row_slots is not None:  # one slot per ROW of the step (rows that do not sample: -1)
                full = [-1] * n
                for r, sl in zip(mi.sample_indices, row_slots):
                    full[r] = sl
                per_row = self._slots_tensor(full)
            g.load(mi.input_tokens, mi.input_positions, md.slot_mapping, md.block_tables, md.seq_lens_tensor, per_row)
            tokens = g.replay()[:n]
            if len(mi.sample_indices) != n:
                tokens = tokens[torch.tensor(mi.sample_indices, dtype=torch.long, device=self.device)]
        elif (graphs is not None and self.mixed_graph_tokens > 0 and not self.capture_logits and row_slots is None
              and mi.input_tokens.shape[0] <= self.mixed_graph_tokens
              and md.seq_lens_tensor.shape[0] <= self.mixed_graph_seqs
              and md.block_tables.shape[0] == md.seq_lens_tensor.shape[0]
              and 0 < md.block_tables.shape[1] <= graphs.max_blocks_per_seq):
            g = graphs.get_mixed(self.mixed_graph_tokens, self.mixed_graph_seqs)
            g.load(mi.input_tokens, mi.input_positions, md.slot_mapping, md.block_tables, md.seq_lens_tensor,
                   md.query_start_loc, mi.sample_indices)
            tokens = g.replay()[:len(mi.sample_indices)]
        else:
            mi.to(self.device)
            hidden = self.model.forward(mi.input_tokens, mi.input_positions, ce.gpu_cache, md)
            hidden_all = hidden
            if len(mi.sample_indices) != hidden.shape[0]:
                hidden = hidden[torch.tensor(mi.sample_indices, dtype=torch.long, device=self.device)]
            logits = self.model.compute_logits(hidden)
            if self.capture_logits:
                self.last_logits = logits.float().cpu()
            prompt_lps = None
            if logprob_rows is not None:
                prompt_lps = self._prompt_logprobs(hidden_all, logprob_rows)
            lps = None
            if logprob_rows is not None and logits.shape[0] > 0:  # before the draw: the sampler appends what it draws
                from ..sampling import SamplingBatch
                rows = [logprob_rows[sid] if sid in logprob_rows and logprob_rows[sid].get("num") is not None
                        else dict(params=None, prompt=(), output=(), eos=None) for sid in mi.sample_seq_ids]
                lps = SamplingBatch(rows, logits.shape[1], self.device).logprobs(logits)
            if row_slots is None or logits.shape[0] == 0:
                tokens = torch.argmax(logits, dim=-1)
            else:
                tokens = self.sampler.sample(logits, self._slots_tensor(row_slots))
            if lps is not None:
                from ..sampling import random_sample, sample_logprobs
                nums = [logprob_rows[sid].get("num") if sid in logprob_rows else None for sid in mi.sample_seq_ids]
                extra = None
                for r, sid in enumerate(mi.sample_seq_ids):  # the further samples of a prompt that forks (n / best_of)
                    k = logprob_rows[sid].get("extra", 0) if sid in logprob_rows else 0
                    if k > 0:
                        gen = None
                        seed = logprob_rows[sid]["params"].seed
                        if seed is not None:
                            gen = torch.Generator(device=self.device).manual_seed(
                                (seed * 0x9E3779B97F4A7C15 + 0x632BE59BD9B4E019) % (1 << 63))
                        rep_lps = lps[r:r + 1].expand(k, -1)
                        toks = random_sample(rep_lps.exp(), gen)
                        dicts = sample_logprobs(rep_lps, toks, [nums[r]] * k)
                        extra = extra or {}
                        extra[sid] = list(zip(toks.tolist(), dicts))
                out = torch.empty(tokens.shape, dtype=tokens.dtype, pin_memory=True)
                out.copy_(tokens, non_blocking=True)
                return ExecuteOutput(out, mi.sample_seq_ids, logprobs=sample_logprobs(lps, tokens, nums),
                                     prompt_logprobs=prompt_lps, extra_samples=extra)
            if prompt_lps:
                out = torch.empty(tokens.shape, dtype=tokens.dtype, pin_memory=True)
                out.copy_(tokens, non_blocking=True)
                return ExecuteOutput(out, mi.sample_seq_ids, prompt_logprobs=prompt_lps)
        out = torch.empty(tokens.shape, dtype=tokens.dtype, pin_memory=True)
        out.copy_(tokens, non_blocking=True)
        return ExecuteOutput(out, mi.sample_seq_ids)


    def _prompt_logprobs(self, hidden_all: torch.Tensor, logprob_rows) -> Optional[Dict[int, List[dict]]]:
        """The prompt half of get_logprobs (sampler.py:863-915) for the prompt chunks of this step whose requests ask:
        the logits of the chunk's positions that predict another PROMPT token (all of them, or all but the last when
        the chunk completes the prompt: that row samples), adjusted by the request's temperature / top-k / top-p / min-p
        but no penalty (sampling_metadata.py:447-456), log_softmax, then {next prompt token: (logprob, rank)} + the n
        most likely tokens per position.  `logprob_rows[seq id]["prompt_rows"]` = (first row of the chunk in the step's
        token batch, number of such positions, their next tokens)."""
        import dataclasses
        from ..sampling import SamplingBatch, sample_logprobs
        out: Dict[int, List[dict]] = {}
        for sid, r in logprob_rows.items():
            pr = r.get("prompt_rows")
            if pr is None or pr[1] == 0:
                continue
            start, n, nxt = pr
            sp = dataclasses.replace(r["params"], presence_penalty=0.0, frequency_penalty=0.0, repetition_penalty=1.0,
                                     min_tokens=0)
            logits = self.model.compute_logits(hidden_all[start:start + n])
            batch = SamplingBatch([dict(params=sp, prompt=(), output=(), eos=None)] * n, logits.shape[1], self.device)
            lps = batch.logprobs(logits)
            out[sid] = sample_logprobs(lps, torch.tensor(nxt, dtype=torch.long, device=self.device),
                                       [r["params"].prompt_logprobs] * n)
        return out or None

    @torch.inference_mode()
    def execute_decode(self, metas, slot: int = 0, num_steps: int = 1, state_slots=None) -> Optional[ExecuteOutput]:
        """A step of decode tokens only, taken straight from the scheduler's metadata to the captured
        graph's staging buffer (DecodeStepArrays): same inputs as `execute(input_builder(...))`, a
        fraction of the host time.  None when the step does not fit the captured shapes.
        num_steps > 1: that many model steps back to back on the device (DecodeGraph.replay_steps); the
        result then holds [num_steps, n] tokens."""
        graphs = self.graph_pools[slot] if self.graph_pools is not None else None
        if graphs is None or self.capture_logits:
            return None
        n = len(metas)
        row_slots = None
        if state_slots is not None:  # {seq id: sampler state slot}: the graph flavour that ends with the sampling kernel
            row_slots = [state_slots.get(next(iter(m.seq_data)), -1) for m in metas]
        g = graphs.get(n, sampler=self.sampler if row_slots is not None else None)
        seq_ids = g.staging.fill(metas, row_slots)
        if seq_ids is None:  # a block table wider than the captured step
            return None
        g.load_staged()
        if num_steps > 1:
            tokens = g.replay_steps(num_steps)[:, :n]
            out = g.next_host_log(num_steps)[:, :n]
            out.copy_(tokens, non_blocking=True)
            return ExecuteOutput(out, seq_ids, num_steps=num_steps)
        tokens = g.replay()[:n]
        out = g.next_host_tokens()[:n]
        out.copy_(tokens, non_blocking=True)
        return ExecuteOutput(out, seq_ids)


    @torch.inference_mode()
    def execute_mixed(self, metas, slot: int = 0) -> Optional[ExecuteOutput]:
        """A step of prompt chunks and decode tokens under the captured mixed graph, staged straight
        from the scheduler's metadata (MixedStepArrays); None when it does not fit."""
        graphs = self.graph_pools[slot] if self.graph_pools is not None else None
        if graphs is None or self.mixed_graph_tokens <= 0 or self.capture_logits:
            return None
        g = graphs.get_mixed(self.mixed_graph_tokens, self.mixed_graph_seqs)
        filled = g.staging.fill(metas)
        if filled is None:
            return None
        seq_ids, _ = filled
        g.load_staged()
        tokens = g.replay()[:len(seq_ids)]
        out = g.next_host_tokens()[:len(seq_ids)]
        out.copy_(tokens, non_blocking=True)
        return ExecuteOutput(out, seq_ids)


class LLMEngine:

    def __init__(self, model_config: ModelConfig, cache_config: CacheConfig,
                 scheduler_config: SchedulerConfig, device: str = "cuda:0", use_hip_graph: bool = True,
                 decode_version: Optional[str] = None, eos_token_id: Optional[int] = None, seed: int = 0):
        self.model_config, self.cache_config, self.scheduler_config = model_config, cache_config, scheduler_config
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.attn_backend = PagedAttnBackend()
        # (multi-step decode over the prefix-caching allocator: the reference's v2 manager under lookahead is replayed
        # bit for bit -- tests/golden/block_manager_v2_cached_lookahead*.json, scheduler_lookahead_prefix_cache_v2.json)
        # steps in flight run on separate streams (a stream per task, core/executor.py:62-93), so
        # each needs its own graph static buffers
        # ("simple_async": the reference's executor runs the queued steps one after the other,
        # core/executor.py:48-60 -- here they keep their own graph buffers but share one stream)
        self.num_slots = (max(1, scheduler_config.max_num_on_the_fly)
                          if scheduler_config.scheduling in ("simple_async", "async", "double_buffer") else 1)
        # chunked prefill with a token budget the decode GEMM takes (<= 64 rows): mixed steps replay a graph
        mixed = (scheduler_config.chunked_prefill_enabled and scheduler_config.max_num_batched_tokens <= 64
                 and cache_config.sliding_window is None and model_config.fuse_decode_ops
                 and cache_config.block_size in (16, 32)
                 and (cache_config.cache_dtype == "auto" or model_config.head_dim % 64 == 0))
        self.worker = Worker(model_config, cache_config, self.attn_backend, device, use_hip_graph,
                             decode_version, scheduler_config.max_model_len, seed, num_slots=self.num_slots,
                             mixed_graph_tokens=scheduler_config.max_num_batched_tokens if mixed else 0,
                             mixed_graph_seqs=min(scheduler_config.max_num_seqs,
                                                  scheduler_config.max_num_batched_tokens) if mixed else 0)
        # with steps on several streams every decode GEMM takes half the CUs and leaves the rest
        # to the other step's kernel (+10 % tokens/s when measured first; since the GEMM's activation
        # path was fixed a lone 128-workgroup GEMM is as fast as a 256-workgroup one and the setting
        # is worth +-1.5 %: profiles/r01_tuning.md)
        concurrent = self.num_slots > 1 and scheduler_config.scheduling != "simple_async"
        gemm_wgs = scheduler_config.gemm_workgroups or (128 if concurrent else 256)
        self.gemm_workgroups = gemm_wgs
        torch.ops._C_amd.set_tuning("gemm_workgroups", gemm_wgs)
        num_gpu, num_cpu = cache_config.num_gpu_blocks, cache_config.num_cpu_blocks
        if num_gpu is None:
            num_gpu, auto_cpu = self.worker.determine_num_available_blocks(scheduler_config)
            num_cpu = auto_cpu if num_cpu is None else num_cpu
        self.worker.initialize_cache(num_gpu, num_cpu or 0)
        chunked = scheduler_config.chunked_prefill_enabled
        self.scheduler = DecodingScheduler(scheduler_config, cache_config, chunked_prefill_enabled=chunked)
        self.input_builder = ModelInputBuilder(scheduler_config, cache_config, self.attn_backend,
                                               cache_config.sliding_window, chunked_prefill_enabled=chunked)
        self.eos_token_id = eos_token_id
        self.seq_counter = 0
        self.groups: Dict[str, SequenceGroup] = {}
        self.seq_to_group: Dict[int, SequenceGroup] = {}
        # async machinery (core/executor.py:48-185)
        if scheduler_config.scheduling == "simple_async":
            self.streams = [torch.cuda.Stream(self.device)] * self.num_slots
        else:
            self.streams = [torch.cuda.Stream(self.device) for _ in range(self.num_slots)]
        self.stream = self.streams[0]
        self.free_slots: "queue.Queue" = queue.Queue()
        for i in range(self.num_slots):
            self.free_slots.put(i)
        # one waiter per slot: a step's result is handed back when ITS stream finishes.  (The
        # reference's single helper waits in submission order, core/executor.py:66-70; when the
        # step submitted second finishes first -- usual with two steps sharing the GPU -- its
        # group then sits idle until the older step is done, and the GPU drains: measured 1.2 ms
        # of every 8 ms with nothing running, tools/trace_step.py.)
        self._done_threads: List[threading.Thread] = []
        self._done_qs: List["queue.Queue"] = [queue.Queue() for _ in range(self.num_slots)]
        self.executor_out: "queue.Queue" = queue.Queue()
        self.num_on_the_fly = 0
        self.fast_decode_inputs = scheduler_config.fast_decode_inputs
        # the engine thread polls the steps' events instead of being woken by a waiter thread (+2 % tokens/s;
        # SchedulerConfig.poll_completion = False brings the per-slot waiter threads back)
        self.poll_completion = scheduler_config.poll_completion
        self.device_sampler = None  # device_sampler.DeviceSampler, created with the first request that needs it
        self._sampler_seed = seed
        self._pending: List[Tuple[int, torch.cuda.Event, SchedulerOutput, ExecuteOutput]] = []
        self._last_event: Dict[int, Optional[torch.cuda.Event]] = {}  # latest step of each slot
        self._fence: Optional[torch.cuda.Event] = None                # latest block-moving step (see _launch)
        self.step_timeout_s = 300.0
        self.step_returns_outputs = True
        # counters a benchmark reads instead of inferring its work: model steps the processed results covered
        # (a burst of k counts k) and tokens actually appended to sequences
        self.stat_model_steps = 0
        self.stat_tokens_appended = 0

    def capture_decode_graphs(self, batch_size: int) -> None:
        """Set-up: capture every slot's HIP graph of a decode step of `batch_size` sequences now rather than
        at the slot's first step.  Each capture runs under the slot's own stream, as the first step would
        (DecodeGraphPool.get refuses to build a graph from the default stream: see there)."""
        if self.worker.graph_pools is None:
            return
        for slot, pool in enumerate(self.worker.graph_pools):
            with torch.cuda.stream(self.streams[slot]):
                pool.get(batch_size)
        torch.cuda.synchronize(self.device)

    # ---- requests ----
    def add_request(self, request_id: str, prompt_token_ids: List[int], max_tokens: Optional[int] = 16,
                    sampling_params=None) -> None:
        """`sampling_params` (engine/sampling_params.py): None = plain greedy with `max_tokens`; otherwise its
        max_tokens / stop_token_ids / ignore_eos are the stop criteria and its other fields drive the sampler."""
        seq = Sequence(self.seq_counter, list(prompt_token_ids), self.cache_config.block_size, self.eos_token_id)
        self.seq_counter += 1
        if sampling_params is not None:
            max_tokens = sampling_params.max_tokens
        g = SequenceGroup(request_id, [seq], time.time(), max_tokens=max_tokens, sampling_params=sampling_params,
                          n=sampling_params.num_samples if sampling_params is not None else 1)
        self.groups[request_id] = g
        self.seq_to_group[seq.seq_id] = g
        self.scheduler.add_request(g)

    def has_unfinished_requests(self) -> bool:
        return self.scheduler.has_unfinished_requests()

    # ---- output processing (decoding/processor/output_processor.py, greedy subset) ----
    def _process_burst(self, sched: SchedulerOutput, out: ExecuteOutput) -> List[RequestOutput]:
        """Output processing of a multi-step decode: sequence i sampled out.sampled[0..k-1, i]; the tokens are
        appended one by one under the stop checks of the single-step path, and what a sequence sampled after it
        finished is dropped (its KV went into lookahead slots that are freed with the sequence)."""
        rows = out.sampled.t().tolist()  # [n][k]
        tok_of = dict(zip(out.sample_seq_ids, rows))
        self.stat_model_steps += out.num_steps
        results: List[RequestOutput] = []
        max_model_len = self.scheduler_config.max_model_len
        eos = self.eos_token_id
        for s in sched.scheduled_seq_groups:
            g = s.seq_group
            seq = g.seqs[0]
            for tok in tok_of[seq.seq_id]:
                seq.data.update_num_computed_tokens(1)
                seq.append_token_id(tok, 0.0)
                self.stat_tokens_appended += 1
                sp = g.sampling_params
                if (eos is not None and tok == eos and not (sp is not None and sp.ignore_eos)) or \
                        (sp is not None and tok in sp.stop_token_ids):
                    seq.status = SequenceStatus.FINISHED_STOPPED
                elif g.max_tokens is not None and seq.get_output_len() >= g.max_tokens:
                    seq.status = SequenceStatus.FINISHED_LENGTH_CAPPED
                elif seq.get_len() >= max_model_len:
                    seq.status = SequenceStatus.FINISHED_LENGTH_CAPPED
                if seq.status > 2:
                    break
            finished = seq.status > 2
            if finished:
                self.scheduler.free_seq(seq)
            if self.step_returns_outputs:
                results.append(RequestOutput(g.request_id, list(seq.get_output_token_ids()), finished,
                                             SequenceStatus.get_finished_reason(seq.status)))
            else:
                results.append(RequestOutput(g.request_id, [], finished))
        if any(r.finished for r in results):
            self.scheduler.free_finished_request([s.seq_group.request_id for s in sched.scheduled_seq_groups])
            for r in results:
                if r.finished:
                    g = self.groups.pop(r.request_id, None)
                    if g is not None:
                        for seq in g.seqs:
                            self.seq_to_group.pop(seq.seq_id, None)
                            if self.device_sampler is not None:
                                self.device_sampler.release(seq.seq_id)
        else:
            for s in sched.scheduled_seq_groups:
                s.seq_group.busy = False
        return results

    def _process(self, sched: SchedulerOutput, out: ExecuteOutput) -> List[RequestOutput]:
        if out.num_steps > 1:
            return self._process_burst(sched, out)
        sampled = out.sampled.tolist() if out.sampled is not None else []
        tok_of = dict(zip(out.sample_seq_ids, sampled))
        lp_of = dict(zip(out.sample_seq_ids, out.logprobs)) if out.logprobs is not None else None
        if out.prompt_logprobs:
            for sid, dicts in out.prompt_logprobs.items():
                g_ = self.seq_to_group.get(sid)
                seq_ = next((q for q in g_.seqs if q.seq_id == sid), None) if g_ is not None else None
                if seq_ is not None:
                    if not seq_.prompt_logprobs:
                        seq_.prompt_logprobs.append(None)  # the first prompt token has none (sampler.py:868-870)
                    seq_.prompt_logprobs.extend(dicts)
        self.stat_model_steps += 1 if sched.scheduled_seq_groups else 0
        self.stat_tokens_appended += len(sampled)
        results: List[RequestOutput] = []
        max_model_len = self.scheduler_config.max_model_len
        for s in sched.scheduled_seq_groups:
            g = s.seq_group
            seqs = g.seqs
            if len(seqs) == 1 and g.n <= 1 and seqs[0].status == SequenceStatus.RUNNING and seqs[0].seq_id in tok_of:
                # one running sequence that sampled a token: the loop below, without its lists
                seq = seqs[0]
                seq.data.update_num_computed_tokens(s.token_chunk_size)
                tok = tok_of[seq.seq_id]
                lp = lp_of.get(seq.seq_id) if lp_of is not None else None
                if lp is not None:
                    seq.output_logprobs.append(lp)
                    seq.append_token_id(tok, lp[tok][0])
                else:
                    seq.append_token_id(tok, 0.0)
                sp = g.sampling_params
                if ((self.eos_token_id is not None and tok == self.eos_token_id and not (sp is not None and sp.ignore_eos))
                        or (sp is not None and tok in sp.stop_token_ids)):
                    seq.status = SequenceStatus.FINISHED_STOPPED
                elif g.max_tokens is not None and seq.get_output_len() >= g.max_tokens:
                    seq.status = SequenceStatus.FINISHED_LENGTH_CAPPED
                elif seq.get_len() >= max_model_len:
                    seq.status = SequenceStatus.FINISHED_LENGTH_CAPPED
                finished = seq.status > 2
                if finished:
                    self.scheduler.free_seq(seq)
                if self.step_returns_outputs:
                    results.append(RequestOutput(g.request_id, list(seq.get_output_token_ids()), finished,
                                                 SequenceStatus.get_finished_reason(seq.status),
                                                 list(seq.output_logprobs) if seq.output_logprobs else None,
                                                 list(seq.prompt_logprobs) if seq.prompt_logprobs else None))
                else:
                    results.append(RequestOutput(g.request_id, [], finished))
                continue
            g.update_num_computed_tokens(s.token_chunk_size)
            sp = g.sampling_params
            show_lp = sp is not None and sp.logprobs is not None

            def take(seq, tok, lp):  # one sampled token of one sequence: the dictionary, the token, the stop checks
                if lp is not None:
                    if show_lp:
                        seq.output_logprobs.append(lp)
                    seq.append_token_id(tok, lp[tok][0])
                else:
                    seq.append_token_id(tok, 0.0)
                if ((self.eos_token_id is not None and tok == self.eos_token_id and not (sp is not None and sp.ignore_eos))
                        or (sp is not None and tok in sp.stop_token_ids)):
                    seq.status = SequenceStatus.FINISHED_STOPPED
                elif g.max_tokens is not None and seq.get_output_len() >= g.max_tokens:
                    seq.status = SequenceStatus.FINISHED_LENGTH_CAPPED
                elif seq.get_len() >= self.scheduler_config.max_model_len:
                    seq.status = SequenceStatus.FINISHED_LENGTH_CAPPED
                if seq.is_finished():
                    self.scheduler.free_seq(seq)

            for seq in list(g.get_seqs(status=SequenceStatus.RUNNING)):
                if seq.seq_id not in tok_of:
                    continue  # a prompt chunk that sampled nothing
                extra = out.extra_samples.get(seq.seq_id) if out.extra_samples else None
                if extra:
                    # SamplingParams.n / best_of: the prompt's last position drew for every sequence of the request
                    # (sampler.py:385-432); the further ones fork from the parent BEFORE it takes its own token
                    # (output_processor.py:84-92: scheduler.fork_seq shares the prompt's blocks copy-on-write)
                    for tok_x, lp_x in extra:
                        child = seq.fork(self.seq_counter)
                        self.seq_counter += 1
                        g.seqs.append(child)
                        g.seqs_dict[child.seq_id] = child
                        self.seq_to_group[child.seq_id] = g
                        self.scheduler.fork_seq(seq, child)
                        take(child, tok_x, lp_x)
                        self.stat_tokens_appended += 1
                take(seq, tok_of[seq.seq_id], lp_of.get(seq.seq_id) if lp_of is not None else None)
            if self.step_returns_outputs:
                ranked = g.seqs
                if g.n > 1 and sp is not None:  # the n most likely of the request's sequences (cumulative log-probability)
                    ranked = sorted(g.seqs, key=lambda q: q.data.cumulative_logprob, reverse=True)[:sp.n]
                seq0 = ranked[0]
                results.append(RequestOutput(g.request_id, list(seq0.get_output_token_ids()), g.is_finished(),
                                             SequenceStatus.get_finished_reason(seq0.status),
                                             list(seq0.output_logprobs) if seq0.output_logprobs else None,
                                             list(seq0.prompt_logprobs) if seq0.prompt_logprobs else None,
                                             [list(q.get_output_token_ids()) for q in ranked] if g.n > 1 else None))
            else:
                results.append(RequestOutput(g.request_id, [], g.is_finished()))
        for g in sched.ignored_seq_groups:
            results.append(RequestOutput(g.request_id, [], True, "length"))
        if any(r.finished for r in results):
            self.scheduler.free_finished_request([s.seq_group.request_id for s in sched.scheduled_seq_groups])
            for r in results:
                if r.finished:
                    g = self.groups.pop(r.request_id, None)
                    if g is not None:
                        for seq in g.seqs:
                            self.seq_to_group.pop(seq.seq_id, None)
                            if self.device_sampler is not None:
                                self.device_sampler.release(seq.seq_id)
        else:  # nobody left: free_finished_request would only clear the busy flags of this step's groups
            for s in sched.scheduled_seq_groups:
                s.seq_group.busy = False
        return results

    # ---- synchronous step (core/llm_engine.py:119-130) ----
    def step(self) -> List[RequestOutput]:
        sched = self.scheduler.schedule()
        if sched is None or sched.is_empty():
            if sched is not None and sched.ignored_seq_groups:
                return self._process(sched, ExecuteOutput(None, []))
            return []
        with torch.cuda.stream(self.stream):
            out = self._execute(sched, 0)
        self.stream.synchronize()
        return self._process(sched, out)

    # ---- asynchronous step (core/llm_engine.py:132-176) ----
    def _execute(self, sched: SchedulerOutput, slot: int) -> ExecuteOutput:
        """Decode-only steps go from the scheduler's metadata to the captured graph directly; every
        other step through the general input builder.  Steps with requests that are not plain greedy carry their
        sampler state slots along and stay on the same paths (fast decode inputs, multi-step bursts included)."""
        state_slots = self._sampler_slots(sched)
        logprob_rows = self._logprob_rows(sched)
        if logprob_rows is not None:  # the general path, one model step (the reference's multi-step has none either)
            return self.worker.execute(self.input_builder(sched), slot, state_slots=state_slots,
                                       logprob_rows=logprob_rows)
        if self.fast_decode_inputs:
            from .input_builder import DecodeStepArrays, MixedStepArrays
            plain = not (sched.blocks_to_swap_in or sched.blocks_to_swap_out or sched.blocks_to_copy)
            metas = sched.seq_group_metadata_list
            if DecodeStepArrays.eligible(metas, plain, self.cache_config.sliding_window):
                out = self.worker.execute_decode(metas, slot, self._burst_steps(sched), state_slots)
                if out is not None:
                    return out
            elif (state_slots is None and self.worker.mixed_graph_tokens > 0
                  and MixedStepArrays.eligible(metas, plain, self.cache_config.sliding_window)):
                out = self.worker.execute_mixed(metas, slot)
                if out is not None:
                    return out
        return self.worker.execute(self.input_builder(sched), slot, state_slots=state_slots)

    def _logprob_rows(self, sched: SchedulerOutput) -> Optional[Dict[int, dict]]:
        """None unless some request of the step asks for log-probabilities; then, per running sequence of such a
        request, what sampling.SamplingBatch needs to restate its adjusted logits."""
        rows: Dict[int, dict] = {}
        wanted = False
        for s in sched.scheduled_seq_groups:
            sp = s.seq_group.sampling_params
            if sp is not None and (sp.logprobs is not None or sp.prompt_logprobs is not None or sp.num_samples > 1):
                wanted = True
                break
        if not wanted:
            return None
        from .input_builder import token_rows
        for sid, m, row, n_rows, ctx, end in token_rows(sched.seq_group_metadata_list, self.cache_config.block_size,
                                                        self.cache_config.sliding_window):
            g = self.groups.get(m.request_id)
            sp = g.sampling_params if g is not None else None
            if sp is None or (sp.logprobs is None and sp.prompt_logprobs is None and sp.num_samples <= 1):
                continue
            seq = next(q for q in g.seqs if q.seq_id == sid)
            entry = dict(params=sp, prompt=seq.prompt_token_ids, output=seq.get_output_token_ids(), eos=self.eos_token_id)
            # what the sampled token's dictionary holds: the request's n; 0 (the token alone) for a request that
            # forks, whose sequences are ranked by cumulative log-probability at the end; None: nothing
            entry["num"] = sp.logprobs if sp.logprobs is not None else (0 if sp.num_samples > 1 else None)
            if m.is_prompt and m.do_sample and sp.num_samples > len(g.seqs):
                entry["extra"] = sp.num_samples - len(g.seqs)  # the prompt's last position draws for every sequence
            if m.is_prompt and sp.prompt_logprobs is not None:
                toks = m.seq_data[sid].get_token_ids()
                # positions ctx .. end - 1 predict tokens ctx + 1 .. end; the one that completes the prompt samples
                n_lp = n_rows - 1 if (m.do_sample and end >= len(toks)) else n_rows
                n_lp = max(0, min(n_lp, len(toks) - 1 - ctx))
                entry["prompt_rows"] = (row, n_lp, list(toks[ctx + 1:ctx + 1 + n_lp]))
            rows[sid] = entry
        return rows or None

    def _sampler_slots(self, sched: SchedulerOutput) -> Optional[Dict[int, int]]:
        """None when every request of the step is plain greedy (the captured arg-max serves it); otherwise
        {seq id: state slot on the device} for the step's requests that are not, their state built on the current
        stream where it does not exist yet (new request, or evicted while it waited)."""
        groups = [s.seq_group for s in sched.scheduled_seq_groups
                  if s.seq_group.sampling_params is not None and not s.seq_group.sampling_params.plain_greedy]
        if not groups:
            return None
        if self.device_sampler is None:
            from ..device_sampler import DeviceSampler
            n = self.worker.sampler_state_slots(self.scheduler_config)  # reserved when the cache was sized
            self.device_sampler = self.worker.sampler = DeviceSampler(self.model_config.vocab_size, self.device, n,
                                                                      seed=self._sampler_seed)

        def pinned():  # sequences of steps in flight and of this step keep their slots
            keep = {seq.seq_id for g in self.groups.values() if g.busy for seq in g.seqs}
            keep.update(seq.seq_id for g in groups for seq in g.seqs)
            return keep
        out: Dict[int, int] = {}
        for g in groups:
            for k, seq in enumerate(g.seqs):
                if seq.status == SequenceStatus.RUNNING:
                    out[seq.seq_id] = self.device_sampler.ensure(seq.seq_id, g.sampling_params, seq.prompt_token_ids,
                                                                 seq.get_output_token_ids(), self.eos_token_id, pinned,
                                                                 salt=k)
        return out or None

    def _burst_steps(self, sched: SchedulerOutput) -> int:
        """Model steps a decode-only step runs on the device before it returns to the host: the configured
        `num_scheduler_steps` when the block manager reserved the slots for it and no sequence would run past
        the model length (positions of a burst: len - 1 .. len + k - 2), else 1."""
        k = self.scheduler_config.num_scheduler_steps
        if k <= 1 or sched.num_lookahead_slots < k - 1:
            return 1
        limit = self.scheduler_config.max_model_len
        for m in sched.seq_group_metadata_list:
            for data in m.seq_data.values():
                if data.get_len() + k - 1 > limit:
                    return 1
        return k

    def _launch(self, sched: SchedulerOutput) -> None:
        """What the reference's async_execute_loop does per task (core/executor.py:62-93): take a
        stream from the pool, launch the step on it without waiting for the previous one -- so up
        to `max_num_on_the_fly` steps overlap on the GPU -- and leave the waiting to a helper.
        Here the launch runs on the engine thread itself: a separate launcher thread has to win
        the GIL from the engine thread first, which is busy preparing the other group's step, so
        both launches ended up back to back and the two steps ran, and finished, in phase --
        leaving the GPU idle for the whole host turnaround (measured: 1.0 ms of every 8 ms).
        Launched inline the steps stay staggered by one turnaround and hide it for each other."""
        slot = self.free_slots.get()
        try:
            t0 = time.perf_counter()
            stream = self.streams[slot]
            # Steps in flight run on their own streams with no ordering between them, which is only safe while
            # they touch disjoint blocks.  A step that MOVES blocks (swap in / out, copy-on-write) or an engine
            # with prefix caching (a later prompt reads blocks an earlier step is still filling; blocks are
            # marked computed at schedule time, scheduler.py:924-926) is ordered on the device instead: it waits
            # for every step in flight, and every later step waits for it.  No host wait.
            moves = bool(sched.blocks_to_swap_in or sched.blocks_to_swap_out or sched.blocks_to_copy)
            fence = moves or self.cache_config.enable_prefix_caching
            with torch.cuda.stream(stream):
                if self._fence is not None:
                    if self._fence.query():
                        self._fence = None
                    else:
                        stream.wait_event(self._fence)
                if fence:
                    for other, last in self._last_event.items():
                        if other != slot and last is not None and not last.query():
                            stream.wait_event(last)
                out = self._execute(sched, slot)
                ev = torch.cuda.Event()
                ev.record(stream)
            self._last_event[slot] = ev
            if fence:
                self._fence = ev
            out.execute_begin_ts = t0
            if self.poll_completion:
                self._pending.append((slot, ev, sched, out))
            else:
                self._done_qs[slot].put((slot, ev, sched, out))
        except Exception:
            self.free_slots.put(slot)
            raise

    def _done_loop(self, done_q: "queue.Queue") -> None:
        torch.cuda.set_device(self.device)
        while True:
            item = done_q.get()
            if item is None:
                return
            slot, ev, sched, out = item
            try:
                ev.synchronize()  # results are on the host
                out.execute_end_ts = time.perf_counter()
                self.free_slots.put(slot)
                self.executor_out.put((sched, out))
            except Exception as e:
                self.free_slots.put(slot)
                self.executor_out.put(e)

    def ensure_start_execute_loop(self) -> None:
        if not self._done_threads and not self.poll_completion:
            self._done_threads = [threading.Thread(target=self._done_loop, args=(q,), daemon=True)
                                  for q in self._done_qs]
            for t in self._done_threads:
                t.start()

    def async_step(self, schedule_more: bool = True) -> List[RequestOutput]:
        """schedule_more=False only collects a result (used to drain the pipeline)."""
        self.ensure_start_execute_loop()
        # keep up to max_num_on_the_fly steps queued behind the one that is executing
        limit = min(self.scheduler_config.max_num_on_the_fly, self.num_slots)  # a step needs a slot's buffers
        while schedule_more and self.num_on_the_fly < limit:
            sched = self.scheduler.schedule()
            if sched is None or sched.is_empty():
                break
            self._launch(sched)
            self.num_on_the_fly += 1
        if self.num_on_the_fly == 0:
            return []
        if self.poll_completion:
            # the engine thread watches the steps' events itself: no waiter thread to wake, no queue hop
            # between a step finishing and its group's next step being prepared
            spins = 0
            t_wait = None
            while True:
                for i, (slot, ev, sched, out) in enumerate(self._pending):
                    if ev.query():
                        del self._pending[i]
                        self.free_slots.put(slot)
                        out.execute_end_ts = time.perf_counter()
                        self.num_on_the_fly -= 1
                        return self._process(sched, out)
                spins += 1
                if spins & 0xfff == 0:  # a hung step must not spin the engine thread for ever
                    now = time.perf_counter()
                    t_wait = t_wait or now
                    if now - t_wait > self.step_timeout_s:
                        raise RuntimeError(f"no step completed within {self.step_timeout_s:.0f} s "
                                           f"({len(self._pending)} in flight)")
        item = self.executor_out.get()
        if isinstance(item, Exception):
            raise item
        self.num_on_the_fly -= 1
        sched, out = item
        return self._process(sched, out)

    def shutdown(self) -> None:
        if self._done_threads:
            for q in self._done_qs:
                q.put(None)
            for t in self._done_threads:
                t.join(timeout=5)
            self._done_threads = []

    # ---- synthetic context (benchmarks): mark prompts as computed and fill their KV ----
    def prefill_synthetic(self, seed: int = 0) -> None:
        """Admit every waiting request as if its prompt had been prefilled: blocks are
        allocated through the scheduler's block manager, K/V rows are random values written
        through `reshape_and_cache`, and the first output token is a random id.  Used by
        bench.py to start decode at a given context length without timing prompt processing."""
        from .. import _custom_ops as ops
        from ..attention.backend import compute_slot_mapping
        from ..paged_attn import PagedAttention
        gen = torch.Generator(device=self.device).manual_seed(seed)
        cfg = self.model_config
        sched = self.scheduler
        admitted: List[SequenceGroup] = []
        while sched.waiting:
            g = sched.waiting[0]
            if sched.block_manager.can_allocate(g).name != "OK":
                break
            sched.waiting.popleft()
            sched.block_manager.allocate(g)
            for seq in g.get_seqs(status=SequenceStatus.WAITING):
                seq.status = SequenceStatus.RUNNING
            sched.running.append(g)
            admitted.append(g)
        slot_mapping: List[int] = []
        for g in admitted:
            seq = g.seqs[0]
            table = {seq.seq_id: sched.block_manager.get_block_table(seq)}
            compute_slot_mapping(False, slot_mapping, seq.seq_id, seq.get_len(), 0, 0,
                                 self.cache_config.block_size, table)
        slots = torch.tensor(slot_mapping, dtype=torch.int64, device=self.device)
        T = slots.numel()
        for kv in self.worker.cache_engine.gpu_cache:
            kc, vc = PagedAttention.split_kv_cache(kv, cfg.num_key_value_heads, cfg.head_dim)
            for s0 in range(0, T, 8192):
                n = min(8192, T - s0)
                k = (torch.randn(n, cfg.num_key_value_heads, cfg.head_dim, generator=gen, device=self.device) * 0.5).to(cfg.dtype)
                v = (torch.randn(n, cfg.num_key_value_heads, cfg.head_dim, generator=gen, device=self.device) * 0.5).to(cfg.dtype)
                ops.reshape_and_cache(k, v, kc, vc, slots[s0:s0 + n], self.cache_config.cache_dtype, 1.0, 1.0)
        for g in admitted:
            seq = g.seqs[0]
            seq.data.update_num_computed_tokens(seq.get_len())
            seq.append_token_id(int(torch.randint(0, cfg.vocab_size, (1,), generator=gen, device=self.device)), 0.0)
        torch.cuda.synchronize(self.device)
