"""Host side of the device-resident sampler (csrc/sampler.hip, `lvllm_sample_rows`).

The reference rebuilds its sampling tensors on the host every step that has penalties (sampler.py:113-125: "we
cannot reuse sampling tensors, since output_tokens changes between decode runs") -- padded prompt / output token
matrices, H2D copies, ~25 torch launches on the logits.  Here a request that is not plain greedy owns a STATE SLOT on
the device for as long as it lives: its SamplingParams as a 128-byte record and one int32 per vocabulary entry (bit
31: in the prompt; low bits: occurrences in the output).  The sampling kernel reads the slot and appends the token it
drew, so a decode step needs one int32 per row from the host (the slot number, -1 = plain greedy) and a burst of k
model steps needs nothing in between.  A slot is a pure function of host-side data (params, prompt, outputs so far),
so it can be dropped and rebuilt at any step boundary: slots of sequences that are not in a step in flight are
evicted LRU when the pool is full.

Mirrors, for the engine, the job of SamplingMetadata / SamplingTensors (decoding/backends/sampling_metadata.py).
"""
import random
import struct
from typing import Dict, List, Optional, Sequence

import torch

from . import _custom_ops  # noqa: F401  (registers torch.ops._C_amd; raises if the HIP library is missing)

PARAMS_BYTES = 128
MAX_BANNED = 20
_FMT = "<6f2iQ2i20i"
assert struct.calcsize(_FMT) == PARAMS_BYTES


def pack_params(sp, vocab_size: int, output_len: int, seed: int, eos: Optional[int]) -> bytes:
    """SamplingParams -> the kernel's record (include/lvllm_hip.h, lvllm_sample_rows)."""
    banned: List[int] = []
    if sp.min_tokens > 0:  # sampler.py:238-277: stop tokens (and EOS unless ignored) cannot appear before min_tokens
        banned = sorted(set(sp.stop_token_ids) | ({eos} if eos is not None and not sp.ignore_eos else set()))
        if len(banned) > MAX_BANNED:
            raise ValueError(f"min_tokens with {len(banned)} stop tokens: the device sampler bans at most {MAX_BANNED}")
    top_k = vocab_size if sp.top_k == -1 else min(sp.top_k, vocab_size)
    return struct.pack(_FMT, float(sp.temperature), float(sp.top_p), float(sp.min_p), float(sp.presence_penalty),
                       float(sp.frequency_penalty), float(sp.repetition_penalty), int(top_k), int(sp.min_tokens),
                       int(seed) & 0xFFFFFFFFFFFFFFFF, int(output_len), len(banned),
                       *(banned + [0] * (MAX_BANNED - len(banned))))


class DeviceSampler:

    def __init__(self, vocab_size: int, device, num_slots: int = 64, seed: int = 0):
        self.vocab_size, self.device, self.num_slots = vocab_size, torch.device(device), num_slots
        self.params = torch.zeros(num_slots, PARAMS_BYTES, dtype=torch.uint8, device=self.device)
        self.counts = torch.zeros(num_slots, vocab_size, dtype=torch.int32, device=self.device)
        self._slot_of: Dict[int, int] = {}
        self._owner: List[Optional[int]] = [None] * num_slots
        self._last_used: List[int] = [0] * num_slots
        self._clock = 0
        self._rng = random.Random(seed)  # seeds of requests that did not bring one
        self._seeds: Dict[int, int] = {}
        self._scratch: Dict[tuple, torch.Tensor] = {}
        self.evictions = 0

    # ---- slots ----
    def slot_of(self, seq_id: int) -> Optional[int]:
        return self._slot_of.get(seq_id)

    def release(self, seq_id: int) -> None:
        s = self._slot_of.pop(seq_id, None)
        if s is not None:
            self._owner[s] = None
        self._seeds.pop(seq_id, None)

    def ensure(self, seq_id: int, sp, prompt_ids: Sequence[int], output_ids: Sequence[int], eos: Optional[int],
               pinned=(), salt: int = 0) -> int:
        """The slot of `seq_id`, built on the CURRENT stream from the request's histories when it has none (new
        request, or evicted while it waited).  `pinned`: sequence ids whose slots must stay (steps in flight, the step
        being built), or a callable returning them -- asked only when a slot has to be evicted."""
        self._clock += 1
        s = self._slot_of.get(seq_id)
        if s is not None:
            self._last_used[s] = self._clock
            return s
        s = self._take_slot(pinned)
        seed = self._seeds.get(seq_id)
        if seed is None:
            seed = sp.seed if sp.seed is not None else self._rng.getrandbits(64)
            if salt:  # the further sequences of a request that forks (SamplingParams.n): streams of their own
                seed = (seed * 0x9E3779B97F4A7C15 + salt * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
            self._seeds[seq_id] = seed
        rec = pack_params(sp, self.vocab_size, len(output_ids), seed, eos)
        pin = self.device.type == "cuda"
        host = torch.frombuffer(bytearray(rec), dtype=torch.uint8)
        self.params[s].copy_(host.pin_memory() if pin else host, non_blocking=True)

        def dev(ids):
            t = torch.tensor(list(ids), dtype=torch.long)
            return (t.pin_memory() if pin else t).to(self.device, non_blocking=True)
        torch.ops._C_amd.sampler_init_row(self.counts[s], dev(prompt_ids), dev(output_ids))
        self._slot_of[seq_id] = s
        self._owner[s] = seq_id
        self._last_used[s] = self._clock
        return s

    def _take_slot(self, pinned) -> int:
        for s, o in enumerate(self._owner):
            if o is None:
                return s
        pinned = set(pinned() if callable(pinned) else pinned)
        victims = [s for s, o in enumerate(self._owner) if o not in pinned]
        if not victims:
            raise RuntimeError(f"device sampler: all {self.num_slots} state slots belong to the step being built")
        s = min(victims, key=lambda i: self._last_used[i])
        self._slot_of.pop(self._owner[s], None)
        self._owner[s] = None
        self.evictions += 1
        return s

    # ---- sampling ----
    def new_scratch(self, rows: int, device=None) -> torch.Tensor:
        """Working rows of a launch: the vocabulary (4-aligned) + the 64-float tail in which the workgroups that share a
        row's first pass meet (include/lvllm_hip.h, lvllm_sample_rows) -- zeroed once, the kernel leaves it zeroed.
        One per launch in flight (per captured graph): launches must not share it."""
        return torch.zeros(rows, ((self.vocab_size + 3) & ~3) + 64, dtype=torch.float32, device=device or self.device)

    def scratch_for(self, rows: int, device=None) -> torch.Tensor:
        """The eager path's working rows, kept per (rows, stream): a buffer belongs to one launch at a time, launches
        on one stream are ordered, and steps in flight run on their own streams."""
        device = device or self.device
        stream = torch.cuda.current_stream(device).cuda_stream if torch.device(device).type == "cuda" else 0
        key = (rows, str(device), stream)
        t = self._scratch.get(key)
        if t is None:
            t = self._scratch[key] = self.new_scratch(rows, device)
        return t

    def sample(self, logits: torch.Tensor, state_slot: torch.Tensor, tokens_out: Optional[torch.Tensor] = None,
               scratch: Optional[torch.Tensor] = None, processed_out: Optional[torch.Tensor] = None,
               update_state: bool = True) -> torch.Tensor:
        """tokens [rows] int64 for `logits` [rows, vocab]; state_slot int32 [rows] on the device (-1 = plain greedy)."""
        rows = logits.shape[0]
        if tokens_out is None:
            tokens_out = torch.empty(rows, dtype=torch.long, device=logits.device)
        if scratch is None:
            scratch = self.scratch_for(rows, logits.device)
        torch.ops._C_amd.sample_rows(tokens_out, logits, state_slot, self.params, self.counts, scratch, processed_out,
                                     update_state)
        return tokens_out
