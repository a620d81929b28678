"""Sequences and sequence groups as the block manager and the scheduler see them.

The subset of light_vllm/decoding/schema/sequence.py (:37-69 SequenceStatus, :95-245
SequenceData, :247-395 Sequence, :397-575 SequenceGroup) that the KV-cache path reads:
token ids, lengths, computed-token bookkeeping, per-block content hashes, status and
group membership.  Sampling state, logprobs and detokenisation are out of this path.

The block managers of this package only use this interface by duck typing
(`seq_id`, `n_blocks`, `get_len()`, `get_token_ids()`, `data.get_len()`,
`data.get_num_computed_tokens()`, `hash_of_block(i)`, `num_hashed_tokens_of_block(i)`,
`is_finished()`, group: `request_id`, `get_seqs(status)`, `num_seqs(status)`,
`get_max_num_running_seqs()`, `is_prefill()`, `seqs_dict`), so the reference's own
Sequence / SequenceGroup objects can be handed to them unchanged.
"""
import enum
from typing import Dict, List, Optional, Tuple


class SequenceStatus(enum.IntEnum):
    WAITING = 0
    RUNNING = 1
    SWAPPED = 2
    # everything after SWAPPED is a finished state (sequence.py:41-46)
    FINISHED_STOPPED = 3
    FINISHED_LENGTH_CAPPED = 4
    FINISHED_ABORTED = 5
    FINISHED_IGNORED = 6

    @staticmethod
    def is_finished(status: "SequenceStatus") -> bool:
        return status > SequenceStatus.SWAPPED

    @staticmethod
    def get_finished_reason(status: "SequenceStatus") -> Optional[str]:
        return {SequenceStatus.FINISHED_STOPPED: "stop",
                SequenceStatus.FINISHED_LENGTH_CAPPED: "length",
                SequenceStatus.FINISHED_ABORTED: "abort",
                SequenceStatus.FINISHED_IGNORED: "length"}.get(status)


class SequenceStage(enum.Enum):
    PREFILL = enum.auto()
    DECODE = enum.auto()


class SequenceData:
    """Token ids of one sequence and how many of them have been run through the model."""

    __slots__ = ("_prompt", "_output", "_all", "_num_computed", "_stage", "cumulative_logprob")

    def __init__(self, prompt_token_ids: List[int], output_token_ids: Optional[List[int]] = None):
        self._prompt: Tuple[int, ...] = tuple(prompt_token_ids)
        self._output: List[int] = list(output_token_ids or [])
        self._all: List[int] = list(self._prompt) + self._output
        self._num_computed = 0
        self._stage = SequenceStage.PREFILL
        self.cumulative_logprob = 0.0

    @property
    def prompt_token_ids(self) -> Tuple[int, ...]:
        return self._prompt

    @property
    def output_token_ids(self) -> Tuple[int, ...]:
        return tuple(self._output)

    def append_token_id(self, token_id: int, logprob: float = 0.0) -> None:
        self._output.append(token_id)
        self._all.append(token_id)
        self.cumulative_logprob += logprob

    def get_len(self) -> int:
        return len(self._all)

    def get_prompt_len(self) -> int:
        return len(self._prompt)

    def get_output_len(self) -> int:
        return len(self._output)

    def get_token_ids(self) -> List[int]:
        return self._all

    def get_prefix_token_ids(self, num_tokens: int):
        """Hashable prefix (sequence.py:186-195)."""
        n = len(self._prompt)
        if num_tokens > n:
            return (self._prompt, tuple(self._output[:num_tokens - n]))
        return (self._prompt[:num_tokens], None)

    def get_num_computed_tokens(self) -> int:
        return self._num_computed

    def update_num_computed_tokens(self, num_new_computed_tokens: int) -> None:
        self._num_computed += num_new_computed_tokens
        assert self._num_computed <= self.get_len(), (self._num_computed, self.get_len())
        if self.get_num_uncomputed_tokens() == 0:
            self._stage = SequenceStage.DECODE

    def reset_state_for_recompute(self) -> None:
        self._num_computed = 0
        self._stage = SequenceStage.PREFILL

    def get_num_uncomputed_tokens(self) -> int:
        # prompt + output: a recomputed sequence prefills both (sequence.py:218-223)
        return self.get_len() - self._num_computed

    def get_last_token_id(self) -> int:
        return self._all[-1]

    @property
    def stage(self) -> SequenceStage:
        return self._stage


class Sequence:

    def __init__(self, seq_id: int, prompt_token_ids: List[int], block_size: int,
                 eos_token_id: Optional[int] = None) -> None:
        self.seq_id = seq_id
        self.block_size = block_size
        self.eos_token_id = eos_token_id
        self.data = SequenceData(prompt_token_ids)
        self.status = SequenceStatus.WAITING
        self.stop_reason = None
        self.output_logprobs: List[dict] = []  # one {token id: (logprob, rank)} per output token when asked for
        self.prompt_logprobs: List[Optional[dict]] = []  # [None] + one per further prompt token when asked for

    @property
    def n_blocks(self) -> int:
        return (self.get_len() + self.block_size - 1) // self.block_size

    @property
    def prompt_token_ids(self) -> Tuple[int, ...]:
        return self.data.prompt_token_ids

    def hash_of_block(self, logical_idx: int) -> int:
        """Content hash of blocks 0..logical_idx (sequence.py:300-308): equal for two
        sequences exactly when their first (logical_idx+1)*block_size tokens are equal."""
        return hash(self.data.get_prefix_token_ids(self.num_hashed_tokens_of_block(logical_idx)))

    def num_hashed_tokens_of_block(self, logical_idx: int) -> int:
        return logical_idx * self.block_size + self.block_size

    def reset_state_for_recompute(self) -> None:
        self.data.reset_state_for_recompute()

    def append_token_id(self, token_id: int, logprob: float = 0.0) -> None:
        self.data.append_token_id(token_id, logprob)

    def get_len(self) -> int:
        return self.data.get_len()

    def get_prompt_len(self) -> int:
        return self.data.get_prompt_len()

    def get_output_len(self) -> int:
        return self.data.get_output_len()

    def get_token_ids(self) -> List[int]:
        return self.data.get_token_ids()

    def get_last_token_id(self) -> int:
        return self.data.get_last_token_id()

    def get_output_token_ids(self) -> Tuple[int, ...]:
        return self.data.output_token_ids

    def is_finished(self) -> bool:
        return self.status > 2  # SequenceStatus.SWAPPED: every later state is a finished one

    def fork(self, new_seq_id: int) -> "Sequence":
        child = Sequence(new_seq_id, list(self.data.prompt_token_ids), self.block_size, self.eos_token_id)
        for t in self.data.output_token_ids:
            child.data.append_token_id(t)
        child.data._num_computed = self.data._num_computed
        child.data._stage = self.data._stage
        child.status = self.status
        return child

    def get_num_new_tokens(self) -> int:
        """1 for a decoding sequence, else the not yet computed tokens (sequence.py:377-386)."""
        if self.data.stage == SequenceStage.DECODE:
            return 1
        return self.data.get_num_uncomputed_tokens()

    def is_prefill(self) -> bool:
        return self.data.stage == SequenceStage.PREFILL

    def __repr__(self) -> str:
        return f"Sequence(seq_id={self.seq_id}, status={self.status.name}, num_blocks={self.n_blocks})"


class SequenceGroup:
    """Sequences generated from one prompt (sequence.py:397-575)."""

    def __init__(self, request_id: str, seqs: List[Sequence], arrival_time: float = 0.0,
                 max_tokens: Optional[int] = None, n: int = 1, sampling_params=None) -> None:
        self.request_id = request_id
        # None or plain greedy: the captured step's arg-max; otherwise the sampler's front half runs on the
        # logits of this group's rows (engine/sampling_params.py, sampling.py)
        self.sampling_params = sampling_params
        self.seqs = seqs
        self.seqs_dict: Dict[int, Sequence] = {s.seq_id: s for s in seqs}
        self.arrival_time = arrival_time
        self.max_tokens = max_tokens
        self.n = n  # sampling_params.n of the reference: sequences forked after the prompt
        self.first_scheduled_time: Optional[float] = None
        # set by the scheduler while a step that contains this group is in flight
        # (scheduler.py:874; lets the async engine schedule ahead)
        self.busy = False

    @property
    def prompt_token_ids(self) -> Tuple[int, ...]:
        return self.seqs[0].prompt_token_ids

    def maybe_set_first_scheduled_time(self, t: float) -> None:
        if self.first_scheduled_time is None:
            self.first_scheduled_time = t

    def get_max_num_running_seqs(self) -> int:
        """Upper bound of sequences that run in parallel for the rest of this request's
        lifetime (sequence.py:487-498)."""
        seqs = self.seqs
        if len(seqs) == 1 and self.n <= 1:  # the common case, answered without building lists
            return 0 if seqs[0].status > 2 else 1
        if self.n > len(seqs):
            return self.n  # still in the prompt stage: n sequences will be forked
        return self.num_unfinished_seqs()

    def get_seqs(self, status: Optional[SequenceStatus] = None) -> List[Sequence]:
        seqs = self.seqs
        if status is None:
            return seqs
        if len(seqs) == 1:
            return seqs if seqs[0].status == status else []
        return [s for s in seqs if s.status == status]

    def is_encoder_decoder(self) -> bool:
        return False

    def get_encoder_seq(self):
        return None

    def get_unfinished_seqs(self) -> List[Sequence]:
        return [s for s in self.seqs if not s.is_finished()]

    def get_finished_seqs(self) -> List[Sequence]:
        return [s for s in self.seqs if s.is_finished()]

    def update_num_computed_tokens(self, num_new_computed_tokens: int) -> None:
        for s in self.seqs:
            if not s.is_finished():
                s.data.update_num_computed_tokens(num_new_computed_tokens)

    def get_num_uncomputed_tokens(self) -> int:
        return sum(s.data.get_num_uncomputed_tokens() for s in self.seqs if not s.is_finished())

    def num_seqs(self, status: Optional[SequenceStatus] = None) -> int:
        return len(self.seqs) if status is None else len(self.get_seqs(status))

    def num_unfinished_seqs(self) -> int:
        return len(self.get_unfinished_seqs())

    def num_finished_seqs(self) -> int:
        return len(self.get_finished_seqs())

    def find(self, seq_id: int) -> Sequence:
        return self.seqs_dict[seq_id]

    def add(self, seq: Sequence) -> None:
        assert seq.seq_id not in self.seqs_dict
        self.seqs_dict[seq.seq_id] = seq
        self.seqs.append(seq)

    def remove(self, seq_id: int) -> None:
        seq = self.seqs_dict.pop(seq_id)
        self.seqs.remove(seq)

    def is_finished(self) -> bool:
        seqs = self.seqs
        if len(seqs) == 1:
            return seqs[0].status > 2
        return all(s.status > 2 for s in seqs)

    def is_prefill(self) -> bool:
        return self.seqs[0].is_prefill()  # all sequences of a group share the stage

    def __repr__(self) -> str:
        return f"SequenceGroup(request_id={self.request_id}, num_seqs={len(self.seqs)})"
