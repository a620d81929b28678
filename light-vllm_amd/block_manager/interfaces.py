"""Block-space-manager interface of the decoding scheduler.

Method names, argument meaning and return values follow
light_vllm/decoding/core/interfaces.py:10-115 (AllocStatus, BlockSpaceManager ABC and its
"v1"/"v2" factory) so that the reference's DecodingScheduler can be pointed at these classes.
"""
import enum
from abc import ABC, abstractmethod
from typing import List, Sequence as GenericSequence, Tuple


class AllocStatus(enum.Enum):
    """can_allocate / can_swap_in verdicts: OK now; LATER (fits once blocks free up);
    NEVER (larger than the whole cache)."""
    OK = enum.auto()
    LATER = enum.auto()
    NEVER = enum.auto()


class BlockSpaceManager(ABC):

    @staticmethod
    def get_block_space_manager_class(version: str):
        version = version.lower()
        if version == "v1":
            from .v1 import BlockSpaceManagerV1
            return BlockSpaceManagerV1
        if version == "v2":
            from .v2 import BlockSpaceManagerV2
            return BlockSpaceManagerV2
        raise ValueError(f"Unknown version {version=}")

    @abstractmethod
    def can_allocate(self, seq_group) -> AllocStatus: ...

    @abstractmethod
    def allocate(self, seq_group) -> None: ...

    @abstractmethod
    def can_append_slots(self, seq_group, num_lookahead_slots: int) -> bool: ...

    @abstractmethod
    def append_slots(self, seq, num_lookahead_slots: int) -> List[Tuple[int, int]]: ...

    @abstractmethod
    def fork(self, parent_seq, child_seq) -> None: ...

    @abstractmethod
    def can_swap_in(self, seq_group, num_lookahead_slots: int) -> AllocStatus: ...

    @abstractmethod
    def swap_in(self, seq_group) -> List[Tuple[int, int]]: ...

    @abstractmethod
    def can_swap_out(self, seq_group) -> bool: ...

    @abstractmethod
    def swap_out(self, seq_group) -> List[Tuple[int, int]]: ...

    @abstractmethod
    def free(self, seq) -> None: ...

    @abstractmethod
    def get_block_table(self, seq) -> List[int]: ...

    @abstractmethod
    def get_num_free_gpu_blocks(self) -> int: ...

    @abstractmethod
    def get_num_free_cpu_blocks(self) -> int: ...

    @abstractmethod
    def access_all_blocks_in_seq(self, seq, access_time: float) -> None: ...

    @abstractmethod
    def get_common_computed_block_ids(self, seqs) -> GenericSequence[int]: ...

    @abstractmethod
    def mark_blocks_as_computed(self, seq_group) -> None: ...
