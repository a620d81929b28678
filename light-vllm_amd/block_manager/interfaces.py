"""What the decoding scheduler asks of a KV-cache block manager.

The method names, argument meaning and return values are those of
light_vllm/decoding/core/interfaces.py:10-115 (AllocStatus, the BlockSpaceManager ABC and its
"v1"/"v2" factory), so the reference's DecodingScheduler can be pointed at the classes of this
package.  Block numbers are indices into the paged tensors of engine/cache_engine.py; every list
of pairs is (source block, destination block).
"""
import enum
from abc import ABC, abstractmethod
from importlib import import_module
from typing import List, Sequence as GenericSequence, Tuple

BlockPairs = List[Tuple[int, int]]

# version string -> (module of this package, class name); imported on demand so that the v1
# manager does not drag in the v2 allocator stack and vice versa
_MANAGERS = {"v1": (".v1", "BlockSpaceManagerV1"), "v2": (".v2", "BlockSpaceManagerV2")}


class AllocStatus(enum.Enum):
    """Verdict of can_allocate / can_swap_in."""
    OK = enum.auto()     # fits now
    LATER = enum.auto()  # fits once running groups release blocks
    NEVER = enum.auto()  # larger than the whole cache: the scheduler drops the request


class BlockSpaceManager(ABC):

    @staticmethod
    def get_block_space_manager_class(version: str):
        try:
            module, name = _MANAGERS[version.lower()]
        except KeyError:
            raise ValueError(f"Unknown version {version=}") from None
        return getattr(import_module(module, __package__), name)

    # ---- admission of a waiting group (prompt blocks) ----
    @abstractmethod
    def can_allocate(self, seq_group) -> AllocStatus:
        ...

    @abstractmethod
    def allocate(self, seq_group) -> None:
        ...

    # ---- growth of a running group, one decode step (+ lookahead) at a time ----
    @abstractmethod
    def can_append_slots(self, seq_group, num_lookahead_slots: int) -> bool:
        ...

    @abstractmethod
    def append_slots(self, seq, num_lookahead_slots: int) -> BlockPairs:
        """Returns the copy-on-write copies the step must do before it writes."""

    @abstractmethod
    def fork(self, parent_seq, child_seq) -> None:
        ...

    # ---- preemption by swapping: device <-> host ----
    @abstractmethod
    def can_swap_out(self, seq_group) -> bool:
        ...

    @abstractmethod
    def swap_out(self, seq_group) -> BlockPairs:
        ...

    @abstractmethod
    def can_swap_in(self, seq_group, num_lookahead_slots: int) -> AllocStatus:
        ...

    @abstractmethod
    def swap_in(self, seq_group) -> BlockPairs:
        ...

    # ---- release and queries ----
    @abstractmethod
    def free(self, seq) -> None:
        ...

    @abstractmethod
    def get_block_table(self, seq) -> List[int]:
        ...

    @abstractmethod
    def get_num_free_gpu_blocks(self) -> int:
        ...

    @abstractmethod
    def get_num_free_cpu_blocks(self) -> int:
        ...

    # ---- prefix caching ----
    @abstractmethod
    def access_all_blocks_in_seq(self, seq, access_time: float) -> None:
        ...

    @abstractmethod
    def get_common_computed_block_ids(self, seqs) -> GenericSequence[int]:
        ...

    @abstractmethod
    def mark_blocks_as_computed(self, seq_group) -> None:
        ...
