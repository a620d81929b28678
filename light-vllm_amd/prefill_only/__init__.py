"""The prefill-only (encode-only) workflow of the reference (light_vllm/prefill_only/, BASELINE
config 4): requests are whole prompts, one forward pass each, no KV cache.  Scheduler, schemas and
metadata mirror the reference's classes; attention is this package's HIP varlen kernel.  One GPU:
`engine.PrefillOnlyEngine`; several: `dp_executor.DataParallelEncodeEngine`, one scheduler feeding a queue that N
worker processes (one per GPU) drain -- the reference's data-parallel executor
(prefill_only/executor/gpu_data_parallelism_executor.py) with processes for its threads."""
from .config import PrefillOnlySchedulerConfig  # noqa: F401
from .scheduler import (PrefillOnlyRequestOutput, PrefillOnlyScheduler, PrefillOnlySchedulerOutput,  # noqa: F401
                        PrefillOnlySchedulingBudget, Request, SchedulableRequest)
