"""Configuration of the decode engine (the subset of light_vllm/decoding/config.py and the HF
model config that the paged-attention decode path reads)."""
from dataclasses import dataclass
from typing import Optional

import torch


@dataclass
class ModelConfig:
    """Llama / Qwen2-shaped decoder (RMSNorm, NeoX RoPE, SwiGLU, GQA): the only decode-only
    family the reference registers (light_vllm/decode_only/modelzoo/qwen2.py)."""
    hidden_size: int = 4096
    intermediate_size: int = 14336
    num_hidden_layers: int = 32
    num_attention_heads: int = 32
    num_key_value_heads: int = 8
    vocab_size: int = 128256
    rms_norm_eps: float = 1e-5
    rope_theta: float = 500000.0
    max_position_embeddings: int = 8192
    qkv_bias: bool = False          # Qwen2: True (qwen2.py:114-121); Llama: False
    dtype: torch.dtype = torch.bfloat16
    pack_weights: bool = True       # keep an MFMA-ordered copy of each projection for decode
    fuse_decode_ops: bool = True    # rope+cache write in one launch, split-K sum inside add+norm
    # the individual fused launches of a decode step (each bit-identical to what it replaces; off = A/B runs)
    rope_in_attention: bool = True  # rotary_embedding + reshape_and_cache inside the attention launch
    rope_in_attention_fp8: bool = True   # the same over an fp8 KV cache (bit-identical too; +2 % on config 5 once the
                                         # new token's cache stores left the prologue, profiles/r03_tuning.md section 10)
    swiglu_epilogue: bool = True    # silu_and_mul in the gate_up projection's epilogue
    qkv_reduce_in_rope: bool = True  # mixed steps of 33..64 rows: the QKV projection's split-K slabs are summed by the
                                     # rope + cache-write launch (lvllm_rotary_embedding_and_cache_splitk): 9 launches
                                     # per layer instead of 10, bit-identical
    argmax_epilogue: bool = True    # greedy arg-max in the lm_head projection's epilogue
    stream_gemm_max_rows: int = 256  # 65..this many rows: projections through lvllm_stream_gemm where it wins
    # decode steps of at least this many rows split o_proj's K over workgroups and leave the fp32 partials to the
    # add + norm launch (as the down projection always does); below, K is split inside a workgroup only
    o_proj_partials_min_rows: int = 33
    # "fp8": W8A8 projections (BASELINE config 5): per-tensor e4m3 weights, static per-tensor
    # activation scales calibrated once on a random batch (the reference's activation_scheme="static")
    quantization: Optional[str] = None
    # W8A8 decode steps: activations quantised once by their producer (norm launches, SwiGLU epilogue) and handed to
    # the projections as fp8 (bit-identical to quantising inside every projection; off = A/B runs)
    fp8_activations_once: bool = True

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @staticmethod
    def llama3_8b() -> "ModelConfig":
        return ModelConfig()

    @staticmethod
    def tiny(dtype=torch.bfloat16) -> "ModelConfig":
        return ModelConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                           num_attention_heads=4, num_key_value_heads=2, vocab_size=512,
                           max_position_embeddings=2048, dtype=dtype)


@dataclass
class CacheConfig:
    block_size: int = 16
    gpu_memory_utilization: float = 0.9
    swap_space_bytes: int = 4 << 30
    cache_dtype: str = "auto"
    num_gpu_blocks: Optional[int] = None
    num_cpu_blocks: Optional[int] = None
    sliding_window: Optional[int] = None
    enable_prefix_caching: bool = False
    # bytes between the end of a block and the start of the next in each K / V plane (a multiple of 16).  Blocks a power
    # of two apart (32 KiB for 8 kv heads x 128 x 16 tokens of bf16) put tile t of EVERY sequence of a freshly
    # filled cache on the same HBM channels -- sequence i's blocks start i x 2 MiB in -- and the decode attention
    # launch runs 12 % slower than over scattered blocks; 1 KiB of padding (3 % of the cache) removes that
    # (profiles/r03_tuning.md section 9).  The ops take the stride from the tensors; 0 = the reference's dense layout;
    # None = 1/32 of a block's bytes in one plane, in multiples of 256 (1 KiB for the shapes above, 512 bytes for their
    # fp8 cache, where 512 measured as well as 1 024).
    block_pad_bytes: Optional[int] = None


@dataclass
class SchedulerConfig:
    max_num_batched_tokens: int = 8192
    max_num_seqs: int = 256
    max_model_len: int = 8192
    use_v2_block_manager: bool = False
    scheduling: str = "sync"          # sync | simple_async | async | double_buffer
    max_num_on_the_fly: Optional[int] = None  # None: 3 for "double_buffer", else 2 (decoding/config.py:149-155)
    preemption_mode: Optional[str] = None  # None | "swap" | "recompute"
    chunked_prefill_enabled: bool = False  # decoding/config.py: prompts are cut to the token budget
    # slots reserved per decoding sequence beyond its known tokens (decoding/config.py:127,160; the
    # reference plumbs them through its scheduler for lookahead / speculative decoding, scheduler.py:1095-1106)
    num_lookahead_slots: int = 0
    # > 1: a decode step runs this many model steps back to back on the device (advance_step between
    # them, csrc/prepare_inputs/advance_step.cu) and returns to the host once; needs the lookahead slots
    # for the tokens in between, hence the v2 block manager (v1 refuses lookahead, block_manager_v1.py:353-355)
    num_scheduler_steps: int = 1
    # workgroups of a decode GEMM (lvllm_set_tuning "gemm_workgroups"): None = 256 for one stream, 128 when
    # steps run side by side on several streams (each GEMM then leaves CUs to the other steps' kernels)
    gemm_workgroups: Optional[int] = None
    fast_decode_inputs: bool = True   # decode / mixed steps staged straight from the scheduler's metadata
    poll_completion: bool = True      # the engine thread polls the steps' events (False: per-slot waiter threads)

    def __post_init__(self) -> None:
        if self.num_scheduler_steps < 1:
            raise ValueError(f"num_scheduler_steps {self.num_scheduler_steps} must be positive")
        if self.num_scheduler_steps > 1:
            self.num_lookahead_slots = max(self.num_lookahead_slots, self.num_scheduler_steps - 1)
        if self.num_lookahead_slots < 0:
            raise ValueError(f"num_lookahead_slots ({self.num_lookahead_slots}) must be greater than or equal to 0.")
        if self.num_lookahead_slots > 0 and not self.use_v2_block_manager:
            raise ValueError("lookahead slots (multi-step decode) need use_v2_block_manager=True: "
                             "BlockSpaceManagerV1 does not support lookahead allocation")
        if self.num_lookahead_slots > 0 and self.preemption_mode == "swap":
            # the reference's own swap_out raises on a table that ends in reserved, still empty lookahead blocks
            # (block/common.py:207 via naive_block.py:335; tests/bm_driver.py): no trace to replay, so no claim
            raise ValueError("lookahead slots (multi-step decode) with preemption_mode='swap' are not supported: "
                             "preempted sequences are recomputed")
        if self.max_num_on_the_fly is None:
            self.max_num_on_the_fly = 3 if self.scheduling == "double_buffer" else 2
        if self.max_num_on_the_fly < 1:  # the reference insists on >= 2 (:190-193); one step in flight is
            raise ValueError(f"max_num_on_the_fly {self.max_num_on_the_fly} must be positive")  # allowed here
