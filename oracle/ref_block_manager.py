"""Imports the reference's pure-Python KV-cache block manager from /root/reference through a
stub loader.  TEST INFRASTRUCTURE ONLY, dev container only (the reference does not travel):
used by oracle/make_golden.py to record golden traces into tests/golden/.

Plain `import light_vllm` fails here with ordinary Python errors (py3.10 lacks
typing.assert_never; light_vllm/__init__.py eagerly imports the engine, which needs msgspec
and pydantic, neither installed).  The harness therefore (reference untouched):
  * aliases typing.assert_never to typing_extensions.assert_never,
  * pre-registers `light_vllm` as an empty namespace package rooted at the reference tree
    (so its __init__ is skipped),
  * registers permissive stubs for `msgspec` and `pydantic`,
then imports the block-manager and sequence modules themselves, unmodified.
"""
import os
import sys
import types
import typing

REFERENCE = "/root/reference"


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE, "light_vllm", "decoding", "core"))


def load():
    """Returns a namespace with BlockSpaceManagerV1/V2, Sequence, SequenceGroup, SequenceStatus,
    Logprob, TextOnlyInputs, AllocStatus of the reference."""
    if not available():
        raise RuntimeError("/root/reference is not present (GPU box?): use the committed golden traces")
    import typing_extensions
    if not hasattr(typing, "assert_never"):
        typing.assert_never = typing_extensions.assert_never
    if "light_vllm" not in sys.modules:
        pkg = types.ModuleType("light_vllm")
        pkg.__path__ = [os.path.join(REFERENCE, "light_vllm")]
        sys.modules["light_vllm"] = pkg
    if "msgspec" not in sys.modules:
        m = types.ModuleType("msgspec")

        class Struct:
            def __init_subclass__(cls, **kwargs):
                super().__init_subclass__()

            def __init__(self, **kw):
                for k, v in kw.items():
                    setattr(self, k, v)

        m.Struct = Struct
        m.field = lambda *a, **k: k.get("default", None)
        m.Meta = lambda *a, **k: None
        sys.modules["msgspec"] = m
    if "pydantic" not in sys.modules:
        p = types.ModuleType("pydantic")
        p.BaseModel = object
        p.Field = lambda *a, **k: None
        sys.modules["pydantic"] = p
    from light_vllm.core.schema.engine_io import TextOnlyInputs
    from light_vllm.decoding.core.block_manager_v1 import BlockSpaceManagerV1
    from light_vllm.decoding.core.block_manager_v2 import BlockSpaceManagerV2
    from light_vllm.decoding.core.interfaces import AllocStatus
    from light_vllm.decoding.schema.sequence import (Logprob, Sequence, SequenceGroup,
                                                     SequenceStatus)
    return types.SimpleNamespace(
        BlockSpaceManagerV1=BlockSpaceManagerV1, BlockSpaceManagerV2=BlockSpaceManagerV2,
        Sequence=Sequence, SequenceGroup=SequenceGroup, SequenceStatus=SequenceStatus,
        Logprob=Logprob, TextOnlyInputs=TextOnlyInputs, AllocStatus=AllocStatus)


def load_input_builder():
    """The reference's per-step input builder and attention-plugin classes (SURVEY a13 / boundary B):
    ModelInputForGPUBuilder (decoding/processor/model_input_builder.py:105-378), the flash backend's
    metadata builder (decoding/backends/attention/backends/flash_attn.py:208-365), the slot-mapping
    helpers (backends/utils.py:31-75), the abstract plugin classes (backends/abstract.py:15-166) and the
    sequence metadata they consume.  flash_attn.py imports two callables from the third-party package
    `vllm_flash_attn` (requirements.txt:32, absent here) at module level; neither is reached by the
    metadata builder, so the harness registers an empty module of that name carrying the two names.
    The module-level `pin_memory` flags (a GPU probe) are set to False: pinning needs a GPU."""
    ns = load()
    if "vllm_flash_attn" not in sys.modules:
        m = types.ModuleType("vllm_flash_attn")
        m.flash_attn_varlen_func = m.flash_attn_with_kvcache = None
        sys.modules["vllm_flash_attn"] = m
    from light_vllm.decoding.backends.attention.backends import abstract, flash_attn, utils
    from light_vllm.decoding.processor import model_input_builder
    from light_vllm.decoding.schema.sequence import SequenceData, SequenceGroupMetadata
    flash_attn.pin_memory = False
    model_input_builder.pin_memory = False
    ns.abstract, ns.flash_attn, ns.attn_utils = abstract, flash_attn, utils
    ns.model_input_builder = model_input_builder
    ns.SequenceData, ns.SequenceGroupMetadata = SequenceData, SequenceGroupMetadata
    return ns


def ref_build_model_input(ns, scenario, attn_backend=None):
    """Runs one tests/ib_driver.py scenario through the reference's ModelInputForGPUBuilder with the
    given attention backend class (default: the reference's own flash backend) and returns its
    DecodingModelInputForGPU."""
    backend = attn_backend or ns.flash_attn.DecodeOnlyFlashAttentionBackend
    window = scenario["sliding_window"]
    model_config = types.SimpleNamespace(get_sliding_window=lambda: window)
    scheduler_config = types.SimpleNamespace(use_v2_block_manager=scenario["use_v2_block_manager"],
                                             chunked_prefill_enabled=scenario["chunked_prefill_enabled"])
    cache_config = types.SimpleNamespace(block_size=scenario["block_size"])
    builder = ns.model_input_builder.ModelInputForGPUBuilder(
        model_config=model_config, scheduler_config=scheduler_config, cache_config=cache_config,
        attn_backend=backend, device="cpu")
    for g in scenario["groups"]:
        seq_data = {}
        for sid, prompt, out, comp in zip(g["seq_ids"], g["prompts"], g["outputs"], g["num_computed"]):
            d = ns.SequenceData(prompt, out)
            d.update_num_computed_tokens(comp)
            seq_data[sid] = d
        builder.add_seq_group(ns.SequenceGroupMetadata(
            request_id=g["request_id"], is_prompt=g["is_prompt"], seq_data=seq_data, sampling_params=None,
            block_tables=dict(zip(g["seq_ids"], g["block_tables"])), do_sample=g["do_sample"],
            token_chunk_size=g["token_chunk_size"], computed_block_nums=g["computed_block_nums"] or None))
    return builder.build()
