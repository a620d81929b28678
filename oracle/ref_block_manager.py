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
