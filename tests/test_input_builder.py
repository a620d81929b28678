"""SURVEY a13 + boundary B, pinned to the reference.

tests/golden/input_builder.json holds what the REFERENCE's ModelInputForGPUBuilder
(decoding/processor/model_input_builder.py:105-378) and the metadata builder of its wired attention backend
(decoding/backends/attention/backends/flash_attn.py:208-365; slot mapping backends/utils.py:31-75) produce
for 120 seeded steps (tests/ib_driver.py: prompts, chunks, decodes, forked groups, recomputed prompts, prefix
hits, v1/v2 sliding windows, block sizes 8/16/32).  Everything is integer work: the comparison is exact,
dtypes included.

CPU only.  The tests marked `needs_reference` also run the reference's classes live (dev container);
they skip on the GPU box, where /root/reference does not exist.
"""
import inspect
import json
import os

import numpy as np
import pytest
import torch

import ib_driver
from light_vllm_amd.attention.backend import (PagedAttnBackend, PagedAttnImpl, PagedAttnMetadata,
                                              PagedAttnMetadataBuilder)
from light_vllm_amd.engine.config import CacheConfig, SchedulerConfig
from light_vllm_amd.engine.input_builder import DecodeStepArrays, MixedStepArrays, ModelInputBuilder
from light_vllm_amd.engine.scheduler import SequenceGroupMetadata
from light_vllm_amd.engine.sequence import SequenceData

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "input_builder.json")
with open(GOLDEN) as f:
    SCENARIOS = json.load(f)["scenarios"]

needs_reference = pytest.mark.skipif(not os.path.isdir("/root/reference/light_vllm"),
                                     reason="the reference tree is only present in the dev container")


def our_metas(sc):
    metas = []
    for g in sc["groups"]:
        seq_data = {}
        for sid, prompt, out, comp in zip(g["seq_ids"], g["prompts"], g["outputs"], g["num_computed"]):
            d = SequenceData(list(prompt), list(out))
            d.update_num_computed_tokens(comp)
            seq_data[sid] = d
        chunk = g["token_chunk_size"]
        if chunk is None:  # sequence.py:633-637 of the reference
            chunk = next(iter(seq_data.values())).get_len() if g["is_prompt"] else 1
        metas.append(SequenceGroupMetadata(
            request_id=g["request_id"], is_prompt=g["is_prompt"], seq_data=seq_data,
            block_tables=dict(zip(g["seq_ids"], g["block_tables"])), do_sample=g["do_sample"],
            token_chunk_size=chunk, computed_block_nums=list(g["computed_block_nums"])))
    return metas


def our_builder(sc, prompt_block_tables):
    cc = CacheConfig(block_size=sc["block_size"], num_gpu_blocks=8192, num_cpu_blocks=0,
                     sliding_window=sc["sliding_window"])
    scfg = SchedulerConfig(use_v2_block_manager=sc["use_v2_block_manager"],
                           chunked_prefill_enabled=sc["chunked_prefill_enabled"])
    b = ModelInputBuilder(scfg, cc, PagedAttnBackend(), sc["sliding_window"],
                          chunked_prefill_enabled=sc["chunked_prefill_enabled"])
    b.prompt_block_tables = prompt_block_tables
    return b


def assert_same(got, exp, name, skip=()):
    for f in ib_driver.ARRAY_FIELDS + ib_driver.SCALAR_FIELDS + ib_driver.LIST_FIELDS:
        if f in skip:
            continue
        assert got[f] == exp[f], f"{name}: {f} differs\n got {got[f]}\n exp {exp[f]}"
    for f, dt in exp["dtypes"].items():
        if f not in skip:
            assert got["dtypes"][f] == dt, f"{name}: dtype of {f}: {got['dtypes'][f]} vs {dt}"


@pytest.mark.parametrize("sc", SCENARIOS, ids=[s["name"] for s in SCENARIOS])
def test_input_arrays_equal_the_reference(sc):
    """Reference semantics (plain prompts carry no block table, flash_attn.py:262-273): every field."""
    mi = our_builder(sc, prompt_block_tables=False).prepare_model_input(our_metas(sc))
    assert_same(ib_driver.record(mi), sc["expect"], sc["name"])


@pytest.mark.parametrize("sc", SCENARIOS, ids=[s["name"] for s in SCENARIOS])
def test_default_builder_differs_only_in_prompt_block_tables(sc):
    """The shipped default hands plain prompts their block table too (the HIP prefill kernel reads the
    chunk back from the paged cache): every other field, and the decode rows of the block tables,
    equal the reference."""
    mi = our_builder(sc, prompt_block_tables=sc["sliding_window"] is None).prepare_model_input(our_metas(sc))
    got, exp = ib_driver.record(mi), sc["expect"]
    assert_same(got, exp, sc["name"], skip=("block_tables",))
    n = exp["num_prefills"]
    gt, et = got["block_tables"], exp["block_tables"]
    for row_g, row_e in zip(gt[n:], et[n:]):  # decode rows: same entries (width may be padded further)
        w = len(row_e)
        assert row_g[:w] == row_e and not any(row_g[w:])
    for i, (row_g, row_e) in enumerate(zip(gt[:n], et[:n])):
        if any(row_e):  # the reference gave this prompt a table (prefix hit / chunked): same entries
            k = max(j for j, b in enumerate(row_e) if b) + 1
            assert row_g[:k] == row_e[:k]


def test_decode_step_arrays_equal_the_reference():
    """The staging fast path of captured decode steps (DecodeStepArrays) against the reference's arrays."""
    n = 0
    for sc in SCENARIOS:
        metas = our_metas(sc)
        if not DecodeStepArrays.eligible(metas, True, sc["sliding_window"]):
            continue
        exp = sc["expect"]
        B = len(metas)
        W = len(exp["block_tables"][0]) + 2
        arr = DecodeStepArrays(B + 3, W, sc["block_size"])
        ids = arr.fill(metas)
        assert ids == [g["seq_ids"][0] for g in sc["groups"]]
        assert arr.input_ids[:B].tolist() == exp["input_tokens"]
        assert arr.positions[:B].tolist() == exp["input_positions"]
        assert arr.slot_mapping[:B].tolist() == exp["slot_mapping"]
        assert arr.seq_lens[:B].tolist() == exp["seq_lens_tensor"]
        for i, row in enumerate(exp["block_tables"]):
            k = len(sc["groups"][i]["block_tables"][0])
            assert arr.block_tables[i, :k].tolist() == row[:k]
        assert (arr.slot_mapping[B:] == -1).all() and (arr.seq_lens[B:] == 0).all()
        n += 1
    assert n >= 8


def test_mixed_step_arrays_equal_the_reference():
    """The staging fast path of captured mixed steps (MixedStepArrays) against the reference's arrays."""
    n = 0
    for sc in SCENARIOS:
        metas = our_metas(sc)
        if not sc["chunked_prefill_enabled"] or not MixedStepArrays.eligible(metas, True, sc["sliding_window"]):
            continue
        exp = sc["expect"]
        T, S = len(exp["input_tokens"]), len(metas)
        W = max(len(g["block_tables"][0]) for g in sc["groups"])
        arr = MixedStepArrays(T + 5, S + 2, W + 1, sc["block_size"])
        filled = arr.fill(metas)
        assert filled is not None
        ids, ntok = filled
        assert ntok == T
        assert arr.input_ids[:T].tolist() == exp["input_tokens"]
        assert arr.positions[:T].tolist() == exp["input_positions"]
        assert arr.slot_mapping[:T].tolist() == exp["slot_mapping"]
        assert arr.seq_lens[:S].tolist() == exp["seq_lens_tensor"]
        assert arr.query_start_loc[:S + 1].tolist() == exp["query_start_loc"]
        assert ids == [g["seq_ids"][0] for g in sc["groups"] if g["do_sample"]]
        n += 1
    assert n >= 8


# ---------------- live against the reference's classes (dev container) ----------------

@pytest.fixture(scope="module")
def ref_ns():
    from oracle import ref_block_manager
    return ref_block_manager.load_input_builder()


@needs_reference
def test_golden_file_is_what_the_reference_produces_now(ref_ns):
    from oracle import ref_block_manager
    fresh = ib_driver.make_scenarios()
    assert len(fresh) == len(SCENARIOS)
    for sc, gold in zip(fresh, SCENARIOS):
        assert sc["groups"] == gold["groups"]
        assert ib_driver.record(ref_block_manager.ref_build_model_input(ref_ns, sc)) == gold["expect"]


@needs_reference
def test_reference_input_builder_drives_our_attention_backend(ref_ns):
    """Boundary B the way the reference uses it: ITS ModelInputForGPUBuilder constructs OUR metadata builder
    through `attn_backend.make_metadata_builder(weakref.proxy(self))` (model_input_builder.py:199-200) and
    calls build() over its own inter_data_list.  Output: our PagedAttnMetadata with the reference's values."""
    from oracle import ref_block_manager

    class RefSemantics(PagedAttnBackend):  # plain prompts without a table, as the flash backend does
        @classmethod
        def make_metadata_builder(cls, input_builder):
            b = PagedAttnMetadataBuilder(input_builder)
            b.prompt_block_tables = False
            return b

    for sc in SCENARIOS:
        mi = ref_block_manager.ref_build_model_input(ref_ns, sc, attn_backend=RefSemantics)
        assert isinstance(mi.attn_metadata, PagedAttnMetadata)
        assert_same(ib_driver.record(mi), sc["expect"], sc["name"])
        md = mi.attn_metadata
        # the split views the Impl consumes (abstract.py:91-103)
        p, d = md.prefill_metadata, md.decode_metadata
        assert (p is None) == (md.num_prefills == 0) and (d is None) == (md.num_decode_tokens == 0)
        if p is not None and d is not None:
            assert p.slot_mapping.numel() == md.num_prefill_tokens
            assert d.seq_lens_tensor.tolist() == sc["expect"]["seq_lens_tensor"][md.num_prefills:]


def _params(fn):
    return [(p.name, p.kind, p.default) for p in inspect.signature(fn).parameters.values()]


@needs_reference
def test_plugin_classes_match_the_reference_abstract_interface(ref_ns):
    """Every abstract method of DecodeOnlyAttentionBackend / Impl / Metadata / MetadataBuilder
    (backends/abstract.py:15-166) exists on the PagedAttn* classes with the same parameter names, order,
    kinds and defaults; extra parameters of ours must be optional and trail."""
    A = ref_ns.abstract
    pairs = [(A.DecodeOnlyAttentionBackend, PagedAttnBackend), (A.DecodeOnlyAttentionImpl, PagedAttnImpl),
             (A.DecodeOnlyAttentionMetadataBuilder, PagedAttnMetadataBuilder),
             (A.DecodeOnlyAttentionMetadata, PagedAttnMetadata)]
    checked = 0
    for ref_cls, ours in pairs:
        names = set(getattr(ref_cls, "__abstractmethods__", ()))
        names |= {n for n in ("make_metadata", "make_metadata_builder") if hasattr(ref_cls, n)}
        for name in sorted(names):
            assert hasattr(ours, name), f"{ours.__name__} lacks {name}"
            ref_attr = inspect.getattr_static(ref_cls, name)
            our_attr = inspect.getattr_static(ours, name)
            if isinstance(ref_attr, property):
                assert isinstance(our_attr, property), f"{ours.__name__}.{name} must be a property"
                checked += 1
                continue
            assert type(ref_attr) is type(our_attr) or not isinstance(ref_attr, (staticmethod, classmethod)), \
                f"{ours.__name__}.{name}: {type(our_attr).__name__} vs {type(ref_attr).__name__}"
            rp, op = _params(getattr(ref_cls, name)), _params(getattr(ours, name))
            if name == "forward":  # attn_type's default is the reference's enum member; ours accepts None
                rp = [(n, k, d if n != "attn_type" else None) for n, k, d in rp]
            assert op[:len(rp)] == rp, f"{ours.__name__}.{name}: {op} vs {rp}"
            for n, k, d in op[len(rp):]:
                assert d is not inspect.Parameter.empty, f"{ours.__name__}.{name}: extra parameter {n} needs a default"
            checked += 1
    assert checked >= 14
    # the metadata dataclass: the reference's required fields first, in its order (abstract.py:75-89;
    # flash_attn.py:76-137 adds the rest)
    import dataclasses
    ref_fields = [f.name for f in dataclasses.fields(ref_ns.flash_attn.DecodeOnlyFlashAttentionMetadata)]
    our_fields = [f.name for f in dataclasses.fields(PagedAttnMetadata)]
    assert our_fields[:len(ref_fields)] == ref_fields, (our_fields, ref_fields)


@needs_reference
def test_slot_mapping_helpers_equal_the_reference(ref_ns):
    from light_vllm_amd.attention import backend as ours
    U = ref_ns.attn_utils
    rng = np.random.default_rng(5)
    for _ in range(300):
        bs = int(rng.choice([8, 16, 32]))
        seq_len = int(rng.integers(1, 200))
        ctx = int(rng.integers(0, seq_len))
        window = None if rng.random() < 0.5 else int(rng.integers(4, 64))
        is_prompt = bool(rng.random() < 0.6)
        use_v2 = bool(rng.random() < 0.5) or (window is not None and ctx > 0)
        qlen = seq_len - ctx if is_prompt else 1
        table = {7: rng.integers(0, 500, size=(seq_len + bs - 1) // bs).tolist()}
        a = U.compute_slot_mapping_start_idx(is_prompt, qlen, ctx, window, use_v2)
        b = ours.compute_slot_mapping_start_idx(is_prompt, qlen, ctx, window, use_v2)
        assert a == b
        for profile in (False, True):
            ra, rb = [], []
            U.compute_slot_mapping(profile, ra, 7, seq_len, ctx, a, bs, table)
            ours.compute_slot_mapping(profile, rb, 7, seq_len, ctx, a, bs, table)
            assert ra == rb
    assert U.is_block_tables_empty(None) and ours.is_block_tables_empty(None)
    assert U.is_block_tables_empty({1: None}) and ours.is_block_tables_empty({1: None})
    assert not U.is_block_tables_empty({1: [3]}) and not ours.is_block_tables_empty({1: [3]})
    assert U.PAD_SLOT_ID == ours.PAD_SLOT_ID


@needs_reference
def test_rocm_platform_has_the_reference_platform_interface(ref_ns):
    """RocmPlatform answers every method of light_vllm/platforms/interface.py:31-106 with the same
    parameters, and installing it replaces the UnspecifiedPlatform the reference picks on ROCm torch -- in the
    modules that already bound the name too (gpu_worker's dtype check indexes the capability)."""
    import light_vllm.platforms as P
    from light_vllm.platforms.interface import Platform
    from light_vllm.decoding.worker import gpu_worker
    from light_vllm_amd import platforms as ours
    for name, member in inspect.getmembers(Platform, predicate=lambda m: inspect.isfunction(m) or inspect.ismethod(m)):
        if name.startswith("__"):
            continue
        assert hasattr(ours.RocmPlatform, name), name
        assert _params(getattr(ours.RocmPlatform, name)) == _params(member), name
        assert isinstance(inspect.getattr_static(ours.RocmPlatform, name), classmethod) == \
            isinstance(inspect.getattr_static(Platform, name), classmethod), name
    assert [m.name for m in ours.PlatformEnum] == [m.name for m in P.PlatformEnum]
    assert ours.DeviceCapability(9, 5).to_int() == 95 and ours.DeviceCapability._fields == ("major", "minor")
    before = P.current_platform
    assert before.get_device_capability() is None  # the failure mode of F8
    try:
        plat = ours.install()
        assert P.current_platform is plat and gpu_worker.current_platform is plat
        assert plat.is_rocm() and plat.is_cuda_alike() and not plat.is_cuda()
    finally:
        P.current_platform = before
        gpu_worker.current_platform = before


@pytest.mark.parametrize("sc", SCENARIOS, ids=[s["name"] for s in SCENARIOS])
def test_token_rows_name_the_rows_the_builder_lays_out(sc):
    """`token_rows` (the log-probability path's map from a sequence to its rows of the step's token batch) against the
    builder itself on every recorded scenario: rows are consecutive, cover the batch, and hold the sequence's tokens
    [context, end) -- chunks, prefix-cache hits and decode rows alike."""
    from light_vllm_amd.engine.input_builder import token_rows
    metas = our_metas(sc)
    b = our_builder(sc, prompt_block_tables=False)
    mi = b.prepare_model_input(metas)
    rows = token_rows(metas, b.block_size, sc["sliding_window"])
    toks = mi.input_tokens.tolist() if mi.input_tokens is not None else []
    pos = mi.input_positions.tolist() if mi.input_positions is not None else []
    at = 0
    for sid, m, row, n, ctx, end in rows:
        assert row == at and n == end - ctx
        data = m.seq_data[sid]
        assert toks[row:row + n] == list(data.get_token_ids()[ctx:end])
        assert pos[row:row + n] == list(range(ctx, end))
        at += n
    assert at == len(toks)
