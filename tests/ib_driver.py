"""Seeded scenarios for the per-step input arrays: token ids, positions, slot mapping,
block tables, sequence lengths, query/seq start offsets.

A scenario is plain data -- a configuration and a list of scheduled sequence groups (prompt and output
tokens, computed-token count, block table, chunk size, computed prefix blocks) in the order the
reference's scheduler emits them (prompts first, then decodes: decoding/scheduler.py:856-930).
`oracle/make_golden.py input_builder` materialises each scenario with the REFERENCE's own classes
(SequenceData, SequenceGroupMetadata, ModelInputForGPUBuilder + DecodeOnlyFlashAttentionMetadataBuilder:
model_input_builder.py:105-378, flash_attn.py:208-365, backends/utils.py:31-75) and records what they
produce into tests/golden/input_builder.json; tests/test_input_builder.py materialises the same
scenarios with this package's classes and compares every array element by element.

TEST INFRASTRUCTURE ONLY.
"""
import random
from typing import Dict, List, Optional

CONFIGS = [
    # name, block_size, sliding_window, use_v2_block_manager, chunked_prefill, prefix_hits
    ("plain_bs16", 16, None, False, False, False),
    ("plain_bs8_v2", 8, None, True, False, False),
    ("plain_bs32", 32, None, False, False, False),
    ("chunked_bs16", 16, None, False, True, False),
    ("chunked_bs32_v2", 32, None, True, True, False),
    ("prefix_bs16", 16, None, False, False, True),
    ("prefix_bs8_v2", 8, None, True, False, True),
    ("window_v1_bs16", 16, 48, False, False, False),
    ("window_v2_bs16", 16, 40, True, False, False),
    ("window_v2_bs8_chunked", 8, 20, True, True, False),
]
STEPS_PER_CONFIG = 12


def _table(rng: random.Random, n_blocks: int, pool: int, ring: Optional[int] = None, nulls: int = 0) -> List[int]:
    """A block table of n_blocks entries: distinct random physical ids; with `ring` the v1 sliding-window
    reuse (block i >= ring repeats block i % ring, block_manager_v1.py:236-240); with `nulls` leading
    entries replaced by one shared null block id (v2, block_table.py:133-143)."""
    ids = rng.sample(range(pool), min(n_blocks, pool))
    while len(ids) < n_blocks:
        ids.append(rng.randrange(pool))
    if ring:
        ids = [ids[i % ring] for i in range(n_blocks)]
    if nulls:
        ids = [pool] * min(nulls, n_blocks) + ids[min(nulls, n_blocks):]
    return ids


def make_scenarios(seed: int = 1234) -> List[Dict]:
    rng = random.Random(seed)
    scenarios: List[Dict] = []
    seq_counter = 0
    for name, bs, window, use_v2, chunked, prefix in CONFIGS:
        for step in range(STEPS_PER_CONFIG):
            groups: List[Dict] = []
            decode_only = step % 4 == 3
            n_prompts = 0 if decode_only else rng.randint(0 if step % 2 else 1, 3)
            n_decodes = rng.randint(1 if n_prompts == 0 else 0, 5)
            if not chunked and n_prompts > 0 and step % 3 != 0:
                n_decodes = 0  # the default policy never mixes prompts and decodes
            pool = 4096
            for _ in range(n_prompts):
                plen = rng.choice([1, 2, bs - 1, bs, bs + 1, 3 * bs, 3 * bs + 5, rng.randint(1, 6 * bs)])
                prompt = [rng.randrange(32000) for _ in range(plen)]
                outputs: List[int] = []
                if rng.random() < 0.2:  # preempted by recompute: the "prompt" now covers its outputs too
                    outputs = [rng.randrange(32000) for _ in range(rng.randint(1, bs + 2))]
                    plen += len(outputs)
                computed, chunk, do_sample, cbn = 0, None, True, []
                if chunked:
                    computed = rng.choice([0, 0, rng.randint(0, plen - 1)])
                    chunk = rng.randint(1, plen - computed)
                    do_sample = computed + chunk == plen
                n_blocks = (plen + bs - 1) // bs
                ring = None
                nulls = 0
                if window is not None and not use_v2:
                    ring = (window + bs - 1) // bs
                table = _table(rng, n_blocks, pool, ring, nulls)
                if prefix and plen > bs and rng.random() < 0.7:
                    cbn = table[:rng.randint(1, (plen - 1) // bs)]
                groups.append(dict(request_id=f"p{seq_counter}", is_prompt=True, seq_ids=[seq_counter],
                                   prompts=[prompt], outputs=[outputs], num_computed=[computed],
                                   block_tables=[table], token_chunk_size=chunk, computed_block_nums=cbn,
                                   do_sample=do_sample))
                seq_counter += 1
            for _ in range(n_decodes):
                n_seqs = 1 if rng.random() < 0.8 else rng.randint(2, 3)  # a forked group decodes n sequences
                plen = rng.randint(1, 5 * bs)
                prompt = [rng.randrange(32000) for _ in range(plen)]
                ids, prompts, outs, comps, tables = [], [], [], [], []
                for _s in range(n_seqs):
                    olen = rng.choice([1, 2, bs, rng.randint(1, 4 * bs)])
                    total = plen + olen
                    n_blocks = (total + bs - 1) // bs
                    ring = nulls = None
                    if window is not None and not use_v2:
                        ring = (window + bs - 1) // bs
                    if window is not None and use_v2:
                        nulls = max(0, (total - 1 - window) // bs)
                    ids.append(seq_counter)
                    seq_counter += 1
                    prompts.append(prompt)
                    outs.append([rng.randrange(32000) for _ in range(olen)])
                    comps.append(total - 1)
                    tables.append(_table(rng, n_blocks, pool, ring, nulls or 0))
                groups.append(dict(request_id=f"d{ids[0]}", is_prompt=False, seq_ids=ids, prompts=prompts,
                                   outputs=outs, num_computed=comps, block_tables=tables,
                                   token_chunk_size=1, computed_block_nums=[], do_sample=True))
            scenarios.append(dict(name=f"{name}_{step}", block_size=bs, sliding_window=window,
                                  use_v2_block_manager=use_v2, chunked_prefill_enabled=chunked,
                                  groups=groups))
    return scenarios


ARRAY_FIELDS = ("input_tokens", "input_positions", "slot_mapping", "block_tables", "seq_lens_tensor",
                "query_start_loc", "seq_start_loc", "context_lens_tensor")
SCALAR_FIELDS = ("num_prefills", "num_prefill_tokens", "num_decode_tokens", "max_query_len",
                 "max_prefill_seq_len", "max_decode_seq_len")
LIST_FIELDS = ("seq_lens", "query_lens")


def record(model_input) -> Dict:
    """The comparable content of a built model input (either side's classes: same attribute names)."""
    md = model_input.attn_metadata
    out = {"input_tokens": model_input.input_tokens.tolist(),
           "input_positions": model_input.input_positions.tolist(),
           "seq_lens": list(model_input.seq_lens), "query_lens": list(model_input.query_lens)}
    for f in ARRAY_FIELDS[2:]:
        out[f] = getattr(md, f).tolist()
    for f in SCALAR_FIELDS:
        out[f] = int(getattr(md, f))
    out["dtypes"] = {f: str(getattr(md, f).dtype) for f in ARRAY_FIELDS[2:]}
    out["dtypes"]["input_tokens"] = str(model_input.input_tokens.dtype)
    out["dtypes"]["input_positions"] = str(model_input.input_positions.dtype)
    return out
