"""The JSON line bench.py printed on the GPU box (committed as profiles/r01_bench_line.json) carries every
field of the driver's contract, with the types and internal consistency the contract states."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_recorded_bench_line_has_the_contract_fields():
    with open(os.path.join(ROOT, "profiles", "r01_bench_line.json")) as f:
        d = json.loads(f.read())
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        base = json.load(f)
    assert d["metric"] == base["metric"]
    for k, t in (("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str)):
        assert isinstance(d[k], t), (k, d[k])
    assert d["vs_baseline"] is None and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert "workload" in d["config"] and "model" not in d["config"]
    # value = tokens of all ranks / time: 32 sequences per step
    assert abs(d["value"] - 32 * d["n_gpus"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 2e-3
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) / r["achieved"] < 2e-3
    assert r["traffic"] is None or 0.9 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.5
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] and c["sample"]
