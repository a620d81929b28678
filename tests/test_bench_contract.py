"""bench.py's contract: step planning (bursts divide the timed region exactly), the N > 1 self-launch, the
JSON line's fields -- on the code, not on a recording -- and (GPU) the whole script on a tiny model, including
the ctypes kernel leg with a 16-bit and an fp8 KV cache (the path that once handed 2-byte strides to a 1-byte
cache: profiles/r01_tuning.md, "memory access fault")."""
import json
import os
import subprocess
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def args(**kw):
    d = dict(steps=64, warmup=16, num_scheduler_steps=8, on_the_fly=2, scheduling="async", gpus=1)  # two in flight
    d.update(kw)
    return types.SimpleNamespace(**d)


def test_bursts_divide_the_timed_region_exactly():
    b = load_bench()
    assert b.plan_steps(args()) == (8, 8, 2)
    assert b.plan_steps(args(steps=60, warmup=10)) == (6, 5, 2)       # 60 = 10 x 6, 10 = 2 x 5
    assert b.plan_steps(args(steps=20, warmup=5)) == (5, 5, 2)
    assert b.plan_steps(args(steps=7, warmup=3)) == (7, 3, 1)         # one burst: nothing else can be in flight
    assert b.plan_steps(args(num_scheduler_steps=1)) == (1, 1, 2)
    assert b.plan_steps(args(scheduling="sync")) == (8, 8, 1)
    assert b.plan_steps(args(warmup=0)) == (8, 1, 2)
    for n in range(1, 70):
        for k in (1, 2, 3, 8):
            d = b.largest_divisor_at_most(n, k)
            assert 1 <= d <= k and n % d == 0
            assert all(n % e for e in range(d + 1, min(k, n) + 1))


def test_gpus_n_started_as_one_process_launches_n_ranks(monkeypatch):
    """`python bench.py --gpus 4` without WORLD_SIZE starts 4 ranks through torch.distributed.run as a CHILD
    process (never an exec) on 127.0.0.1 and passes its own arguments through."""
    b = load_bench()
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=7)

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "16"])
    rc = b.spawn_replicas(args(gpus=4))
    cmd = seen["cmd"]
    assert rc == 7
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "16"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def check_line(d, batch):
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        base = json.load(f)
    assert d["metric"] == base["metric"]
    for k, t in (("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str)):
        assert isinstance(d[k], t), (k, d[k])
    assert d["vs_baseline"] is None and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert "workload" in d["config"] and "model" not in d["config"]
    # value = tokens of all ranks / time; ms_per_step is per MODEL step whatever the burst length
    assert abs(d["value"] - batch * d["n_gpus"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 2e-3
    assert d["steps"] % d["config"]["num_scheduler_steps"] == 0
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # (avg_launch_us is printed with two decimals: allow for that on launches of a few microseconds)
    tol = max(2e-3, 0.006 / r["avg_launch_us"], 0.06 / r["achieved"])  # ... and `achieved` with one
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) / r["achieved"] < tol
    assert r["traffic"] is None or 0.9 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.5
    assert "single pass" in r["kernel"] or "partition pass" in r["kernel"]


def test_recorded_bench_lines_have_the_contract_fields():
    """Every line committed under profiles/ (one per round) still satisfies the contract."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[2-9]_bench_line*.json")))
    for p in paths:
        with open(p) as f:
            d = json.loads(f.read())
        check_line(d, 32)
        c = d["cpu_baseline"]
        assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] and c["sample"]


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--kv-cache-dtype", "fp8"], ["--num-scheduler-steps", "1", "--scheduling", "sync"]],
                         ids=["default", "fp8_kv", "single_step_sync"])
def test_bench_runs_end_to_end_on_a_tiny_model(extra):
    """The whole script -- engine, bursts, ctypes kernel leg (16-bit and fp8 caches), GEMM leg, per-op baselines --
    in a child process on the tiny model; the line it prints satisfies the contract."""
    env = dict(os.environ)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--tiny", "--steps", "24", "--warmup", "8",
                        "--batch-size", "8", "--context", "64", "--kernel-iters", "64"] + extra,
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    check_line(line, 8)
    # 24-step regions are short: four of them, each exactly 24 model steps; `value` is the median one
    tr = line["timed_regions"]
    assert tr["count"] == 4 and len(tr["tokens_per_s"]) == 4 and tr["value"] == "median"
    assert line["value"] == sorted(tr["tokens_per_s"])[2] and line["ms_per_step"] in tr["ms_per_step"]
    # the launch the step really makes (rope + cache write + attention in one kernel) beside the plain one
    ins = line["roofline"]["in_step"]
    assert ins is None or (ins["avg_launch_us"] > 0 and "ROPE" in ins["kernel"] and 0 < ins["frac"] < 1.5)
    if "sync" not in extra:  # the headline at two steps in flight, three reported beside it
        assert line["config"]["max_num_on_the_fly"] == 2
        o = line["other_settings"]["max_num_on_the_fly=3"]
        assert o["value"] > 0 and o["sequences_resident"] == 24
    if not extra:  # BASELINE configs 3, 4, 5 beside the headline, each with its rate, ms/step and bytes (or FLOPs) / time
        o = line["other_settings"]
        for key, unit in (("config3_chunked_prefill", "tokens/s"), ("config5_fp8_weights_fp8_kv", "tokens/s"),
                          ("config4_encode_only", "sequences/s")):
            e = o[key]
            assert e["value"] > 0 and e["unit"] == unit and e["ms_per_step"] > 0 and e["steps"] > 0 and "BASELINE config" in e["config"]
            h = e["hbm"]
            assert h["peak"] == 8000.0 and h["unit"] == "GB/s" and abs(h["frac"] - h["achieved"] / 8000.0) < 1e-3
        assert o["config3_chunked_prefill"]["requests_per_s"] > 0
        assert o["config3_chunked_prefill"]["short_run"]["value"] > 0  # the 96-prompt run rounds 2 - 4 quoted
        f8 = o["config5_fp8_weights_fp8_kv"]
        assert f8["roofline_attention"]["bound"] == "hbm" and f8["roofline_attention"]["avg_launch_us"] > 0
        assert set(f8["roofline_projections"]["per_shape"]) == {"qkv", "o", "gate_up", "down"}
        m = o["config4_encode_only"]["mfma"]
        # (the tiny model's FLOPs are a few 1e-5 of the peak: the fraction may round to 0.0000)
        assert m["peak"] == 2500.0 and m["unit"] == "TFLOP/s" and 0 <= m["frac"] < 1
        sd = o["sampled_decode"]  # the headline workload with sampled requests (the device-side sampler in the step)
        assert sd["value"] > 0 and sd["unit"] == "tokens/s" and sd["ms_per_step"] > 0 and "top-p" in sd["config"]
    else:
        assert not any(k.startswith("config") or k == "sampled_decode" for k in line.get("other_settings", {}))
    assert line["cpu_baseline"]["value"] > 0
    ops = line["ops_baseline"]["ops"]
    assert set(ops) == {"reshape_and_cache", "rms_norm", "fused_add_rms_norm", "rotary_embedding", "silu_and_mul"}
    assert all(v["gpu_us"] > 0 and v["cpu_us"] > 0 for v in ops.values())


@pytest.mark.gpu
def test_bench_gpus_2_launches_its_ranks_and_sums_them():
    """`python bench.py --gpus 2` as ONE plain process: it starts two ranks through torch.distributed.run, they
    meet at the barriers, the slowest clock and the summed tokens make one line with n_gpus = 2.  On a one-GPU box
    both ranks share cuda:0 and the process group is gloo (RCCL refuses two ranks on one device); everything else
    is the path the driver's multi-GPU run takes."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--tiny", "--steps", "16",
                        "--warmup", "8", "--batch-size", "8", "--context", "64", "--kernel-iters", "64",
                        "--replica-backend", "gloo", "--single-device"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0)"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 16 and d["config"]["parallelism"].startswith("dp2")
    check_line(d, 8)  # value = 2 ranks x 8 sequences per step / the slowest rank's time
    assert d["cpu_baseline"] is None  # rank 0 times the CPU baseline at N = 1 only


@pytest.mark.gpu
def test_prefill_roofline_leg_reports_the_kernel_against_the_mfma_peak():
    """`roofline_prefill` of the bench line (row f-1): FLOPs = 4 D (visible pairs) H over the HIP-event time of the
    launch, against the dense bf16 MFMA peak; the leg runs the shipped dispatch (32x32-MFMA body at this size)."""
    import torch
    bench = load_bench()
    eng = types.SimpleNamespace(
        model_config=types.SimpleNamespace(num_attention_heads=8, num_key_value_heads=2, head_dim=128, dtype=torch.bfloat16),
        cache_config=types.SimpleNamespace(block_size=16), device=torch.device("cuda:0"))
    r = bench.prefill_leg(eng, qlen=1024, iters=3)
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert r["flops_per_launch"] == 4.0 * 128 * (1024 * 1025 // 2) * 8
    assert abs(r["achieved"] - r["flops_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e12) <= 0.02 * r["achieved"] + 0.1
    assert abs(r["frac"] - r["achieved"] / 2500.0) < 1e-3 and 0 < r["frac"] < 1
    assert r["min_launch_us"] <= r["avg_launch_us"]
