"""profiles/r01_pmc_gemm.json from the two counter_collection.csv files of tools/pmc_gemm.py
(usage: pmc_gemm_summary.py <fetch.csv> <write.csv> <out.json> [--w8])."""
import csv, json, sys
from collections import defaultdict
shapes = [("qkv", 6144, 4096), ("o", 4096, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336)]
def per_launch(path, counter):
    vals = []
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "skinny_gemm_kernel" in r["Kernel_Name"]:
            vals.append(float(r["Counter_Value"]))
    return vals
W8 = "--w8" in sys.argv
f, w = per_launch(sys.argv[1], "FETCH_SIZE"), per_launch(sys.argv[2], "WRITE_SIZE")
assert len(f) == len(w) == 16 * len(shapes), (len(f), len(w))
out = {"command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python tools/pmc_gemm.py  (second pass: --pmc WRITE_SIZE)",
       "kernel": "lvllm::skinny_gemm_kernel<BF16, 2, 8, packed, W8, XQ> (fp8 weights, fp8 activations in; gate_up: SwiGLU + fp8 out; "
                 "down: fp32 split-K slabs out)" if W8 else "lvllm::skinny_gemm_kernel<BF16, 2, 16, packed>", "M": 32,
       "correction": "gfx950: FETCH_SIZE reports half the bytes of wide coalesced streaming reads (MI355X_MICROARCH.md, HBM) -> x2; WRITE_SIZE exact",
       "shapes": {}}
for i, (name, N, K) in enumerate(shapes):
    fk = sum(f[16 * i + 4:16 * i + 16]) / 12  # skip the first launches (cold instruction / TLB state)
    wk = sum(w[16 * i + 4:16 * i + 16]) / 12
    hbm = (2 * fk + wk) * 1024
    algo = N * K * 2 + 32 * K * 2 + 32 * N * 2 + (4 * 32 * N * 4 if K > 4096 else 0)  # + fp32 split-K partials of down
    if W8:
        algo = N * K + 32 * K + (32 * (N // 2) if name == "gate_up" else 4 * 32 * N * 4 if K > 4096 else 32 * N * 2)
    out["shapes"][name] = {"FETCH_SIZE_KB_per_launch": round(fk, 1), "WRITE_SIZE_KB_per_launch": round(wk, 1),
                           "hbm_bytes_per_launch": int(hbm), "algorithmic_bytes_per_launch": algo,
                           "traffic_over_algorithmic": round(hbm / algo, 4)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["shapes"], indent=1))
