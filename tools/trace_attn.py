"""Per-workgroup timeline of decode-attention launches from a -DLVLLM_TRACE build (tools/build_variant.sh trace_attn
attention_bf16.hip -DLVLLM_TRACE): trains of back-to-back launches at the metric's shape; for every launch the spread of
workgroup starts, the workgroup durations and the tail, and the gap to the next launch.  100 MHz wall clock."""
import argparse, ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import light_vllm_amd  # noqa
from light_vllm_amd import _custom_ops as ops

ap = argparse.ArgumentParser()
ap.add_argument("--kv", default="fp8")
ap.add_argument("--bs", type=int, default=32)
ap.add_argument("--seq", type=int, default=1024)
ap.add_argument("--launches", type=int, default=64)
a = ap.parse_args()
dev = "cuda:0"
B, H, KVH, D, BS, L = a.bs, 32, 8, 128, 16, a.seq
nblk = L // BS
NB = B * nblk + 7
torch.manual_seed(0)
caches = []
for i in range(8):
    if a.kv == "fp8":
        kc = (torch.randn(NB, KVH, D // 16, BS, 16, device=dev) * 0.5).to(torch.float8_e4m3fn).view(torch.uint8)
        vc = (torch.randn(NB, KVH, D, BS, device=dev) * 0.5).to(torch.float8_e4m3fn).view(torch.uint8)
    else:
        kc = (torch.randn(NB, KVH, D // 8, BS, 8, device=dev) * 0.5).to(torch.bfloat16)
        vc = (torch.randn(NB, KVH, D, BS, device=dev) * 0.5).to(torch.bfloat16)
    caches.append((kc, vc, torch.randperm(NB, device=dev)[: B * nblk].view(B, nblk).to(torch.int32)))
q = (torch.randn(B, H, D, device=dev) * 0.5).to(torch.bfloat16)
sl = torch.full((B,), L, dtype=torch.int32, device=dev)
out = torch.zeros_like(q)
P = (L + 511) // 512
tmp = torch.zeros(B, H, P, D, dtype=torch.bfloat16, device=dev)
es = torch.zeros(B, H, P, dtype=torch.float32, device=dev)
ml = torch.zeros_like(es)
for i in range(a.launches + 32):
    kc, vc, bt = caches[i % 8]
    ops.paged_attention_v2(out, es, ml, tmp, q, kc, vc, KVH, 1 / math.sqrt(D), bt, sl, BS, L, None, a.kv, 1.0, 1.0)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.path.join(os.path.dirname(light_vllm_amd.__file__), "lib", "liblvllm_hip.so"))
NREC = 1 << 20
buf = np.zeros(3 * NREC, dtype=np.uint64)
head = ctypes.c_uint(0)
assert lib.lvllm_trace_read_attn(buf.ctypes.data_as(ctypes.c_void_p), ctypes.byref(head)) == 0
rec = buf.reshape(NREC, 3)[: head.value]
rec = rec[(rec[:, 2] >> 48) == 2]
grid = int((rec[0, 2] >> 24) & 0xffffff)
rec = rec[np.argsort(rec[:, 0], kind="stable")]
n = len(rec) // grid
rows = []
for i in range(n - a.launches, n):
    r = rec[i * grid:(i + 1) * grid]
    s, e = r[:, 0].astype(np.int64), r[:, 1].astype(np.int64)
    rows.append((s.min(), s.max(), e.min(), e.max(), np.median(e - s), (e - s).max(), (e - s).min()))
rows = np.array(rows, dtype=np.float64) / 100.0  # us
span = rows[:, 3] - rows[:, 0]
gap = rows[1:, 0] - rows[:-1, 3]
period = rows[1:, 0] - rows[:-1, 0]
print(f"{a.kv}: {grid} workgroups per launch, {len(rows)} launches")
print(f"  launch period (first start -> next first start)  {np.median(period):6.2f} us")
print(f"  span first start -> last end                      {np.median(span):6.2f} us")
print(f"  gap last end -> next first start                  {np.median(gap):6.2f} us")
print(f"  start spread (last start - first start)           {np.median(rows[:, 1] - rows[:, 0]):6.2f} us")
print(f"  workgroup duration median / min / max             {np.median(rows[:, 4]):6.2f} / {np.median(rows[:, 6]):6.2f} / {np.median(rows[:, 5]):6.2f} us")
print(f"  tail (last end - first end)                       {np.median(rows[:, 3] - rows[:, 2]):6.2f} us")
