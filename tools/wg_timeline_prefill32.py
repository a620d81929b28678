"""Per-workgroup timeline of one launch of the 32x32-MFMA prefill body from a -DLVLLM_PREFILL32_STAMPS=2 build
(LVLLM_PREFILL32_WG_FILE=...): where the launch's time goes outside the tile loop -- prologues, epilogues, the
gaps between the workgroups a CU runs one after the other, and the tail.  Clock: 100 MHz (10 ns)."""
import sys
from collections import defaultdict

rows = []
for line in open(sys.argv[1]):
    f = [int(x) for x in line.split()]
    if f[1] == 0:
        continue
    rows.append(dict(wg=f[0], t0=f[1], t1=f[2], t2=f[3], t3=f[4], tiles=f[5], hw=f[6]))
base = min(r["t0"] for r in rows)
end = max(r["t3"] for r in rows)
us = lambda t: t / 100.0
print(f"{len(rows)} workgroups, launch span {us(end - base):.1f} us")
tot_tiles = sum(r["tiles"] for r in rows)
loop = sum(r["t2"] - r["t1"] for r in rows)
print(f"tiles walked {tot_tiles}; loop time per tile {us(loop) / tot_tiles:.3f} us; "
      f"prologue mean {us(sum(r['t1'] - r['t0'] for r in rows)) / len(rows):.2f} us, "
      f"epilogue mean {us(sum(r['t3'] - r['t2'] for r in rows)) / len(rows):.2f} us")
cus = defaultdict(list)
for r in rows:
    cus[r["hw"] & ~0x3f0000000f if False else (r["hw"] >> 32, (r["hw"] >> 8) & 0xf, (r["hw"] >> 13) & 0x7)].append(r)  # (xcc, cu, se)
print(f"{len(cus)} distinct (XCC, CU, SE) ids")
busy, gaps, ends, firsts = [], [], [], []
for k, lst in cus.items():
    lst.sort(key=lambda r: r["t0"])
    busy.append(sum(r["t3"] - r["t0"] for r in lst))
    firsts.append(lst[0]["t0"] - base)
    ends.append(lst[-1]["t3"] - base)
    for a, b in zip(lst, lst[1:]):
        gaps.append(b["t0"] - a["t3"])
ends.sort()
print(f"first workgroup of a CU starts {us(min(firsts)):.1f} .. {us(max(firsts)):.1f} us after the first of all")
print(f"CU busy (sum of its workgroups) min {us(min(busy)):.1f} mean {us(sum(busy) / len(busy)):.1f} max {us(max(busy)):.1f} us")
if gaps:
    print(f"gap between consecutive workgroups on a CU: mean {us(sum(gaps) / len(gaps)):.2f} us, max {us(max(gaps)):.2f} us ({len(gaps)} gaps)")
print(f"CU finish times: 10 % {us(ends[len(ends) // 10]):.1f}  50 % {us(ends[len(ends) // 2]):.1f}  90 % {us(ends[9 * len(ends) // 10]):.1f}  last {us(ends[-1]):.1f} us")
per = defaultdict(int)
for k, lst in cus.items():
    per[len(lst)] += 1
print("workgroups per CU:", dict(sorted(per.items())))
