"""Rebuild launches from gpurun_out/trace_step.npz (tools/trace_step.py) and print a timeline window
plus per-kernel statistics: duration alone vs overlapped, gaps in each stream."""
import sys, numpy as np
from collections import defaultdict
path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/trace_step.npz"
z = np.load(path)
NAMES = {1: "gemm", 2: "attn", 3: "norm", 4: "normK", 5: "rope"}
recs = np.concatenate([z[k] for k in z.files]).astype(np.int64)
start, end, meta = recs[:, 0], recs[:, 1], recs[:, 2]
kid, grid, blk = meta >> 48, (meta >> 24) & 0xffffff, meta & 0xffffff
order = np.argsort(start, kind="stable")
launches = []   # dict(kid, grid, s0, s1, e1, n)
open_ = defaultdict(list)
for i in order:
    key = (int(kid[i]), int(grid[i]))
    lst = open_[key]
    for L in lst:
        if blk[i] not in L["blocks"]:
            break
    else:
        L = dict(kid=key[0], grid=key[1], s0=start[i], s1=start[i], e0=end[i], e1=end[i], blocks=set())
        lst.append(L); launches.append(L)
    L["blocks"].add(int(blk[i])); L["s1"] = max(L["s1"], start[i]); L["e1"] = max(L["e1"], end[i]); L["e0"] = min(L["e0"], end[i])
    if len(L["blocks"]) == L["grid"]:
        lst.remove(L)
full = [L for L in launches if len(L["blocks"]) == L["grid"]]
full.sort(key=lambda L: L["s0"])
t_base = full[len(full) // 2]["s0"]
print(len(launches), "launches,", len(full), "complete")
# timeline window
span = float(sys.argv[2]) if len(sys.argv) > 2 else 400.0
win = [L for L in full if 0 <= (L["s0"] - t_base) / 100 < span]
print("   start     end    dur  ramp  kernel grid")
for L in win:
    print("%8.1f %8.1f %6.1f %5.1f  %-6s %d" % ((L["s0"] - t_base) / 100, (L["e1"] - t_base) / 100, (L["e1"] - L["s0"]) / 100,
                                             (L["s1"] - L["s0"]) / 100, NAMES[L["kid"]], L["grid"]))

# ---- idle gaps of the whole GPU (no instrumented kernel running) and a coarse activity profile ----
iv = sorted((L["s0"], L["e1"]) for L in full)
gaps = []
cur_e = iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        gaps.append((cur_e, s))
    cur_e = max(cur_e, e)
tot = (iv[-1][1] - iv[0][0]) / 100
idle = sum(b - a for a, b in gaps) / 100
print("\ntrace span %.0f us, idle (no instrumented kernel) %.0f us = %.1f %%" % (tot, idle, 100 * idle / tot))
big = [(a, b) for a, b in gaps if b - a > 1000]
print("gaps > 10 us: %d, total %.0f us" % (len(big), sum(b - a for a, b in big) / 100))
for a, b in big[:40]:
    print("  at %9.1f us: %6.1f us" % ((a - iv[0][0]) / 100, (b - a) / 100))

# ---- steady state: the last 40 ms ----
t_end = iv[-1][1]
w0 = t_end - 40000 * 100
sel = [L for L in full if L["s0"] >= w0]
iv2 = sorted((L["s0"], L["e1"]) for L in sel)
gaps2 = []
cur_e = iv2[0][1]
for s, e in iv2[1:]:
    if s > cur_e:
        gaps2.append((cur_e, s))
    cur_e = max(cur_e, e)
tot = (iv2[-1][1] - iv2[0][0]) / 100
idle = sum(b - a for a, b in gaps2) / 100
print("\nlast 40 ms: idle %.0f us = %.1f %%; gaps > 10 us: %s" % (idle, 100 * idle / tot,
      ["%.0f@%.0f" % ((b - a) / 100, (a - w0) / 100) for a, b in gaps2 if b - a > 1000][:60]))
busy1 = 0  # time with exactly one launch running
ev = sorted([(s, 1) for s, e in iv2] + [(e, -1) for s, e in iv2])
depth, last, hist = 0, ev[0][0], defaultdict(int)
for t, d in ev:
    hist[depth] += t - last
    last = t; depth += d
print("concurrency histogram (us):", {k: round(v / 100) for k, v in sorted(hist.items())})
by = defaultdict(list)
for L in sel:
    by[(NAMES[L["kid"]], L["grid"])].append((L["e1"] - L["s0"]) / 100)
for k, v in sorted(by.items()):
    print("  %-6s grid %4d: n=%5d  dur med %.1f  mean %.1f  min %.1f  max %.1f" % (k[0], k[1], len(v), np.median(v), np.mean(v), min(v), max(v)))

if len(sys.argv) > 3:  # timeline around the k-th big gap of the steady-state window
    k = int(sys.argv[3])
    bigs = [(a, b) for a, b in gaps2 if b - a > 30000]
    a, b = bigs[k]
    print("\naround the gap at %.0f us (%.0f us long):" % ((a - w0) / 100, (b - a) / 100))
    for L in sel:
        if a - 60000 <= L["s0"] <= b + 150000:
            print("%9.1f %9.1f %6.1f %5.1f  %-6s %d" % ((L["s0"] - a) / 100, (L["e1"] - a) / 100, (L["e1"] - L["s0"]) / 100,
                                                     (L["s1"] - L["s0"]) / 100, NAMES[L["kid"]], L["grid"]))
