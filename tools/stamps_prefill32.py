"""Prints the phase timeline of one workgroup of the 32x32-MFMA prefill body from a LVLLM_PREFILL32_STAMPS=1 build.
Stamps per wave and tile j: 0 loop top, 1 after the early publish, 2 after K.Q^T, 3 after the late stash + publish,
4 after the softmax, 5 after P.V; 6 / 7 = just before / after the s_barrier that publishes tile j (early waves:
inside 0..1; late waves: inside 2..3 of tile j-1)."""
import sys

rows = {}
for line in open(sys.argv[1]):
    f = line.split()
    rows[(int(f[0]), int(f[1]))] = [int(x) for x in f[2:]]
t0 = min(v[0] for v in rows.values() if v[0])
tiles = range(12, 15)
print("barrier of tile j: arrival and release per wave (cycles from t0)")
for t in tiles:
    print(f"  tile {t}: " + "  ".join(f"w{w} {rows[(w, t)][6] - t0:6d}->{rows[(w, t)][7] - t0:6d}" for w in range(8) if rows[(w, t)][6]))
for w in range(8):
    for t in tiles:
        v = rows[(w, t)]
        if not v[0]:
            continue
        names = sys.argv[2].split(",") if len(sys.argv) > 2 else ["publish", "qk", "mid", "softmax", "pv"]
        print(f"wave {w} tile {t}: top {v[0] - t0:6d} | " + " | ".join(f"{n} {v[i + 1] - v[i]:5d}" for i, n in enumerate(names))
              + f" | barrier wait {v[7] - v[6]:5d}")
