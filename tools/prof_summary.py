"""Cuts rocprofv3 output down to what profiles/ keeps.

  python tools/prof_summary.py stats <dir> <out.csv>             the *_kernel_stats.csv of a --stats run: top rows, kernel names
                                                                 shortened to their template head (torch's RNG kernels are 5 kB each)
  python tools/prof_summary.py pmc <fetch_dir> <write_dir> <kernel substring> <algorithmic bytes> <out.json>
                                                                 per-launch HBM bytes of one kernel from two --pmc passes
                                                                 (FETCH_SIZE, WRITE_SIZE; KB units), with the gfx950 correction of
                                                                 /opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE x 2 for wide
                                                                 coalesced streaming reads)
  python tools/prof_summary.py counters <kernel substring> <out.json> <dir>...
                                                                 per-launch means of every counter found under the dirs
                                                                 (one --pmc pass each) for one kernel
"""
import csv
import glob
import json
import os
import sys


def find(d, pattern):
    hits = sorted(glob.glob(os.path.join(d, "**", pattern), recursive=True))
    if not hits:
        raise SystemExit(f"no {pattern} under {d}")
    return hits[-1]


def short(name):
    name = name.replace("void ", "")
    cut = name.find("(")
    if cut > 0:
        name = name[:cut]
    return name if len(name) <= 140 else name[:137] + "..."


def stats(d, out):
    rows = list(csv.DictReader(open(find(d, "*kernel_stats.csv"))))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows[:28]:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], f"{float(r['AverageNs']):.1f}", r["Percentage"],
                        r["MinNs"], r["MaxNs"], f"{float(r['StdDev']):.1f}"])
    print(open(out).read())


def pmc_mean(d, kernel, counter):
    vals = []
    for r in csv.DictReader(open(find(d, "*counter_collection.csv"))):
        if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
            vals.append(float(r["Counter_Value"]))
    if not vals:
        raise SystemExit(f"no {counter} rows for {kernel} under {d}")
    vals = vals[len(vals) // 4:]  # drop the warm-up quarter
    return sum(vals) / len(vals), len(vals)


def pmc(fetch_dir, write_dir, kernel, algo, out):
    fetch_kb, n = pmc_mean(fetch_dir, kernel, "FETCH_SIZE")
    write_kb, _ = pmc_mean(write_dir, kernel, "WRITE_SIZE")
    hbm = int(2 * fetch_kb * 1024 + write_kb * 1024)
    res = {"kernel": kernel, "launches": n, "FETCH_SIZE_KB_per_launch": round(fetch_kb, 2),
           "WRITE_SIZE_KB_per_launch": round(write_kb, 2),
           "correction": "gfx950: FETCH_SIZE reports half the bytes of wide coalesced streaming reads "
                         "(MI355X_MICROARCH.md, HBM) -> x2; WRITE_SIZE exact for 16-byte-per-lane stores",
           "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": int(algo),
           "traffic_over_algorithmic": round(hbm / float(algo), 5)}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


def counters(kernel, out, dirs):
    res = {}
    for d in dirs:
        acc = {}
        for r in csv.DictReader(open(find(d, "*counter_collection.csv"))):
            if kernel in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for name, vals in acc.items():
            vals = vals[len(vals) // 4:]
            res[name] = {"launches": len(vals), "mean": sum(vals) / len(vals)}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "counters":
        counters(sys.argv[2], sys.argv[3], sys.argv[4:])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], float(sys.argv[5]), sys.argv[6])
