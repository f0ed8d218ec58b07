"""Readable digest of a rocprofv3 `*_kernel_stats.csv`: short kernel names, calls, average microseconds.
  python benchmarks/stats_summary.py stats.csv [--updates N] [--top 25] [--out digest.csv]
--updates N: also prints microseconds per learn() update (total time of the kernel / N)."""
import argparse
import csv
import re
import sys


def short(name: str) -> str:
    name = name.strip('"')
    if name.startswith("Cijk_"):
        m = re.search(r"MT(\d+x\d+x\d+)", name)
        return f"rocBLAS {name[:14]}..MT{m.group(1) if m else '?'}"
    name = re.sub(r"^void\s+", "", name)
    tmpl = ""
    m = re.match(r"([A-Za-z0-9_:]+)\s*(<[^(]*>)?\(", name)
    if m:
        base = m.group(1).split("::")[-1]
        tmpl = m.group(2) or ""
        if len(tmpl) > 24:
            tmpl = tmpl[:21] + "..>"
        return base + tmpl
    return name[:60]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--updates", type=int, default=0)
    ap.add_argument("--top", type=int, default=25)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.csv)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    out = []
    for r in rows[:a.top]:
        rec = {"kernel": short(r["Name"]), "calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 3),
               "min_us": round(float(r["MinNs"]) / 1e3, 3), "pct": round(float(r["Percentage"]), 2)}
        if a.updates:
            rec["us_per_update"] = round(float(r["TotalDurationNs"]) / 1e3 / a.updates, 3)
        out.append(rec)
    cols = list(out[0].keys())
    w = csv.DictWriter(open(a.out, "w") if a.out else sys.stdout, fieldnames=cols)
    w.writeheader()
    w.writerows(out)
    if a.updates:
        print(f"# sum of the listed kernels: {sum(r['us_per_update'] for r in out):.2f} us per update", file=sys.stderr)


if __name__ == "__main__":
    main()
