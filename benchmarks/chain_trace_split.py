"""Post-processing for `rocprofv3 --kernel-trace --output-format csv -- python chain_probe.py`: per-kernel average
duration in the first half of the run (true dependencies) and in the second half (frozen inputs)."""
import collections
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kind"] == "KERNEL_DISPATCH"]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
by = collections.defaultdict(list)
for r in rows:
    by[r["Kernel_Name"].split("(")[0][:48]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print(f"{'kernel':50s} {'n':>5s} {'real us':>8s} {'frozen us':>9s} {'delta':>6s}")
tot = [0.0, 0.0]
for k, d in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    if len(d) < 300:
        continue
    h = len(d) // 2
    a = sorted(d[:h])[h // 2] / 1e3          # medians
    b = sorted(d[h:])[(len(d) - h) // 2] / 1e3
    tot[0] += a
    tot[1] += b
    print(f"{k:50s} {len(d):5d} {a:8.2f} {b:9.2f} {a - b:6.2f}")
print(f"{'sum of medians':50s} {'':5s} {tot[0]:8.2f} {tot[1]:9.2f} {tot[0] - tot[1]:6.2f}")
