"""Per-dispatch view of the update chain from a rocprofv3 --kernel-trace CSV: duration histogram per kernel and the gaps between
consecutive dispatches (start of one - end of the one before). Usage: python benchmarks/trace_gaps.py <kernel_trace.csv>"""
import csv, sys, collections
import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = {"layer1_kernel<6, true": "L1", "linear_stats": "G2", "layer2_head": "HD", "gemm_bundle": "GB", "bwd_finish": "FN"}


def tag(n):
    for k, v in names.items():
        if k in n:
            return v
    return None


dur = collections.defaultdict(list)
gap = collections.defaultdict(list)
prev = None
for r in rows:
    t = tag(r["Kernel_Name"])
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t:
        dur[t].append((e - s) / 1000.0)
        if prev and prev[0]:
            gap[prev[0] + ">" + t].append((s - prev[1]) / 1000.0)
    prev = (t, e)
for t, v in dur.items():
    v = np.array(v)
    print(f"{t}: n {len(v)}  mean {v.mean():.2f}  p5 {np.percentile(v, 5):.2f}  p50 {np.percentile(v, 50):.2f}  p95 {np.percentile(v, 95):.2f}  max {v.max():.2f}")
for t, v in sorted(gap.items(), key=lambda kv: -len(kv[1]))[:8]:
    v = np.array(v)
    print(f"gap {t}: n {len(v)}  mean {v.mean():.2f}  p5 {np.percentile(v, 5):.2f}  p50 {np.percentile(v, 50):.2f}  p95 {np.percentile(v, 95):.2f}")
